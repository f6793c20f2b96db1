// th_trace3c.h — the HYBRID closest-hit walk: every ray is answered exactly as accel/bvh.jl:212-258 answers it on the CANONICAL tree (the reference's own
// construction, th_bvh_ref.h, or the tree a host handed over), while the bulk of the rays never touch that tree.
//
// Why two trees.  intersect!(bvh, ray) keeps the LAST accepted primitive: a primitive is accepted iff its own test passes with the CURRENT t_max
// (triangle_mesh.jl:211-214 rejects `t_scaled > t_max * det`, equality accepts; sphere.jl:137), a subtree is entered iff its box passes bounds.jl:186-200
// with the current t_max — and a sphere entered from inside returns its far root WITHOUT looking at t_max (sphere.jl:137-138).  So the
// answer depends on the visiting order, i.e. on the topology, for (a) rays that start inside a sphere, (b) rays with two acceptable primitives at (nearly)
// the same t, (c) rays that graze a leaf's box so closely that `tx_min < t_max` is decided by rounding.  The reference's builder (12 buckets born as the
// point 0, range-length weights: A.6) makes a tree that costs 2x the library's binned-SAH tree to walk.  This kernel walks the library's tree — the
// ACCELERATOR — and returns, per ray, either an answer together with a certificate that EVERY valid tree gives it, or the ray itself on a fallback list
// that k_trace3 (th_trace2.h) then walks on the canonical tree in the reference's order.
//
// The certificate.  Everything in the reference's tests except the comparisons with t_max is a function of (ray, primitive) or (ray, box) alone.  Call a
// primitive p a CANDIDATE when the t_max-free clauses of bounds.jl:186-198 pass on the box of the canonical LEAF that holds it and the t_max-free part of
// its own test passes; t_p (the t it would set) and tau_p (the smallest t_max that accepts it, within 4 ulps of t_p) do not depend on the walk.  Commit
// makes every accelerator leaf (part of) one canonical leaf with bit for bit that leaf's box — regrouping the accelerator's leaves where the two builders drew
// them differently (tu_scene.hip conform_accelerator) — and verifies it, and for rays without a zero direction component a leaf box that passes the t_max-free clauses implies that every ancestor box — in any tree whose
// boxes nest — passes them, and enters no later than the leaf (the slab products are monotonic in the box planes).  So the SET of candidates is the same in
// both trees; only the t_max clauses see the order.  With D = the largest coordinate offset between the ray origin and the scene bound (the reach of the
// ray's Float32 arithmetic; em / tight_scale) and kz the ray's dominant axis (every hit lies inside the scene, so t <= D |1 / d[kz]|),
//     dt = 2^-16 D |1 / d[kz]|      (kCertDt: 256 ulps of the largest possible t — the rounding between t_p, tau_p and the slab products is a few ulps of it)
// the walk keeps t_lim = T + 2 dt, T = the t of the last accepted candidate (the INCUMBENT) or the ray's own t_max, and
//     * tests every primitive it reaches against the relaxed limit t_lim.  A candidate p found that way is ACCEPTED iff t_p <= T - 2 dt (= t_lim - 4 dt) and the EXACT
//       entry distance of p's leaf box is <= t_p + dt (the reference, holding any t_max > t_p + dt, enters the leaf).  Otherwise the ray is FLAGGED: p lies within 2 dt of
//       the incumbent or of the ray's own t_max, or its leaf box lets the reference in only beyond it — the order decides (shared edges, coincident surfaces, grazed boxes);
//     * culls a box c only when entry(c) >= t_lim + mb, a bound below which no candidate inside c can have its t.  Two facts, for every triangle candidate p under c:
//       (i) the ray point at the computed t_p lies within `growth` = 2^-20 D + 2^-18 sq_flat of p's boxes, per axis (kCertGrow: the sheared vertex coordinates carry <= 5
//       ulps of D each; kCertFlat x sq_flat: what the edge functions' rounding moves the point for a FLAT triangle — zero extent along some axis, walls and floors of any
//       size —, sq_flat = the scene's largest L^3 / 2A over such triangles), so t_p >= entry(c) - growth x max |1 / d|;  (ii) for the others, t_p = sum(e_k z_k sz) / sum(e_k)
//       is a convex combination of the vertices' own depths along kz, all inside c's kz slab: t_p differs from the true depth of that point by at most the triangle's
//       kz extent <= mle_small[kz] (the largest extent of a non-flat leaf box along kz, a scene constant) times |1 / d[kz]|.  mb = mle_small[kz] |1 / d[kz]| + growth max |1 / d|,
//       one value per ray (AXIS: the growth term per axis instead, from the grown box: rays that start far outside the scene).  A ray whose growth term exceeds kCertCap
//       times the extent term (near-axis-parallel: max |1 / d| ~ 1e3) would overshoot every hit by that much: it is flagged when it is fetched;
//     * tests the scene's spheres (<= kCertMaxSpheres, else no hybrid mode) for every ray BEFORE its walk, all of them (the chunk pre-pass below): no box on a sphere's
//       path is then ever culled — the Float32 quadratic reports hits up to 1e-3 |o - c| outside the sphere's box, no margin derived from the boxes bounds where they lie.
//       A full sphere seen from outside is a candidate like any other (same acceptance rule, its own leaf box).  A full sphere entered from INSIDE (sphere.jl:137-138:
//       the far root t1 is returned whatever t_max is — every ray reflected off or refracted into a sphere starts that way) is ALWAYS accepted by the reference when it
//       tests it — and it always does: its leaf box holds the origin (required here: entry <= 0), no t_max culls its path — and OVERWRITES what the reference held.  So
//       of the other candidates only those the reference tests AFTER that sphere count, and that is a function of the canonical tree and the direction signs alone:
//       each accelerator primitive record carries, per sphere, the split axis of the canonical node where its root path parts from the sphere's and the child it is in
//       (3 bits each, the ORDER WORD, set at commit; axis code 3 = the sphere's own leaf, before / behind it); bvh.jl:239-246 visits the second child first iff
//       d[axis] < 0.  A candidate that does not count is skipped, one that does is held to the acceptance rule against T = t1.  A ray inside one sphere that meets a
//       second sphere candidate is flagged;
//     * flags a ray that meets a clipped sphere (sphere.jl:143-149 can return a root that ignores t_max), and a ray with a zero or non-finite direction component, a
//       non-finite origin or a NaN t_max (0 x Inf = NaN breaks the monotonicity argument).
// Claim: an unflagged ray's answer w (or "miss") is what the reference's walk returns on any tree over the same leaves.
//   (1) every other candidate p that counts has t_p, tau_p > T_w + dt: if it was tested it failed the relaxed limit of the moment (>= T_w + 2 dt) — or was accepted and
//       later replaced (each acceptance lowers T by >= 2 dt); if it never was, a box above it was culled with entry >= t_lim + mb, hence t_p >= t_lim >= T_w + 2 dt;
//   (2) on the other tree, before w is tested t_max is the ray's own (>= t_w + 2 dt: the acceptance rule) or some t_p > t_w + dt; w's leaf box enters at <= t_w + dt and
//       every ancestor no later: w is reached, and accepted (tau_w <= t_w + 4 ulp);  (3) afterwards every other candidate that counts is rejected (tau_p > t_w), and the
//       ones that do not count were tested before the sphere that overwrote them.
// The accelerator's own visiting order is irrelevant to the claim; it keeps k_trace3's (near child first by the split axis' sign) because that schedule is tuned.
//
// The chunk pre-pass.  A wave takes rays from the queue in chunks of kChunk; when it takes a chunk it runs ALL of the chunk's rays against ALL spheres, 64 rays at a
// time with every lane (lanes in the middle of a walk included: their walk state just stays in its registers), and leaves {t, canonical slot, state} in the ray's hit
// record: state = accepted-sphere bit, the id of the sphere the ray starts inside of, or "flagged".  A fetch reads the record back (past L1: same wave, other lane;
// the stores are only waited for — an agent-scope fence here would write the XCD's L2 back, measured +70 % kernel time).  Testing the spheres at each fetch instead
// (a dozen lanes at a time, the whole wave paying transforms and quadratics) cost 3x k_trace3's refill; in the leaf phase the sphere code sets the walk's registers.
//
// Primitive records of the accelerator are in ITS leaf order and carry the CANONICAL slot in the second record's .w lane: hits, shading records, the
// inspection API and the oracle all speak canonical slots.
#pragma once
#include "th_trace2.h"
#include "th_trace8.h"  // FallbackList

namespace th {

#ifndef TH_TRACE3C_WAVES
#define TH_TRACE3C_WAVES 6
#endif
#ifndef TH_TRACE3C_AXIS_LESS
#define TH_TRACE3C_AXIS_LESS 0  // 1: the per-axis form (camera rays far outside the scene) one wave per SIMD less
#endif
#ifndef TH_TRACE3C_REFILL
#define TH_TRACE3C_REFILL TH_TRACE_REFILL  // idle lanes of a wave that trigger a refill
#endif
#ifndef TH_TRACE3C_LDS
#define TH_TRACE3C_LDS 11
#endif
#ifndef TH_TRACE3C_LEAF_WAIT
#define TH_TRACE3C_LEAF_WAIT 32
#endif
#ifndef TH_TRACE3C_POP_MIN
#define TH_TRACE3C_POP_MIN 8
#endif
#ifndef TH_TRACE3C_MAX_A
#define TH_TRACE3C_MAX_A 8
#endif
constexpr float kCertDt = 1.52587890625e-5f;     // 2^-16: dt = kCertDt D |1 / d[kz]| (256 ulps of the largest possible t)
constexpr float kCertGrow = 9.5367431640625e-7f;  // 2^-20 (16 ulps): the sheared vertex coordinates x' = fl(fl(v_x - o_x) + fl(S_x fl(v_z - o_z))) carry <= 5 ulps of D each (one for each
                                                  // subtraction, two for S_x, one for the product), so the point of the TRUE triangle with the computed barycentrics lies within 5 ulps of D per
                                                  // lateral axis of the ray point at the computed t — which therefore lies inside the primitive's boxes grown by this x D per axis
constexpr float kCertFlat = 3.814697265625e-6f;   // 2^-18 (64 ulps): … plus what the edge functions' own rounding (<= 3 ulps of each product) moves that point: for a FLAT primitive (in an
                                                  // axis-aligned plane: its thin direction in the sheared frame IS a coordinate axis, the products are long x thin) at most 36 ulps of L^3 / 2A
                                                  // whatever the viewing angle; for the others the kz-extent bound is used instead (mle_small)
constexpr uint32_t kCertOrderSpheres = 10;       // the order word of a primitive record holds 3 bits per sphere: the first ten spheres of a scene have them; a ray that starts INSIDE a later one goes to the reference-order walk
constexpr uint32_t kCertMaxSpheres = 32;         // every ray is tested against every sphere before its walk (a box test each, all lanes): beyond this the canonical tree alone
#ifndef TH_CERT_CAP
#define TH_CERT_CAP 64.0f
#endif
constexpr float kCertCap = TH_CERT_CAP;                  // a ray whose growth margin in t units exceeds this x (the kz extent margin) goes to the reference-order walk at once (near-axis-parallel
                                                  // rays: |1 / d| ~ 1e3 and more): the accelerator walk would overshoot every hit by that much

typedef float v2f __attribute__((ext_vector_type(2)));  // (two products per instruction: v_pk_add_f32 / v_pk_mul_f32)
// {a.x - b[H], a.y - b[H]} and {a.x x b[H], a.y x b[H]}: the packed instructions with ONE half of the second operand feeding both lanes (op_sel), so that a ray's origin and
// reciprocal direction live in three register pairs instead of six (the compiler's own selection duplicates each component into a pair of its own).  IEEE single operations, each
// rounded once: the same numbers as the scalar forms.
template <int H>
TH_D v2f pk_sub_h(v2f a, v2f b) {  // (used by the four-wide step: th_trace3c4.h)
    v2f r;
    if (H == 0)
        asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    else
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// min / max of values that came out of those instructions: written out, because the compiler cannot see that an asm result is no signalling NaN and would put a
// canonicalising v_max_f32 x, x in front of every operand (IEEE mode).  No NaN reaches them (header: rays with a zero or non-finite component never walk here).
TH_D float amin(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
TH_D float amax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
TH_D float amin3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
TH_D float amax3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
template <int H>
TH_D v2f pk_mul_h(v2f a, v2f b) {
    v2f r;
    if (H == 0)
        asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
    else
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

struct CertScene {               // what the certificate needs beside the accelerator's WideScene
    const float* sphere_boxes;   // per sphere id: the box of the canonical leaf that holds it (6 floats)
    const uint32_t* sphere_slots;  // per sphere id: its canonical slot
    uint32_t n_spheres;          // every ray is tested against all of them before its walk (k_trace3c's chunk pre-pass): never more than kCertMaxSpheres (32) in a hybrid scene
    const void* sphere_cert;     // SphereCert[n_spheres]: what those tests read, one contiguous record per sphere
    const float* slot_boxes;     // one-leaf accelerator: per canonical slot, the box of the canonical leaf that holds it (6 floats)
    float inv_tight;             // 1 / WideScene::tight_scale: D = em x inv_tight
    float mle_small[3];          // per axis: the largest extent of a NON-FLAT leaf box along that axis — bounds |computed t - the depth of the ray's point on the (perturbed) triangle|
                                 // of every primitive in such a leaf (both lie between the vertices' depths)
    float sq_flat;               // FLAT leaves (zero extent in some axis: walls, floors — any size): the largest L^3 / (2 A) of their triangles (L longest edge, A area), which bounds
                                 // how far the edge functions' rounding moves the ray point at the computed t off the triangle: 2^-19 x this / |d^[flat axis]|
};

// What k_trace3c reads only when a ray is fetched or handed to the fallback lists lives in HBM behind ONE pointer (scalar loads where it is used) instead of in the kernel's
// argument list: the walk runs at the SGPR limit (lane masks, the queue, the scene), and every argument that stays live across the loop is spilled through v_writelane / v_readlane.
struct CertCold {
    const float* sphere_boxes;
    const uint32_t* sphere_slots;
    uint32_t* fb_list;
    uint32_t* fb_counts;
    uint32_t n_spheres, fb_cap;
    float mle_small[3];
    float sq_flat, inv_tight;
    float pad;
};
struct SphereCert;
struct CertHot {   // … and what the walk itself, or every fetch, needs
    float kdt;     // kCertDt / tight_scale: dt = kdt x em x |1 / d[kz]|
    float kgrow;   // kCertGrow / tight_scale
    float gflat;   // kCertFlat x sq_flat: growth = kgrow x em + gflat
    uint32_t n_spheres;
    const SphereCert* spheres;  // one contiguous record per sphere (a single burst of scalar loads each)
    float mle[3];               // CertScene::mle_small
};
template <int TH_ONE_COPY = 0> __global__ void k_store_cert_cold(CertCold* dst, CertCold v) { *dst = v; }
// "count_visits": one thread, between the certified walk and the fallback walk of a launch (phase 0) and after the fallback walk (phase 1): what the closest-hit visit
// counters gained during the fallback walk goes to nodes_fallback / prims_fallback
template <int TH_ONE_COPY = 0> __global__ void k_hybrid_count_mark(Counters* c, int phase) {
    if (phase == 0) {
        c->nodes_seen = c->nodes_closest;
        c->prims_seen = c->prims_closest;
    } else {
        c->nodes_fallback += c->nodes_closest - c->nodes_seen;
        c->prims_fallback += c->prims_closest - c->prims_seen;
    }
}

// bounds.jl:186-198 on one child box, as slab_test2 (its t_max-free clauses, plus the two tight clauses), returning the exact entry distance and, with AXIS, the entry distance
// of the box GROWN by `grow` (a length) in every axis
template <bool AXIS>
TH_D bool slab_test3(float bx0, float by0, float bz0, float bx1, float by1, float bz1, f3 o, f3 inv_d, float em, float grow, bool tight, bool negx, bool negy, bool negz, float& tmin_out, float& tgrown_out) {
    const float tx_min = ((negx ? bx1 : bx0) - o.x) * inv_d.x;
    const float tx_max = ((negx ? bx0 : bx1) - o.x) * inv_d.x;
    const float ty_min = ((negy ? by1 : by0) - o.y) * inv_d.y;
    const float ty_max = ((negy ? by0 : by1) - o.y) * inv_d.y;
    const float tz_min = ((negz ? bz1 : bz0) - o.z) * inv_d.z;
    const float tz_max = ((negz ? bz0 : bz1) - o.z) * inv_d.z;
    const bool miss_xy = (tx_min > ty_max) | (ty_min > tx_max);     // bounds.jl:188
    const float a = ty_min > tx_min ? ty_min : tx_min;              // :189
    const float b = ty_max > tx_max ? ty_max : tx_max;              // :190
    const bool miss_z = (a > tz_max) | (tz_min > b);                // :194
    const float t_in = tz_min > a ? tz_min : a;                     // :196
    const float t_out = tz_max < b ? tz_max : b;                    // :197
    const float exit_xy = fminf(__fmaf_rn(em, fabsf(inv_d.x), tx_max), __fmaf_rn(em, fabsf(inv_d.y), ty_max));
    const bool miss_tight = tight & ((__fmaf_rn(-em, fabsf(inv_d.z), tz_min) > exit_xy) | (exit_xy < 0.0f));
    tmin_out = t_in;
    tgrown_out = AXIS ? fmaxf(fmaxf(__fmaf_rn(-grow, fabsf(inv_d.x), tx_min), __fmaf_rn(-grow, fabsf(inv_d.y), ty_min)), __fmaf_rn(-grow, fabsf(inv_d.z), tz_min)) : t_in;
    return !(miss_xy | miss_z | miss_tight) & (t_out > 0.0f);
}

// One record per sphere for the sphere pre-pass of k_trace3c: everything a test reads, contiguous (one burst of scalar loads per sphere, no dependent second trip)
struct SphereCert {
    float box[6];       // the box of the canonical leaf that holds the sphere
    float radius;
    uint32_t slot;      // its canonical slot
    float o2w_inv[16];  // SphereRec::o2w_inv (world_to_object.m)
    uint32_t never_clipped, pad[3];
};
static_assert(sizeof(SphereCert) == 112, "SphereCert layout");

// sphere.jl:125-158 up to the roots: 0 = no candidate within t_lim, 1 = candidate at t (a full sphere seen from outside: accepted iff t0 <= t_max, sets t0),
// 3 = a full sphere entered from INSIDE: the reference takes the far root t1 WHATEVER t_max is (sphere.jl:137-138) — a candidate that is always accepted (the caller's
// "sticky" rule), 2 = a clipped sphere (:143-149 may return a root that ignores t_max after a clipped first one) or a NaN root: the ray goes to the reference-order walk
template <bool FULL_ONLY>
TH_D int sphere_candidate_m(const float* o2w_inv, float radius, bool never_clipped, f3 o, f3 d, float t_lim, float& t);
template <bool FULL_ONLY>
TH_D int sphere_candidate_c(const SphereRec& s, f3 o, f3 d, float t_lim, float& t) {
    return sphere_candidate_m<FULL_ONLY>(s.o2w_inv, s.radius, s.never_clipped != 0u, o, d, t_lim, t);
}
template <bool FULL_ONLY>
TH_D int sphere_candidate_m(const float* o2w_inv, float radius, bool never_clipped, f3 o, f3 d, float t_lim, float& t) {
    const f3 oo = xf_point(o2w_inv, o);
    const f3 od = xf_vec(o2w_inv, d);
    const float nd = norm(od);
    const float a = nd * nd;
    const float b = dot(2.0f * oo, od);
    const float no = norm(oo);
    const float c = no * no - radius * radius;
    float t0, t1;
    if (!solve_quadratic(a, b, c, t0, t1)) return 0;
    if (!(t1 >= 0.0f)) return t1 < 0.0f ? 0 : 2;  // (a NaN root: let the reference-order walk decide)
    if (!FULL_ONLY && !never_clipped) return 2;
    if (!(t0 >= 0.0f)) {
        if (!(t0 < 0.0f)) return 2;
        t = t1;
        return 3;
    }
    if (t0 > t_lim) return 0;
    t = t0;
    return 1;
}

// rays for the reference-order walk: appended to the fallback lists (kSeg lists, one counter each — a single counter word serialises at ~88 atomics per microsecond; a wave
// starts at its own list and moves on while a list is full: together they hold as many entries as the queue has rays).  Whole wave; returns the number of rays appended.
// Not inlined: its loop and the list's addresses then stay out of the walk's registers.
TH_D uint32_t fallback_append(FallbackList fb, bool to_fb, uint32_t idx, uint32_t first_list) {
    uint32_t fseg = first_list, n = 0;
    for (int tries = 0; tries < kSeg && __ballot(to_fb) != 0ull; ++tries) {
        const uint32_t j = wave_compact(to_fb, &fb.counts[fseg * kCtrStride]);
        const bool put = to_fb && j < fb.cap;
        if (put) {
            fb.list[(size_t)fseg * fb.cap + j] = idx;
            to_fb = false;
        }
        n += (uint32_t)__popcll(__ballot(put));
        fseg = (fseg + 1) % kSeg;
    }
    return n;
}

// AXIS: the cull bound from the box grown per axis (two fma + max3 more per child) instead of the scalar margin — for launches whose rays start far outside the scene
// (camera rays 50 scene sizes away: D, hence the scalar margin, is 50x larger, while their binding slab is almost always the dominant axis'); TraceOut::far_hint picks it.
// Register budget: the node reference and its primitive count stay packed in ONE word (the child / stack word format), what is only read at a ray's fetch or at a hand-over lives
// behind one pointer (CertCold), the spheres are tested in the chunk pre-pass: the certificate costs the walk two live values (the per-ray margin and the current node's entry distance;
// t_lim takes t_max's place).
template <bool COUNT, bool FULL_ONLY, bool BIG = false, bool AXIS = false>
__global__ __launch_bounds__(kBlock, (BIG || (AXIS && TH_TRACE3C_AXIS_LESS)) ? TH_TRACE3C_WAVES - 1 : TH_TRACE3C_WAVES) void k_trace3c(DeviceScene sc /* prims: the accelerator's order */, WideScene ws /* the accelerator */, CertHot ch,
                                                                                               const CertCold* __restrict__ cold, SegQueue q, const float4* __restrict__ ro,
                                                                                               const float4* __restrict__ rd, const float* __restrict__ tmax_or_null, TraceOut out,
                                                                                               uint32_t* __restrict__ work, uint2* __restrict__ overflow, Counters* ctr) {
    constexpr int kLds = TH_TRACE3C_LDS;
    constexpr uint32_t kLeafBit = 1u << 24;  // a node word >= this (and != kRefNone) is a leaf: ref | count << 24
    __shared__ uint2 s_stk[kLds][kBlock];  // {child word, entry distance}: one 8-byte LDS access per push / pop
    // per-lane state that is only touched when a ray is fetched, accepted or finished lives in LDS, not in registers (the walk runs at the 80-VGPR line of six waves per
    // SIMD; a scratch spill costs a trip to memory, an LDS word 64 cycles): the ray's queue index, its state word, the entry distance of the node in hand
    __shared__ uint32_t s_idx[kBlock];
    __shared__ uint32_t s_st[kBlock];
    __shared__ float s_ex[kBlock];
    __shared__ SegView sv;
    seg_load(q, sv);
    const uint32_t tid = threadIdx.x;
    const uint32_t gthreads = gridDim.x * kBlock;
    const uint32_t gtid = blockIdx.x * kBlock + tid;
    const uint32_t lane = lane_id();

    bool active = false, exhausted = false, to_fb = false;
    uint32_t wseg = __builtin_amdgcn_readfirstlane((gtid >> 6) % kSeg), dry = 0, pool_next = 0, pool_end = 0;  // wave-uniform
    uint32_t cur = kRefNone;  // kRefNone, an interior node's index (< 2^24), or a leaf word
    int sp = 0;
    f3 o = splat3(0.0f), inv_d = splat3(0.0f);
    float em = 0.0f;
    RayShear shear{0, 0.0f, 0.0f, 0.0f};
    // (the direction signs are read off inv_d where they are needed: a ray with a zero component, the one case where sign(1 / d) is not sign(d), never walks here)
#define negx (inv_d.x < 0.0f)
#define negy (inv_d.y < 0.0f)
#define negz (inv_d.z < 0.0f)
    float t_lim = 0.0f;   // t_max + 2 dt, t_max = the t of the last accepted candidate (or the ray's own t_max): what the primitive tests accept up to; a box is culled when the lower
                          // bound of what it holds — its entry distance minus the margin — reaches it
    float mb = 0.0f;      // per ray: the margin of the lower bound: non-flat primitives their kz extent, everything the growth in the entering axis (with AXIS the growth is applied per
                          // axis to the box instead, and is not in here)
    // s_ex[tid]: entry distance of the node in `cur` (the reference's tx_min of its box)
    // s_st[tid]: 0: nothing accepted yet, 1: a candidate is; bits 8..: 1 + the sphere the ray started INSIDE of (header: what the reference tests before that sphere does not count)
    uint32_t nn = 0, np = 0;
    uint32_t n_fb = 0;    // wave-uniform
    unsigned long long n_why[4] = {0ull, 0ull, 0ull, 0ull};  // COUNT: why rays went to the canonical tree — 0 direction / finiteness / cap, 1 a sphere (clipped, inside two), 2 near tie / guard; [3]: rays that start INSIDE a sphere and were certified (the order word)
    uint32_t why = 0;

#ifdef TH_DIAG_PHASES
    unsigned long long ph_cyc[4] = {0, 0, 0, 0}, ph_lan[4] = {0, 0, 0, 0}, ph_cnt[4] = {0, 0, 0, 0};  // refill (+ hand-over), pop, node, leaf (tools/phase_probe.py)
#endif
    auto margin_t = [&]() { return ch.kdt * em * fabsf(shear.sz); };       // dt, from what is live (D = em / tight_scale)
    auto growth = [&]() { return __fmaf_rn(ch.kgrow, em, ch.gflat); };     // the length by which the ray point at a primitive's computed t can lie outside the primitive's boxes
    auto inv_max = [&]() { return fmaxf(fmaxf(fabsf(inv_d.x), fabsf(inv_d.y)), fabsf(inv_d.z)); };

    while (true) {
        // ---- rays for the reference-order walk: appended to the fallback lists ----
        if (__ballot(to_fb) != 0ull) {
            const FallbackList fb{uniform_load(&cold->fb_list, 0), uniform_load(&cold->fb_counts, 0), uniform_load(&cold->fb_cap, 0)};
            n_fb += fallback_append(fb, to_fb, s_idx[tid], __builtin_amdgcn_readfirstlane((gtid >> 6) % kSeg));
            to_fb = false;
        }
        // ---- refill idle lanes (as k_trace3) ----
        const unsigned long long idle = __ballot(!active);
        const uint32_t n_idle = (uint32_t)__popcll(idle);
        if (n_idle == 64u || (!exhausted && n_idle >= (uint32_t)TH_TRACE3C_REFILL)) {
            TH_PHASE_BEGIN();
            if (!exhausted) {
                if (pool_next >= pool_end) {
                    const uint32_t cnt = __builtin_amdgcn_readfirstlane(sv.count[wseg]);
                    uint32_t base = cnt, take = (uint32_t)kChunk;
                    if (lane == 0 && cnt != 0u) {
                        const uint32_t at = __hip_atomic_load(&work[wseg * kCtrStride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (at < cnt) {
#if TH_TRACE3_SMALL_CHUNKS
                            if (cnt - at < (uint32_t)TH_TRACE3_SMALL_CHUNKS * (gthreads >> 6) / (uint32_t)kSeg * (uint32_t)kChunk) take = (uint32_t)kChunk / 4u;
#endif
                            base = atomicAdd(&work[wseg * kCtrStride], take);
                        }
                    }
                    base = __builtin_amdgcn_readfirstlane(base);
                    take = __builtin_amdgcn_readfirstlane(take);
                    if (base < cnt) {
                        pool_next = base;
                        pool_end = min(base + take, cnt);
                        dry = 0;
                        // ---- the chunk's SPHERE PRE-PASS: every ray of the chunk this wave now owns against every sphere of the scene, 64 rays at a time with ALL lanes (the lanes that
                        //      are in the middle of a walk work too: their own state just stays where it is).  A sphere is then never hidden from the certificate — whatever the boxes on
                        //      its path do — and the walk skips sphere primitives.  Done here rather than when a ray is fetched (a dozen lanes at a time, the whole wave paying the
                        //      transforms, quadratics and their scalar loads: three times k_trace3's refill cost) or in the leaf phase (where the sphere code sets the walk's register
                        //      count).  The outcome waits in the ray's hit record: {t, slot, state} ----
                        if (ch.n_spheres != 0u) {
                            const uint32_t n_chunk = pool_end - pool_next;
#pragma unroll 1
                            for (uint32_t i0 = 0; i0 < n_chunk; i0 += 64u) {
                                const bool valid = i0 + lane < n_chunk;
                                uint32_t pidx = valid ? seg_phys(q, wseg, pool_next + i0 + lane) : 0u;
                                if (valid && q.indirect) pidx = q.indirect[pidx];
                                float4 po4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), pd4 = make_float4(0.0f, 0.0f, 1.0f, 0.0f);
                                if (valid) {
                                    po4 = ro[pidx];
                                    pd4 = rd[pidx];
                                }
                                const f3 po = mk3(po4.x, po4.y, po4.z), pd = mk3(pd4.x, pd4.y, pd4.z);
                                const f3 pinv = mk3(1.0f / pd.x, 1.0f / pd.y, 1.0f / pd.z);
                                const bool pnx = pd.x < 0.0f, pny = pd.y < 0.0f, pnz = pd.z < 0.0f;
                                const float pem = slab_margin(ws.root_box, ws.tight_scale, po);
                                const float pdt = ch.kdt * pem * fabsf(ray_shear(pd).sz);
                                float p_lim = ((valid && tmax_or_null) ? tmax_or_null[pidx] : kInf) + 2.0f * pdt;
                                uint32_t pst = 0u;
                                float4 prec = make_float4(kInf, __int_as_float(-1), 0.0f, 0.0f);
                                bool pflag = false;
#pragma unroll 1
                                for (uint32_t ks = 0; ks < ch.n_spheres; ++ks) {
                                    const SphereCert sr = uniform_load(ch.spheres, ks);  // wave-uniform: scalar loads, one burst
                                    float ex;
                                    if (valid && !pflag && slab_test2(sr.box[0], sr.box[1], sr.box[2], sr.box[3], sr.box[4], sr.box[5], po, pinv, pem, false, pnx, pny, pnz, ex)) {
                                        if (COUNT) np++;
                                        float t_c = 0.0f;
                                        // a sphere the ray starts inside of is taken whatever the limit is (sphere.jl:137-138); one seen from outside up to the relaxed limit
                                        const int r = sphere_candidate_m<FULL_ONLY>(sr.o2w_inv, sr.radius, sr.never_clipped != 0u, po, pd, p_lim, t_c);
                                        if (r == 2 || (r != 0 && (pst >> 8) != 0u)) {
                                            pflag = true;  // clipped; or the ray starts inside a sphere AND meets another one below that sphere's far root: left to the reference's order
                                        } else if (r != 0) {
                                            // accepted iff it lies 2 dt below the incumbent (p_lim - 4 dt; the ray's own t_max at first) and its leaf box lets the reference in by then.
                                            // A sphere the ray starts INSIDE of (r == 3): the reference takes its far root whenever it tests it — and it always does: the box holds the
                                            // origin (required: ex <= 0), so no t_max culls its path — and forgets what it held; the ray remembers the sphere (state), and of what the
                                            // walk finds only what the reference tests AFTER that sphere counts (the order word of the primitive records, below)
                                            if (!(t_c <= p_lim - 4.0f * pdt) || !(ex <= (r == 3 ? 0.0f : t_c + pdt)) || (r == 3 && ks >= kCertOrderSpheres)) {  // (a sphere without order bits: inside rays to the reference's order)
                                                pflag = true;
                                            } else {
                                                if (COUNT && r == 3) n_why[3]++;  // (not a fallback: rays that start inside a sphere and stay on the accelerator)
                                                p_lim = t_c + 2.0f * pdt;
                                                pst = 1u | (r == 3 ? (ks + 1u) << 8 : 0u);
                                                prec = make_float4(t_c, __uint_as_float(sr.slot), 0.0f, 0.0f);
                                            }
                                        }
                                    }
                                }
                                prec.z = __uint_as_float(pflag ? 0x80000000u : pst);
                                if (valid) out.hits[pidx] = prec;
                            }
                            __builtin_amdgcn_s_waitcnt(0);
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // (the stores have reached L2: no cache is written back or invalidated — an agent-scope fence writes the XCD's whole L2 back.)  The records are read back past L1 when the rays are fetched
                        }
                    } else {
                        pool_next = pool_end = 0;
                        wseg = (wseg + 1) % kSeg;
                        if (++dry >= (uint32_t)kSeg) exhausted = true;
                    }
                }
                const uint32_t avail = pool_end - pool_next;
                if (avail && !active) {
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                    if (rank < avail) {
                        uint32_t idx = seg_phys(q, wseg, pool_next + rank);
                        if (q.indirect) idx = q.indirect[idx];
                        s_idx[tid] = idx;
                        uint32_t st = 0u;
                        const float4 o4 = ro[idx], d4 = rd[idx];
                        o = mk3(o4.x, o4.y, o4.z);
                        const f3 d = mk3(d4.x, d4.y, d4.z);
                        inv_d = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                        em = slab_margin(ws.root_box, ws.tight_scale, o);
                        shear = ray_shear(d);
                        const float t_own = tmax_or_null ? tmax_or_null[idx] : kInf;
                        const float dt = margin_t();
                        const float mkz = (shear.kz == 0 ? ch.mle[0] : (shear.kz == 1 ? ch.mle[1] : ch.mle[2])) * fabsf(shear.sz);
                        mb = AXIS ? mkz : __fmaf_rn(growth(), inv_max(), mkz);
                        t_lim = t_own + 2.0f * dt;
                        sp = 0;
                        active = true;
                        if (COUNT) nn++;
                        // what the certificate does not cover goes to the reference-order walk at once: a zero or non-finite direction component (0 x Inf = NaN in the slab
                        // products), a non-finite origin or margin, a NaN t_max — and near-axis-parallel rays, whose scalar margin would make the walk overshoot every hit
                        // (kCertCap; with AXIS the margin is per axis: no cap)
                        const bool plain = d.x != 0.0f && d.y != 0.0f && d.z != 0.0f && mb < kInf && fabsf(o.x) < kInf && fabsf(o.y) < kInf && fabsf(o.z) < kInf && fabsf(inv_d.x) < kInf &&
                                           fabsf(inv_d.y) < kInf && fabsf(inv_d.z) < kInf && t_own == t_own && (AXIS || mb - mkz <= kCertCap * mkz + dt);
                        float tmin;
                        if (!plain) {
                            to_fb = true;
                            active = false;
                            if (COUNT) n_why[0]++;
                        } else if (ws.root_ref != kRefNone && slab_test2(ws.root_box[0], ws.root_box[1], ws.root_box[2], ws.root_box[3], ws.root_box[4], ws.root_box[5], o, inv_d, em, false, negx, negy, negz, tmin)) {
                            cur = ws.root_ref | (ws.root_cnt << 24);  // (the root is not culled by t: the reference's clause `tmin < t_max` holds whenever anything inside can be accepted)
                            s_ex[tid] = tmin;
                            // the sphere pre-pass of the chunk left this ray's state in its hit record: an accepted sphere (the incumbent), the sphere it starts inside of, or
                            // "to the reference-order walk"
                            bool flagged = false;
                            if (ch.n_spheres != 0u) {
                                const float* recp = reinterpret_cast<const float*>(&out.hits[idx]);
                                st = __float_as_uint(__hip_atomic_load(recp + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                                flagged = (st >> 31) != 0u;
                                st &= 0x7fffffffu;
                                if (st & 1u) t_lim = __hip_atomic_load(recp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 2.0f * dt;
                            }
                            s_st[tid] = st;
                            if (flagged) {
                                to_fb = true;
                                active = false;
                                if (COUNT) n_why[1]++;
                            }
                        } else {
                            cur = kRefNone;
                            s_st[tid] = 0u;
                        }
                    }
                }
                pool_next += min(n_idle, avail);
            }
            TH_PHASE_END(0, n_idle);
            if (__ballot(active) == 0ull) {
                if (__ballot(to_fb) != 0ull) continue;  // flush first
                if (exhausted) break;
                continue;
            }
        }
        // ---- phase A: interior steps and pops; lanes holding a leaf wait (k_trace3's schedule).  A lane WITHOUT a node (both children failed and the stack top was dead, a leaf
        //      that left a dead top) takes part in the step's tail instead of a pop section of its own: the tail reads the stack top anyway (k_trace3's in-step pop) — one entry per
        //      round, dead ones dropped; what is left of the pop section is the delivery of the rays whose stack is empty ----
#pragma unroll 1
        for (int it = 0; it < TH_TRACE3C_MAX_A; ++it) {
#ifdef TH_DIAG_PHASES
            const unsigned long long ph_pop_m = __ballot(active && cur == kRefNone && sp == 0);
            const unsigned long long ph_t_pop = __builtin_readcyclecounter();
#endif
            if (active && cur == kRefNone && sp == 0) {  // the walk is over: a certified hit (stored when it was accepted) or a certified miss
                active = false;
                const uint32_t fst = s_st[tid] & 3u;  // bit 0: a hit is held; bit 1: the walk stored it (else it is the pre-pass's sphere record, whose third lane holds the state)
                if (fst == 0u) out.hits[s_idx[tid]] = make_float4(kInf, __int_as_float(-1), 0.0f, 0.0f);
                else if (fst == 1u) reinterpret_cast<float*>(&out.hits[s_idx[tid]])[2] = 0.0f;
            }
#ifdef TH_DIAG_PHASES
            {
                const unsigned long long now = __builtin_readcyclecounter();
                ph_cyc[1] += now - ph_t_pop;
                ph_lan[1] += (unsigned long long)__popcll(ph_pop_m);
                ph_cnt[1] += 1ull;
            }
            const unsigned long long ph_node_m = __ballot(active && (cur < kLeafBit || cur == kRefNone));
            const unsigned long long ph_t_node = __builtin_readcyclecounter();
#endif
            const bool stepping = active && cur < kLeafBit;
            if (stepping || (active && cur == kRefNone)) {
                // interior: one 64-byte burst, both child boxes (every lane of the section loads, a lane that only pops the root's node: no zero-initialised registers, th_trace3c4.h)
                const float4* np = ws.wnodes + 4 * (size_t)(stepping ? cur : 0u);
                const float4 a0 = np[0], a1 = np[1], a2 = np[2], a3 = np[3];
                uint32_t top_enc = kRefNone;
                float top_tm = kInf;
                if (sp > 0) {
                    if (sp - 1 < kLds) {
                        const uint2 e = s_stk[sp - 1][tid];
                        top_enc = e.x;
                        top_tm = __uint_as_float(e.y);
                    } else if (sp - 1 < kStack2Total) {
                        const uint2 e = overflow[(size_t)(sp - 1 - kLds) * gthreads + gtid];
                        top_enc = e.x;
                        top_tm = __uint_as_float(e.y);
                    }
                }
                const float t_cull = t_lim + mb;
                bool any_child = false;
                float ex_new = 0.0f;
                cur = kRefNone;
                if (stepping) {
                    if (COUNT) nn += 2;
                    const uint32_t lenc = __float_as_uint(a3.x), renc = __float_as_uint(a3.y), meta = __float_as_uint(a3.z);
                    float bl, br;    // per child: what is compared with t_cull — +Inf for a missed child
                    float vl, vr;    // … and the entry distance that travels with it (only read for a child that is entered)
                    bool neg;        // the second child first
                    const float gr = AXIS ? growth() : 0.0f;
                    float gl, gr_;
                    const bool hl = slab_test3<AXIS>(a0.x, a0.z, a1.x, a0.y, a0.w, a1.y, o, inv_d, em, gr, !(meta & 4u), negx, negy, negz, vl, gl);  // (the accelerator's node layout: each axis' two planes side by side)
                    const bool hr = slab_test3<AXIS>(a1.z, a2.x, a2.z, a1.w, a2.y, a2.w, o, inv_d, em, gr, !(meta & 8u), negx, negy, negz, vr, gr_);
                    // per child: its exact entry distance travels with it; what is compared with t_cull is that distance (with AXIS: the entry of the grown box); a missed child: +Inf.
                    // (Boxes on a sphere's path keep the reference's clauses alone — bits 2 / 3 — but are culled like any other: the spheres themselves were tested at the fetch.)
                    bl = hl ? (AXIS ? gl : vl) : kInf;
                    br = hr ? (AXIS ? gr_ : vr) : kInf;
                    const uint32_t axis = meta & 3u;
                    neg = axis == 0 ? negx : (axis == 1 ? negy : negz);
                    const float bn = neg ? br : bl, bf = neg ? bl : br;
                    const float vn = AXIS ? (neg ? vr : vl) : bn, vf = AXIS ? (neg ? vl : vr) : bf;  // (for a child that is entered the two are the same number unless AXIS)
                    const uint32_t nenc = neg ? renc : lenc, fenc = neg ? lenc : renc;
                    const bool go_n = bn < t_cull, go_f = bf < t_cull;
                    // t_max never goes up in THIS walk (a ray that could see it raised is flagged and leaves): an entry that fails now fails at pop time
                    if (go_n && go_f) {
                        if (sp < kLds) {
                            s_stk[sp][tid] = make_uint2(fenc, __float_as_uint(vf));
                        } else if (sp < kStack2Total) {
                            overflow[(size_t)(sp - kLds) * gthreads + gtid] = make_uint2(fenc, __float_as_uint(vf));
                        }
                        sp++;
                    }
                    any_child = go_n || go_f;
                    cur = any_child ? (go_n ? nenc : fenc) : kRefNone;
                    ex_new = go_n ? vn : vf;
                }
                if (!any_child && sp > 0) {  // nothing was pushed in this step: the top read above is still the top
                    sp--;
                    const float t_pop = AXIS ? __fmaf_rn(growth(), inv_max(), t_cull) : t_cull;
                    if (top_tm < t_pop && sp < kStack2Total) {
                        cur = top_enc;
                        ex_new = top_tm;
                    }
                }
                s_ex[tid] = ex_new;
            }
#ifdef TH_DIAG_PHASES
            ph_cyc[2] += __builtin_readcyclecounter() - ph_t_node;
            ph_lan[2] += (unsigned long long)__popcll(ph_node_m);
            ph_cnt[2] += 1ull;
#endif
            const uint32_t n_desc = (uint32_t)__popcll(__ballot(active && (cur < kLeafBit || cur == kRefNone)));
            if (n_desc <= (uint32_t)TH_TRACE3C_LEAF_WAIT) break;
        }
        // ---- phase B: leaves ----
#ifdef TH_DIAG_PHASES
        const unsigned long long ph_leaf_m = __ballot(active && cur >= kLeafBit && cur != kRefNone);
        const unsigned long long ph_t_leaf = __builtin_readcyclecounter();
#endif
        if (active && cur >= kLeafBit && cur != kRefNone) {
            bool flagged = false;
            const uint32_t leaf_ref = cur & 0x00ffffffu, leaf_cnt = cur >> 24;
            uint32_t top_enc = kRefNone;
            float top_tm = kInf;
            if (sp > 0) {
                if (sp - 1 < kLds) {
                    const uint2 e = s_stk[sp - 1][tid];
                    top_enc = e.x;
                    top_tm = __uint_as_float(e.y);
                } else if (sp - 1 < kStack2Total) {
                    const uint2 e = overflow[(size_t)(sp - 1 - kLds) * gthreads + gtid];
                    top_enc = e.x;
                    top_tm = __uint_as_float(e.y);
                }
            }
            for (uint32_t k = 0; k < leaf_cnt; ++k) {
                const uint32_t slot = leaf_ref + k;
                const float4 p0 = sc.prims[3 * slot];
                const float4 p1 = sc.prims[3 * slot + 1], p2 = sc.prims[3 * slot + 2];
                asm volatile("" ::"v"(p1.x), "v"(p1.y), "v"(p1.z), "v"(p1.w), "v"(p2.x), "v"(p2.y), "v"(p2.z), "v"(p2.w));  // one burst (th_trace2.h "one fetch per leaf")
                const uint32_t meta = __float_as_uint(p0.w);
                if (COUNT) np++;
                TriTest tt;
                // (spheres were tested when the ray was fetched)
                if (!(meta & (PRIM_SPHERE | PRIM_DEGENERATE)) && tri_intersect_sheared<true>(mk3(p0.x, p0.y, p0.z), mk3(p1.x, p1.y, p1.z), mk3(p2.x, p2.y, p2.z), o, shear, t_lim, &tt)) {
                    // a candidate below the relaxed limit.  A ray that started inside sphere s: only what the reference tests AFTER s counts — s overwrites the rest (header); the
                    // primitive's order word (the third record's .w lane) holds, per sphere, the split axis of the canonical node where their paths part and the child it is in
                    bool counts = true;
                    const uint32_t st = s_st[tid];
                    if (st >> 8) {
                        const uint32_t ow = __float_as_uint(p2.w) >> (3u * ((st >> 8) - 1u));
                        const uint32_t ax = ow & 3u;
                        const bool second = (ow & 4u) != 0u;  // the primitive sits in the second child there (axis 3: in the sphere's own leaf, behind it)
                        counts = ax == 3u ? second : (second != (ax == 0u ? negx : (ax == 1u ? negy : negz)));  // bvh.jl:239-246: the second child is visited first iff d[axis] < 0
                    }
                    if (counts) {
                        const float dt = margin_t();
                        // accepted iff it lies 2 dt below the incumbent (t_lim - 4 dt; the ray's own t_max at first) AND its leaf box lets the reference in by t + dt (the guard); a NaN fails
                        if (!(tt.t <= t_lim - 4.0f * dt) || !(s_ex[tid] <= tt.t + dt)) {
                            if (COUNT && !flagged) why = 2u;
                            flagged = true;
                        } else if (!flagged) {
                            t_lim = tt.t + 2.0f * dt;
                            s_st[tid] = st | 3u;
                            out.hits[s_idx[tid]] = make_float4(out.bary_mode ? tt.bary.z : tt.t, p1.w /* the canonical slot */, tt.bary.x, tt.bary.y);  // stored at once: a nearer candidate overwrites it
                        }
                    }
                }
            }
            cur = kRefNone;
            if (flagged) {  // the reference-order walk decides this ray
                active = false;
                to_fb = true;
                sp = 0;
                if (COUNT) n_why[why & 3u]++;
            } else if (sp > 0) {  // the next stack entry against the limit the leaf left
                sp--;
                const float t_pop = t_lim + (AXIS ? __fmaf_rn(growth(), inv_max(), mb) : mb);
                if (top_tm < t_pop && sp < kStack2Total) {
                    cur = top_enc;
                    s_ex[tid] = top_tm;
                }
            }
        }
#ifdef TH_DIAG_PHASES
        ph_cyc[3] += __builtin_readcyclecounter() - ph_t_leaf;
        ph_lan[3] += (unsigned long long)__popcll(ph_leaf_m);
        ph_cnt[3] += 1ull;
#endif
    }
#ifdef TH_DIAG_PHASES
    if (lane == 0)
        for (int k4 = 0; k4 < 4; ++k4) {
            atomicAdd(&g_phase[3 * k4], ph_cyc[k4]);
            atomicAdd(&g_phase[3 * k4 + 1], ph_lan[k4]);
            atomicAdd(&g_phase[3 * k4 + 2], ph_cnt[k4]);
        }
#endif
    if (ctr) {
        if (blockIdx.x == 0 && threadIdx.x == 0 && !q.no_total) atomicAdd(&ctr->closest_total, (unsigned long long)seg_total(sv));
        if (lane_id() == 0 && n_fb) atomicAdd(&ctr->fallback_total, (unsigned long long)n_fb);
        if (COUNT) {
            const unsigned long long sn = wave_sum(nn), spr = wave_sum(np);
            if (lane_id() == 0) {
                atomicAdd(&ctr->nodes_closest, sn);
                atomicAdd(&ctr->prims_closest, spr);
            }
            for (int k = 0; k < 4; ++k) {
                const unsigned long long w = wave_sum(n_why[k]);
                if (lane_id() == 0 && w) atomicAdd(&ctr->fallback_why[k], w);
            }
        }
    }
}
#undef negx
#undef negy
#undef negz

// ---- one-leaf accelerator (scenes of at most "tiny_scene_prims" primitives: S-cornell, the shadows scene) -------------------------------------------------
// The accelerator is the list of all primitives in canonical slot order, walked with a wave-uniform index (k_trace_leaf's scheme: scalar loads, no stack).
// Nothing is culled, so every candidate is seen; what remains of the certificate: a primitive counts only when the t_max-free clauses pass on the box of ITS
// canonical leaf (cs.slot_boxes), and the acceptance / flag rule of the header (a candidate below g + dt is accepted iff max(t, entry of its leaf box) <= t_max - dt).  Flagged rays go to the fallback list, which k_trace3 walks on the canonical tree.
#ifndef TH_TRACE_LEAF_C_WAVES
#define TH_TRACE_LEAF_C_WAVES TH_TRACE_LEAF_WAVES
#endif
#ifndef TH_LEAF_C_DEFER
#define TH_LEAF_C_DEFER 2  // 1: the canonical leaf boxes are tested once per ray, for the candidate it ends up holding; 0: for every candidate as it is found (round 4); 2: 1 with the spheres in
                           // a loop of their own and a branch-free triangle loop (round 6: S-cornell closest-hit 18.9 -> 17.65 ms per 64 spp, 17.5 -> 15.9 without the second stream; 60 VGPRs)
#endif
template <bool COUNT, bool FULL_ONLY>
__global__ __launch_bounds__(kBlock, FULL_ONLY ? TH_TRACE_LEAF_C_WAVES : 4) void k_trace_leaf_c(DeviceScene sc /* canonical records */, WideScene ws /* root box; root_ref / root_cnt = all slots */,
                                                                              CertScene cs, SegQueue q, const float4* __restrict__ ro, const float4* __restrict__ rd,
                                                                              const float* __restrict__ tmax_or_null, TraceOut out, Counters* ctr, FallbackList fb) {
    __shared__ SegView sv;
    seg_load(q, sv);
    const uint32_t total = sv.prefix[kSeg];
    const uint32_t first = ws.root_ref, cnt = ws.root_cnt;
    const uint32_t gtid = blockIdx.x * kBlock + threadIdx.x;
    uint32_t nn = 0, np = 0;
    unsigned long long n_fb = 0;
    uint32_t seg = 0;  // (carried over the iterations: seg_locate_from)
    for (uint32_t flat = gtid; flat < total; flat += gridDim.x * kBlock) {
        uint32_t lb;
        seg_locate_from(sv, flat & ~63u, seg, lb);
        const uint32_t local = lb + (flat & 63u);
        const bool valid = local < sv.count[seg];
        uint32_t idx = valid ? seg_phys(q, seg, local) : 0u;
        if (valid && q.indirect) idx = q.indirect[idx];
        float4 o4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), d4 = make_float4(0.0f, 0.0f, 1.0f, 0.0f);
        if (valid) {
            o4 = ro[idx];
            d4 = rd[idx];
        }
        const f3 o = mk3(o4.x, o4.y, o4.z), d = mk3(d4.x, d4.y, d4.z);
        const f3 inv_d = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        const bool negx = d.x < 0.0f, negy = d.y < 0.0f, negz = d.z < 0.0f;
        float t_max = (valid && tmax_or_null) ? tmax_or_null[idx] : kInf;
        const RayShear shear = ray_shear(d, inv_d);
        const float D = slab_margin(ws.root_box, 1.0f, o);
        const float dt = kCertDt * D * fabsf(shear.sz);
        float t_lim = t_max + 2.0f * dt;  // t_max + 2 dt (header): the relaxed limit of the primitive tests
        bool live = false, flagged = false;
        if (valid) {
            const bool plain = d.x != 0.0f && d.y != 0.0f && d.z != 0.0f && dt < kInf && fabsf(o.x) < kInf && fabsf(o.y) < kInf && fabsf(o.z) < kInf && fabsf(inv_d.x) < kInf && fabsf(inv_d.y) < kInf &&
                               fabsf(inv_d.z) < kInf && t_max == t_max;
            float tmin;
            if (COUNT) nn++;
            if (!plain)
                flagged = true;
            else  // the root box (bvh.jl:226): the same box in every tree
                live = slab_test2(ws.root_box[0], ws.root_box[1], ws.root_box[2], ws.root_box[3], ws.root_box[4], ws.root_box[5], o, inv_d, 0.0f, false, negx, negy, negz, tmin) && tmin < t_lim;
        }
        bool found = false, sticky = false;
#if TH_LEAF_C_DEFER == 2
        // LEAN (round 6): the spheres first, in a loop of their own (the scene's sphere list), then the triangles in a loop whose body is the test and five selects — the generic
        // candidate record (one path for sphere and triangle candidates) and its lane masks cost 1.44 x k_trace_leaf's instructions per primitive (profiles/r6).  The order in
        // which candidates are met is free (header: the acceptance rule flags what the order could decide).
        uint32_t best_slot = 0u;
        float4 best_r4 = make_float4(kInf, __int_as_float(-1), 0.0f, 0.0f);
#pragma unroll 1
        for (uint32_t ks = 0; ks < cs.n_spheres; ++ks) {
            if (__ballot(live) == 0ull) break;
            const uint32_t slot = uniform_load(cs.sphere_slots, ks);
            const float4 p0 = uniform_load(sc.prims, 3 * slot);
            const SphereRec sr = uniform_load(sc.spheres, __float_as_uint(p0.x));
            if (live) {
                if (COUNT) np++;
                float t_c = 0.0f;
                const int r = sphere_candidate_c<FULL_ONLY>(sr, o, d, t_lim, t_c);
                if (r == 2) {
                    flagged = true;  // a clipped sphere (or a NaN root) on the ray's line: the order decides
                } else if (r != 0) {
                    if (sticky || !(t_c <= t_max - 2.0f * dt)) {
                        flagged = true;
                    } else {
                        t_max = t_c;
                        t_lim = t_c + 2.0f * dt;
                        found = true;
                        sticky = r == 3;
                        best_slot = slot;
                        best_r4 = make_float4(t_c, __int_as_float((int)slot), 0.0f, 0.0f);
                    }
                }
                if (flagged) live = false;
            }
        }
#pragma unroll 1
        for (uint32_t k = 0; k < cnt; ++k) {
            if (__ballot(live) == 0ull) break;
            const uint32_t slot = first + k;  // wave-uniform: scalar loads
            const float4 p0 = uniform_load(sc.prims, 3 * slot);
            const uint32_t meta = __float_as_uint(p0.w);
            if (meta & (PRIM_SPHERE | PRIM_DEGENERATE)) {  // (uniform)
                if (COUNT && live && (meta & PRIM_DEGENERATE)) np++;
                continue;
            }
            const float4 p1 = uniform_load(sc.prims, 3 * slot + 1), p2 = uniform_load(sc.prims, 3 * slot + 2);
            if (COUNT && live) np++;
            TriTest tt;
            tt.t = 0.0f;
            tt.bary = mk3(0.0f, 0.0f, 0.0f);
            const bool hit = live && tri_intersect_sheared<true>(mk3(p0.x, p0.y, p0.z), mk3(p1.x, p1.y, p1.z), mk3(p2.x, p2.y, p2.z), o, shear, t_lim, &tt);
            // accepted iff it lies 2 dt below the incumbent and no sphere the ray started inside of holds the ray; a candidate that is not accepted flags the ray
            const bool acc = hit && !sticky && (tt.t <= t_max - 2.0f * dt);
            flagged = flagged || (hit && !acc);
            live = live && !(hit && !acc);
            found = found || acc;
            t_max = acc ? tt.t : t_max;
            t_lim = acc ? tt.t + 2.0f * dt : t_lim;
            best_slot = acc ? slot : best_slot;
            best_r4.x = acc ? (out.bary_mode ? tt.bary.z : tt.t) : best_r4.x;
            best_r4.y = acc ? __int_as_float((int)slot) : best_r4.y;
            best_r4.z = acc ? tt.bary.x : best_r4.z;
            best_r4.w = acc ? tt.bary.y : best_r4.w;
        }
        if (valid && found && !flagged) {  // the reference reaches the holder's leaf (bounds.jl:186-198 on its box, t_max aside) and enters it by t + dt (the guard); a sphere entered from inside: its box holds the origin
            const float* bx = cs.slot_boxes + 6 * (size_t)best_slot;
            float ex;
            if (COUNT) nn++;
            if (!slab_test2(bx[0], bx[1], bx[2], bx[3], bx[4], bx[5], o, inv_d, 0.0f, false, negx, negy, negz, ex) || !(ex <= (sticky ? 0.0f : t_max + dt))) flagged = true;
        }
        if (valid && found && !flagged) out.hits[idx] = best_r4;
#elif TH_LEAF_C_DEFER
        // The leaf boxes are looked at ONCE, for the candidate the ray ends up holding (header "one-leaf accelerator"): a candidate the reference cannot reach (its own test
        // passes, the t_max-free clauses on its leaf's box do not) may ride as the incumbent for a while — whatever it displaced or hid lies farther than what finally holds the
        // ray, or the final check sends the ray to the reference-order walk.
        uint32_t best_slot = 0u;
        float4 best_r4 = make_float4(kInf, __int_as_float(-1), 0.0f, 0.0f);  // the record the ray holds, stored once at the end (stored at every acceptance: 17.9 against 17.4 ms per 64 spp
                                                                             // on S-cornell).  Measured and dropped (profiles/r5/r5_trace3c4_experiments.txt): the leaf-box clauses in front
                                                                             // of every primitive (11.9 -> 4.7 tests per ray, 19.1 ms), the next slot's scalar loads issued a slot ahead (18.9 ms)
#pragma unroll 1
        for (uint32_t k = 0; k < cnt; ++k) {
            if (__ballot(live) == 0ull) break;
            const uint32_t slot = first + k;  // wave-uniform: scalar loads
            const float4 p0 = uniform_load(sc.prims, 3 * slot);
            const uint32_t meta = __float_as_uint(p0.w);
            float t_c = 0.0f;
            float4 r4 = make_float4(0.0f, __int_as_float((int)slot), 0.0f, 0.0f);
            bool cand = false, inside = false;
            if (meta & PRIM_SPHERE) {
                const SphereRec sr = uniform_load(sc.spheres, __float_as_uint(p0.x));
                if (live) {
                    if (COUNT) np++;
                    const int r = sphere_candidate_c<FULL_ONLY>(sr, o, d, t_lim, t_c);
                    if (r == 1 || r == 3) {
                        cand = true;
                        inside = r == 3;
                        r4.x = t_c;
                    } else if (r == 2) {
                        flagged = true;  // a clipped sphere (or a NaN root) on the ray's line: the order decides
                    }
                }
            } else if (!(meta & PRIM_DEGENERATE)) {
                const float4 p1 = uniform_load(sc.prims, 3 * slot + 1), p2 = uniform_load(sc.prims, 3 * slot + 2);
                if (live) {
                    if (COUNT) np++;
                    TriTest tt;
                    if (tri_intersect_sheared<true>(mk3(p0.x, p0.y, p0.z), mk3(p1.x, p1.y, p1.z), mk3(p2.x, p2.y, p2.z), o, shear, t_lim, &tt)) {
                        cand = true;
                        t_c = tt.t;
                        r4 = make_float4(out.bary_mode ? tt.bary.z : tt.t, __int_as_float((int)slot), tt.bary.x, tt.bary.y);
                    }
                }
            } else if (COUNT && live) {
                np++;
            }
            if (cand) {
                // accepted iff it lies 2 dt below the incumbent (the ray's own t_max at first) and no sphere the ray started inside of holds the ray (that one the reference
                // takes whatever t_max is: what it tests afterwards is the order's business)
                if (sticky || !(t_c <= t_max - 2.0f * dt)) {
                    flagged = true;
                } else {
                    t_max = t_c;
                    t_lim = t_c + 2.0f * dt;
                    found = true;
                    sticky = inside;
                    best_slot = slot;
                    best_r4 = r4;
                }
            }
            if (flagged) live = false;
        }
        if (valid && found && !flagged) {  // the reference reaches the holder's leaf (bounds.jl:186-198 on its box, t_max aside) and enters it by t + dt (the guard); a sphere entered from inside: its box holds the origin
            const float* bx = cs.slot_boxes + 6 * (size_t)best_slot;
            float ex;
            if (COUNT) nn++;
            if (!slab_test2(bx[0], bx[1], bx[2], bx[3], bx[4], bx[5], o, inv_d, 0.0f, false, negx, negy, negz, ex) || !(ex <= (sticky ? 0.0f : t_max + dt))) flagged = true;
        }
        if (valid && found && !flagged) out.hits[idx] = best_r4;
#else
#pragma unroll 1
        for (uint32_t k = 0; k < cnt; ++k) {
            if (__ballot(live) == 0ull) break;
            const uint32_t slot = first + k;  // wave-uniform: scalar loads
            const float4 p0 = uniform_load(sc.prims, 3 * slot);
            const uint32_t meta = __float_as_uint(p0.w);
            float t_c = 0.0f;
            float4 r4 = make_float4(0.0f, __int_as_float((int)slot), 0.0f, 0.0f);
            bool cand = false, inside = false, clipped = false;
            if (meta & PRIM_SPHERE) {
                const SphereRec sr = uniform_load(sc.spheres, __float_as_uint(p0.x));
                if (live) {
                    if (COUNT) np++;
                    const int r = sphere_candidate_c<FULL_ONLY>(sr, o, d, t_lim, t_c);
                    if (r == 1 || r == 3) {
                        cand = true;
                        inside = r == 3;
                        r4.x = t_c;
                    } else if (r == 2) {
                        clipped = true;  // — but only a sphere the reference can reach at all (its leaf's t_max-free clauses) makes the order matter
                    }
                }
            } else if (!(meta & PRIM_DEGENERATE)) {
                const float4 p1 = uniform_load(sc.prims, 3 * slot + 1), p2 = uniform_load(sc.prims, 3 * slot + 2);
                if (live) {
                    if (COUNT) np++;
                    TriTest tt;
                    if (tri_intersect_sheared<true>(mk3(p0.x, p0.y, p0.z), mk3(p1.x, p1.y, p1.z), mk3(p2.x, p2.y, p2.z), o, shear, t_lim, &tt)) {
                        cand = true;
                        t_c = tt.t;
                        r4 = make_float4(out.bary_mode ? tt.bary.z : tt.t, __int_as_float((int)slot), tt.bary.x, tt.bary.y);
                    }
                }
            } else if (COUNT && live) {
                np++;
            }
            if (__ballot(cand | clipped) != 0ull) {  // (a few primitives per ray: the box of the primitive's canonical leaf is fetched only now — wave-uniform, scalar loads)
                const float* bx = cs.slot_boxes + 6 * (size_t)slot;
                const float b0 = uniform_load(bx, 0), b1 = uniform_load(bx, 1), b2 = uniform_load(bx, 2), b3 = uniform_load(bx, 3), b4 = uniform_load(bx, 4), b5 = uniform_load(bx, 5);
                float ex;
                if (COUNT && cand) nn++;
                if ((cand | clipped) && slab_test2(b0, b1, b2, b3, b4, b5, o, inv_d, 0.0f, false, negx, negy, negz, ex)) {  // the reference reaches this primitive's leaf at all
                    if (clipped || sticky || !(t_c <= t_max - 2.0f * dt) || !(ex <= t_c + dt)) {
                        flagged = true;
                    } else {
                        t_max = t_c;
                        t_lim = t_c + 2.0f * dt;
                        found = true;
                        sticky = inside;
                        out.hits[idx] = r4;  // stored at once (a later accepted candidate overwrites it; a flagged ray's record is rewritten by the fallback walk)
                    }
                }
            }
            if (flagged) live = false;
        }
#endif
        const bool to_fb = valid && flagged;
        if (__ballot(to_fb) != 0ull) {
            uint32_t fseg = __builtin_amdgcn_readfirstlane((gtid >> 6) % kSeg);
            bool pending = to_fb;
            for (int tries = 0; tries < kSeg && __ballot(pending) != 0ull; ++tries) {
                const uint32_t j = wave_compact(pending, &fb.counts[fseg * kCtrStride]);
                if (pending && j < fb.cap) {
                    fb.list[(size_t)fseg * fb.cap + j] = idx;
                    n_fb++;
                    pending = false;
                }
                fseg = (fseg + 1) % kSeg;
            }
        }
        if (valid && !flagged && !found) out.hits[idx] = make_float4(kInf, __int_as_float(-1), 0.0f, 0.0f);
    }
    if (ctr) {
        if (blockIdx.x == 0 && threadIdx.x == 0 && !q.no_total) atomicAdd(&ctr->closest_total, (unsigned long long)seg_total(sv));
        const unsigned long long sfb = wave_sum(n_fb);
        if (lane_id() == 0 && sfb) atomicAdd(&ctr->fallback_total, sfb);
        if (COUNT) {
            const unsigned long long sn = wave_sum(nn), spr = wave_sum(np);
            if (lane_id() == 0) {
                atomicAdd(&ctr->nodes_closest, sn);
                atomicAdd(&ctr->prims_closest, spr);
            }
        }
    }
}

// ---- any-hit rays of a one-leaf accelerator -----------------------------------------------------------------------------------------------------------
// intersect_p(bvh, ray) is a boolean and the ray's t_max never changes during it (bvh.jl:260-299): a primitive stops the ray iff its canonical leaf's box
// passes bounds.jl:186-200 with that t_max (its ancestors then pass too: monotonic) and its own test accepts — the same in every tree, in any order.  So the
// leaf is walked in k_any_leaf's order (solid angle at the lights, two stages), a primitive that accepts the ray additionally has to pass its leaf's box, and
// only rays with a zero direction component (NaN products) go to the canonical tree.
template <bool COUNT, bool FULL_ONLY>
__global__ __launch_bounds__(kBlock, FULL_ONLY ? TH_TRACE_LEAF_WAVES : 4) void k_any_leaf_c(DeviceScene sc, WideScene ws, CertScene cs, SegQueue q, const float4* __restrict__ ro,
                                                                            const float4* __restrict__ rd, const float* __restrict__ tmax_or_null, TraceOut out, Counters* ctr,
                                                                            FallbackList fb) {
    __shared__ SegView sv;
    __shared__ uint32_t s_ring[kBlock / 64][128];
    seg_load(q, sv);
    const uint32_t total = sv.prefix[kSeg];
    const uint32_t first = ws.root_ref, cnt = ws.root_cnt;
    const uint32_t lane = lane_id(), wv = threadIdx.x >> 6;
    const uint32_t gtid = blockIdx.x * kBlock + threadIdx.x;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const uint32_t n_a = min(cnt, (uint32_t)TH_LEAF_STAGE_A);
    uint32_t nn = 0, np = 0;
    unsigned long long n_fb = 0;
    uint32_t ring_head = 0, ring_cnt = 0;  // wave-uniform
    auto run = [&](uint32_t idx, f3 o, f3 d, uint32_t k0, uint32_t k1, bool& live) {
        const float t_max = (live && tmax_or_null) ? tmax_or_null[idx] : kInf;
        const RayShear shear = ray_shear(d);
        const f3 inv_d = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        bool found = false;
#pragma unroll 1
        for (uint32_t k = k0; k < k1; ++k) {
            if (__ballot(live) == 0ull) break;
            const uint32_t slot = first + (ws.leaf_order ? uniform_load(ws.leaf_order, k) : k);  // wave-uniform: scalar loads
            const float4 p0 = uniform_load(sc.prims, 3 * slot);
            const uint32_t meta = __float_as_uint(p0.w);
            const float* bx = cs.slot_boxes + 6 * (size_t)slot;
            const float b0 = uniform_load(bx, 0), b1 = uniform_load(bx, 1), b2 = uniform_load(bx, 2), b3 = uniform_load(bx, 3), b4 = uniform_load(bx, 4), b5 = uniform_load(bx, 5);
            if (COUNT && lane == 0) np++;
            bool acc = false;
            if (meta & PRIM_SPHERE) {
                const SphereRec sr = uniform_load(sc.spheres, __float_as_uint(p0.x));
                if (live) {
                    SphereHit sh;
                    acc = sphere_intersect<false, FULL_ONLY>(sr, o, d, t_max, sh);
                }
            } else if (!(meta & PRIM_DEGENERATE)) {
                const float4 p1 = uniform_load(sc.prims, 3 * slot + 1), p2 = uniform_load(sc.prims, 3 * slot + 2);
                if (live) {
                    TriTest tt;
                    acc = tri_intersect_sheared<false>(mk3(p0.x, p0.y, p0.z), mk3(p1.x, p1.y, p1.z), mk3(p2.x, p2.y, p2.z), o, shear, t_max, &tt);
                }
            }
            if (acc) {  // … and the reference reaches the primitive's leaf (bounds.jl:186-200 on its box, the ray's own t_max)
                float tmin;
                if (slab_test2(b0, b1, b2, b3, b4, b5, o, inv_d, 0.0f, false, d.x < 0.0f, d.y < 0.0f, d.z < 0.0f, tmin) && tmin < t_max) {
                    found = true;
                    live = false;
                }
            }
        }
        return found;
    };
    auto deliver = [&](bool on, uint32_t idx, float4 o4, float4 d4, bool found) {
        if (!on) return;
        if (out.L) {
            const uint32_t slot = __float_as_uint(o4.w);
            if (!found) {
                const float4 c = out.contrib[idx];
                float4 l = out.L[slot];
                l.x += c.x;
                l.y += c.y;
                l.z += c.z;
                out.L[slot] = l;
            } else {
                const uint32_t poison = __float_as_uint(d4.w);
                if (poison) {
                    float4 l = out.L[slot];
                    const float nanv = __builtin_nanf("");
                    if (poison & 1u) l.x += nanv;
                    if (poison & 2u) l.y += nanv;
                    if (poison & 4u) l.z += nanv;
                    out.L[slot] = l;
                }
            }
        } else {
            out.occluded[idx] = found ? 1 : 0;
        }
    };
    auto stage_b = [&](uint32_t n) {
        const bool on = lane < n;
        const uint32_t idx = on ? s_ring[wv][(ring_head + lane) & 127u] : 0u;
        float4 o4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), d4 = make_float4(0.0f, 0.0f, 1.0f, 0.0f);
        if (on) {
            o4 = ro[idx];
            d4 = rd[idx];
        }
        bool live = on;
        const bool found = run(idx, mk3(o4.x, o4.y, o4.z), mk3(d4.x, d4.y, d4.z), n_a, cnt, live);
        deliver(on, idx, o4, d4, found);
    };
    uint32_t seg = 0;  // (carried over the iterations: seg_locate_from)
    for (uint32_t flat = gtid; flat < total; flat += gridDim.x * kBlock) {
        uint32_t lb;
        seg_locate_from(sv, flat & ~63u, seg, lb);
        const uint32_t local = lb + (flat & 63u);
        const bool valid = local < sv.count[seg];
        uint32_t idx = valid ? seg_phys(q, seg, local) : 0u;
        if (valid && q.indirect) idx = q.indirect[idx];
        float4 o4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), d4 = make_float4(0.0f, 0.0f, 1.0f, 0.0f);
        if (valid) {
            o4 = ro[idx];
            d4 = rd[idx];
        }
        const f3 o = mk3(o4.x, o4.y, o4.z), d = mk3(d4.x, d4.y, d4.z);
        // a zero direction component: NaN products, the monotonicity argument fails — the canonical tree decides
        const bool to_fb = valid && !(d.x != 0.0f && d.y != 0.0f && d.z != 0.0f);
        if (__ballot(to_fb) != 0ull) {
            uint32_t fseg = __builtin_amdgcn_readfirstlane((gtid >> 6) % kSeg);
            bool pending = to_fb;
            for (int tries = 0; tries < kSeg && __ballot(pending) != 0ull; ++tries) {
                const uint32_t j = wave_compact(pending, &fb.counts[fseg * kCtrStride]);
                if (pending && j < fb.cap) {
                    fb.list[(size_t)fseg * fb.cap + j] = idx;
                    n_fb++;
                    pending = false;
                }
                fseg = (fseg + 1) % kSeg;
            }
        }
        bool live = false;
        if (valid && !to_fb) {
            const float t_max = tmax_or_null ? tmax_or_null[idx] : kInf;
            const f3 inv_d = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
            float tmin;
            if (COUNT) nn++;
            live = slab_test2(ws.root_box[0], ws.root_box[1], ws.root_box[2], ws.root_box[3], ws.root_box[4], ws.root_box[5], o, inv_d, 0.0f, false, d.x < 0.0f, d.y < 0.0f, d.z < 0.0f, tmin) &&
                   tmin < t_max;
        }
        const bool found = run(idx, o, d, 0u, n_a, live);
        const bool park = live && n_a < cnt;
        deliver(valid && !to_fb && !park, idx, o4, d4, found);
        const unsigned long long m = __ballot(park);
        if (m) {
            if (park) s_ring[wv][(ring_head + ring_cnt + (uint32_t)__popcll(m & lt_mask)) & 127u] = idx;
            ring_cnt += (uint32_t)__popcll(m);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (ring_cnt >= 64u) {
                stage_b(64u);
                ring_head = (ring_head + 64u) & 127u;
                ring_cnt -= 64u;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
        }
    }
    if (ring_cnt) stage_b(ring_cnt);
    if (ctr) {
        // every ray of the queue is counted here, once: the launch that walks the fallback list is told not to (SegQueue::no_total)
        if (COUNT) {
            const unsigned long long sn = wave_sum(nn), spr = wave_sum(np);
            if (lane_id() == 0) {
                atomicAdd(&ctr->nodes_shadow, sn);
                atomicAdd(&ctr->prims_shadow, spr);
            }
        }
        if (blockIdx.x == 0 && threadIdx.x == 0 && !q.no_total) atomicAdd(&ctr->shadow_total, (unsigned long long)seg_total(sv));
    }
    (void)n_fb;
}

}  // namespace th
