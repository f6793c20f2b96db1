// th_trace2.h — traversal kernel v2: same results, bit for bit, as accel/bvh.jl:212-299 walked in the reference's order
// (k_trace_closest / k_trace_any in th_kernels.h are the literal form), restructured for gfx950:
//
//  * "children-in-parent" nodes (64 B, one 4 x dwordx4 burst per interior node): both child boxes + child refs + split
//    axis.  Every node's box is still tested with the reference's slab arithmetic (bounds.jl:186-206) exactly once; the
//    far child is tested when its parent is fetched instead of when it is popped, so its t_max-dependent clause
//    (`tx_min < ray.t_max`) is deferred: tx_min travels on the stack and is compared with the CURRENT t_max at pop time,
//    which is what the reference evaluates then (t_max can even grow: sphere.jl:137-138 / primitive.jl:17, A.8).
//    Leaves are not fetched at all (a leaf is a child ref = first primitive slot + count).  Dependent loads per ray
//    halve.
//  * persistent waves with per-lane ray replacement: a lane whose ray finished takes the next ray index from the wave's
//    chunk (kChunk indices per atomic on one of kSeg per-segment cursors; ballot + popcount ranks) instead of idling until
//    the slowest of its 63 neighbours is done — visit counts per ray are heavy-tailed (grazing rays over a height field
//    visit 100-1000x the median).
//  * stack entries {ref, tx_min}: 16 levels per lane in LDS as stack[level][lane] (conflict-free), deeper levels in a
//    global overflow slab laid out [level][thread] (coalesced); 64 levels in total like bvh.jl:222.
#pragma once
#include "th_kernels.h"

namespace th {

#ifndef TH_STACK2_LDS
#define TH_STACK2_LDS 16
#endif
#ifndef TH_TRACE2_MIN_WAVES
#define TH_TRACE2_MIN_WAVES 1
#endif
constexpr int kStack2Lds = TH_STACK2_LDS;  // stack levels per lane kept in LDS
constexpr int kStack2Total = 64;
// k_trace3 occupancy: waves per SIMD asked of the compiler and stack levels kept in LDS (levels x 2 KB per block of 256 lanes).
// Measured on the S-mesh frame (ms): closest-hit 4 waves / 16 levels 360, 5 / 12 (spills ~10 of its 106 VGPRs) 330, 6 / 10 397;
// any-hit (93 VGPRs) 4 / 16: 170, 5 / 12: 155, 6 / 10: 144.  End of round 2: the closest-hit variant has come down to 85 VGPRs (hits
// stored at once, the triangle permutation by selects, …), five short of six waves: 6 / 12 (80 VGPRs + 20 B scratch) 271 ms against
// 5 / 12 282 ms, 6 / 10 272, 7 / 8 284.
#ifndef TH_TRACE3_WAVES_CLOSEST
#define TH_TRACE3_WAVES_CLOSEST 6
#endif
#ifndef TH_TRACE3_LDS_CLOSEST
#define TH_TRACE3_LDS_CLOSEST 12
#endif
#ifndef TH_TRACE3_WAVES_ANY
#define TH_TRACE3_WAVES_ANY 6
#endif
#ifndef TH_TRACE3_LDS_ANY
#define TH_TRACE3_LDS_ANY 10
#endif
constexpr int kStack3MinLds = TH_TRACE3_LDS_CLOSEST < TH_TRACE3_LDS_ANY ? TH_TRACE3_LDS_CLOSEST : TH_TRACE3_LDS_ANY;
constexpr int kStackMinLds = kStack2Lds < kStack3MinLds ? kStack2Lds : kStack3MinLds;  // the global overflow slab holds the levels above this
// levels per thread of the global overflow slab every traversal launch gets (the EXPERIMENTS build's two-rays-per-lane kernel, th_trace4.h with 8 LDS levels, needs 2 x 56)
#ifdef TRHIP_EXPERIMENTS
constexpr int kStackSlabLevels = (kStack2Total - kStackMinLds) > 112 ? (kStack2Total - kStackMinLds) : 112;
#else
constexpr int kStackSlabLevels = kStack2Total - kStackMinLds;
#endif
constexpr uint32_t kRefNone = 0xffffffffu;
#ifndef TH_TRACE_REFILL
#define TH_TRACE_REFILL 12  // idle lanes of a wave that trigger a refill from the queue
#endif
constexpr int kChunk = 256;  // ray indices a wave takes from a segment cursor per atomic (== kSegGran)

struct WideScene {            // device view of the v2 node array
    const float4* wnodes;     // 4 float4 per interior node
    float root_box[6];        // bounds of flat node 0 (tested first, bvh.jl:226)
    uint32_t root_ref, root_cnt;  // root_cnt > 0: the root is a leaf with that many primitives starting at root_ref
    uint32_t n_wnodes;
    const float4* w4nodes;    // hybrid accelerator only: the same tree four children wide, 8 float4 per node (th_trace3c4.h); null = none
    uint32_t n_w4nodes;
    float tight_scale;        // slab_test2's margin as a fraction of the ray's reach (2^-14); 0: the reference's loose test alone
    const uint32_t* leaf_order;  // one-leaf scenes: the order in which any-hit rays test the leaf's primitives (k_any_leaf); null = slot order
    uint32_t leaf_tight;      // every triangle leaf's box is exactly the union of its triangles' boxes (k_trace7's cheap interior test needs it)
    float sphere_lag;         // k_trace7: 64 ulp / (smallest world-space sphere radius), so that lag_s = sphere_lag x D^2 x max |1 / d|; 0 without spheres
    // k_trace3's postponed leaves (TH_TRACE3_SPEC): only for rays whose t_max cannot go UP, i.e. whose origin lies outside every sphere's (grown) world bound (A.18)
    uint32_t spec_spheres;    // number of boxes below; 0xffffffff: postponement off (option "trace3_spec" = 0, or more spheres than boxes)
    float spec_box[8][6];
};

// The t_max-independent part of bounds.jl:186-206; returns false when the box is certainly missed, otherwise tx_min (to be
// compared with t_max by the caller: `tx_min < ray.t_max`).
//
// The reference's test is LOOSE: `ty_max > tx_max && (tx_max = ty_max)` (bounds.jl:190) keeps the LARGER of the x and y
// exits, so the earlier of the two never bounds the z entry and a box wholly behind the ray in x or y still passes.  That only
// costs visits — a box the ray does not pierce holds no primitive the ray can hit — but it costs a lot: a near-horizontal
// ray leaving a height field visits every box whose y range holds it (measured 8·10⁴ node fetches for single rays of the
// 1 M-triangle scene, the whole traversal tail).  The kernels here therefore AND the reference's test with the two clauses it
// lost, made conservative: z entry ≤ min(x exit, y exit) and min(x exit, y exit) ≥ 0, evaluated on the box GROWN by
// `em` = 2⁻¹⁴ · (largest |coordinate offset| between the ray origin and the scene bound), per axis in t units (slab_margin).
// Why that is result-neutral: the reference accepts a primitive only through its own float test (watertight triangle,
// triangle_mesh.jl:187-243; sphere quadratic, sphere.jl:120-150), whose accepted hit point lies within a few ulps of
// |v - o| ≤ D of the primitive (translate, shear and edge-function roundings: < 40 ε D), hence inside its leaf box and
// every ancestor box grown by 2⁻¹⁴ D = 1024 ε D; the slab arithmetic's own error (3 ε per t) is far below the same margin.
// So every box that holds an acceptable hit passes; boxes the reference rejects are still rejected (its clauses are all
// kept), and the visit ORDER among the boxes that remain is unchanged, so equal-t ties resolve as before (A.6).
// Spheres are the exception: the fp32 quadratic (sphere.jl:120-150) carries an absolute error ~ 8 ε |o - c|² in its discriminant,
// so the reference "hits" spheres the ray passes at up to ~ |o - c| sqrt(8 ε) ≈ 10⁻³ |o - c|; boxes on the path to a sphere
// (bit 18 / 19 of the node's packed word, set at upload) keep the reference's test alone (`tight` = false), as does the root.
// NaN (0 · Inf on a face-grazing axis-parallel ray) never rejects: the added comparisons are false on NaN, as the reference's.
// The literal kernels (traversal 1, th_kernels.h) keep the reference's test alone and serve as the on-device A/B.
TH_D float slab_margin(const float* __restrict__ root_box, float scale, f3 o) {
    const float D = fmaxf(fmaxf(fmaxf(fabsf(root_box[0] - o.x), fabsf(root_box[3] - o.x)), fmaxf(fabsf(root_box[1] - o.y), fabsf(root_box[4] - o.y))),
                          fmaxf(fabsf(root_box[2] - o.z), fabsf(root_box[5] - o.z)));
    return D * scale;  // in length units; slab_test2 turns it into t units per axis (x |1 / d|)
}
TH_D bool slab_test2(float bx0, float by0, float bz0, float bx1, float by1, float bz1, f3 o, f3 inv_d, float em, bool tight, bool negx, bool negy, bool negz, float& tmin_out) {
    // branch-free on purpose: every value is computed, the clauses are combined with `|` — the whole 64-byte node is then
    // loaded at once (with early returns the compiler sinks the z loads behind the x-y clause: a second dependent round trip)
    const float tx_min = ((negx ? bx1 : bx0) - o.x) * inv_d.x;
    const float tx_max = ((negx ? bx0 : bx1) - o.x) * inv_d.x;
    const float ty_min = ((negy ? by1 : by0) - o.y) * inv_d.y;
    const float ty_max = ((negy ? by0 : by1) - o.y) * inv_d.y;
    const float tz_min = ((negz ? bz1 : bz0) - o.z) * inv_d.z;
    const float tz_max = ((negz ? bz0 : bz1) - o.z) * inv_d.z;
    const bool miss_xy = (tx_min > ty_max) | (ty_min > tx_max);     // bounds.jl:188
    const float a = ty_min > tx_min ? ty_min : tx_min;              // :189
    const float b = ty_max > tx_max ? ty_max : tx_max;              // :190 (the larger exit: see above)
    const bool miss_z = (a > tz_max) | (tz_min > b);                // :194
    const float t_in = tz_min > a ? tz_min : a;                     // :196
    const float t_out = tz_max < b ? tz_max : b;                    // :197
    // margins by explicit fma (one instruction each; their rounding is immaterial); fminf drops a NaN operand: no constraint from that axis
    const float exit_xy = fminf(__fmaf_rn(em, fabsf(inv_d.x), tx_max), __fmaf_rn(em, fabsf(inv_d.y), ty_max));
    const bool miss_tight = tight & ((__fmaf_rn(-em, fabsf(inv_d.z), tz_min) > exit_xy) | (exit_xy < 0.0f));
    tmin_out = t_in;
    return !(miss_xy | miss_z | miss_tight) & (t_out > 0.0f);       // :198 without its t_max clause (the caller's)
}

struct TraceOut {
    float4* hits;            // closest: {t, prim, b1, b2}
    float4* L;               // any-hit + accumulate mode
    const float4* contrib;
    uint8_t* occluded;       // any-hit, plain mode
    uint32_t bary_mode;      // closest, 1: hits.x = third barycentric of a triangle hit instead of t (the shading kernels then
                             // skip re-running the triangle test); spheres keep t
    uint32_t far_hint;       // closest, hybrid mode: 1 = the rays of this launch start far outside the scene (camera rays): k_trace3c's AXIS variant (th_trace3c.h)
    // any-hit, hybrid mode (k_trace3 only): intersect_p is a boolean with a fixed t_max, so a ray WITHOUT a zero direction component gets the same answer on every tree
    // whose leaves carry the canonical leaf boxes (monotonic slab products) — those rays walk the accelerator tree (zero_mode 1: rays with a zero component are skipped
    // and *zero_flag is raised), the others (0 x Inf = NaN in the slab products: tree-dependent) the canonical tree in a second launch (zero_mode 2: only they; the
    // launch returns at once when the flag is down)
    uint32_t zero_mode;
    uint32_t* zero_flag;
    uint32_t any_acc_hint;   // any-hit, hybrid mode, option "any_on_accelerator" = -1 (default): 1 = this integrator's shadow rays are faster on the library's tree (SPPM's camera pass:
                             // 16.0 -> 10.9 ms per 100 iterations of C4; the path tracer's on S-mesh / S-blob measure 6-12 % SLOWER there and keep the canonical tree)
};

// ---- streaming wavefront (DESIGN.md "Stragglers") ---------------------------------------------------------------------------
// Node visits per ray are heavy-tailed; a launch that waits for its slowest ray pays a 10^5-fetch chain alone.  With STREAM
// a ray that exceeds the round's fetch budget is SUSPENDED: its traversal state (current node, stack, best hit so far) and
// what the path needs to go on are copied to a list; the next round's launch resumes the list first, next to its fresh rays
// (longest jobs first), so a straggler costs a lane for a few rounds instead of the whole GPU for its tail.  Resuming
// restores the exact state, so the result of a ray does not depend on where it was cut.
struct SuspendList {   // SoA over `cap` entries
    float4* o;         // closest: ray o | slot        any: o | term index into L
    float4* d;         // closest: ray d | key lo      any: d | poison bits
    float4* b;         // closest: β | key hi          any: contribution β·Ld
    float4* trav;      // t_max, hx, b1, b2
    uint4* st;         // cur, cur_cnt, sp, found << 31 | (hit_prim + 1)
    uint32_t* depth;   // closest: the path's depth tag
    uint2* stack;      // [kStack2Total][cap]: {ref | cnt << 24, tx_min}
    uint32_t cap;
};
struct StreamCtl {
    SuspendList in, out;        // in: suspended in the previous round, resumed first; out: receives this round's suspensions
    const uint32_t* in_count;   // entries of `in`
    uint32_t* in_cursor;        // claim cursor over `in` (zeroed before the launch)
    uint32_t* out_count;        // entries of `out` (zeroed before the launch)
    uint32_t budget_min;        // a ray is suspended after max(budget_min, fresh rays >> budget_shift) interior fetches; 0 = never (drain round)
    uint32_t budget_shift;
    // closest: β and depth tags of the fresh rays (copied when one is suspended), and where a resumed ray that finished
    // re-joins the round: appended to the live queue behind the fresh entries (the launch fetches through frozen counts)
    const float4* beta_in;
    const uint32_t* depth_in;
    float4 *app_o, *app_d, *app_b, *app_hits;
    uint32_t* app_depth;
    uint32_t* app_counts;  // live fill counters of the queue (row of Counters::n_queue)
    uint32_t app_cap;
};

template <bool ANY, bool COUNT, bool STREAM = false>
__global__ __launch_bounds__(kBlock, TH_TRACE2_MIN_WAVES) void k_trace2(DeviceScene sc, WideScene ws, SegQueue q, const float4* __restrict__ ro, const float4* __restrict__ rd, const float* __restrict__ tmax_or_null,
                                                   TraceOut out, uint32_t* __restrict__ work /* kSeg cursors, zeroed */, uint2* __restrict__ overflow, Counters* ctr, uint32_t debug_budget,
                                                   StreamCtl sx = StreamCtl{}) {
    __shared__ uint32_t s_ref[kStack2Lds][kBlock];
    __shared__ float s_tmin[kStack2Lds][kBlock];
    __shared__ SegView sv;
    seg_load(q, sv);
    const uint32_t tid = threadIdx.x;
    const uint32_t gthreads = gridDim.x * kBlock;
    const uint32_t gtid = blockIdx.x * kBlock + tid;
    const uint32_t lane = lane_id();
    const unsigned long long lt_mask = (1ull << lane) - 1ull;

    bool active = false, exhausted = false;
    // wave-uniform: the segment this wave is draining, consecutive drained segments seen, the chunk it currently owns
    uint32_t wseg = (gtid >> 6) % kSeg, dry = 0, pool_next = 0, pool_end = 0;
    uint32_t idx = 0, cur = kRefNone, cur_cnt = 0;
    int sp = 0;
    uint32_t steps = 0;  // interior fetches of the current ray (diagnostic budget)
    f3 o = splat3(0.0f), d = splat3(0.0f), inv_d = splat3(0.0f);
    float em = 0.0f;
    const bool tight_on = ws.tight_scale > 0.0f;
    RayShear shear{0, 0.0f, 0.0f, 0.0f};  // the triangle test's per-ray part (th_device.h)
    bool negx = false, negy = false, negz = false;
    float t_max = 0.0f, b1 = 0.0f, b2 = 0.0f, hx = 0.0f, slot_w = 0.0f, flag_w = 0.0f;
    int hit_prim = -1;
    bool found = false;
    uint32_t nn = 0, np = 0;
    // STREAM: resume bookkeeping
    bool resume_done = !STREAM, resumed = false, no_suspend = false;
    uint32_t rj = 0;
    const uint32_t n_in = STREAM ? min(*sx.in_count, sx.in.cap) : 0u;
    const uint32_t budget = STREAM && sx.budget_min ? max(sx.budget_min, sv.prefix[kSeg] >> sx.budget_shift) : 0u;

    while (true) {
        // ---- refill idle lanes ------------------------------------------------------------------------------------------------
        const unsigned long long idle = __ballot(!active);
        const uint32_t n_idle = (uint32_t)__popcll(idle);
        if (STREAM && !resume_done && n_idle > 0u) {  // suspended rays of the previous round first
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(sx.in_cursor, n_idle);
            base = __shfl(base, 0);
            if (base >= n_in) {
                resume_done = true;
            } else {
                if (!active) {
                    const uint32_t j = base + (uint32_t)__popcll(idle & lt_mask);
                    if (j < n_in) {
                        const float4 o4 = sx.in.o[j], d4 = sx.in.d[j], tv = sx.in.trav[j];
                        const uint4 st = sx.in.st[j];
                        o = mk3(o4.x, o4.y, o4.z);
                        d = mk3(d4.x, d4.y, d4.z);
                        slot_w = o4.w;
                        flag_w = d4.w;
                        inv_d = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                        em = slab_margin(ws.root_box, ws.tight_scale, o);
                        shear = ray_shear(d);
                        negx = d.x < 0.0f;
                        negy = d.y < 0.0f;
                        negz = d.z < 0.0f;
                        t_max = tv.x;
                        hx = tv.y;
                        b1 = tv.z;
                        b2 = tv.w;
                        cur = st.x;
                        cur_cnt = st.y;
                        sp = (int)st.z;
                        found = (st.w >> 31) != 0u;
                        hit_prim = (int)(st.w & 0x7fffffffu) - 1;
                        for (int lvl = 0; lvl < sp && lvl < kStack2Total; ++lvl) {
                            const uint2 e = sx.in.stack[(size_t)lvl * sx.in.cap + j];
                            if (lvl < kStack2Lds) {
                                s_ref[lvl][tid] = e.x;
                                s_tmin[lvl][tid] = __uint_as_float(e.y);
                            } else {
                                overflow[(size_t)(lvl - kStack2Lds) * gthreads + gtid] = e;
                            }
                        }
                        steps = 0;
                        resumed = true;
                        no_suspend = false;
                        rj = j;
                        active = true;
                    }
                }
                continue;  // re-evaluate the idle lanes
            }
        }
        if (n_idle == 64u || (!exhausted && n_idle >= (uint32_t)TH_TRACE_REFILL)) {
            if (!exhausted) {
                if (pool_next >= pool_end) {  // take the next chunk: try this wave's segment, move on when it is drained
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(&work[wseg * kCtrStride], (uint32_t)kChunk);
                    base = __shfl(base, 0);
                    const uint32_t cnt = sv.count[wseg];
                    if (base < cnt) {
                        pool_next = base;
                        pool_end = min(base + (uint32_t)kChunk, cnt);
                        dry = 0;
                    } else {
                        pool_next = pool_end = 0;
                        wseg = (wseg + 1) % kSeg;
                        if (++dry >= (uint32_t)kSeg) exhausted = true;
                    }
                }
                const uint32_t avail = pool_end - pool_next;
                if (avail && !active) {
                    const uint32_t rank = (uint32_t)__popcll(idle & lt_mask);
                    if (rank < avail) {
                        idx = seg_phys(q, wseg, pool_next + rank);
                        const float4 o4 = ro[idx], d4 = rd[idx];
                        o = mk3(o4.x, o4.y, o4.z);
                        d = mk3(d4.x, d4.y, d4.z);
                        slot_w = o4.w;
                        flag_w = d4.w;
                        inv_d = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                        em = slab_margin(ws.root_box, ws.tight_scale, o);
                        shear = ray_shear(d);
                        negx = d.x < 0.0f;
                        negy = d.y < 0.0f;
                        negz = d.z < 0.0f;
                        t_max = tmax_or_null ? tmax_or_null[idx] : kInf;
                        sp = 0;
                        steps = 0;
                        found = false;
                        hit_prim = -1;
                        b1 = b2 = 0.0f;
                        resumed = false;
                        no_suspend = false;
                        active = true;
                        float tmin;
                        if (COUNT) nn++;
                        if (ws.root_ref != kRefNone && slab_test2(ws.root_box[0], ws.root_box[1], ws.root_box[2], ws.root_box[3], ws.root_box[4], ws.root_box[5], o, inv_d, em, false, negx, negy, negz, tmin) &&
                            tmin < t_max) {
                            cur = ws.root_ref;
                            cur_cnt = ws.root_cnt;
                        } else {
                            cur = kRefNone;
                        }
                    }
                }
                pool_next += min(n_idle, avail);
            }
            if (__ballot(active) == 0ull) {
                if (exhausted && resume_done) break;
                continue;  // nothing fetched yet (chunk ran dry / segment drained): try again
            }
        }
        // ---- a few traversal steps ------------------------------------------------------------------------------------------
#pragma unroll 1
        for (int rep = 0; rep < 4; ++rep) {
            if (!active) continue;
            bool pop = true;
            bool finished = false;
            if (cur == kRefNone) {
                pop = true;
            } else if (cur_cnt > 0) {
                // leaf: primitives in slot order, later equal-t hit wins (bvh.jl:229-237, triangle_mesh.jl:211-214)
                for (uint32_t k = 0; k < cur_cnt; ++k) {
                    const uint32_t slot = cur + k;
                    const float4 p0 = sc.prims[3 * slot];
                    const float4 p1 = sc.prims[3 * slot + 1], p2 = sc.prims[3 * slot + 2];
                    const uint32_t meta = __float_as_uint(p0.w);
                    if (COUNT) np++;
                    if (meta & PRIM_SPHERE) {
                        SphereHit sh;
                        if (sphere_intersect<false>(sc.spheres[__float_as_uint(p0.x)], o, d, t_max, sh)) {
                            if (ANY) {
                                found = true;
                                finished = true;
                                break;
                            }
                            t_max = sh.t;
                            found = true;
                            hit_prim = (int)slot;
                            b1 = b2 = 0.0f;
                            hx = sh.t;
                        }
                    } else {
                        TriTest tt;
                        if (!(meta & PRIM_DEGENERATE) && tri_intersect_sheared<!ANY>(mk3(p0.x, p0.y, p0.z), mk3(p1.x, p1.y, p1.z), mk3(p2.x, p2.y, p2.z), o, shear, t_max, &tt)) {
                            if (ANY) {
                                found = true;
                                finished = true;
                                break;
                            }
                            t_max = tt.t;
                            found = true;
                            hit_prim = (int)slot;
                            b1 = tt.bary.x;
                            b2 = tt.bary.y;
                            hx = out.bary_mode ? tt.bary.z : tt.t;
                        }
                    }
                }
            } else {
                // interior: one 64-byte burst, both child boxes
                const float4 a0 = ws.wnodes[4 * (size_t)cur], a1 = ws.wnodes[4 * (size_t)cur + 1], a2 = ws.wnodes[4 * (size_t)cur + 2], a3 = ws.wnodes[4 * (size_t)cur + 3];
                if (COUNT) nn += 2;  // two node boxes tested (the reference would visit these two nodes)
                if (debug_budget && ++steps > debug_budget) {  // DIAGNOSTIC ONLY (option "debug_trace_budget"): abandon the ray -> wrong result
                    sp = 0;
                    cur = kRefNone;
                    continue;
                }
                if (STREAM && budget && !no_suspend && ++steps > budget) {  // suspend: the next round goes on from exactly here
                    const uint32_t j = atomicAdd(sx.out_count, 1u);
                    if (j < sx.out.cap) {
                        sx.out.o[j] = make_float4(o.x, o.y, o.z, slot_w);
                        sx.out.d[j] = make_float4(d.x, d.y, d.z, flag_w);
                        if (ANY) {
                            sx.out.b[j] = resumed ? sx.in.b[rj] : out.contrib[idx];
                        } else {
                            sx.out.b[j] = resumed ? sx.in.b[rj] : sx.beta_in[idx];
                            sx.out.depth[j] = resumed ? sx.in.depth[rj] : sx.depth_in[idx];
                            if (!resumed) out.hits[idx] = make_float4(0.0f, __int_as_float(-2), 0.0f, 0.0f);  // pending: the shading kernel skips it
                        }
                        sx.out.trav[j] = make_float4(t_max, hx, b1, b2);
                        sx.out.st[j] = make_uint4(cur, cur_cnt, (uint32_t)sp, (found ? 0x80000000u : 0u) | (uint32_t)(hit_prim + 1));
                        for (int lvl = 0; lvl < sp && lvl < kStack2Total; ++lvl) {
                            uint2 e;
                            if (lvl < kStack2Lds)
                                e = make_uint2(s_ref[lvl][tid], __float_as_uint(s_tmin[lvl][tid]));
                            else
                                e = overflow[(size_t)(lvl - kStack2Lds) * gthreads + gtid];
                            sx.out.stack[(size_t)lvl * sx.out.cap + j] = e;
                        }
                        active = false;
                        continue;
                    }
                    no_suspend = true;  // the list is full: this ray runs to its end here
                }
                float tl, tr;
                const bool hl = slab_test2(a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, o, inv_d, em, tight_on && !(__float_as_uint(a3.z) & 4u), negx, negy, negz, tl);
                const bool hr = slab_test2(a1.z, a1.w, a2.x, a2.y, a2.z, a2.w, o, inv_d, em, tight_on && !(__float_as_uint(a3.z) & 8u), negx, negy, negz, tr);
                const uint32_t lenc = __float_as_uint(a3.x), renc = __float_as_uint(a3.y), axis = __float_as_uint(a3.z) & 3u;
                const uint32_t lref = lenc & 0x00ffffffu, rref = renc & 0x00ffffffu, lcnt = lenc >> 24, rcnt = renc >> 24;
                const bool neg = axis == 0 ? negx : (axis == 1 ? negy : negz);  // bvh.jl:239: dir_is_neg[split_axis] == 2 -> second child first
                const bool hn = neg ? hr : hl, hf = neg ? hl : hr;
                const float tn = neg ? tr : tl, tf = neg ? tl : tr;
                const uint32_t nref = neg ? rref : lref, fref = neg ? lref : rref, ncnt = neg ? rcnt : lcnt, fcnt = neg ? lcnt : rcnt;
                const bool near_ok = hn && tn < t_max;
                if (near_ok) {
                    if (hf) {  // push the far child with its tx_min; the t_max clause is re-evaluated at pop time
                        const uint32_t enc = fref | (fcnt << 24);
                        if (sp < kStack2Lds) {
                            s_ref[sp][tid] = enc;
                            s_tmin[sp][tid] = tf;
                        } else if (sp < kStack2Total) {
                            overflow[(size_t)(sp - kStack2Lds) * gthreads + gtid] = make_uint2(enc, __float_as_uint(tf));
                        }
                        sp++;
                    }
                    cur = nref;
                    cur_cnt = ncnt;
                    pop = false;
                } else if (hf && tf < t_max) {  // near child missed: the reference pops the far child next, with the same t_max
                    cur = fref;
                    cur_cnt = fcnt;
                    pop = false;
                }
            }
            if (pop && !finished) {
                finished = true;
                while (sp > 0) {
                    sp--;
                    uint32_t enc;
                    float tm;
                    if (sp < kStack2Lds) {
                        enc = s_ref[sp][tid];
                        tm = s_tmin[sp][tid];
                    } else if (sp < kStack2Total) {
                        const uint2 e = overflow[(size_t)(sp - kStack2Lds) * gthreads + gtid];
                        enc = e.x;
                        tm = __uint_as_float(e.y);
                    } else {
                        continue;  // beyond 64 levels the reference throws (bvh.jl:222); entries were dropped
                    }
                    if (tm < t_max) {
                        cur = enc & 0x00ffffffu;
                        cur_cnt = enc >> 24;
                        finished = false;
                        break;
                    }
                }
            }
            if (finished) {
                active = false;
                if (ANY) {
                    if (out.L) {
                        const uint32_t slot = __float_as_uint(slot_w);
                        if (!found) {
                            const float4 c = (STREAM && resumed) ? sx.in.b[rj] : out.contrib[idx];
                            float4 l = out.L[slot];
                            l.x += c.x;
                            l.y += c.y;
                            l.z += c.z;
                            out.L[slot] = l;
                        } else {
                            const uint32_t poison = __float_as_uint(flag_w);
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
                } else if (STREAM && resumed) {  // re-join the round: append path + hit behind the fresh entries of the live queue
                    const uint32_t seg = rj % (uint32_t)kSeg;
                    const uint32_t k = atomicAdd(&sx.app_counts[seg * kCtrStride], 1u);
                    if (k < sx.app_cap) {  // cannot fail: every segment keeps list-capacity / kSeg + 1 spare entries
                        const uint32_t a = seg * sx.app_cap + k;
                        sx.app_o[a] = make_float4(o.x, o.y, o.z, slot_w);
                        sx.app_d[a] = make_float4(d.x, d.y, d.z, flag_w);
                        sx.app_b[a] = sx.in.b[rj];
                        sx.app_depth[a] = sx.in.depth[rj];
                        sx.app_hits[a] = make_float4(found ? hx : kInf, __int_as_float(found ? hit_prim : -1), b1, b2);
                    }
                } else {
                    out.hits[idx] = make_float4(found ? hx : kInf, __int_as_float(found ? hit_prim : -1), b1, b2);
                }
            }
        }
    }
    if (ctr) {
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(ANY ? &ctr->shadow_total : &ctr->closest_total, (unsigned long long)seg_total(sv));
        if (COUNT) {
            const unsigned long long sn = wave_sum(nn), spr = wave_sum(np);
            if (lane_id() == 0) {
                atomicAdd(ANY ? &ctr->nodes_shadow : &ctr->nodes_closest, sn);
                atomicAdd(ANY ? &ctr->prims_shadow : &ctr->prims_closest, spr);
            }
        }
    }
}

// ---- k_trace_leaf: scenes committed as ONE leaf (th_bvh.h, tiny_scene_prims: S-cornell, the shadows scene) ---------------------
// Every ray tests the same primitives in the same order, so there is nothing to schedule: a wave takes 64 queue entries, tests
// the root box (bvh.jl:226) and walks the leaf with a wave-uniform index — the primitive and sphere records arrive through
// scalar loads, the triangle / sphere branch is uniform, no stack, no per-lane replacement, ~60 VGPRs.  Same operations per ray
// in the same order as k_trace2 on this scene (which it replaces there): results are bit-identical (parity tests, traversal 2/3
// vs 1).  Any-hit lanes stop at their first accepted primitive (intersect_p returns, bvh.jl:283-287); the wave ends when all have.
#ifndef TH_TRACE_LEAF_WAVES
#define TH_TRACE_LEAF_WAVES 5
#endif
template <bool ANY, bool COUNT, bool FULL_ONLY>
__global__ __launch_bounds__(kBlock, FULL_ONLY ? TH_TRACE_LEAF_WAVES : 4) void k_trace_leaf(DeviceScene sc, WideScene ws, SegQueue q, const float4* __restrict__ ro, const float4* __restrict__ rd,
                                                                            const float* __restrict__ tmax_or_null, TraceOut out, Counters* ctr) {
    __shared__ SegView sv;
    seg_load(q, sv);
    const uint32_t total = sv.prefix[kSeg];
    const uint32_t first = ws.root_ref, cnt = ws.root_cnt;
    uint32_t nn = 0, np = 0;
    uint32_t seg = 0;  // (carried over the iterations: seg_locate_from)
    for (uint32_t flat = blockIdx.x * kBlock + threadIdx.x; flat < total; flat += gridDim.x * kBlock) {
        uint32_t lb;
        seg_locate_from(sv, flat & ~63u, seg, lb);
        const uint32_t local = lb + (flat & 63u);
        const bool valid = local < sv.count[seg];
        const uint32_t idx = valid ? seg_phys(q, seg, local) : 0u;
        float4 o4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), d4 = make_float4(0.0f, 0.0f, 1.0f, 0.0f);
        if (valid) {
            o4 = ro[idx];
            d4 = rd[idx];
        }
        const f3 o = mk3(o4.x, o4.y, o4.z), d = mk3(d4.x, d4.y, d4.z);
        float t_max = (valid && tmax_or_null) ? tmax_or_null[idx] : kInf;
        bool live = false;  // still has primitives to test
        if (valid) {
            const f3 inv_d = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
            float tmin;
            if (COUNT) nn++;
            live = slab_test2(ws.root_box[0], ws.root_box[1], ws.root_box[2], ws.root_box[3], ws.root_box[4], ws.root_box[5], o, inv_d, 0.0f, false, d.x < 0.0f, d.y < 0.0f, d.z < 0.0f, tmin) &&
                   tmin < t_max;
        }
        const RayShear shear = ray_shear(d);
        bool found = false;
        int hit_prim = -1;
        float hx = 0.0f, b1 = 0.0f, b2 = 0.0f;
#pragma unroll 1
        for (uint32_t k = 0; k < cnt; ++k) {
            if (__ballot(live) == 0ull) break;
            const uint32_t slot = first + k;  // wave-uniform: scalar loads
            const float4 p0 = uniform_load(sc.prims, 3 * slot);
            const uint32_t meta = __float_as_uint(p0.w);
            if (meta & PRIM_SPHERE) {
                const SphereRec sr = uniform_load(sc.spheres, __float_as_uint(p0.x));
                if (live) {
                    if (COUNT) np++;
                    SphereHit sh;
                    if (sphere_intersect<false, FULL_ONLY>(sr, o, d, t_max, sh)) {
                        found = true;
                        if (ANY) {
                            live = false;
                        } else {
                            t_max = sh.t;
                            hit_prim = (int)slot;
                            b1 = b2 = 0.0f;
                            hx = sh.t;
                        }
                    }
                }
            } else if (!(meta & PRIM_DEGENERATE)) {
                const float4 p1 = uniform_load(sc.prims, 3 * slot + 1), p2 = uniform_load(sc.prims, 3 * slot + 2);
                if (live) {
                    if (COUNT) np++;
                    TriTest tt;
                    if (tri_intersect_sheared<!ANY>(mk3(p0.x, p0.y, p0.z), mk3(p1.x, p1.y, p1.z), mk3(p2.x, p2.y, p2.z), o, shear, t_max, &tt)) {
                        found = true;
                        if (ANY) {
                            live = false;
                        } else {
                            t_max = tt.t;
                            hit_prim = (int)slot;
                            b1 = tt.bary.x;
                            b2 = tt.bary.y;
                            hx = out.bary_mode ? tt.bary.z : tt.t;
                        }
                    }
                }
            } else if (COUNT && live) {
                np++;  // k_trace2 counts the degenerate triangle it skips
            }
        }
        if (!valid) continue;
        if (ANY) {
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
        } else {
            out.hits[idx] = make_float4(found ? hx : kInf, __int_as_float(found ? hit_prim : -1), b1, b2);
        }
    }
    if (ctr) {
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(ANY ? &ctr->shadow_total : &ctr->closest_total, (unsigned long long)seg_total(sv));
        if (COUNT) {
            const unsigned long long sn = wave_sum(nn), spr = wave_sum(np);
            if (lane_id() == 0) {
                atomicAdd(ANY ? &ctr->nodes_shadow : &ctr->nodes_closest, sn);
                atomicAdd(ANY ? &ctr->prims_shadow : &ctr->prims_closest, spr);
            }
        }
    }
}

// ---- k_any_leaf: any-hit rays of a one-leaf scene ------------------------------------------------------------------------------------
// intersect_p is a boolean, so the order in which the leaf's primitives are tried is free: `ws.leaf_order` lists them by the solid angle
// they subtend at the lights (a shadow ray runs from the surface through the light, t_max = Inf: what the light sees stops it).  And, as in
// k_any_occluders below, a wave that walks the whole leaf for the last of its 64 rays runs most tests with a handful of lanes: stage A
// tries the first TH_LEAF_STAGE_A primitives on the rays as they come, the rays still looking are parked in a per-wave ring in LDS and
// go through the rest of the leaf 64 at a time (stage B).  S-cornell any-hit: 27 -> see DESIGN.md §4.
#ifndef TH_LEAF_STAGE_A
#define TH_LEAF_STAGE_A 3
#endif
template <bool COUNT, bool FULL_ONLY>
__global__ __launch_bounds__(kBlock, FULL_ONLY ? TH_TRACE_LEAF_WAVES : 4) void k_any_leaf(DeviceScene sc, WideScene ws, SegQueue q, const float4* __restrict__ ro, const float4* __restrict__ rd,
                                                                          const float* __restrict__ tmax_or_null, TraceOut out, Counters* ctr) {
    __shared__ SegView sv;
    __shared__ uint32_t s_ring[kBlock / 64][128];
    seg_load(q, sv);
    const uint32_t total = sv.prefix[kSeg];
    const uint32_t first = ws.root_ref, cnt = ws.root_cnt;
    const uint32_t lane = lane_id(), wv = threadIdx.x >> 6;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const uint32_t n_a = min(cnt, (uint32_t)TH_LEAF_STAGE_A);
    uint32_t nn = 0, np = 0;
    uint32_t ring_head = 0, ring_cnt = 0;  // wave-uniform
    // primitives order[k0 .. k1) on this lane's ray while it is `live`; returns "accepted by one of them"
    auto run = [&](uint32_t idx, f3 o, f3 d, uint32_t k0, uint32_t k1, bool& live) {
        const float t_max = (live && tmax_or_null) ? tmax_or_null[idx] : kInf;
        const RayShear shear = ray_shear(d);
        bool found = false;
#pragma unroll 1
        for (uint32_t k = k0; k < k1; ++k) {
            if (__ballot(live) == 0ull) break;
            const uint32_t slot = first + (ws.leaf_order ? uniform_load(ws.leaf_order, k) : k);  // wave-uniform: scalar loads
            const float4 p0 = uniform_load(sc.prims, 3 * slot);
            const uint32_t meta = __float_as_uint(p0.w);
            if (COUNT && lane == 0) np++;  // primitive records FETCHED: one scalar fetch serves the wave's 64 rays (what the byte model of bench.py counts)
            if (meta & PRIM_SPHERE) {
                const SphereRec sr = uniform_load(sc.spheres, __float_as_uint(p0.x));
                if (live) {
                    SphereHit sh;
                    if (sphere_intersect<false, FULL_ONLY>(sr, o, d, t_max, sh)) {
                        found = true;
                        live = false;
                    }
                }
            } else if (!(meta & PRIM_DEGENERATE)) {
                const float4 p1 = uniform_load(sc.prims, 3 * slot + 1), p2 = uniform_load(sc.prims, 3 * slot + 2);
                if (live) {
                    TriTest tt;
                    if (tri_intersect_sheared<false>(mk3(p0.x, p0.y, p0.z), mk3(p1.x, p1.y, p1.z), mk3(p2.x, p2.y, p2.z), o, shear, t_max, &tt)) {
                        found = true;
                        live = false;
                    }
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
    auto stage_b = [&](uint32_t n) {  // the first n parked rays through the rest of the leaf
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
    for (uint32_t flat = blockIdx.x * kBlock + threadIdx.x; flat < total; flat += gridDim.x * kBlock) {
        uint32_t lb;
        seg_locate_from(sv, flat & ~63u, seg, lb);
        const uint32_t local = lb + (flat & 63u);
        const bool valid = local < sv.count[seg];
        const uint32_t idx = valid ? seg_phys(q, seg, local) : 0u;
        float4 o4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), d4 = make_float4(0.0f, 0.0f, 1.0f, 0.0f);
        if (valid) {
            o4 = ro[idx];
            d4 = rd[idx];
        }
        const f3 o = mk3(o4.x, o4.y, o4.z), d = mk3(d4.x, d4.y, d4.z);
        bool live = false;  // still has primitives to test
        if (valid) {
            const float t_max = tmax_or_null ? tmax_or_null[idx] : kInf;
            const f3 inv_d = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
            float tmin;
            if (COUNT) nn++;
            live = slab_test2(ws.root_box[0], ws.root_box[1], ws.root_box[2], ws.root_box[3], ws.root_box[4], ws.root_box[5], o, inv_d, 0.0f, false, d.x < 0.0f, d.y < 0.0f, d.z < 0.0f, tmin) &&
                   tmin < t_max;
        }
        const bool found = run(idx, o, d, 0u, n_a, live);
        const bool park = live && n_a < cnt;  // passed the root box, not stopped yet, primitives left
        deliver(valid && !park, idx, o4, d4, found);
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
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&ctr->shadow_total, (unsigned long long)seg_total(sv));
        if (COUNT) {
            const unsigned long long sn = wave_sum(nn), spr = wave_sum(np);
            if (lane_id() == 0) {
                atomicAdd(&ctr->nodes_shadow, sn);
                atomicAdd(&ctr->prims_shadow, spr);
            }
        }
    }
}

// ---- k_any_occluders: the scene's largest triangles first, for any-hit rays -----------------------------------------------------
// intersect_p(bvh, ray) is a boolean — "some primitive the walk reaches accepts the ray" — so the order of the walk is free.  A
// shadow ray of the reference has t_max = Inf (A.8): in an interior (every BASELINE scene is a mesh in a closed Cornell box) it ends
// on a wall BEYOND the light at the latest, after walking the mesh's hierarchy all the way.  This pre-pass tests the ≤ 16 largest
// triangles of the scene (picked at upload: the walls) with a wave-uniform loop — scalar loads, all lanes busy, like k_trace_leaf —
// and resolves the ray as occluded when one of them accepts it AND the reference's own box test (bounds.jl:180-200, the loose one)
// passes on that triangle's leaf box.  The second clause makes the shortcut exact: every ancestor box contains the leaf box, the
// slab products are monotonic in the box planes (x - o and its product with 1/d round monotonically), so every clause of the test
// that passes on the leaf box passes on each ancestor: the reference's walk does reach that leaf (or returns true before).  Rays
// with a zero direction component (0 · Inf = NaN breaks the monotonicity argument) and rays no large triangle stops go on to
// k_trace3 through per-segment survivor lists (SegQueue::indirect).  S-mesh any-hit 134 -> see DESIGN.md §4.
struct OccluderSet {
    const uint32_t* slots;  // ordered primitive slots of the occluders
    const float* boxes;     // their leaf node's bounds, 6 floats each
    uint32_t n;
};
// Two stages.  Ordered by the solid angle they subtend at the lights, the first two or three occluders stop most rays (2.3 tests per
// ray on the S-mesh scene) — but a wave that walks the whole list for the last of its 64 rays runs all the tests with a handful of
// lanes.  Stage A therefore tests only the first TH_OCC_STAGE_A occluders on the rays as they come; the rays still looking are parked in
// a per-wave ring in LDS and, 64 at a time, go through the rest of the list in stage B with all lanes busy at its start.
#ifndef TH_OCC_STAGE_A
#define TH_OCC_STAGE_A 3
#endif
template <bool COUNT>
__global__ __launch_bounds__(kBlock, 5) void k_any_occluders(DeviceScene sc, OccluderSet oc, SegQueue q, const float4* __restrict__ ro, const float4* __restrict__ rd,
                                                             const float* __restrict__ tmax_or_null, TraceOut out, uint32_t* __restrict__ surv, uint32_t* __restrict__ surv_counts,
                                                             uint32_t surv_cap, Counters* ctr) {
    __shared__ SegView sv;
    __shared__ uint32_t s_ring[kBlock / 64][128];
    seg_load(q, sv);
    const uint32_t total = sv.prefix[kSeg];
    const uint32_t lane = lane_id(), wv = threadIdx.x >> 6;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    // the survivors of a wave go to one list: chunk c of the flat work space is taken by wave c mod (waves of the grid, a multiple of kSeg),
    // so list s receives the survivors of every kSeg-th chunk: at most total / kSeg + 64 <= surv_cap entries
    const uint32_t seg_out = q.counts ? ((blockIdx.x * kBlock + threadIdx.x) >> 6) % kSeg : 0u;  // a dense queue (kernel-level API) has one list
    const uint32_t n_a = min(oc.n, (uint32_t)TH_OCC_STAGE_A);
    uint32_t nn = 0, np = 0;
    uint32_t ring_head = 0, ring_cnt = 0;  // wave-uniform
    // occluders [k0, k1) on this lane's ray (if `on`); writes the result of a ray that is stopped; returns "stopped"
    auto run = [&](bool on, uint32_t idx, float4 o4, float4 d4, uint32_t k0, uint32_t k1, bool& live) {
        const f3 o = mk3(o4.x, o4.y, o4.z), d = mk3(d4.x, d4.y, d4.z);
        const float t_max = (on && tmax_or_null) ? tmax_or_null[idx] : kInf;
        const f3 inv_d = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        const RayShear shear = ray_shear(d);
        bool found = false;
#pragma unroll 1
        for (uint32_t k = k0; k < k1; ++k) {
            if (__ballot(live) == 0ull) break;
            const uint32_t slot = uniform_load(oc.slots, k);
            const float4 p0 = uniform_load(sc.prims, 3 * slot), p1 = uniform_load(sc.prims, 3 * slot + 1), p2 = uniform_load(sc.prims, 3 * slot + 2);
            const float b0 = uniform_load(oc.boxes, 6 * k), b1 = uniform_load(oc.boxes, 6 * k + 1), b2 = uniform_load(oc.boxes, 6 * k + 2), b3 = uniform_load(oc.boxes, 6 * k + 3),
                        b4 = uniform_load(oc.boxes, 6 * k + 4), b5 = uniform_load(oc.boxes, 6 * k + 5);
            if (COUNT && lane == 0) {  // records FETCHED: one scalar fetch of the triangle and of its leaf box serves the wave's 64 rays
                np++;
                nn++;
            }
            if (live) {
                TriTest tt;
                if (tri_intersect_sheared<false>(mk3(p0.x, p0.y, p0.z), mk3(p1.x, p1.y, p1.z), mk3(p2.x, p2.y, p2.z), o, shear, t_max, &tt)) {
                    float tmin;
                    if (slab_test2(b0, b1, b2, b3, b4, b5, o, inv_d, 0.0f, false, d.x < 0.0f, d.y < 0.0f, d.z < 0.0f, tmin) && tmin < t_max) {
                        found = true;
                        live = false;
                    }
                }
            }
        }
        if (on && found) {
            if (out.L) {
                const uint32_t poison = __float_as_uint(d4.w);
                if (poison) {
                    const uint32_t slot = __float_as_uint(o4.w);
                    float4 l = out.L[slot];
                    const float nanv = __builtin_nanf("");
                    if (poison & 1u) l.x += nanv;
                    if (poison & 2u) l.y += nanv;
                    if (poison & 4u) l.z += nanv;
                    out.L[slot] = l;
                }
            } else {
                out.occluded[idx] = 1;
            }
        }
        return found;
    };
    auto survivors = [&](bool survive, uint32_t idx) {  // called by the whole wave
        const uint32_t j = wave_compact(survive, &surv_counts[seg_out * kCtrStride]);
        if (survive && j < surv_cap) surv[seg_out * surv_cap + j] = idx;
    };
    auto stage_b = [&](uint32_t cnt) {  // the first cnt parked rays through the rest of the list
        const bool on = lane < cnt;
        const uint32_t idx = on ? s_ring[wv][(ring_head + lane) & 127u] : 0u;
        float4 o4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), d4 = make_float4(0.0f, 0.0f, 1.0f, 0.0f);
        if (on) {
            o4 = ro[idx];
            d4 = rd[idx];
        }
        bool live = on;
        const bool found = run(on, idx, o4, d4, n_a, oc.n, live);
        survivors(on && !found, idx);
    };
    uint32_t seg = 0;  // (carried over the iterations: seg_locate_from)
    for (uint32_t flat = blockIdx.x * kBlock + threadIdx.x; flat < total; flat += gridDim.x * kBlock) {
        uint32_t lb;
        seg_locate_from(sv, flat & ~63u, seg, lb);
        const uint32_t local = lb + (flat & 63u);
        const bool valid = local < sv.count[seg];
        const uint32_t idx = valid ? seg_phys(q, seg, local) : 0u;
        float4 o4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), d4 = make_float4(0.0f, 0.0f, 1.0f, 0.0f);
        if (valid) {
            o4 = ro[idx];
            d4 = rd[idx];
        }
        const bool testable = valid && d4.x != 0.0f && d4.y != 0.0f && d4.z != 0.0f;  // a zero component: straight to k_trace3 (see above)
        bool live = testable;
        const bool found = run(valid, idx, o4, d4, 0u, n_a, live);
        const bool park = testable && !found && n_a < oc.n;
        survivors(valid && !found && !park, idx);
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
        // every ray of the queue is counted here, once (the survivors' launch is told not to: SegQueue::no_total); a count of the resolved
        // rays per wave was 8 192 atomics on one word at the end of every launch
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&ctr->shadow_total, (unsigned long long)seg_total(sv));
        if (COUNT) {
            const unsigned long long sn = wave_sum(nn), spr = wave_sum(np);
            if (lane_id() == 0) {
                atomicAdd(&ctr->nodes_shadow, sn);
                atomicAdd(&ctr->prims_shadow, spr);
            }
        }
    }
}

// ---- k_trace3: k_trace2 with the work of a wave re-grouped ("while-while") ------------------------------------------------------
// Measured on the 1 M-triangle scene, k_trace2 keeps 11.6 of 64 lanes busy per VALU instruction: in every step the lanes at an
// interior node, the lanes at a leaf and the lanes popping run one after the other.  Here a wave alternates between two
// phases: (A) lanes pop and step through interior nodes while lanes that reached a leaf wait, until at most TH_TRACE3_LEAF_WAIT lanes
// are still descending; (B) every lane that holds a leaf tests its primitives.  Each ray performs exactly the operations it
// performs in k_trace2, in the same order; only the interleaving across lanes differs, so the results are the same bit for bit.
#ifndef TH_TRACE3_LEAF_WAIT
#define TH_TRACE3_LEAF_WAIT 32  // measured (128 spp, S-blob / S-mesh frame ms): 20: 694 / 2250, 32: 654 / 2207, 40: 659 / 2202, 48: 696 / 2265
#endif
#ifndef TH_TRACE3_POP_MIN
#define TH_TRACE3_POP_MIN 8  // with the in-step pops: 1 -> 8 measured -1 % (S-mesh 411.2 -> 407.4 ms), 12 / 16 the same
#endif
#ifndef TH_TRACE3_MAX_A
#define TH_TRACE3_MAX_A 8
#endif
#ifndef TH_TRACE3_INLINE_POP
#define TH_TRACE3_INLINE_POP 1
#endif
#ifndef TH_TRACE3_LEAF_BURST
#define TH_TRACE3_LEAF_BURST 1
#endif
#ifndef TH_TRACE3_LEAF_PREFETCH
#define TH_TRACE3_LEAF_PREFETCH 0
#endif
#ifndef TH_TRACE3_SMALL_CHUNKS
#define TH_TRACE3_SMALL_CHUNKS 2  // quarter chunks once fewer than this many full chunks per wave of the segment's share are left; 0 = always full chunks
#endif
// Postponed leaves (closest-hit): a lane that reaches a leaf parks it (`pend`) and goes on descending instead of idling until the wave's next leaf
// phase; the leaf phase tests the parked leaf first.  The lane's t_max is then STALE (too large) while it descends: it visits boxes the reference would
// have culled — never fewer — and whatever leaf it finds there is tested against the clause `tx_min < t_max` again once the parked leaf has been
// tested (a child's entry distance is never below its parent's, so a leaf under a box the reference culled fails that clause).  The primitives are
// still tested in the reference's order with the reference's t_max, so the results are the same bit for bit.  Needs t_max to be non-increasing:
// rays that start inside a sphere's bound (sphere.jl:137-138 returns t1 without looking at t_max, A.18) do not postpone.
// Measured (round 3, S-mesh, 256 spp; profiles/r3/r3aa_postponed_leaves_ab.txt): exact (the 220 parity / scale / edge tests pass with it), 3 % more boxes per ray
// (48.8 against 47.2) — and no gain: closest-hit 295.0 ms with the lanes parking, 295.0 ms with the same binary and the option off, 271.4 ms without the code
// (its three extra live values turn 20 bytes of cold scratch into 100).  The wave's phase schedule already keeps the leaf phase short; compiled out.
#ifndef TH_TRACE3_SPEC
#define TH_TRACE3_SPEC 0
#endif
#ifdef TH_DIAG_PHASES
// DIAGNOSTIC build (tools/phase_probe.py): wave cycles and active lanes per phase of k_trace3, summed over the waves of all launches
static __device__ unsigned long long g_phase[16];
#define TH_PHASE_BEGIN() const unsigned long long ph_t0 = __builtin_readcyclecounter()
#define TH_PHASE_END(slot, lanes)                                                   \
    do {                                                                            \
        ph_cyc[slot] += __builtin_readcyclecounter() - ph_t0;                       \
        ph_lan[slot] += (unsigned long long)(lanes);                                \
        ph_cnt[slot] += 1ull;                                                       \
    } while (0)
#else
#define TH_PHASE_BEGIN()
#define TH_PHASE_END(slot, lanes)
#endif
// BIG (closest-hit only): scenes whose nodes and primitives exceed the last-level cache (10.5 M triangles: 1.2 GB) run one wave per SIMD
// fewer — there the sixth wave's extra streams cost more in misses than they hide (365 vs 374 ms; the 1 M-triangle scene: 282 vs 271 ms)
// PAIRS: the node array is the hybrid mode's accelerator, whose nodes keep each axis' two planes side by side (tu_scene.hip build_accelerator; th_trace3c.h "The step")
template <bool ANY, bool COUNT, bool FULL_ONLY, bool BIG = false, bool PAIRS = false>
__global__ __launch_bounds__(kBlock, ANY ? TH_TRACE3_WAVES_ANY : (BIG ? TH_TRACE3_WAVES_CLOSEST - 1 : TH_TRACE3_WAVES_CLOSEST)) void k_trace3(DeviceScene sc, WideScene ws, SegQueue q, const float4* __restrict__ ro, const float4* __restrict__ rd, const float* __restrict__ tmax_or_null,
                                                   TraceOut out, uint32_t* __restrict__ work, uint2* __restrict__ overflow, Counters* ctr) {
    constexpr int kLds = ANY ? TH_TRACE3_LDS_ANY : TH_TRACE3_LDS_CLOSEST;
    __shared__ uint32_t s_ref[kLds][kBlock];
    __shared__ float s_tmin[kLds][kBlock];
    __shared__ SegView sv;
    if (ANY && out.zero_mode == 2u && *out.zero_flag == 0u) return;  // (uniform: no ray of this launch has a zero direction component)
    seg_load(q, sv);
    const uint32_t tid = threadIdx.x;
    const uint32_t gthreads = gridDim.x * kBlock;
    const uint32_t gtid = blockIdx.x * kBlock + tid;
    const uint32_t lane = lane_id();

    bool active = false, exhausted = false;
    uint32_t wseg = __builtin_amdgcn_readfirstlane((gtid >> 6) % kSeg), dry = 0, pool_next = 0, pool_end = 0;  // wave-uniform: scalar registers
    uint32_t idx = 0, cur = kRefNone, cur_cnt = 0;
    int sp = 0;
    f3 o = splat3(0.0f), inv_d = splat3(0.0f);  // the direction itself is not kept: only spheres need it (reloaded there)
    float em = 0.0f;
    const bool tight_on = ws.tight_scale > 0.0f;
    RayShear shear{0, 0.0f, 0.0f, 0.0f};
    bool negx = false, negy = false, negz = false;
    float t_max = 0.0f, slot_w = 0.0f, flag_w = 0.0f;
    bool found = false;
    uint32_t nn = 0, np = 0;
    constexpr bool SPEC = !ANY && TH_TRACE3_SPEC != 0;
    uint32_t pend = kRefNone;  // SPEC: the parked leaf (ref | count << 24)
    float cur_tm = 0.0f;       // SPEC: entry distance of a leaf reached while another one is parked
    bool spec_ok = false;
#if TH_TRACE3_LEAF_PREFETCH
    uint32_t pf0 = 0, pf1 = 0;  // landing registers of the leaf prefetch, never read
#endif
#ifdef TH_DIAG_PHASES
    unsigned long long ph_cyc[4] = {0, 0, 0, 0}, ph_lan[4] = {0, 0, 0, 0}, ph_cnt[4] = {0, 0, 0, 0};  // refill, pop, node, leaf
    unsigned long long ph_load = 0;  // of the leaf cycles: until the primitive's three records have arrived
#endif
#ifdef TH_DIAG_RAY_VISITS
    uint32_t rn = 0;  // DIAGNOSTIC build only: interior fetches of the current ray, delivered in hits[].x (tools/visit_probe.py)
#endif

    while (true) {
        // ---- refill idle lanes (as k_trace2) ----------------------------------------------------------------------------------
        const unsigned long long idle = __ballot(!active);
        const uint32_t n_idle = (uint32_t)__popcll(idle);
        if (n_idle == 64u || (!exhausted && n_idle >= (uint32_t)TH_TRACE_REFILL)) {
            TH_PHASE_BEGIN();
            if (!exhausted) {
                if (pool_next >= pool_end) {
                    // an empty segment, or one whose cursor already ran past its count, needs no atomic: the waves that arrive when the queue is
                    // drained (all of them, at the end of every launch) would otherwise queue 32 returning atomics each on the same 32 words
                    const uint32_t cnt = __builtin_amdgcn_readfirstlane(sv.count[wseg]);
                    uint32_t base = cnt, take = (uint32_t)kChunk;
                    if (lane == 0 && cnt != 0u) {
                        const uint32_t at = __hip_atomic_load(&work[wseg * kCtrStride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (at < cnt) {
                            // towards the end of a segment the waves take a quarter of a chunk: a wave that leaves with the last 256 rays works through four
                            // generations of them alone while the others idle (≈ 0.4 ms per launch, a tenth of a 32-spp launch: profiles/r3/r3ad_*)
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
                    } else {
                        pool_next = pool_end = 0;
                        wseg = (wseg + 1) % kSeg;
                        if (++dry >= (uint32_t)kSeg) exhausted = true;
                    }
                }
                const uint32_t avail = pool_end - pool_next;
                if (avail && !active) {
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));  // idle lanes below this one (no mask kept in registers)
                    if (rank < avail) {
                        idx = seg_phys(q, wseg, pool_next + rank);
                        if (q.indirect) idx = q.indirect[idx];
                        const float4 o4 = ro[idx], d4 = rd[idx];
                        o = mk3(o4.x, o4.y, o4.z);
                        const f3 d = mk3(d4.x, d4.y, d4.z);
                        slot_w = o4.w;
                        flag_w = d4.w;
                        inv_d = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                        em = slab_margin(ws.root_box, ws.tight_scale, o);
                        shear = ray_shear(d);
                        negx = d.x < 0.0f;
                        negy = d.y < 0.0f;
                        negz = d.z < 0.0f;
                        t_max = tmax_or_null ? tmax_or_null[idx] : kInf;
                        sp = 0;
                        found = false;
                        active = true;
                        if (SPEC) {
                            pend = kRefNone;
                            spec_ok = ws.spec_spheres != 0xffffffffu;
                            for (uint32_t k = 0; k < 8u; ++k)
                                if (k < ws.spec_spheres && o.x >= ws.spec_box[k][0] && o.y >= ws.spec_box[k][1] && o.z >= ws.spec_box[k][2] && o.x <= ws.spec_box[k][3] && o.y <= ws.spec_box[k][4] &&
                                    o.z <= ws.spec_box[k][5])
                                    spec_ok = false;
                        }
                        float tmin;
#ifdef TH_DIAG_RAY_VISITS
                        rn = 0;
#endif
                        if (COUNT) nn++;
                        if (ws.root_ref != kRefNone && slab_test2(ws.root_box[0], ws.root_box[1], ws.root_box[2], ws.root_box[3], ws.root_box[4], ws.root_box[5], o, inv_d, em, false, negx, negy, negz, tmin) &&
                            tmin < t_max) {
                            cur = ws.root_ref;
                            cur_cnt = ws.root_cnt;
                        } else {
                            cur = kRefNone;
                            cur_cnt = 0;
                        }
                        if (ANY && out.zero_mode) {  // hybrid any-hit (TraceOut): mode 1 leaves the rays with a zero direction component to the second launch, mode 2 takes only those
                            const bool z = d.x == 0.0f || d.y == 0.0f || d.z == 0.0f;
                            if (z == (out.zero_mode == 1u)) {
                                if (z) *out.zero_flag = 1u;
                                active = false;  // not delivered here
                                cur = kRefNone;
                                cur_cnt = 0;
                            }
                        }
                    }
                }
                pool_next += min(n_idle, avail);
            }
            TH_PHASE_END(0, n_idle);
            if (__ballot(active) == 0ull) {
                if (exhausted) break;
                continue;
            }
        }
        // ---- phase A: pop / interior steps; lanes holding a leaf wait ---------------------------------------------------------------
#pragma unroll 1
        for (int it = 0; it < TH_TRACE3_MAX_A; ++it) {
            bool finished = false;
#ifdef TH_DIAG_PHASES
            const unsigned long long ph_pop_m = __ballot(active && cur == kRefNone);
            const unsigned long long ph_t_pop = __builtin_readcyclecounter();
#endif
            // lanes whose node is done pop their stack — together: the pop section runs only when TH_TRACE3_POP_MIN lanes wait for it (or
            // nobody can take an interior step), instead of in every round for the handful of lanes that happen to need it
            // SPEC: a lane whose stack is empty while a leaf is parked has nothing to pop: it waits for the leaf phase
            const bool wants_pop = active && cur == kRefNone && !(SPEC && pend != kRefNone && sp == 0);
            const bool pop_now = (uint32_t)__popcll(__ballot(wants_pop)) >= (uint32_t)TH_TRACE3_POP_MIN || __ballot(active && cur != kRefNone && cur_cnt == 0) == 0ull;
            if (pop_now && wants_pop) {  // pop the next entry whose tx_min is still below t_max (bvh.jl:247-250 with the deferred clause)
                finished = true;
                while (sp > 0) {
                    sp--;
                    uint32_t enc;
                    float tm;
                    if (sp < kLds) {
                        enc = s_ref[sp][tid];
                        tm = s_tmin[sp][tid];
                    } else if (sp < kStack2Total) {
                        const uint2 e = overflow[(size_t)(sp - kLds) * gthreads + gtid];
                        enc = e.x;
                        tm = __uint_as_float(e.y);
                    } else {
                        continue;
                    }
                    if (tm < t_max) {
                        cur = enc & 0x00ffffffu;
                        cur_cnt = enc >> 24;
                        if (SPEC) cur_tm = tm;
                        finished = false;
                        break;
                    }
                }
                if (SPEC && pend != kRefNone) finished = false;  // the parked leaf is still to be tested
                if (SPEC && spec_ok && pend == kRefNone && cur != kRefNone && cur_cnt > 0) {  // a popped leaf: park it, the next pop round goes on
                    pend = cur | (cur_cnt << 24);
                    cur = kRefNone;
                    cur_cnt = 0;
                }
            }
            if (finished) {  // the ray is done: deliver (as k_trace2)
                active = false;
                if (ANY) {
                    if (out.L) {
                        const uint32_t slot = __float_as_uint(slot_w);
                        if (!found) {
                            const float4 c = out.contrib[idx];
                            float4 l = out.L[slot];
                            l.x += c.x;
                            l.y += c.y;
                            l.z += c.z;
                            out.L[slot] = l;
                        } else {
                            const uint32_t poison = __float_as_uint(flag_w);
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
                } else {
                    if (!found) out.hits[idx] = make_float4(kInf, __int_as_float(-1), 0.0f, 0.0f);  // a hit was stored when it was accepted
#ifdef TH_DIAG_RAY_VISITS
                    out.hits[idx].x = (float)rn;
#endif
                }
            }
#ifdef TH_DIAG_PHASES
            {
                const unsigned long long now = __builtin_readcyclecounter();
                ph_cyc[1] += now - ph_t_pop;
                ph_lan[1] += (unsigned long long)__popcll(ph_pop_m);
                ph_cnt[1] += 1ull;
            }
            const unsigned long long ph_node_m = __ballot(active && cur != kRefNone && cur_cnt == 0);
            const unsigned long long ph_t_node = __builtin_readcyclecounter();
#endif
            if (active && cur != kRefNone && cur_cnt == 0) {  // interior: one 64-byte burst, both child boxes
                const float4 a0 = ws.wnodes[4 * (size_t)cur], a1 = ws.wnodes[4 * (size_t)cur + 1], a2 = ws.wnodes[4 * (size_t)cur + 2], a3 = ws.wnodes[4 * (size_t)cur + 3];
#if TH_TRACE3_INLINE_POP
                // the stack top, read while the node is on its way: when neither child is entered the lane continues with it in this same
                // step instead of idling through a pop section of its own at the next round (that section ran for ~9 of 64 lanes and took a
                // fifth of the kernel's cycles); a dead top (tx_min >= t_max) is dropped and the pop section goes on from there, as before
                uint32_t top_enc = kRefNone;
                float top_tm = kInf;
                if (sp > 0) {
                    if (sp - 1 < kLds) {
                        top_enc = s_ref[sp - 1][tid];
                        top_tm = s_tmin[sp - 1][tid];
                    } else if (sp - 1 < kStack2Total) {
                        const uint2 e = overflow[(size_t)(sp - 1 - kLds) * gthreads + gtid];
                        top_enc = e.x;
                        top_tm = __uint_as_float(e.y);
                    }
                }
#endif
                if (COUNT) nn += 2;
#ifdef TH_DIAG_RAY_VISITS
                rn++;
#endif
                const uint32_t lenc = __float_as_uint(a3.x), renc = __float_as_uint(a3.y), meta = __float_as_uint(a3.z);
                float tl, tr;
                const bool hl = PAIRS ? slab_test2(a0.x, a0.z, a1.x, a0.y, a0.w, a1.y, o, inv_d, em, tight_on && !(meta & 4u), negx, negy, negz, tl)
                                      : slab_test2(a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, o, inv_d, em, tight_on && !(meta & 4u), negx, negy, negz, tl);
                const bool hr = PAIRS ? slab_test2(a1.z, a2.x, a2.z, a1.w, a2.y, a2.w, o, inv_d, em, tight_on && !(meta & 8u), negx, negy, negz, tr)
                                      : slab_test2(a1.z, a1.w, a2.x, a2.y, a2.z, a2.w, o, inv_d, em, tight_on && !(meta & 8u), negx, negy, negz, tr);
                // a missed child gets tx_min = +Inf: it then fails `tx_min < t_max` like a hit one beyond t_max (a NaN tx_min fails it too, as in bounds.jl:198)
                const float tlh = hl ? tl : kInf, trh = hr ? tr : kInf;
                const uint32_t axis = meta & 3u;
                const bool neg = axis == 0 ? negx : (axis == 1 ? negy : negz);  // bvh.jl:239: dir_is_neg[split_axis] == 2 -> second child first
                const float tn = neg ? trh : tlh, tf = neg ? tlh : trh;
                const uint32_t nenc = neg ? renc : lenc, fenc = neg ? lenc : renc;
                const bool go_n = tn < t_max, go_f = tf < t_max;
                // The far child waits on the stack; its `tx_min < t_max` clause is evaluated when the entry is popped, as bvh.jl:226 does at
                // visit time.  A child that fails the clause NOW fails it then too — unless t_max can go up in between: t_max is not
                // monotonic in the reference (a sphere entered from inside returns t1 without looking at t_max, sphere.jl:137-138).  Any-hit rays,
                // whose t_max never changes, skip the dead entries.  (Doing the same for closest-hit rays that start outside every sphere's
                // bound was measured: no gain — dead entries are not what the pop phase costs.)
                if (go_n & (ANY ? go_f : (tf < kInf))) {
                    if (sp < kLds) {
                        s_ref[sp][tid] = fenc;
                        s_tmin[sp][tid] = tf;
                    } else if (sp < kStack2Total) {
                        overflow[(size_t)(sp - kLds) * gthreads + gtid] = make_uint2(fenc, __float_as_uint(tf));
                    }
                    sp++;
                }
                const uint32_t nxt = go_n ? nenc : fenc;
                const bool any_child = go_n | go_f;
                cur = any_child ? (nxt & 0x00ffffffu) : kRefNone;
                cur_cnt = any_child ? (nxt >> 24) : 0u;
                if (SPEC) cur_tm = go_n ? tn : tf;
#if TH_TRACE3_INLINE_POP
                if (!any_child && sp > 0) {  // nothing was pushed in this step: the top read above is still the top
                    sp--;
                    if (top_tm < t_max && sp < kStack2Total) {
                        cur = top_enc & 0x00ffffffu;
                        cur_cnt = top_enc >> 24;
                        if (SPEC) cur_tm = top_tm;
                    }
                }
#endif
                if (SPEC && spec_ok && pend == kRefNone && cur != kRefNone && cur_cnt > 0) {  // reached a leaf: park it and go on (the pop section finds the next node)
                    pend = cur | (cur_cnt << 24);
                    cur = kRefNone;
                    cur_cnt = 0;
                }
#if TH_TRACE3_LEAF_PREFETCH
                // the lane now waits for phase B with a leaf in hand: touch its first primitive's record (48 bytes, possibly across two lines)
                // so that phase B finds it in the cache.  The loaded words are never read; the asm hides the loads from the compiler, hence
                // the explicit wait in front: every load the compiler tracks is then older than these two and its counting stays right.
                if (cur != kRefNone && cur_cnt > 0) {
                    const float4* rec = sc.prims + 3 * (size_t)cur;
                    asm volatile("s_waitcnt vmcnt(0)\n\tglobal_load_dword %0, %2, off\n\tglobal_load_dword %1, %2, off offset:44" : "+v"(pf0), "+v"(pf1) : "v"(rec) : "memory");
                }
#endif
            }
#ifdef TH_DIAG_PHASES
            ph_cyc[2] += __builtin_readcyclecounter() - ph_t_node;
            ph_lan[2] += (unsigned long long)__popcll(ph_node_m);
            ph_cnt[2] += 1ull;
#endif
            // lanes that can go on without touching a leaf; when few are left, everybody's leaves are tested together
            const uint32_t n_desc = (uint32_t)__popcll(__ballot(active && cur_cnt == 0 && !(SPEC && pend != kRefNone && cur == kRefNone && sp == 0)));
            if (n_desc <= (uint32_t)TH_TRACE3_LEAF_WAIT) break;
        }
        // ---- phase B: leaves, primitives in slot order, later equal-t hit wins (bvh.jl:229-237, triangle_mesh.jl:211-214) ----------
#ifdef TH_DIAG_PHASES
        const unsigned long long ph_leaf_m = __ballot(active && ((SPEC && pend != kRefNone) || (cur != kRefNone && cur_cnt > 0)));
        const unsigned long long ph_t_leaf = __builtin_readcyclecounter();
#endif
        const bool has_pend = SPEC && pend != kRefNone;  // the parked leaf comes first: it was reached first
        if (active && (has_pend || (cur != kRefNone && cur_cnt > 0))) {
            bool hit_any = false;
            const uint32_t leaf_ref = has_pend ? (pend & 0x00ffffffu) : cur, leaf_cnt = has_pend ? (pend >> 24) : cur_cnt;
#if TH_TRACE3_INLINE_POP
            // the stack top, read while the primitives are on their way: the leaf's lanes pop it at the end of this phase, together
            uint32_t top_enc = kRefNone;
            float top_tm = kInf;
            if (sp > 0) {
                if (sp - 1 < kLds) {
                    top_enc = s_ref[sp - 1][tid];
                    top_tm = s_tmin[sp - 1][tid];
                } else if (sp - 1 < kStack2Total) {
                    const uint2 e = overflow[(size_t)(sp - 1 - kLds) * gthreads + gtid];
                    top_enc = e.x;
                    top_tm = __uint_as_float(e.y);
                }
            }
#endif
            for (uint32_t k = 0; k < leaf_cnt; ++k) {
                const uint32_t slot = leaf_ref + k;
                const float4 p0 = sc.prims[3 * slot];
                const float4 p1 = sc.prims[3 * slot + 1], p2 = sc.prims[3 * slot + 2];
#if TH_TRACE3_LEAF_BURST
                // all three records in one burst: left alone, the compiler sinks the loads of p1 / p2 below the sphere / degenerate test on
                // p0.w — a second dependent round trip per primitive
                asm volatile("" ::"v"(p1.x), "v"(p1.y), "v"(p1.z), "v"(p2.x), "v"(p2.y), "v"(p2.z));
#endif
#ifdef TH_DIAG_PHASES
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                ph_load += __builtin_readcyclecounter() - ph_t_leaf;
#endif
                const uint32_t meta = __float_as_uint(p0.w);
                if (COUNT) np++;
                if (meta & PRIM_SPHERE) {
                    // rare: fetch the direction again, and rebuild what derives from it afterwards — that way none of it is live across
                    // the sphere code (transform, quadratic, divisions), which otherwise sets the kernel's register count
                    const float4 d4 = rd[idx];
                    const f3 d = mk3(d4.x, d4.y, d4.z);
                    SphereHit sh;
                    const bool sphere_hit = sphere_intersect<false, FULL_ONLY>(sc.spheres[__float_as_uint(p0.x)], o, d, t_max, sh);
                    inv_d = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                    em = slab_margin(ws.root_box, ws.tight_scale, o);
                    shear = ray_shear(d);
                    if (sphere_hit) {
                        if (ANY) {
                            hit_any = true;
                            break;
                        }
                        t_max = sh.t;
                        found = true;
                        out.hits[idx] = make_float4(sh.t, __int_as_float((int)slot), 0.0f, 0.0f);  // stored at once: a later accepted hit overwrites it
                    }
                } else {
                    TriTest tt;
                    if (!(meta & PRIM_DEGENERATE) && tri_intersect_sheared<!ANY>(mk3(p0.x, p0.y, p0.z), mk3(p1.x, p1.y, p1.z), mk3(p2.x, p2.y, p2.z), o, shear, t_max, &tt)) {
                        if (ANY) {
                            hit_any = true;
                            break;
                        }
                        t_max = tt.t;
                        found = true;
                        out.hits[idx] = make_float4(out.bary_mode ? tt.bary.z : tt.t, __int_as_float((int)slot), tt.bary.x, tt.bary.y);
                    }
                }
            }
            bool keep_cur = false;
            if (has_pend) {
                pend = kRefNone;
                if (cur != kRefNone && cur_cnt > 0) {
                    // the lane also holds a leaf it reached while this one was parked, i.e. with a stale t_max: its clause again (bvh.jl:226), against the
                    // t_max the reference has at this point; if it stands the leaf is parked in turn (first in line) and the lane goes on
                    if (cur_tm < t_max) pend = cur | (cur_cnt << 24);
                } else if (cur != kRefNone) {
                    keep_cur = true;  // an interior node the lane was about to step into: it stays
                }
            }
            if (!keep_cur) {
                cur = kRefNone;
                cur_cnt = 0;
            }
            if (ANY && hit_any) {  // intersect_p returns at the first accepted primitive: drop the stack, the pop in phase A delivers
                found = true;
                sp = 0;
            }
#if TH_TRACE3_INLINE_POP
            else if (!keep_cur && sp > 0) {  // the next stack entry, against the t_max the leaf left (bvh.jl:226 at pop time); a dead one is dropped, phase A goes on from there
                sp--;
                if (top_tm < t_max && sp < kStack2Total) {
                    cur = top_enc & 0x00ffffffu;
                    cur_cnt = top_enc >> 24;
                    if (SPEC) cur_tm = top_tm;
                    if (SPEC && spec_ok && pend == kRefNone && cur_cnt > 0) {  // a leaf again: park it
                        pend = cur | (cur_cnt << 24);
                        cur = kRefNone;
                        cur_cnt = 0;
                    }
                }
            }
#endif
        }
#ifdef TH_DIAG_PHASES
        ph_cyc[3] += __builtin_readcyclecounter() - ph_t_leaf;
        ph_lan[3] += (unsigned long long)__popcll(ph_leaf_m);
        ph_cnt[3] += 1ull;
#endif
    }
#ifdef TH_DIAG_PHASES
    {
        float f = (float)ph_load;  // a lane's view of the wave's clock; the float rounding does not matter here
        for (int off = 32; off > 0; off >>= 1) f = fmaxf(f, __shfl_xor(f, off));
        ph_load = (unsigned long long)f;
    }
    if (!ANY && lane == 0) atomicAdd(&g_phase[12], ph_load);
    if (!ANY && lane == 0)
        for (int k4 = 0; k4 < 4; ++k4) {
            atomicAdd(&g_phase[3 * k4], ph_cyc[k4]);
            atomicAdd(&g_phase[3 * k4 + 1], ph_lan[k4]);
            atomicAdd(&g_phase[3 * k4 + 2], ph_cnt[k4]);
        }
#endif
    if (ctr) {
        if (blockIdx.x == 0 && threadIdx.x == 0 && !q.no_total) atomicAdd(ANY ? &ctr->shadow_total : &ctr->closest_total, (unsigned long long)seg_total(sv));
        if (COUNT) {
            const unsigned long long sn = wave_sum(nn), spr = wave_sum(np);
            if (lane_id() == 0) {
                atomicAdd(ANY ? &ctr->nodes_shadow : &ctr->nodes_closest, sn);
                atomicAdd(ANY ? &ctr->prims_shadow : &ctr->prims_closest, spr);
            }
        }
    }
}

}  // namespace th
