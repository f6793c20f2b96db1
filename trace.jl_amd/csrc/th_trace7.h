// th_trace7.h — closest-hit traversal in FRONT-TO-BACK order with detection of the rays whose answer depends on the reference's visiting
// order, which are handed to the reference-order walk (k_trace3).  Option "traversal" = 7.
//
// Why an order-free walk can be exact.  accel/bvh.jl:212-258 keeps the last accepted primitive; a primitive is accepted iff its own test passes
// with the CURRENT t_max (triangle_mesh.jl:187-243: `t_scaled > t_max * det` rejects, equality accepts; sphere.jl:125-158), and a subtree is
// entered iff its box passes bounds.jl:180-200 with the current t_max.  Everything in those tests except the comparison with t_max is a function
// of (ray, primitive) or (ray, box) alone.  Call a primitive a CANDIDATE when the t_max-free part of its test passes and the t_max-free part of
// the box test passes for all its ancestors; its t is a number that does not depend on the walk.  Let w be the candidate of smallest t and `gap`
// a margin (below).  If no other candidate has t <= t_w + gap, EVERY visiting order returns w with the same (t, barycentrics):
//   * w is reached: when an ancestor box of w is popped, t_max is the ray's own or the t of an accepted candidate — all of them > t_w + gap —
//     and the box's entry is <= t_w + (the box test's slack, far below gap);
//   * w is accepted when it is tested (same reason), and after that every other candidate is rejected (t > t_w + gap against t_max = t_w).
// So the walk is free to visit near children first BY ENTRY DISTANCE (the reference picks by the split axis' sign, bvh.jl:239-246), to cull
// with t_best + margin at push and pop time, and to never revisit: rays with a second candidate inside the gap — exact ties on shared edges and
// vertices, coincident surfaces, slivers — are FLAGGED and re-traced by k_trace3, which resolves them in the reference's order.  Also flagged:
// a ray that starts inside a sphere (its t_max can go UP mid-walk), reaches a clipped sphere, or grazes a sphere's limb (the
// Float32 quadratic "hits" up to 1e-3 |o - c| outside the sphere's box: whether the reference reaches that leaf depends on its boxes); a
// candidate within the gap of the ray's own finite t_max; rays with a zero / non-finite direction component (NaN in the slab products).
//
// The box test keeps the reference's ARITHMETIC — (plane - o) * (1 / d) per plane, the larger of the x / y exits (bounds.jl:190, A.17), the clause
// structure — so "the t_max-free part passes" means exactly what it means in the reference; only the near / far plane SELECTS by the direction's
// sign become min / max of the two products (the same two floats: rounding is monotonic), and the two clauses the reference lost are ANDed on
// the box grown by the margin, as in slab_test2 (th_trace2.h): ~35 VALU instructions per box instead of ~55.
//
// CHEAP (the default when every leaf box is the exact union of its triangles' boxes — any tree built by this library): bounds.jl:186-200 is monotonic in the box —
// a box that passes keeps passing when it grows (every plane product moves the right way, rounding is monotonic) — so "all ancestors pass" is implied by "the
// LEAF's box passes".  Interior boxes then only have to be CONSERVATIVE: they get a plain slab test in fma form on the box grown by m (18 VALU instructions),
// which never rejects a box that holds an acceptable hit; the reference's exact test runs once per leaf reached, on the box recomputed from the leaf's
// vertices (min / max are exact).  Subtrees that hold a sphere keep the exact test on every box (their leaves' boxes come from the sphere records).
//
// gap = 4 m, cull margin = 5 m with m = 2^-17 x D x max |1 / d|, D = the largest coordinate offset between the ray origin and the scene bound: the watertight
// triangle test accepts rays that pass within ~40 ulps of |v - o| <= D of the triangle (translate, shear, edge functions: th_trace2.h), so a box's entry lags a hit
// inside it by less than 40 ulps of D in the entering axis — a third of m — and the roundings of `t_max * det` and `ts * (1 / det)` are ulps of t.  A candidate culled
// by a box (entry > t_best + 5 m) therefore has t > t_best + 4 m >= t_w + gap: nothing inside the gap is ever skipped.  (The box tests' own margin stays slab_test2's
// 2^-14 D per axis: it only widens boxes.)  Boxes on a sphere's path are culled with `lag_s` more slack: the Float32 quadratic can report a hit up to
// ~64 ulp x D^2 / r outside the sphere (the limb rule in sphere_candidate7 flags those rays when they are tested; the slack makes sure they ARE tested).
#pragma once
#include "th_trace2.h"
#include "th_trace8.h"  // FallbackList

namespace th {

#ifndef TH_TRACE7_WAVES
#define TH_TRACE7_WAVES 5  // 6 waves (80 VGPRs) spill 100 B per lane in the CHEAP variant: 4.78 ms against 4.07 per 8 M bounce rays
#endif
#ifndef TH_TRACE7_LDS
#define TH_TRACE7_LDS 12
#endif
#ifndef TH_TRACE7_LEAF_WAIT
#define TH_TRACE7_LEAF_WAIT 32
#endif
#ifndef TH_TRACE7_MAX_A
#define TH_TRACE7_MAX_A 8
#endif
#ifndef TH_TRACE7_POP_MIN
#define TH_TRACE7_POP_MIN 8
#endif

// One child box: the reference's slab arithmetic and clauses (bounds.jl:186-198 without the t_max clause) AND the two tight clauses on the box grown
// by m (slab_test2).  Returns the entry distance, +Inf when the box is missed.  No NaN can occur: rays with a zero direction component never get here.
TH_D float slab_entry7(float bx0, float by0, float bz0, float bx1, float by1, float bz1, f3 o, f3 inv_d, f3 ma /* margin per axis, in t units: em |1 / d| */, bool tight) {
    const float x0 = (bx0 - o.x) * inv_d.x, x1 = (bx1 - o.x) * inv_d.x;
    const float y0 = (by0 - o.y) * inv_d.y, y1 = (by1 - o.y) * inv_d.y;
    const float z0 = (bz0 - o.z) * inv_d.z, z1 = (bz1 - o.z) * inv_d.z;
    const float lox = fminf(x0, x1), hix = fmaxf(x0, x1);  // tx_min / tx_max of bounds.jl:186-187 whatever the sign of d.x
    const float loy = fminf(y0, y1), hiy = fmaxf(y0, y1);
    const float loz = fminf(z0, z1), hiz = fmaxf(z0, z1);
    const bool miss_xy = (lox > hiy) | (loy > hix);          // :188
    const float a = fmaxf(lox, loy);                         // :189
    const float b = fmaxf(hix, hiy);                         // :190 (the LARGER exit)
    const bool miss_z = (a > hiz) | (loz > b);               // :194
    const float t_in = fmaxf(loz, a);                        // :196
    const float t_out = fminf(hiz, b);                       // :197
    // the two clauses bounds.jl:190 lost, on the box grown by em per axis (slab_test2): z entry <= min(x exit, y exit), and that exit >= 0
    const float exit_xy = fminf(hix + ma.x, hiy + ma.y);
    const bool miss_tight = tight & ((loz - ma.z > exit_xy) | (exit_xy < 0.0f));
    const bool hit = !(miss_xy | miss_z | miss_tight) & (t_out > 0.0f);  // :198 without `t_in < t_max` (the caller's, with its margin)
    return hit ? t_in : kInf;
}

// CHEAP interior boxes: the standard slab test, one fma per plane (t = plane / d - o / d), on the box grown by m; conservative for every box that holds a hit
// the primitive tests accept (th_trace2.h, slab_test2's argument) — nothing else is asked of it.
TH_D float slab_entry_cheap(float bx0, float by0, float bz0, float bx1, float by1, float bz1, f3 inv_d, f3 noid, f3 ma) {
    const float x0 = __fmaf_rn(bx0, inv_d.x, noid.x), x1 = __fmaf_rn(bx1, inv_d.x, noid.x);
    const float y0 = __fmaf_rn(by0, inv_d.y, noid.y), y1 = __fmaf_rn(by1, inv_d.y, noid.y);
    const float z0 = __fmaf_rn(bz0, inv_d.z, noid.z), z1 = __fmaf_rn(bz1, inv_d.z, noid.z);
    // every slab grown by em in its own axis (em |1 / d_axis| in t units): a single scalar margin would inflate the other two axes of a grazing ray's boxes
    const float t_in = fmaxf(fmaxf(fminf(x0, x1) - ma.x, fminf(y0, y1) - ma.y), fminf(z0, z1) - ma.z);
    const float t_out = fminf(fminf(fmaxf(x0, x1) + ma.x, fmaxf(y0, y1) + ma.y), fmaxf(z0, z1) + ma.z);
    const bool hit = (t_in <= t_out) & (t_out >= 0.0f);  // a NaN (the empty leaf's box) fails both
    return hit ? t_in : kInf;  // the entry of the GROWN box: never later than the exact box's
}

// sphere.jl:125-158 up to the roots, as sphere_intersect (th_device.h); 0 = no candidate inside t_lim, 1 = candidate at t, 2 = the ray must be re-traced
// in the reference's order (a clipped sphere; the limb, where the quadratic accepts rays outside the box), 3 = the origin is inside the sphere: the reference
// accepts t1 whatever t_max is, so it ends up as the answer in every visiting order iff it is the nearest candidate — the caller flags the ray otherwise
template <bool FULL_ONLY>
TH_D int sphere_candidate7(const SphereRec& s, f3 o, f3 d, float t_lim, float& t) {
    const f3 oo = xf_point(s.o2w_inv, o);
    const f3 od = xf_vec(s.o2w_inv, d);
    const float nd = norm(od);
    const float a = nd * nd;
    const float b = dot(2.0f * oo, od);
    const float no = norm(oo);
    const float c = no * no - s.radius * s.radius;
    float t0, t1;
    if (!solve_quadratic(a, b, c, t0, t1)) {
        // no real root in Float32 — but a discriminant this close to zero is the limb as well (the sign of b*b - 4ac is rounding): the reference's own
        // answer is "miss" whatever the order, so nothing to flag
        return 0;
    }
    if (t1 < 0.0f) return 0;
    if (!FULL_ONLY && !s.never_clipped) return 2;
    if (t0 < 0.0f) {  // the origin is inside: the reference takes t1 WITHOUT looking at t_max (sphere.jl:137-138, A.18) — a candidate that can raise t_max
        t = t1;
        return 3;
    }
    if (t0 > t_lim) return 0;
    // the limb: b*b - 4ac carries an absolute error of a few ulps of b*b and of 4 a |o|^2; a discriminant within 64 ulps of those cannot tell a grazing hit from a
    // near miss, and a "hit" at the point of closest approach of a near miss lies outside the sphere's box — whether the reference reaches that leaf depends on
    // the boxes on the way.  Above the bound the ray truly pierces the sphere: the hit point is on it, inside every ancestor box.
    const float disc = b * b - 4 * a * c;
    if (disc < 3.8e-6f * (b * b + 4 * a * (no * no))) return 2;
    t = t0;
    return 1;
}

template <bool COUNT, bool FULL_ONLY, bool BIG, bool CHEAP>
__global__ __launch_bounds__(kBlock, BIG ? TH_TRACE7_WAVES - 1 : TH_TRACE7_WAVES) void k_trace7(DeviceScene sc, WideScene ws, SegQueue q, const float4* __restrict__ ro, const float4* __restrict__ rd,
                                                                                               const float* __restrict__ tmax_or_null, TraceOut out, uint32_t* __restrict__ work,
                                                                                               uint2* __restrict__ overflow, Counters* ctr, FallbackList fb) {
    constexpr int kLds = TH_TRACE7_LDS;
    __shared__ uint32_t s_ref[kLds][kBlock];
    __shared__ float s_tin[kLds][kBlock];
    __shared__ SegView sv;
    seg_load(q, sv);
    const uint32_t tid = threadIdx.x;
    const uint32_t gthreads = gridDim.x * kBlock;
    const uint32_t gtid = blockIdx.x * kBlock + tid;
    const uint32_t lane = lane_id();

    bool active = false, exhausted = false, to_fb = false;
    uint32_t wseg = __builtin_amdgcn_readfirstlane((gtid >> 6) % kSeg), dry = 0, pool_next = 0, pool_end = 0;  // wave-uniform
    uint32_t idx = 0, fb_idx = 0, cur = kRefNone, cur_cnt = 0;
    int sp = 0;
    f3 o = splat3(0.0f), inv_d = splat3(0.0f);
    f3 noid = splat3(0.0f);  // CHEAP: -o / d per axis
    RayShear shear{0, 0.0f, 0.0f, 0.0f};
    f3 ma = splat3(0.0f);   // the box tests' margin per axis in t units: em |1 / d_axis| (em = 2^-14 D, slab_test2's)
    float m = 0.0f;         // the tie / cull margin in t units: 2^-17 D max |1 / d| (header)
    float lag_s = 0.0f;     // how far a box on a sphere's path can be entered behind that sphere's (limb) hit: its boxes are culled with this much more slack
    float t_best = kInf;    // smallest candidate t so far (the hit record in out.hits belongs to it)
    float t_second = kInf;  // smallest t among the other candidates seen
    float t_cull = kInf;    // min(t_best, the ray's own t_max) + 5 m: boxes entered beyond it and candidates beyond it do not matter
    float t_own = kInf;     // the ray's own t_max
    float t_raise = kInf;   // smallest t of a sphere entered from inside (it ignores t_max): must end up as the nearest candidate, or the ray is flagged
    bool flagged = false;
    uint32_t nn = 0, np = 0;
    unsigned long long n_fb = 0;
    // COUNT (option "count_visits"): why rays went to the fallback list — 0: zero / non-finite direction or origin, 1: a sphere (origin inside, limb, clipped),
    // 2: a candidate within the gap of the ray's own t_max, 3: a second candidate within the gap of the nearest
    uint32_t why = 0;
    unsigned long long n_why[4] = {0ull, 0ull, 0ull, 0ull};

    while (true) {
        // ---- rays to re-trace in the reference's order: appended to the fallback list (wave-wide) ----
        // kSeg lists of fb.cap entries, one counter each (a single counter word serialises at ~88 atomics per microsecond: 3 ms for the flagged rays of one
        // launch); a wave starts at its own segment and moves on while a list is full — together they hold as many entries as the queue has rays
        if (__ballot(to_fb) != 0ull) {
            uint32_t fseg = __builtin_amdgcn_readfirstlane((gtid >> 6) % kSeg);
            for (int tries = 0; tries < kSeg && __ballot(to_fb) != 0ull; ++tries) {
                const uint32_t j = wave_compact(to_fb, &fb.counts[fseg * kCtrStride]);
                if (to_fb && j < fb.cap) {
                    fb.list[(size_t)fseg * fb.cap + j] = fb_idx;
                    n_fb++;
                    to_fb = false;
                }
                fseg = (fseg + 1) % kSeg;
            }
            to_fb = false;  // (cannot be left over: the lists' total capacity is the queue's)
        }
        // ---- refill idle lanes (as k_trace3) ----
        const unsigned long long idle = __ballot(!active);
        const uint32_t n_idle = (uint32_t)__popcll(idle);
        if (n_idle == 64u || (!exhausted && n_idle >= (uint32_t)TH_TRACE_REFILL)) {
            if (!exhausted) {
                if (pool_next >= pool_end) {
                    const uint32_t cnt = __builtin_amdgcn_readfirstlane(sv.count[wseg]);
                    uint32_t base = cnt;
                    if (lane == 0 && cnt != 0u && __hip_atomic_load(&work[wseg * kCtrStride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < cnt)
                        base = atomicAdd(&work[wseg * kCtrStride], (uint32_t)kChunk);
                    base = __builtin_amdgcn_readfirstlane(base);
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
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                    if (rank < avail) {
                        idx = seg_phys(q, wseg, pool_next + rank);
                        if (q.indirect) idx = q.indirect[idx];
                        const float4 o4 = ro[idx], d4 = rd[idx];
                        o = mk3(o4.x, o4.y, o4.z);
                        const f3 d = mk3(d4.x, d4.y, d4.z);
                        inv_d = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                        const float em = slab_margin(ws.root_box, ws.tight_scale, o);
                        ma = mk3(em * fabsf(inv_d.x), em * fabsf(inv_d.y), em * fabsf(inv_d.z));
                        const float D = em / ws.tight_scale, inv_max = fmaxf(fmaxf(fabsf(inv_d.x), fabsf(inv_d.y)), fabsf(inv_d.z));
                        m = 7.62939453125e-6f * D * inv_max;  // 2^-17 D max |1 / d|: 128 ulps of D against the <= 40 ulps the primitive tests' hit points can lie outside their boxes
                        lag_s = ws.sphere_lag * (D * D) * inv_max;
                        shear = ray_shear(d);
                        if (CHEAP) noid = mk3(-(o.x * inv_d.x), -(o.y * inv_d.y), -(o.z * inv_d.z));
                        t_own = tmax_or_null ? tmax_or_null[idx] : kInf;
                        t_best = t_second = t_raise = kInf;
                        t_cull = t_own + 5.0f * m;
                        sp = 0;
                        flagged = false;
                        active = true;
                        if (COUNT) nn++;
                        // what the argument above does not cover goes to k_trace3 at once: a zero or non-finite direction component (0 x Inf = NaN in the slab
                        // products; the reference's selects and this kernel's min / max then disagree), a non-finite origin or margin
                        // … and rays that start far outside the scene (a camera 50 scene sizes away): every margin here scales with D, the reach of the ray's
                        // Float32 arithmetic, and at D >> scene size the gap swallows the scene's detail — most such rays would be flagged anyway, and they are the
                        // coherent ones k_trace3 walks fastest
                        const float extent = fmaxf(fmaxf(ws.root_box[3] - ws.root_box[0], ws.root_box[4] - ws.root_box[1]), ws.root_box[5] - ws.root_box[2]);
                        const bool plain = d.x != 0.0f && d.y != 0.0f && d.z != 0.0f && m < kInf && m == m && fabsf(o.x) < kInf && fabsf(o.y) < kInf && fabsf(o.z) < kInf &&
                                           fabsf(inv_d.x) < kInf && fabsf(inv_d.y) < kInf && fabsf(inv_d.z) < kInf && D <= 8.0f * extent;
                        if (COUNT) why = 0;
                        if (!plain) {
                            to_fb = true;
                            fb_idx = idx;
                            active = false;
                            if (COUNT) n_why[0]++;
                        } else {
                            const float t_in = ws.root_ref != kRefNone
                                                   ? slab_entry7(ws.root_box[0], ws.root_box[1], ws.root_box[2], ws.root_box[3], ws.root_box[4], ws.root_box[5], o, inv_d, ma, false)
                                                   : kInf;
                            if (t_in < kInf && t_in <= t_cull) {  // (+Inf = missed; t_cull is +Inf until a candidate is found: `<=` alone would let it through)
                                cur = ws.root_ref;
                                cur_cnt = ws.root_cnt;
                            } else {
                                cur = kRefNone;
                                cur_cnt = 0;
                            }
                        }
                    }
                }
                pool_next += min(n_idle, avail);
            }
            if (__ballot(active) == 0ull) {
                if (__ballot(to_fb) != 0ull) continue;  // flush first
                if (exhausted) break;
                continue;
            }
        }
        // ---- phase A: pops and interior steps; lanes holding a leaf wait ----
#pragma unroll 1
        for (int it = 0; it < TH_TRACE7_MAX_A; ++it) {
            bool finished = false;
            const bool pop_now = (uint32_t)__popcll(__ballot(active && cur == kRefNone)) >= (uint32_t)TH_TRACE7_POP_MIN || __ballot(active && cur != kRefNone && cur_cnt == 0) == 0ull;
            if (pop_now && active && cur == kRefNone) {  // the next entry that still matters
                finished = true;
                while (sp > 0) {
                    sp--;
                    uint32_t enc;
                    float tin;
                    if (sp < kLds) {
                        enc = s_ref[sp][tid];
                        tin = s_tin[sp][tid];
                    } else {
                        const uint2 e = overflow[(size_t)(sp - kLds) * gthreads + gtid];
                        enc = e.x;
                        tin = __uint_as_float(e.y);
                    }
                    if (tin <= t_cull) {
                        cur = enc & 0x00ffffffu;
                        cur_cnt = enc >> 24;
                        finished = false;
                        break;
                    }
                }
            }
            if (finished) {  // the walk is over: a miss, a clean hit (stored when it was found), or a ray for the reference-order walk
                active = false;
                const bool found = t_best < kInf;
                const bool tie = found && t_second <= t_best + 4.0f * m;
                const bool raised = t_raise < kInf && t_best < t_raise;  // a sphere entered from inside lost to a nearer candidate: the reference's answer depends on its order
                if (flagged || tie || raised) {
                    to_fb = true;
                    fb_idx = idx;
                    if (COUNT) n_why[(flagged || raised) ? (why == 2u ? 2 : 1) : 3]++;
                } else if (!found) {
                    out.hits[idx] = make_float4(kInf, __int_as_float(-1), 0.0f, 0.0f);
                }
            }
            if (active && cur != kRefNone && cur_cnt == 0) {  // interior: one 64-byte burst, both child boxes
                const float4 a0 = ws.wnodes[4 * (size_t)cur], a1 = ws.wnodes[4 * (size_t)cur + 1], a2 = ws.wnodes[4 * (size_t)cur + 2], a3 = ws.wnodes[4 * (size_t)cur + 3];
                // the stack top, read while the node is on its way (as k_trace3): taken in this same step when neither child is entered
                uint32_t top_enc = kRefNone;
                float top_tin = kInf;
                if (sp > 0) {
                    if (sp - 1 < kLds) {
                        top_enc = s_ref[sp - 1][tid];
                        top_tin = s_tin[sp - 1][tid];
                    } else {
                        const uint2 e = overflow[(size_t)(sp - 1 - kLds) * gthreads + gtid];
                        top_enc = e.x;
                        top_tin = __uint_as_float(e.y);
                    }
                }
                if (COUNT) nn += 2;
                const uint32_t lenc = __float_as_uint(a3.x), renc = __float_as_uint(a3.y), meta = __float_as_uint(a3.z);
                float tl, tr;
                if (CHEAP) {
                    // grown by 2^-17 D per axis (an eighth of the tight clauses' margin; the hits the primitive tests accept lie within 40 ulps of D of their boxes)
                    const f3 mc = mk3(0.125f * ma.x, 0.125f * ma.y, 0.125f * ma.z);
                    tl = slab_entry_cheap(a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, inv_d, noid, mc);
                    tr = slab_entry_cheap(a1.z, a1.w, a2.x, a2.y, a2.z, a2.w, inv_d, noid, mc);
                } else {
                    tl = slab_entry7(a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, o, inv_d, ma, true);
                    tr = slab_entry7(a1.z, a1.w, a2.x, a2.y, a2.z, a2.w, o, inv_d, ma, true);
                }
                if (meta & 12u) {
                    // a child on a sphere's path (rare): the reference's exact test without the tight clauses — a sphere's Float32 "hit" may lie outside its
                    // box — and lag_s more slack against the cull distance (ordering and culling then see the entry that much earlier)
                    if (meta & 4u) tl = slab_entry7(a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, o, inv_d, ma, false) - lag_s;
                    if (meta & 8u) tr = slab_entry7(a1.z, a1.w, a2.x, a2.y, a2.z, a2.w, o, inv_d, ma, false) - lag_s;
                }
                // front to back by entry distance; both entries behind the origin (the ray starts inside both boxes — every ray leaving a surface does near its
                // own leaf): the side the ray travels towards first, i.e. the reference's rule (bvh.jl:239-246: second child first where d[split axis] < 0)
                const uint32_t axis = meta & 3u;
                const bool neg = (axis == 0u ? inv_d.x : (axis == 1u ? inv_d.y : inv_d.z)) < 0.0f;
                const float cl = fmaxf(tl, 0.0f), cr = fmaxf(tr, 0.0f);
                const bool l_first = cl == cr ? !neg : cl < cr;
                const float tn = l_first ? tl : tr, tf = l_first ? tr : tl;  // (a missed child is +Inf: never first unless both are)
                const uint32_t nenc = l_first ? lenc : renc, fenc = l_first ? renc : lenc;
                const bool go_n = (tn < kInf) & (tn <= t_cull), go_f = (tf < kInf) & (tf <= t_cull);  // +Inf = missed (t_cull is +Inf until a candidate is found); tn <= tf: go_f implies go_n
                if (go_f) {
                    if (sp < kLds) {
                        s_ref[sp][tid] = fenc;
                        s_tin[sp][tid] = tf;
                    } else if (sp < kStack2Total) {
                        overflow[(size_t)(sp - kLds) * gthreads + gtid] = make_uint2(fenc, __float_as_uint(tf));
                    } else {
                        flagged = true;  // cannot happen on a committed tree (depth <= 64 is checked there); never drop an entry silently
                    }
                    if (sp < kStack2Total) sp++;
                }
                cur = go_n ? (nenc & 0x00ffffffu) : kRefNone;
                cur_cnt = go_n ? (nenc >> 24) : 0u;
                if (!go_n && sp > 0) {  // nothing was pushed: the top read above is still the top
                    sp--;
                    if (top_tin <= t_cull) {
                        cur = top_enc & 0x00ffffffu;
                        cur_cnt = top_enc >> 24;
                    }
                }
            }
            const uint32_t n_desc = (uint32_t)__popcll(__ballot(active && cur_cnt == 0));
            if (n_desc <= (uint32_t)TH_TRACE7_LEAF_WAIT) break;
        }
        // ---- phase B: leaves ----
        if (active && cur != kRefNone && cur_cnt > 0) {
            uint32_t top_enc = kRefNone;
            float top_tin = kInf;
            if (sp > 0) {
                if (sp - 1 < kLds) {
                    top_enc = s_ref[sp - 1][tid];
                    top_tin = s_tin[sp - 1][tid];
                } else {
                    const uint2 e = overflow[(size_t)(sp - 1 - kLds) * gthreads + gtid];
                    top_enc = e.x;
                    top_tin = __uint_as_float(e.y);
                }
            }
            // one primitive against the ray (the candidate bookkeeping of the header)
            auto test_prim = [&](uint32_t slot, float4 p0, float4 p1, float4 p2) {
                const uint32_t meta = __float_as_uint(p0.w);
                if (COUNT) np++;
                float t_c = kInf;
                float4 rec = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                bool cand = false;
                if (meta & PRIM_SPHERE) {
                    const float4 d4 = rd[idx];  // rare: the direction is not kept (as k_trace3)
                    const f3 d = mk3(d4.x, d4.y, d4.z);
                    const int r = sphere_candidate7<FULL_ONLY>(sc.spheres[__float_as_uint(p0.x)], o, d, t_cull, t_c);
                    inv_d = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                    shear = ray_shear(d);
                    if (r == 2) flagged = true;
                    cand = r == 1 || r == 3;
                    // r == 3 (origin inside): whoever is tested after it is measured against ITS t, whoever came before is overwritten — unless something nearer
                    // exists, the answer is this sphere in every order.  Something nearer (found before or after) makes the order matter: flagged
                    if (r == 3) {
                        if (t_c >= t_best) flagged = true;
                        t_raise = fminf(t_raise, t_c);
                    }
                    rec = make_float4(t_c, __int_as_float((int)slot), 0.0f, 0.0f);
                } else {
                    TriTest tt;
                    if (!(meta & PRIM_DEGENERATE) && tri_intersect_sheared<true>(mk3(p0.x, p0.y, p0.z), mk3(p1.x, p1.y, p1.z), mk3(p2.x, p2.y, p2.z), o, shear, t_cull, &tt)) {
                        cand = true;
                        t_c = tt.t;
                        rec = make_float4(out.bary_mode ? tt.bary.z : tt.t, __int_as_float((int)slot), tt.bary.x, tt.bary.y);
                    }
                }
                if (cand) {
                    if (t_c > t_own - 4.0f * m) {
                        if (COUNT) why = 2u;
                        flagged = true;  // within the gap of the ray's own t_max (or beyond it): the reference's `t_scaled > t_max * det` decides, not this walk
                    } else if (t_c < t_best) {
                        t_second = fminf(t_second, t_best);
                        t_best = t_c;
                        t_cull = fminf(t_cull, t_c + 5.0f * m);
                        out.hits[idx] = rec;  // stored at once: a nearer candidate overwrites it
                    } else {
                        t_second = fminf(t_second, t_c);
                    }
                }
            };
            if (CHEAP && cur_cnt <= 2u) {
                // The reference enters this leaf iff ITS box test passes (bounds.jl:186-198; the ancestors then pass too: header) — on the leaf's box, which is the
                // union of its triangles' boxes (verified at upload: WideScene::leaf_tight).  One or two primitives (97 % of the leaves of the trees built here): both
                // records in one burst, the box from their vertices, then the tests from the same registers — one trip to memory per leaf, as k_trace3's.
                const bool two = cur_cnt == 2u;
                const uint32_t sb = two ? cur + 1u : cur;  // (cnt == 1: the second record is the first again — same line, no branch around the loads)
                const float4 a0 = sc.prims[3 * cur], a1 = sc.prims[3 * cur + 1], a2 = sc.prims[3 * cur + 2];
                const float4 b0 = sc.prims[3 * sb], b1 = sc.prims[3 * sb + 1], b2 = sc.prims[3 * sb + 2];
                asm volatile("" ::"v"(a1.x), "v"(a2.x), "v"(b0.x), "v"(b1.x), "v"(b2.x));
                const bool has_sphere = ((__float_as_uint(a0.w) | __float_as_uint(b0.w)) & PRIM_SPHERE) != 0u;
                bool enter = true;
                if (!has_sphere) {  // (a leaf that holds a sphere was reached through the exact tests of the sphere's path: nothing more to decide)
                    const float bx0 = fminf(fminf(fminf(a0.x, a1.x), a2.x), fminf(fminf(b0.x, b1.x), b2.x)), bx1 = fmaxf(fmaxf(fmaxf(a0.x, a1.x), a2.x), fmaxf(fmaxf(b0.x, b1.x), b2.x));
                    const float by0 = fminf(fminf(fminf(a0.y, a1.y), a2.y), fminf(fminf(b0.y, b1.y), b2.y)), by1 = fmaxf(fmaxf(fmaxf(a0.y, a1.y), a2.y), fmaxf(fmaxf(b0.y, b1.y), b2.y));
                    const float bz0 = fminf(fminf(fminf(a0.z, a1.z), a2.z), fminf(fminf(b0.z, b1.z), b2.z)), bz1 = fmaxf(fmaxf(fmaxf(a0.z, a1.z), a2.z), fmaxf(fmaxf(b0.z, b1.z), b2.z));
                    enter = slab_entry7(bx0, by0, bz0, bx1, by1, bz1, o, inv_d, ma, false) < kInf;
                }
                if (enter) {
                    test_prim(cur, a0, a1, a2);
                    if (two) test_prim(cur + 1u, b0, b1, b2);
                }
            } else {
                bool enter = true;
                if (CHEAP) {  // a leaf of three or more: read twice (the second time from L1)
                    float bx0 = kInf, by0 = kInf, bz0 = kInf, bx1 = -kInf, by1 = -kInf, bz1 = -kInf;
                    bool has_sphere = false;
                    for (uint32_t k = 0; k < cur_cnt; ++k) {
                        const uint32_t slot = cur + k;
                        const float4 p0 = sc.prims[3 * slot];
                        const float4 p1 = sc.prims[3 * slot + 1], p2 = sc.prims[3 * slot + 2];
                        asm volatile("" ::"v"(p1.x), "v"(p1.y), "v"(p1.z), "v"(p2.x), "v"(p2.y), "v"(p2.z));
                        has_sphere = has_sphere || (__float_as_uint(p0.w) & PRIM_SPHERE) != 0u;
                        bx0 = fminf(bx0, fminf(fminf(p0.x, p1.x), p2.x));
                        by0 = fminf(by0, fminf(fminf(p0.y, p1.y), p2.y));
                        bz0 = fminf(bz0, fminf(fminf(p0.z, p1.z), p2.z));
                        bx1 = fmaxf(bx1, fmaxf(fmaxf(p0.x, p1.x), p2.x));
                        by1 = fmaxf(by1, fmaxf(fmaxf(p0.y, p1.y), p2.y));
                        bz1 = fmaxf(bz1, fmaxf(fmaxf(p0.z, p1.z), p2.z));
                    }
                    if (!has_sphere) enter = slab_entry7(bx0, by0, bz0, bx1, by1, bz1, o, inv_d, ma, false) < kInf;
                }
                for (uint32_t k = 0; enter && k < cur_cnt; ++k) {
                    const uint32_t slot = cur + k;
                    const float4 p0 = sc.prims[3 * slot];
                    const float4 p1 = sc.prims[3 * slot + 1], p2 = sc.prims[3 * slot + 2];
                    asm volatile("" ::"v"(p1.x), "v"(p1.y), "v"(p1.z), "v"(p2.x), "v"(p2.y), "v"(p2.z));  // one burst (th_trace2.h "one fetch per leaf")
                    test_prim(slot, p0, p1, p2);
                }
            }
            cur = kRefNone;
            cur_cnt = 0;
            if (flagged) sp = 0;  // the reference-order walk decides this ray: nothing left to do here
            if (sp > 0) {  // the next stack entry against the margin the leaf left
                sp--;
                if (top_tin <= t_cull) {
                    cur = top_enc & 0x00ffffffu;
                    cur_cnt = top_enc >> 24;
                }
            }
        }
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
            for (int k = 0; k < 4; ++k) {
                const unsigned long long w = wave_sum(n_why[k]);
                if (lane_id() == 0 && w) atomicAdd(&ctr->fallback_why[k], w);
            }
        }
    }
}

}  // namespace th
