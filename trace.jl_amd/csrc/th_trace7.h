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
// gap = 4 m, cull margin = 5 m with m = 2^-14 x (largest coordinate offset between the ray origin and the scene bound) x max |1 / d|: the slack
// of the box entry against a hit inside the box (<= m, the tight clauses' margin) plus the rounding of `t_max * det` and `ts * (1 / det)`.
// A candidate culled by a box (entry > t_best + 5 m) has t > t_best + 4 m >= t_w + gap: nothing inside the gap is ever skipped.
#pragma once
#include "th_trace2.h"
#include "th_trace8.h"  // FallbackList

namespace th {

#ifndef TH_TRACE7_WAVES
#define TH_TRACE7_WAVES 6
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
TH_D float slab_entry7(float bx0, float by0, float bz0, float bx1, float by1, float bz1, f3 o, f3 inv_d, float m, bool tight) {
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
    const float exit_xy = fminf(hix, hiy);
    const bool miss_tight = tight & ((loz > exit_xy + (m + m)) | (exit_xy < -m));
    const bool hit = !(miss_xy | miss_z | miss_tight) & (t_out > 0.0f);  // :198 without `t_in < t_max` (the caller's, with its margin)
    return hit ? t_in : kInf;
}

// CHEAP interior boxes: the standard slab test, one fma per plane (t = plane / d - o / d), on the box grown by m; conservative for every box that holds a hit
// the primitive tests accept (th_trace2.h, slab_test2's argument) — nothing else is asked of it.
TH_D float slab_entry_cheap(float bx0, float by0, float bz0, float bx1, float by1, float bz1, f3 inv_d, f3 noid, float m) {
    const float x0 = __fmaf_rn(bx0, inv_d.x, noid.x), x1 = __fmaf_rn(bx1, inv_d.x, noid.x);
    const float y0 = __fmaf_rn(by0, inv_d.y, noid.y), y1 = __fmaf_rn(by1, inv_d.y, noid.y);
    const float z0 = __fmaf_rn(bz0, inv_d.z, noid.z), z1 = __fmaf_rn(bz1, inv_d.z, noid.z);
    const float t_in = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
    const float t_out = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
    const bool hit = (t_in <= t_out + (m + m)) & (t_out >= -m);  // a NaN (the empty leaf's box) fails both
    return hit ? t_in : kInf;
}

// sphere.jl:125-158 up to the roots, as sphere_intersect (th_device.h); 0 = no candidate inside t_lim, 1 = candidate at t, 2 = the ray must be re-traced
// in the reference's order (origin inside the sphere: t_max is ignored, A.18; a clipped sphere; the limb, where the quadratic accepts rays outside the box)
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
    if (t0 > t_lim || t1 < 0.0f) return 0;
    if (t0 < 0.0f) return 2;
    if (!FULL_ONLY && !s.never_clipped) return 2;
    const float disc = b * b - 4 * a * c;
    if (disc < 0.01f * (b * b)) return 2;  // chord shorter than a tenth of the diameter's: the limb
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
    bool cur_exact = true;   // CHEAP: the node in `cur` was reached through the reference's exact box tests (a sphere's path); a triangle leaf reached otherwise tests its own box first
    RayShear shear{0, 0.0f, 0.0f, 0.0f};
    float m = 0.0f;         // the margin in t units
    float t_best = kInf;    // smallest candidate t so far (the hit record in out.hits belongs to it)
    float t_second = kInf;  // smallest t among the other candidates seen
    float t_cull = kInf;    // min(t_best, the ray's own t_max) + 5 m: boxes entered beyond it and candidates beyond it do not matter
    float t_own = kInf;     // the ray's own t_max
    bool flagged = false;
    uint32_t nn = 0, np = 0;
    unsigned long long n_fb = 0;

    while (true) {
        // ---- rays to re-trace in the reference's order: appended to the fallback list (wave-wide) ----
        if (__ballot(to_fb) != 0ull) {
            const uint32_t j = wave_compact(to_fb, &fb.counts[0]);
            if (to_fb) {
                fb.list[j] = fb_idx;
                n_fb++;
            }
            to_fb = false;
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
                        m = em * fmaxf(fmaxf(fabsf(inv_d.x), fabsf(inv_d.y)), fabsf(inv_d.z));
                        shear = ray_shear(d);
                        if (CHEAP) noid = mk3(-(o.x * inv_d.x), -(o.y * inv_d.y), -(o.z * inv_d.z));
                        cur_exact = true;
                        t_own = tmax_or_null ? tmax_or_null[idx] : kInf;
                        t_best = t_second = kInf;
                        t_cull = t_own + 5.0f * m;
                        sp = 0;
                        flagged = false;
                        active = true;
                        if (COUNT) nn++;
                        // what the argument above does not cover goes to k_trace3 at once: a zero or non-finite direction component (0 x Inf = NaN in the slab
                        // products; the reference's selects and this kernel's min / max then disagree), a non-finite origin or margin
                        const bool plain = d.x != 0.0f && d.y != 0.0f && d.z != 0.0f && m < kInf && m == m && fabsf(o.x) < kInf && fabsf(o.y) < kInf && fabsf(o.z) < kInf &&
                                           fabsf(inv_d.x) < kInf && fabsf(inv_d.y) < kInf && fabsf(inv_d.z) < kInf;
                        if (!plain) {
                            to_fb = true;
                            fb_idx = idx;
                            active = false;
                        } else {
                            const float t_in = ws.root_ref != kRefNone
                                                   ? slab_entry7(ws.root_box[0], ws.root_box[1], ws.root_box[2], ws.root_box[3], ws.root_box[4], ws.root_box[5], o, inv_d, m, false)
                                                   : kInf;
                            if (t_in <= t_cull) {
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
                        if (CHEAP) cur_exact = tin == -kInf;
                        finished = false;
                        break;
                    }
                }
            }
            if (finished) {  // the walk is over: a miss, a clean hit (stored when it was found), or a ray for the reference-order walk
                active = false;
                const bool found = t_best < kInf;
                if (flagged || (found && t_second <= t_best + 4.0f * m)) {
                    to_fb = true;
                    fb_idx = idx;
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
                    tl = slab_entry_cheap(a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, inv_d, noid, m);
                    tr = slab_entry_cheap(a1.z, a1.w, a2.x, a2.y, a2.z, a2.w, inv_d, noid, m);
                    if (meta & 12u) {  // a child on a sphere's path (rare): the reference's exact test, no tight clauses
                        if (meta & 4u) tl = slab_entry7(a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, o, inv_d, m, false);
                        if (meta & 8u) tr = slab_entry7(a1.z, a1.w, a2.x, a2.y, a2.z, a2.w, o, inv_d, m, false);
                    }
                } else {
                    tl = slab_entry7(a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, o, inv_d, m, !(meta & 4u));
                    tr = slab_entry7(a1.z, a1.w, a2.x, a2.y, a2.z, a2.w, o, inv_d, m, !(meta & 8u));
                }
                // A subtree that holds a sphere is never culled by distance: the Float32 quadratic accepts rays that pass OUTSIDE the sphere's box (by up to
                // 1e-3 |o - c|), and bounds.jl:190's loose test can pass such a box with an entry far beyond the sphere's t — the reference reaches that leaf
                // or not depending on its order.  Entered whenever the reference's t_max-free clauses pass (entry -Inf), those rays get flagged at the leaf.
                if ((meta & 4u) && tl < kInf) tl = -kInf;
                if ((meta & 8u) && tr < kInf) tr = -kInf;
                const bool l_first = tl <= tr;  // front to back by entry distance
                const float tn = l_first ? tl : tr, tf = l_first ? tr : tl;
                const uint32_t nenc = l_first ? lenc : renc, fenc = l_first ? renc : lenc;
                const bool go_n = tn <= t_cull, go_f = tf <= t_cull;  // tn <= tf: go_f implies go_n
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
                if (CHEAP) cur_exact = tn == -kInf;
                if (!go_n && sp > 0) {  // nothing was pushed: the top read above is still the top
                    sp--;
                    if (top_tin <= t_cull) {
                        cur = top_enc & 0x00ffffffu;
                        cur_cnt = top_enc >> 24;
                        if (CHEAP) cur_exact = top_tin == -kInf;
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
            bool enter = true;
            if (CHEAP && !cur_exact) {
                // the reference enters this leaf iff ITS box test passes (bounds.jl:186-198; the ancestors then pass too: header) — on the leaf's box, which is the
                // union of its triangles' boxes (verified at upload: WideScene::leaf_tight).  A leaf of several triangles reads them twice (the second time from L1).
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
                if (has_sphere)
                    flagged = true;  // cannot happen (a sphere's leaf is reached through exact tests); never decide such a leaf from a triangle's box
                else
                    enter = slab_entry7(bx0, by0, bz0, bx1, by1, bz1, o, inv_d, m, false) < kInf;
            }
            for (uint32_t k = 0; enter && k < cur_cnt; ++k) {
                const uint32_t slot = cur + k;
                const float4 p0 = sc.prims[3 * slot];
                const float4 p1 = sc.prims[3 * slot + 1], p2 = sc.prims[3 * slot + 2];
                asm volatile("" ::"v"(p1.x), "v"(p1.y), "v"(p1.z), "v"(p2.x), "v"(p2.y), "v"(p2.z));  // one burst (th_trace2.h "one fetch per leaf")
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
                    cand = r == 1;
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
            }
            cur = kRefNone;
            cur_cnt = 0;
            if (flagged) sp = 0;  // the reference-order walk decides this ray: nothing left to do here
            if (sp > 0) {  // the next stack entry against the margin the leaf left
                sp--;
                if (top_tin <= t_cull) {
                    cur = top_enc & 0x00ffffffu;
                    cur_cnt = top_enc >> 24;
                    if (CHEAP) cur_exact = top_tin == -kInf;
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
        }
    }
}

}  // namespace th
