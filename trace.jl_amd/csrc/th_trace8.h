// th_trace8.h — traversal 4: intersect!(bvh, ray) / intersect_p(bvh, ray) (accel/bvh.jl:212-299) over 8-wide nodes.
//
// th_wide8.h states why this visits exactly the leaves the reference's binary walk visits, in its order, with its box and
// primitive arithmetic (leaf box: slab_test2 on the triangle's own bound; triangle: tri_intersect_sheared) — results are
// bit-identical to k_trace3 / the literal kernels / the CPU oracle.  What changes is the work per ray and its shape on a wave64:
//   * one 128-byte line yields eight child boxes: a ray of the 1 M-triangle scene fetches ~8 nodes instead of ~24, and an
//     interior box costs 6 fma + 2 three-operand min/max instead of the reference's 12 subtract / multiply + 10 selects + compares;
//   * children in the binary walk's order by a precomputed permutation per direction octant: no sort, no per-child distance;
//   * a lane is always in one of two states — "next is a node" or "next is a leaf" (chosen right after each step from a
//     per-lane group {child base, triangle base, pending mask}) — and the wave runs whichever step more lanes wait for: node
//     steps and leaf steps each run with the lanes that need them instead of interleaving pop / interior / leaf sections;
//   * stack entries are groups (12 bytes), pushed only when a node still has children pending: ~1 entry per wide level, 8 levels in LDS.
// Per-lane ray replacement, segmented queues and work cursors are those of k_trace2/3 (th_trace2.h).
#pragma once
#include "th_trace2.h"
#include "th_wide8.h"

namespace th {

#ifndef TH_TRACE8_WAVES_CLOSEST
#define TH_TRACE8_WAVES_CLOSEST 4
#endif
#ifndef TH_TRACE8_WAVES_ANY
#define TH_TRACE8_WAVES_ANY 4
#endif
#ifndef TH_TRACE8_LDS
#define TH_TRACE8_LDS 8  // stack levels (groups) per lane kept in LDS: 8 x 12 B x 256 lanes = 24 KB per block
#endif
#ifndef TH_TRACE8_REFILL
#define TH_TRACE8_REFILL 12
#endif
#ifndef TH_TRACE8_NODE_BIAS
#define TH_TRACE8_NODE_BIAS 4  // node step when 4 x (lanes at a node) >= BIAS-weighted lanes at a leaf: 4 = plain majority
#endif
constexpr int kStack8Lds = TH_TRACE8_LDS;
constexpr int kStack8Global = kW8MaxDepth > kStack8Lds ? kW8MaxDepth - kStack8Lds : 0;  // levels per thread in the global slab

struct Wide8Scene {
    const uint4* nodes;  // 8 uint4 (128 B) per node, th_wide8.h
    const float4* tris;  // 3 per triangle: v0 | ordered slot, v1 | meta, v2
    float root_box[6];   // flat node 0 (tested first, bvh.jl:226)
    float tri_box[6];    // n_sph > 0: flat node 2 n_sph, the root of the triangles' subtree
    float sph_box[kW8MaxSpheres][6];  // flat node 2 i + 1: the leaf of sphere i (ordered slot i), first child of chain node 2 i
    uint32_t n_sph;
    uint32_t chain_axis;  // split axis of every chain node (trhip_scene_commit gives them all the same one)
    float tight_scale;    // slab_test2's margin (2^-14 of the ray's reach): > 0 whenever this kernel runs
};
struct FallbackList {   // rays this kernel does not take (th_wide8.h): per-segment lists k_trace3 walks afterwards
    uint32_t* list;     // [kSeg][cap]
    uint32_t* counts;   // [kSeg * kCtrStride]
    uint32_t cap;
};

// position k in the visiting order -> slot: the 3-bit field of `iperm` (slot -> position) that equals k
TH_D uint32_t w8_slot_at(uint32_t iperm, uint32_t k) {
    const uint32_t x = (iperm ^ (k * 0x249249u)) & 0xffffffu;  // fields equal to k become 0
    const uint32_t y = (x | (x >> 1) | (x >> 2)) & 0x249249u;  // bit 3s set <=> field s != 0
    const uint32_t z = ~y & 0x249249u;
    return ((uint32_t)__builtin_ctz(z) * 11u) >> 5;            // bit index 3s -> s (s in 0..7)
}

template <bool ANY, bool COUNT, bool FULL_ONLY>
__global__ __launch_bounds__(kBlock, ANY ? TH_TRACE8_WAVES_ANY : TH_TRACE8_WAVES_CLOSEST) void k_trace8(DeviceScene sc, Wide8Scene ws, SegQueue q, const float4* __restrict__ ro,
                                                                                                      const float4* __restrict__ rd, const float* __restrict__ tmax_or_null, TraceOut out,
                                                                                                      uint32_t* __restrict__ work, uint32_t* __restrict__ overflow, Counters* ctr,
                                                                                                      FallbackList fb) {
    __shared__ uint32_t s_cb[kStack8Lds][kBlock];
    __shared__ uint32_t s_tb[kStack8Lds][kBlock];
    __shared__ uint32_t s_pm[kStack8Lds][kBlock];
    __shared__ SegView sv;
    seg_load(q, sv);
    const uint32_t tid = threadIdx.x;
    const uint32_t gthreads = gridDim.x * kBlock;
    const uint32_t gtid = blockIdx.x * kBlock + tid;
    const uint32_t lane = lane_id();
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    // K_SPH: at the chain of sphere leaves (before the subtree, or — F_POST — behind it); K_FIN: done, waiting for delivery
    enum : uint32_t { K_IDLE = 0, K_NODE = 1, K_LEAF = 2, K_SPH = 3, K_FIN = 4 };
    enum : uint32_t { F_FOUND = 1u, F_POST = 2u };

    bool exhausted = false;
    uint32_t wseg = __builtin_amdgcn_readfirstlane((gtid >> 6) % kSeg), dry = 0, pool_next = 0, pool_end = 0;  // wave-uniform
    uint32_t kind = K_IDLE, ref = 0, idx = 0, flags = 0;
    uint32_t g_cb = 0, g_tb = 0, g_pm = 0;  // current group: first interior child | first triangle + (ni << 24) | slot->position word + (pending positions << 24)
    int sp = 0;
    f3 o = splat3(0.0f), d = splat3(0.0f), inv_d = splat3(0.0f);
    float em = 0.0f, t_max = 0.0f, slot_w = 0.0f, flag_w = 0.0f;
    RayShear shear{0, 0.0f, 0.0f, 0.0f};
    bool negx = false, negy = false, negz = false;
    uint32_t nn = 0, np = 0;

    // choose what the lane does next from its group / stack: a node, a leaf, the sphere leaves behind the subtree, or delivery
    auto select_next = [&]() {
        if ((g_pm >> 24) == 0u) {
            if (sp > 0) {
                sp--;
                if (sp < kStack8Lds) {
                    g_cb = s_cb[sp][tid];
                    g_tb = s_tb[sp][tid];
                    g_pm = s_pm[sp][tid];
                } else {
                    const size_t at = (size_t)(sp - kStack8Lds) * gthreads + gtid;
                    g_cb = overflow[at];
                    g_tb = overflow[(size_t)kStack8Global * gthreads + at];
                    g_pm = overflow[2 * (size_t)kStack8Global * gthreads + at];
                }
            } else {
                kind = (flags & F_POST) ? K_SPH : K_FIN;  // the subtree is done
                return;
            }
        }
        const uint32_t k = (uint32_t)__builtin_ctz(g_pm >> 24);
        g_pm &= ~(1u << (24u + k));
        const uint32_t slot = w8_slot_at(g_pm, k);
        const uint32_t ni = g_tb >> 24;
        if (slot < ni) {
            if ((g_pm >> 24) != 0u) {  // children still pending: the group waits on the stack
                if (sp < kStack8Lds) {
                    s_cb[sp][tid] = g_cb;
                    s_tb[sp][tid] = g_tb;
                    s_pm[sp][tid] = g_pm;
                } else if (sp < kStack8Lds + kStack8Global) {
                    const size_t at = (size_t)(sp - kStack8Lds) * gthreads + gtid;
                    overflow[at] = g_cb;
                    overflow[(size_t)kStack8Global * gthreads + at] = g_tb;
                    overflow[2 * (size_t)kStack8Global * gthreads + at] = g_pm;
                }
                sp++;
            }
            kind = K_NODE;
            ref = g_cb + slot;
            g_pm = 0u;
        } else {
            kind = K_LEAF;
            ref = (g_tb & 0xffffffu) + (slot - ni);
        }
    };
    // enter the triangles' subtree (wide node 0) if its root box passes — a cull only (tight test, th_wide8.h); what comes after it otherwise
    auto enter_subtree = [&]() {
        float tb;
        if (COUNT) nn++;
        if (ws.n_sph == 0u || (slab_test2(ws.tri_box[0], ws.tri_box[1], ws.tri_box[2], ws.tri_box[3], ws.tri_box[4], ws.tri_box[5], o, inv_d, em, true, negx, negy, negz, tb) && tb < t_max)) {
            kind = K_NODE;
            ref = 0u;
            g_pm = 0u;
            sp = 0;
        } else {
            kind = (flags & F_POST) ? K_SPH : K_FIN;
        }
    };
    // The chain of sphere leaves (scenes with spheres).  Flat layout: chain node i = interior {leaf of sphere i, rest}, all with the same split
    // axis; the binary walk enters the leaf first unless dir_is_neg[axis] (bvh.jl:239-246).  A closest-hit ray therefore meets EITHER all sphere
    // leaves (i ascending) and then the triangles' subtree, OR the subtree and then the leaves (i descending).  A leaf is entered iff its box
    // passes the reference's test with the t_max of that moment (bvh.jl:226); the chain's interior boxes only cull (th_wide8.h) and are not
    // tested.  Any-hit rays (t_max constant, order free) take the leaves first.  Wave-uniform loop: the sphere records arrive by scalar loads.
    auto sphere_pass = [&](bool descending) {
        const bool mine = kind == K_SPH && (((flags & F_POST) != 0u) == descending);
        if (__ballot(mine) == 0ull) return;
        bool hit_any = false;
        for (uint32_t k = 0; k < ws.n_sph; ++k) {
            const uint32_t i = descending ? ws.n_sph - 1u - k : k;
            float tmin;
            const bool enter = mine && !hit_any &&
                               slab_test2(ws.sph_box[i][0], ws.sph_box[i][1], ws.sph_box[i][2], ws.sph_box[i][3], ws.sph_box[i][4], ws.sph_box[i][5], o, inv_d, em, false, negx, negy, negz, tmin) &&
                               tmin < t_max;
            if (COUNT && mine) nn++;
            if (__ballot(enter) == 0ull) continue;
            const float4 p0 = uniform_load(sc.prims, 3 * i);  // ordered slot i holds sphere i (trhip_scene_commit)
            const SphereRec sr = uniform_load(sc.spheres, __float_as_uint(p0.x));
            if (enter) {
                if (COUNT) np++;
                SphereHit sh;
                if (sphere_intersect<false, FULL_ONLY>(sr, o, d, t_max, sh)) {
                    if (ANY) {
                        hit_any = true;
                    } else {
                        t_max = sh.t;  // primitive.jl:17, unconditional: may RAISE t_max (A.18) — nothing of this ray is on a stack at this point
                        flags |= F_FOUND;
                        out.hits[idx] = make_float4(sh.t, __int_as_float((int)i), 0.0f, 0.0f);
                    }
                }
            }
        }
        if (mine) {
            if (descending || hit_any) {
                if (hit_any) flags |= F_FOUND;
                kind = K_FIN;
            } else {
                enter_subtree();
            }
        }
    };

    while (true) {
        // ---- event: deliver finished rays, refill, sphere leaves — for many lanes at once ----------------------------------------------------
        // A finished lane WAITS (K_FIN, or K_SPH for the sphere leaves behind the subtree) until enough lanes wait with it: delivery (a read-
        // modify-write of the radiance for shadow rays), the queue fetch and the sphere tests each cost the wave a memory round trip.
        const uint32_t n_step = (uint32_t)__popcll(__ballot(kind == K_NODE || kind == K_LEAF)), n_wait = (uint32_t)__popcll(__ballot(kind == K_FIN || kind == K_SPH));
        if (n_step == 0u || (exhausted ? n_wait : 64u - n_step) >= (uint32_t)TH_TRACE8_REFILL) {
            sphere_pass(true);  // leaves behind the subtree -> K_FIN
            if (kind == K_FIN) {
                kind = K_IDLE;
                const bool found = (flags & F_FOUND) != 0u;
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
                } else if (!found) {
                    out.hits[idx] = make_float4(kInf, __int_as_float(-1), 0.0f, 0.0f);  // a hit was stored when it was accepted
                }
            }
            // refill idle lanes from the queue (as k_trace3)
            const unsigned long long idle = __ballot(kind == K_IDLE);
            const uint32_t n_idle = (uint32_t)__popcll(idle);
            bool to_fallback = false;
            if (!exhausted && n_idle) {
                if (pool_next >= pool_end) {
                    // an empty segment, or one whose cursor already ran past its count, needs no atomic: the waves that arrive when the queue is
                    // drained (all of them, at the end of every launch) would otherwise queue 32 returning atomics each on the same 32 words
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
                if (avail && kind == K_IDLE) {
                    const uint32_t rank = (uint32_t)__popcll(idle & lt_mask);
                    if (rank < avail) {
                        idx = seg_phys(q, wseg, pool_next + rank);
                        if (q.indirect) idx = q.indirect[idx];
                        const float4 o4 = ro[idx], d4 = rd[idx];
                        o = mk3(o4.x, o4.y, o4.z);
                        d = mk3(d4.x, d4.y, d4.z);
                        slot_w = o4.w;
                        flag_w = d4.w;
                        inv_d = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                        t_max = tmax_or_null ? tmax_or_null[idx] : kInf;
                        // rays outside the argument of th_wide8.h: a (nearly) zero direction component (0 * Inf = NaN in the slab products breaks
                        // their monotonicity), non-finite input, an origin absurdly far from the scene — k_trace3 takes them
                        const float kBig = 7.9228163e28f;  // 2^96
                        const float reach = fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z));
                        if (!(fabsf(inv_d.x) < kBig) || !(fabsf(inv_d.y) < kBig) || !(fabsf(inv_d.z) < kBig) || !(reach < 2.0f * kW8MaxCoord) || !(fabsf(d.x) < kBig) || !(fabsf(d.y) < kBig) ||
                            !(fabsf(d.z) < kBig)) {
                            to_fallback = true;
                        } else {
                            em = slab_margin(ws.root_box, ws.tight_scale, o);
                            shear = ray_shear(d);
                            negx = d.x < 0.0f;
                            negy = d.y < 0.0f;
                            negz = d.z < 0.0f;
                            flags = 0u;
                            sp = 0;
                            g_pm = 0u;
                            float tmin;
                            if (COUNT) nn++;
                            kind = K_FIN;
                            if (slab_test2(ws.root_box[0], ws.root_box[1], ws.root_box[2], ws.root_box[3], ws.root_box[4], ws.root_box[5], o, inv_d, em, false, negx, negy, negz, tmin) && tmin < t_max) {
                                const bool neg = ws.chain_axis == 0u ? negx : (ws.chain_axis == 1u ? negy : negz);  // bvh.jl:239: second child (the rest) first
                                if (ws.n_sph != 0u && (ANY || !neg)) {
                                    kind = K_SPH;  // sphere leaves first (below), then the subtree
                                } else {
                                    if (ws.n_sph != 0u) flags |= F_POST;  // the subtree first, the sphere leaves when it is done
                                    enter_subtree();
                                }
                            }
                        }
                    }
                }
                pool_next += min(n_idle, avail);
            }
            {  // hand the rays this kernel does not take to the fallback list (wave-wide: every lane takes part in the ballot)
                const uint32_t j = wave_compact(to_fallback, &fb.counts[wseg * kCtrStride]);
                if (to_fallback && j < fb.cap) fb.list[wseg * fb.cap + j] = idx;
            }
            sphere_pass(false);  // fresh rays whose walk starts at the sphere leaves -> subtree (or K_FIN / the next event)
            if (__ballot(kind != K_IDLE) == 0ull) {
                if (exhausted) break;
                continue;
            }
            if (__ballot(kind == K_NODE || kind == K_LEAF) == 0ull) continue;  // only waiting lanes: the next event serves them
        }
        // ---- one step: the kind more lanes wait for -----------------------------------------------------------------------------------------
        const uint32_t m_node = (uint32_t)__popcll(__ballot(kind == K_NODE)), m_leaf = (uint32_t)__popcll(__ballot(kind == K_LEAF));
        if (m_node * (uint32_t)TH_TRACE8_NODE_BIAS >= m_leaf * 4u && m_node != 0u) {
            if (kind == K_NODE) {
                // one 128-byte line: header, eight quantised child boxes, the visiting order of this ray's octant
                const uint4* np4 = ws.nodes + 8 * (size_t)ref;
                const uint4 w0 = np4[0], w1 = np4[1], w2 = np4[2], w3 = np4[3];
                const uint2 w4 = *reinterpret_cast<const uint2*>(np4 + 4);
                const uint32_t oct = (negx ? 1u : 0u) | (negy ? 2u : 0u) | (negz ? 4u : 0u);
                const uint32_t iperm = reinterpret_cast<const uint32_t*>(np4)[18u + oct];
                if (COUNT) nn++;
                const uint32_t hdr = w0.w;
                // Child planes = p + q * 2^e.  The test is the reference's (bounds.jl:180-200) plus the two clauses it lost, clause for clause
                // as slab_test2 — evaluated on the OUTWARD-rounded box with every plane moved out by s = em / 64 (the arithmetic here differs from
                // the reference's by < 4 ulp of the ray's reach; em = 1024 ulp of it) and the lost clauses on a box grown by em + s: whatever
                // slab_test2 accepts on a leaf box inside this one, this accepts (th_wide8.h).  Folded per axis into t = q * S + A.
                const float s_ = em * 0.015625f, gt = em + s_;
                const float sx = __uint_as_float((hdr & 0xffu) << 23), sy = __uint_as_float(((hdr >> 8) & 0xffu) << 23), sz = __uint_as_float(((hdr >> 16) & 0xffu) << 23);
                const float Sx = inv_d.x * sx, Sy = inv_d.y * sy, Sz = inv_d.z * sz;
                const float ax = __uint_as_float(w0.x) - o.x, ay = __uint_as_float(w0.y) - o.y, az = __uint_as_float(w0.z) - o.z;
                const float gx = negx ? -s_ : s_, gy = negy ? -s_ : s_, gz = negz ? -s_ : s_;
                const float Anx = (ax - gx) * inv_d.x, Any = (ay - gy) * inv_d.y, Anz = (az - gz) * inv_d.z;  // entry side
                const float Afx = (ax + gx) * inv_d.x, Afy = (ay + gy) * inv_d.y, Afz = (az + gz) * inv_d.z;  // exit side
                const float cx = gt * fabsf(inv_d.x), cy = gt * fabsf(inv_d.y), cz = gt * fabsf(inv_d.z);   // the lost clauses' margins in t units
                // near / far plane bytes by the sign of the direction: qlo_x = w1.zw, qlo_y = w2.xy, qlo_z = w2.zw, qhi_x = w3.xy, qhi_y = w3.zw, qhi_z = w4.xy
                const uint32_t nx0 = negx ? w3.x : w1.z, nx1 = negx ? w3.y : w1.w, fx0 = negx ? w1.z : w3.x, fx1 = negx ? w1.w : w3.y;
                const uint32_t ny0 = negy ? w3.z : w2.x, ny1 = negy ? w3.w : w2.y, fy0 = negy ? w2.x : w3.z, fy1 = negy ? w2.y : w3.w;
                const uint32_t nz0 = negz ? w4.x : w2.z, nz1 = negz ? w4.y : w2.w, fz0 = negz ? w2.z : w4.x, fz1 = negz ? w2.w : w4.y;
                uint32_t om = 0u;
#define TH_W8_CHILD(S, NXW, NYW, NZW, FXW, FYW, FZW, SH)                                                                                          \
    {                                                                                                                                             \
        const float ex_ = __fmaf_rn((float)(((NXW) >> (SH)) & 0xffu), Sx, Anx), ey_ = __fmaf_rn((float)(((NYW) >> (SH)) & 0xffu), Sy, Any),       \
                    ez_ = __fmaf_rn((float)(((NZW) >> (SH)) & 0xffu), Sz, Anz);                                                                    \
        const float xx_ = __fmaf_rn((float)(((FXW) >> (SH)) & 0xffu), Sx, Afx), xy_ = __fmaf_rn((float)(((FYW) >> (SH)) & 0xffu), Sy, Afy),       \
                    xz_ = __fmaf_rn((float)(((FZW) >> (SH)) & 0xffu), Sz, Afz);                                                                    \
        const float a_ = fmaxf(ex_, ey_);                     /* bounds.jl:189 */                                                                  \
        const float bmin_ = fminf(xx_, xy_), bmax_ = fmaxf(xx_, xy_); /* :190 keeps the LARGER exit (A.17); the smaller one is the lost clause */ \
        const float tin_ = fmaxf(a_, ez_);                    /* :196 */                                                                           \
        const float bt_ = fminf(xx_ + cx, xy_ + cy);                                                                                               \
        /* a NaN never rejects (every comparison below is false on NaN), as in the reference */                                                   \
        const bool miss_ = (a_ > bmin_) | (a_ > xz_) | (ez_ > bmax_) | (ez_ - cz > bt_) | (bt_ < 0.0f) | (fminf(xz_, bmax_) < 0.0f) | (tin_ >= t_max); \
        om |= (miss_ ? 0u : 1u) << ((iperm >> (3 * (S))) & 7u);                                                                                   \
    }
                TH_W8_CHILD(0, nx0, ny0, nz0, fx0, fy0, fz0, 0)
                TH_W8_CHILD(1, nx0, ny0, nz0, fx0, fy0, fz0, 8)
                TH_W8_CHILD(2, nx0, ny0, nz0, fx0, fy0, fz0, 16)
                TH_W8_CHILD(3, nx0, ny0, nz0, fx0, fy0, fz0, 24)
                TH_W8_CHILD(4, nx1, ny1, nz1, fx1, fy1, fz1, 0)
                TH_W8_CHILD(5, nx1, ny1, nz1, fx1, fy1, fz1, 8)
                TH_W8_CHILD(6, nx1, ny1, nz1, fx1, fy1, fz1, 16)
                TH_W8_CHILD(7, nx1, ny1, nz1, fx1, fy1, fz1, 24)
#undef TH_W8_CHILD
                om &= (1u << (hdr >> 28)) - 1u;  // positions of empty slots lie behind the n children
                g_cb = w1.x;
                g_tb = w1.y | (((hdr >> 24) & 0xfu) << 24);
                g_pm = (iperm & 0xffffffu) | (om << 24);
                select_next();
            }
        } else {
            if (kind == K_LEAF) {
                // the leaf's primitive; its box is the triangle's own bound (world_bound, triangle_mesh.jl:97): min / max are exact
                const float4 p0 = ws.tris[3 * (size_t)ref], p1 = ws.tris[3 * (size_t)ref + 1], p2 = ws.tris[3 * (size_t)ref + 2];
                if (COUNT) np++;
                const float bx0 = fminf(fminf(p0.x, p1.x), p2.x), by0 = fminf(fminf(p0.y, p1.y), p2.y), bz0 = fminf(fminf(p0.z, p1.z), p2.z);
                const float bx1 = fmaxf(fmaxf(p0.x, p1.x), p2.x), by1 = fmaxf(fmaxf(p0.y, p1.y), p2.y), bz1 = fmaxf(fmaxf(p0.z, p1.z), p2.z);
                float tmin;
                // the reference's box test on the leaf (bounds.jl:180-200 + the clauses it lost, th_trace2.h) with the t_max of NOW: bvh.jl:226 when the leaf is popped
                if (slab_test2(bx0, by0, bz0, bx1, by1, bz1, o, inv_d, em, true, negx, negy, negz, tmin) && tmin < t_max) {
                    const uint32_t meta = __float_as_uint(p1.w);
                    TriTest tt;
                    if (!(meta & PRIM_DEGENERATE) && tri_intersect_sheared<!ANY>(mk3(p0.x, p0.y, p0.z), mk3(p1.x, p1.y, p1.z), mk3(p2.x, p2.y, p2.z), o, shear, t_max, &tt)) {
                        flags |= F_FOUND;
                        if (ANY) {  // intersect_p returns at the first accepted primitive (bvh.jl:283-287)
                            sp = 0;
                            g_pm = 0u;
                        } else {
                            t_max = tt.t;
                            out.hits[idx] = make_float4(out.bary_mode ? tt.bary.z : tt.t, p0.w, tt.bary.x, tt.bary.y);  // stored at once: a later accepted hit overwrites it
                        }
                    }
                }
                select_next();
            }
        }
    }
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
