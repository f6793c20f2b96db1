// th_leaf2.h — one-leaf scenes (S-cornell, the shadows scene: th_bvh.h commits scenes of <= tiny_scene_prims primitives as ONE leaf): rays grouped by the primitives
// they can possibly hit, so that a wave runs the reference's tests only for those.
//
// k_trace_leaf (th_trace2.h) walks the leaf with a wave-uniform index: all 64 rays of a wave test all primitives (12 in the Cornell box: ~1 400 VALU instructions per
// ray, 85 % of them on primitives the ray cannot hit).  Skipping a primitive for a LANE saves nothing — the wave still issues the test for its other lanes — and the
// rays of a wave (bounce rays) go everywhere.  Here a block takes 1 024 queue entries at a time and
//   1. gives every ray a candidate mask: bit k = "primitive k of the leaf may accept this ray".  Triangles: the one-fma-per-plane slab test on the triangle's box
//      grown by the margin of th_trace2.h's tight clauses (the hit point the reference's triangle test accepts lies within 40 ulps of the ray's reach of the
//      triangle, hence inside that box: slab_test2's argument, with the same margin).  Spheres: the reference's own arithmetic up to the discriminant of its
//      quadratic (sphere.jl:125-136): `discriminant < 0` is where it returns "no hit" — no bound on what the Float32 quadratic accepts is needed;
//   2. sorts the tile's entries by that mask (a counting sort over a hash of it, in LDS);
//   3. lets every wave take 64 sorted entries and walk the leaf in slot order, running a primitive's test only when some lane's mask has its bit, and only on
//      those lanes.
// Every ray still meets the primitives that can accept it in the reference's order with the reference's t_max; the ones it skips are ones whose test returns
// "no hit" whatever t_max is.  Bit-identical to k_trace_leaf / k_any_leaf (parity tests, traversal 1 vs 2 / 3).
// MEASURED (round 3, profiles/r3/r3m_leaf_sorted_ab.txt) and NOT adopted: S-cornell closest-hit 49.2 -> 60.2 ms, any-hit 15.9 -> 30.8 ms.  A candidate test costs a
// third of the exact test it may save (the divisions, the scalar box loads and the mask bookkeeping included), twelve of them plus the ~5 exact tests a sorted
// wave still runs plus the sort come to what testing all twelve costs; the any-hit kernel, which stops at the first of three solid-angle-ordered primitives for
// most rays, never tested twelve in the first place.  Option "leaf_sorted" (default 0) keeps it reachable for the tests.
#pragma once
#include "th_trace2.h"
#include "th_trace7.h"  // slab_entry_cheap

namespace th {

constexpr uint32_t kLeafTile = 1024;   // queue entries a block sorts at a time
constexpr uint32_t kLeafBins = 128;

// the reference's sphere test up to the point where it first says "no hit" independently of t_max (sphere.jl:125-136; sphere_intersect, th_device.h: same operations in
// the same order, hence the same discriminant)
TH_D bool sphere_has_roots(const SphereRec& s, f3 o, f3 d) {
    const f3 oo = xf_point(s.o2w_inv, o);
    const f3 od = xf_vec(s.o2w_inv, d);
    const float nd = norm(od);
    const float a = nd * nd;
    const float b = dot(2.0f * oo, od);
    const float no = norm(oo);
    const float c = no * no - s.radius * s.radius;
    const float disc = b * b - 4 * a * c;
    return !(disc < 0);
}

template <bool ANY, bool COUNT, bool FULL_ONLY>
__global__ __launch_bounds__(kBlock, 4) void k_leaf_sorted(DeviceScene sc, WideScene ws, SegQueue q, const float4* __restrict__ ro, const float4* __restrict__ rd, const float* __restrict__ tmax_or_null,
                                                           TraceOut out, Counters* ctr, const float* __restrict__ leaf_boxes) {
    __shared__ SegView sv;
    __shared__ uint32_t s_idx[kLeafTile];     // ray index of the tile's entry, 0xffffffff = padding
    __shared__ uint32_t s_mask[kLeafTile];    // its candidate mask; bit 31 = "the root box test passed" (bvh.jl:226)
    __shared__ uint16_t s_sorted[kLeafTile];  // entries grouped by mask
    __shared__ uint32_t s_cnt[kLeafBins], s_off[kLeafBins];
    seg_load(q, sv);
    const uint32_t total = sv.prefix[kSeg];
    const uint32_t first = ws.root_ref, cnt = ws.root_cnt;  // cnt <= 30 (the launcher's condition)
    const uint32_t tid = threadIdx.x, lane = lane_id(), wv = tid >> 6;
    const uint32_t full_mask = cnt >= 32u ? 0x7fffffffu : ((1u << cnt) - 1u);
    uint32_t nn = 0, np = 0;
    const uint32_t n_tiles = (total + kLeafTile - 1) / kLeafTile;
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        if (tid < kLeafBins) s_cnt[tid] = 0;
        __syncthreads();
        // ---- 1. candidate masks ----
        uint32_t rank[kLeafTile / kBlock], bin[kLeafTile / kBlock];
#pragma unroll 1
        for (uint32_t j = 0; j < kLeafTile / kBlock; ++j) {
            const uint32_t e = j * kBlock + tid;
            const uint32_t flat = tile * kLeafTile + e;
            uint32_t idx = 0xffffffffu, mask = 0;
            if (flat < total) {  // `flat & ~63` is wave-uniform: a wave's 64 entries lie in one segment (segments are padded to kSegGran)
                uint32_t seg, lb;
                seg_locate(sv, flat & ~63u, seg, lb);
                const uint32_t local = lb + (flat & 63u);
                if (local < sv.count[seg]) idx = seg_phys(q, seg, local);
            }
            if (idx != 0xffffffffu) {
                const float4 o4 = ro[idx], d4 = rd[idx];
                const f3 o = mk3(o4.x, o4.y, o4.z), d = mk3(d4.x, d4.y, d4.z);
                const float t_max = tmax_or_null ? tmax_or_null[idx] : kInf;
                const f3 inv_d = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                float tmin;
                if (COUNT) nn++;
                const bool live = slab_test2(ws.root_box[0], ws.root_box[1], ws.root_box[2], ws.root_box[3], ws.root_box[4], ws.root_box[5], o, inv_d, 0.0f, false, d.x < 0.0f, d.y < 0.0f, d.z < 0.0f, tmin) &&
                                  tmin < t_max;
                const float em = slab_margin(ws.root_box, ws.tight_scale, o);
                const f3 noid = mk3(-(o.x * inv_d.x), -(o.y * inv_d.y), -(o.z * inv_d.z));
                const f3 ma = mk3(em * fabs_(inv_d.x), em * fabs_(inv_d.y), em * fabs_(inv_d.z));
                if (live) {
                    mask = 0x80000000u;
                    // no margin (option slab_margin_log2 = 0) or a direction with a zero component (0 x Inf in the plane distances): every primitive is a candidate
                    if (!(em > 0.0f) || d.x == 0.0f || d.y == 0.0f || d.z == 0.0f) mask |= full_mask;
                }
                // the loop below is wave-uniform (scalar loads); lanes that are not live, or take everything, just do not look at the result
                const bool want = mask == 0x80000000u;
                if (__ballot(want) != 0ull) {
#pragma unroll 1
                    for (uint32_t k = 0; k < cnt; ++k) {
                        const uint32_t slot = first + k;
                        const uint32_t meta = __float_as_uint(uniform_load(sc.prims, 3 * slot).w);
                        bool cand;
                        if (meta & PRIM_SPHERE) {
                            const SphereRec sr = uniform_load(sc.spheres, __float_as_uint(uniform_load(sc.prims, 3 * slot).x));
                            cand = sphere_has_roots(sr, o, d);
                        } else if (meta & PRIM_DEGENERATE) {
                            cand = false;  // is_degenerate (triangle_mesh.jl:190): the reference's test returns at once
                        } else {
                            const float b0 = uniform_load(leaf_boxes, 6 * k), b1 = uniform_load(leaf_boxes, 6 * k + 1), b2 = uniform_load(leaf_boxes, 6 * k + 2), b3 = uniform_load(leaf_boxes, 6 * k + 3),
                                        b4 = uniform_load(leaf_boxes, 6 * k + 4), b5 = uniform_load(leaf_boxes, 6 * k + 5);
                            cand = slab_entry_cheap(b0, b1, b2, b3, b4, b5, inv_d, noid, ma) < kInf;
                        }
                        if (want && cand) mask |= 1u << k;
                    }
                }
            }
            s_idx[e] = idx;
            s_mask[e] = mask;
            bin[j] = ((mask & 0x7fffffffu) * 2654435761u) >> 25;
            rank[j] = atomicAdd(&s_cnt[bin[j]], 1u);
        }
        __syncthreads();
        // ---- 2. offsets of the bins (128 counters: two waves, one shuffle scan each) ----
        if (tid < kLeafBins) {
            const uint32_t c = s_cnt[tid];
            uint32_t incl = c;
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t up = (uint32_t)__shfl_up((int)incl, off);
                if (lane >= (uint32_t)off) incl += up;
            }
            s_off[tid] = incl - c;                 // exclusive inside the wave
            if (lane == 63u) s_cnt[tid] = incl;   // the wave's total, read below (slot 63 / 127)
        }
        __syncthreads();
        if (tid >= 64u && tid < kLeafBins) s_off[tid] += s_cnt[63];
        __syncthreads();
        // ---- 3. entries grouped by mask ----
        for (uint32_t j = 0; j < kLeafTile / kBlock; ++j) s_sorted[s_off[bin[j]] + rank[j]] = (uint16_t)(j * kBlock + tid);
        __syncthreads();
        // ---- 4. the leaf, in slot order, for 64 entries of (mostly) one mask at a time ----
#pragma unroll 1
        for (uint32_t r = 0; r < kLeafTile / kBlock; ++r) {
            const uint32_t e = s_sorted[(r * (kBlock / 64) + wv) * 64u + lane];
            const uint32_t idx = s_idx[e];
            uint32_t mask = s_mask[e];
            const bool valid = idx != 0xffffffffu;
            if (__ballot(valid) == 0ull) continue;
            float4 o4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), d4 = make_float4(0.0f, 0.0f, 1.0f, 0.0f);
            if (valid) {
                o4 = ro[idx];
                d4 = rd[idx];
            }
            const f3 o = mk3(o4.x, o4.y, o4.z), d = mk3(d4.x, d4.y, d4.z);
            float t_max = (valid && tmax_or_null) ? tmax_or_null[idx] : kInf;
            const RayShear shear = ray_shear(d);
            bool found = false;
            int hit_prim = -1;
            float hx = 0.0f, b1 = 0.0f, b2 = 0.0f;
            mask &= 0x7fffffffu;
            // the primitives some lane of the wave wants, lowest slot first
            uint32_t wave_mask;
            {
                uint32_t m = mask;
                for (int off = 32; off > 0; off >>= 1) m |= (uint32_t)__shfl_xor((int)m, off);
                wave_mask = (uint32_t)__builtin_amdgcn_readfirstlane((int)m);
            }
#pragma unroll 1
            while (wave_mask) {
                const uint32_t k = (uint32_t)__builtin_ctz(wave_mask);
                wave_mask &= wave_mask - 1u;
                const bool mine = (mask >> k) & 1u;
                if (ANY && __ballot(mine) == 0ull) continue;  // the lanes that wanted it have found their hit meanwhile
                const uint32_t slot = first + k;
                const float4 p0 = uniform_load(sc.prims, 3 * slot);
                const uint32_t meta = __float_as_uint(p0.w);
                if (COUNT && lane == 0) np++;
                if (meta & PRIM_SPHERE) {
                    const SphereRec sr = uniform_load(sc.spheres, __float_as_uint(p0.x));
                    if (mine) {
                        SphereHit sh;
                        if (sphere_intersect<false, FULL_ONLY>(sr, o, d, t_max, sh)) {
                            found = true;
                            if (ANY) {
                                mask = 0;
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
                    if (mine) {
                        TriTest tt;
                        if (tri_intersect_sheared<!ANY>(mk3(p0.x, p0.y, p0.z), mk3(p1.x, p1.y, p1.z), mk3(p2.x, p2.y, p2.z), o, shear, t_max, &tt)) {
                            found = true;
                            if (ANY) {
                                mask = 0;
                            } else {
                                t_max = tt.t;
                                hit_prim = (int)slot;
                                b1 = tt.bary.x;
                                b2 = tt.bary.y;
                                hx = out.bary_mode ? tt.bary.z : tt.t;
                            }
                        }
                    }
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
        __syncthreads();  // the next tile reuses the arrays
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

}  // namespace th
