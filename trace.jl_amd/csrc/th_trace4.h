// tracehip — k_trace4: k_trace3 with TWO rays per lane (option "traversal" = 6).
//
// k_trace3 keeps ~26 of 64 lanes busy per VALU instruction: in an interior step the lanes that hold a leaf wait (43 of 64 at work),
// in the leaf phase the lanes that are still descending wait (22 of 64).  Here every lane owns two rays, each with its own state
// (registers) and stack (LDS / global slab), and in every step works on whichever of the two can take that step: an interior step
// finds a descending ray in a lane unless BOTH its rays hold leaves, the leaf phase finds a leaf unless both descend.  Each ray
// still performs exactly the operations k_trace3 performs on it, in the same order (th_trace2.h: slab_test2, tri_intersect_sheared,
// sphere_intersect, near child first, far child's tx_min on the stack and tested at pop time, later equal-t hit wins) — only the
// interleaving changes — so the results are the same bit for bit (parity tests run traversal 6 beside 1, 2, 3, 4).
// What it costs: the working copy of the chosen ray's state is selected / written back with v_cndmask, and two stacks per lane
// halve the waves an LDS budget holds.
//
// MEASURED, and why it is not the default (profiles/r2/r2q_trace4_pmc_summary.txt, 8 M rays x 3 ray sets on the 1 M-triangle scene):
// lanes per VALU instruction 26.9 -> 32.7 as intended (+21 %), but the wave-level VALU instruction count per launch did not drop
// (7.81e8 -> 8.03e8): the selects and write-backs add a quarter to the work of every step, which is what the better packing saves;
// with 4 waves per SIMD instead of 5 (127 VGPRs, 2 x 8 stack levels in LDS) the S-mesh closest-hit pass takes 87 ms against
// k_trace3's 75 (64 spp; 3 waves / 10 levels: 104 ms).  Kept as a tested, bit-exact option, like traversal 4.
#pragma once
#include "th_trace2.h"

namespace th {

#ifndef TH_TRACE4_WAVES
#define TH_TRACE4_WAVES 4
#endif
#ifndef TH_TRACE4_LDS
#define TH_TRACE4_LDS 8  // stack levels per ray in LDS: 2 rays x levels x 2 KB per block of 256 lanes
#endif
#ifndef TH_TRACE4_LEAF_WAIT
#define TH_TRACE4_LEAF_WAIT 32  // phase A ends when at most this many lanes still have a ray that can descend
#endif
#ifndef TH_TRACE4_MAX_A
#define TH_TRACE4_MAX_A 8
#endif
#ifndef TH_TRACE4_REFILL
#define TH_TRACE4_REFILL 32  // idle ray slots (of 128) that trigger a refill
#endif

// levels per resident thread of the global stack slab: what k_trace2 / k_trace3 need of it, or two rays' worth for k_trace4
static_assert(kStackSlabLevels >= 2 * (kStack2Total - TH_TRACE4_LDS), "the overflow slab (th_trace2.h kStackSlabLevels) holds two rays' levels per lane");

struct Ray4 {  // one of a lane's two rays
    uint32_t idx, cur, cur_cnt;
    int sp;
    f3 o, inv_d;
    float em, t_max, slot_w, flag_w;
    RayShear shear;
    uint32_t neg;  // bit 0 / 1 / 2: d.x / d.y / d.z < 0
    bool found, active;
};

template <bool ANY, bool COUNT, bool FULL_ONLY>
__global__ __launch_bounds__(kBlock, TH_TRACE4_WAVES) void k_trace4(DeviceScene sc, WideScene ws, SegQueue q, const float4* __restrict__ ro, const float4* __restrict__ rd, const float* __restrict__ tmax_or_null,
                                                                  TraceOut out, uint32_t* __restrict__ work, uint2* __restrict__ overflow /* 2 x the k_trace3 slab */, Counters* ctr) {
    constexpr int kLds = TH_TRACE4_LDS;
    __shared__ uint32_t s_ref[2][kLds][kBlock];
    __shared__ float s_tmin[2][kLds][kBlock];
    __shared__ SegView sv;
    seg_load(q, sv);
    const uint32_t tid = threadIdx.x;
    const uint32_t gthreads = gridDim.x * kBlock;
    const uint32_t gtid = blockIdx.x * kBlock + tid;
    const uint32_t lane = lane_id();
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const bool tight_on = ws.tight_scale > 0.0f;

    Ray4 R0, R1;
    R0.idx = R1.idx = 0;
    R0.cur = R1.cur = kRefNone;
    R0.cur_cnt = R1.cur_cnt = 0;
    R0.sp = R1.sp = 0;
    R0.o = R1.o = R0.inv_d = R1.inv_d = splat3(0.0f);
    R0.em = R1.em = R0.t_max = R1.t_max = R0.slot_w = R1.slot_w = R0.flag_w = R1.flag_w = 0.0f;
    R0.shear = R1.shear = RayShear{0, 0.0f, 0.0f, 0.0f};
    R0.neg = R1.neg = 0;
    R0.found = R1.found = R0.active = R1.active = false;

    bool exhausted = false;
    uint32_t wseg = __builtin_amdgcn_readfirstlane((gtid >> 6) % kSeg), dry = 0, pool_next = 0, pool_end = 0;  // wave-uniform
    uint32_t nn = 0, np = 0;

    // stack entry `level` of ray slot r of this lane
    auto stack_read = [&](uint32_t r, int level, uint32_t& enc, float& tm) {
        if (level < kLds) {
            enc = s_ref[r][level][tid];
            tm = s_tmin[r][level][tid];
        } else {
            const uint2 e = overflow[((size_t)(level - kLds) * 2u + r) * gthreads + gtid];
            enc = e.x;
            tm = __uint_as_float(e.y);
        }
    };
    auto stack_write = [&](uint32_t r, int level, uint32_t enc, float tm) {
        if (level < kLds) {
            s_ref[r][level][tid] = enc;
            s_tmin[r][level][tid] = tm;
        } else {
            overflow[((size_t)(level - kLds) * 2u + r) * gthreads + gtid] = make_uint2(enc, __float_as_uint(tm));
        }
    };
    // a new ray into slot R (as k_trace3's refill)
    auto load_ray = [&](Ray4& R, uint32_t idx) {
        const float4 o4 = ro[idx], d4 = rd[idx];
        R.idx = idx;
        R.o = mk3(o4.x, o4.y, o4.z);
        const f3 d = mk3(d4.x, d4.y, d4.z);
        R.slot_w = o4.w;
        R.flag_w = d4.w;
        R.inv_d = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        R.em = slab_margin(ws.root_box, ws.tight_scale, R.o);
        R.shear = ray_shear(d);
        R.neg = (d.x < 0.0f ? 1u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 4u : 0u);
        R.t_max = tmax_or_null ? tmax_or_null[idx] : kInf;
        R.sp = 0;
        R.found = false;
        R.active = true;
        float tmin;
        if (COUNT) nn++;
        if (ws.root_ref != kRefNone &&
            slab_test2(ws.root_box[0], ws.root_box[1], ws.root_box[2], ws.root_box[3], ws.root_box[4], ws.root_box[5], R.o, R.inv_d, R.em, false, (R.neg & 1u) != 0, (R.neg & 2u) != 0, (R.neg & 4u) != 0, tmin) &&
            tmin < R.t_max) {
            R.cur = ws.root_ref;
            R.cur_cnt = ws.root_cnt;
        } else {
            R.cur = kRefNone;
            R.cur_cnt = 0;
        }
    };
    // the ray is done: deliver (as k_trace3)
    auto deliver = [&](uint32_t idx, float slot_w, float flag_w, bool found) {
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
        }
    };

    while (true) {
        // ---- refill idle ray slots: slot 0 of the idle lanes first, then slot 1 ---------------------------------------------------
        const uint32_t n_idle = (uint32_t)__popcll(__ballot(!R0.active)) + (uint32_t)__popcll(__ballot(!R1.active));
        if (n_idle == 128u || (!exhausted && n_idle >= (uint32_t)TH_TRACE4_REFILL)) {
            if (!exhausted) {
                auto refill_slot = [&](Ray4& R) {
                    const unsigned long long idle = __ballot(!R.active);
                    const uint32_t ni = (uint32_t)__popcll(idle);
                    if (ni == 0u) return;
                    if (pool_next >= pool_end && !exhausted) {
                        for (int tries = 0; tries < kSeg && pool_next >= pool_end && !exhausted; ++tries) {
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
                    }
                    const uint32_t avail = pool_end - pool_next;
                    if (avail && !R.active) {
                        const uint32_t rank = (uint32_t)__popcll(idle & lt_mask);
                        if (rank < avail) {
                            uint32_t idx = seg_phys(q, wseg, pool_next + rank);
                            if (q.indirect) idx = q.indirect[idx];
                            load_ray(R, idx);
                        }
                    }
                    pool_next += min(ni, avail);
                };
                refill_slot(R0);
                refill_slot(R1);
            }
            if (__ballot(R0.active || R1.active) == 0ull) {
                if (exhausted) break;
                continue;
            }
        }
        // ---- phase A: pop / interior steps on whichever ray of the lane can take one ---------------------------------------------------
#pragma unroll 1
        for (int it = 0; it < TH_TRACE4_MAX_A; ++it) {
            // a ray "descends" when it is active and does not hold a leaf (at an interior node, or waiting to pop)
            const bool d0 = R0.active && R0.cur_cnt == 0, d1 = R1.active && R1.cur_cnt == 0;
            const bool n0 = d0 && R0.cur != kRefNone, n1 = d1 && R1.cur != kRefNone;  // at an interior node
            const bool use1 = n0 ? false : (n1 ? true : !d0);                        // an interior step before a pop; slot 0 before slot 1
            const bool have = use1 ? d1 : d0;
            // working copy
            uint32_t cur = use1 ? R1.cur : R0.cur, cur_cnt = 0u, idx = use1 ? R1.idx : R0.idx;
            int sp = use1 ? R1.sp : R0.sp;
            const f3 o = use1 ? R1.o : R0.o, inv_d = use1 ? R1.inv_d : R0.inv_d;
            const float em = use1 ? R1.em : R0.em, t_max = use1 ? R1.t_max : R0.t_max;
            const uint32_t neg = use1 ? R1.neg : R0.neg;
            const bool found = use1 ? R1.found : R0.found;
            const uint32_t r = use1 ? 1u : 0u;
            bool finished = false;
            if (have && cur == kRefNone) {  // pop the next entry whose tx_min is still below t_max (bvh.jl:247-250 with the deferred clause)
                finished = true;
                while (sp > 0) {
                    sp--;
                    if (sp >= kStack2Total) continue;
                    uint32_t enc;
                    float tm;
                    stack_read(r, sp, enc, tm);
                    if (tm < t_max) {
                        cur = enc & 0x00ffffffu;
                        cur_cnt = enc >> 24;
                        finished = false;
                        break;
                    }
                }
            }
            if (finished) deliver(idx, use1 ? R1.slot_w : R0.slot_w, use1 ? R1.flag_w : R0.flag_w, found);
            if (have && !finished && cur != kRefNone && cur_cnt == 0) {  // interior: one 64-byte burst, both child boxes
                const float4 a0 = ws.wnodes[4 * (size_t)cur], a1 = ws.wnodes[4 * (size_t)cur + 1], a2 = ws.wnodes[4 * (size_t)cur + 2], a3 = ws.wnodes[4 * (size_t)cur + 3];
                uint32_t top_enc = kRefNone;  // the stack top, read while the node is on its way (k_trace3's in-step pop)
                float top_tm = kInf;
                if (sp > 0 && sp - 1 < kStack2Total) stack_read(r, sp - 1, top_enc, top_tm);
                if (COUNT) nn += 2;
                const uint32_t lenc = __float_as_uint(a3.x), renc = __float_as_uint(a3.y), meta = __float_as_uint(a3.z);
                const bool negx = (neg & 1u) != 0, negy = (neg & 2u) != 0, negz = (neg & 4u) != 0;
                float tl, tr;
                const bool hl = slab_test2(a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, o, inv_d, em, tight_on && !(meta & 4u), negx, negy, negz, tl);
                const bool hr = slab_test2(a1.z, a1.w, a2.x, a2.y, a2.z, a2.w, o, inv_d, em, tight_on && !(meta & 8u), negx, negy, negz, tr);
                const float tlh = hl ? tl : kInf, trh = hr ? tr : kInf;
                const uint32_t axis = meta & 3u;
                const bool ng = axis == 0 ? negx : (axis == 1 ? negy : negz);  // bvh.jl:239
                const float tn = ng ? trh : tlh, tf = ng ? tlh : trh;
                const uint32_t nenc = ng ? renc : lenc, fenc = ng ? lenc : renc;
                const bool go_n = tn < t_max, go_f = tf < t_max;
                if (go_n & (ANY ? go_f : (tf < kInf))) {  // see k_trace3: a far child is pushed whenever its box is hit (t_max can go up: A.18)
                    if (sp < kStack2Total) stack_write(r, sp, fenc, tf);
                    sp++;
                }
                const uint32_t nxt = go_n ? nenc : fenc;
                const bool any_child = go_n | go_f;
                cur = any_child ? (nxt & 0x00ffffffu) : kRefNone;
                cur_cnt = any_child ? (nxt >> 24) : 0u;
                if (!any_child && sp > 0) {  // nothing was pushed in this step: the top read above is still the top
                    sp--;
                    if (top_tm < t_max && sp < kStack2Total) {
                        cur = top_enc & 0x00ffffffu;
                        cur_cnt = top_enc >> 24;
                    }
                }
            }
            // write back
            if (have) {
                if (use1) {
                    R1.cur = cur;
                    R1.cur_cnt = cur_cnt;
                    R1.sp = sp;
                    if (finished) R1.active = false;
                } else {
                    R0.cur = cur;
                    R0.cur_cnt = cur_cnt;
                    R0.sp = sp;
                    if (finished) R0.active = false;
                }
            }
            const uint32_t n_desc = (uint32_t)__popcll(__ballot((R0.active && R0.cur_cnt == 0) || (R1.active && R1.cur_cnt == 0)));
            if (n_desc <= (uint32_t)TH_TRACE4_LEAF_WAIT) break;
        }
        // ---- phase B: one leaf per lane, primitives in slot order, later equal-t hit wins (bvh.jl:229-237, triangle_mesh.jl:211-214) ----
        {
            const bool l0 = R0.active && R0.cur != kRefNone && R0.cur_cnt > 0, l1 = R1.active && R1.cur != kRefNone && R1.cur_cnt > 0;
            const bool use1 = !l0;
            const bool have = l0 || l1;
            if (have) {
                const uint32_t r = use1 ? 1u : 0u;
                const uint32_t cur = use1 ? R1.cur : R0.cur, cur_cnt = use1 ? R1.cur_cnt : R0.cur_cnt, idx = use1 ? R1.idx : R0.idx;
                int sp = use1 ? R1.sp : R0.sp;
                const f3 o = use1 ? R1.o : R0.o;
                float t_max = use1 ? R1.t_max : R0.t_max;
                bool found = use1 ? R1.found : R0.found;
                RayShear shear = use1 ? R1.shear : R0.shear;
                bool hit_any = false;
                uint32_t top_enc = kRefNone;  // the stack top, read while the primitives are on their way
                float top_tm = kInf;
                if (sp > 0 && sp - 1 < kStack2Total) stack_read(r, sp - 1, top_enc, top_tm);
                for (uint32_t k = 0; k < cur_cnt; ++k) {
                    const uint32_t slot = cur + k;
                    const float4 p0 = sc.prims[3 * slot];
                    const float4 p1 = sc.prims[3 * slot + 1], p2 = sc.prims[3 * slot + 2];
                    asm volatile("" ::"v"(p1.x), "v"(p1.y), "v"(p1.z), "v"(p2.x), "v"(p2.y), "v"(p2.z));  // one burst (see k_trace3)
                    const uint32_t meta = __float_as_uint(p0.w);
                    if (COUNT) np++;
                    if (meta & PRIM_SPHERE) {
                        const float4 d4 = rd[idx];  // rare: fetch the direction again (see k_trace3)
                        const f3 d = mk3(d4.x, d4.y, d4.z);
                        SphereHit sh;
                        if (sphere_intersect<false, FULL_ONLY>(sc.spheres[__float_as_uint(p0.x)], o, d, t_max, sh)) {
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
                uint32_t ncur = kRefNone, ncnt = 0u;
                if (ANY && hit_any) {  // intersect_p returns at the first accepted primitive: drop the stack, the pop in phase A delivers
                    found = true;
                    sp = 0;
                } else if (sp > 0) {  // the next stack entry, against the t_max the leaf left (bvh.jl:226 at pop time); a dead one is dropped, phase A goes on from there
                    sp--;
                    if (top_tm < t_max && sp < kStack2Total) {
                        ncur = top_enc & 0x00ffffffu;
                        ncnt = top_enc >> 24;
                    }
                }
                if (use1) {
                    R1.cur = ncur;
                    R1.cur_cnt = ncnt;
                    R1.sp = sp;
                    R1.t_max = t_max;
                    R1.found = found;
                } else {
                    R0.cur = ncur;
                    R0.cur_cnt = ncnt;
                    R0.sp = sp;
                    R0.t_max = t_max;
                    R0.found = found;
                }
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
