// th_trace3d.h — k_trace3c (th_trace3c.h: the certified walk of the hybrid mode) with a per-wave LEAF QUEUE.  Option "leaf_queue" (round 4, A/B).
//
// k_trace3c inherits k_trace3's schedule: a lane that reaches a leaf waits for the wave's leaf phase, which then runs with the ~22 lanes that hold one (27 % of the wave
// cycles at a third of the lanes).  The reference's order forces that on k_trace3 — the leaf's primitives must be tested with the t_max of THAT moment.  The certified walk
// is order-free by construction: what it returns is a function of the SET of candidates (the nearest one, unless a second lies within 2 dt of it or a guard fails), so
//   * a lane that reaches a leaf appends {leaf word, exact entry distance, owner lane} to a queue in LDS and goes on with its stack — culling with a limit that is merely
//     not yet as low as it will be (conservative);
//   * when 64 leaves are queued, each lane takes one — whoever's it is: the owner's origin and shear come over the wave's crossbar (ds_bpermute), the outcome goes to the
//     owner's LDS words with atomics: the incumbent's t (atomicMin), the runner-up's t (atomicMin of the loser of every comparison: the second smallest of {the ray's own
//     t_max} + all candidates), a state word (hit held / flagged);
//   * a ray is finished when its stack is empty and none of its leaves is queued; it is FLAGGED (reference-order walk) when a candidate failed its guard (leaf box entered
//     later than t + dt) or the runner-up lies within 2 dt of the incumbent — the rule of th_trace3c.h evaluated on the final state instead of in passing (it flags a subset
//     of what the in-passing rule flags, and still everything the proof needs: every other candidate that counts lies 2 dt above the answer; the ray's own t_max, entered as the
//     first "candidate", brings the own-limit clause with it).
// Everything else — the margins, the sphere pre-pass, the order word for rays that start inside a sphere, the fallback lists — is k_trace3c's, as is the proof.
#pragma once
#include "th_trace3c.h"

namespace th {

#ifndef TH_TRACE3D_WAVES
#define TH_TRACE3D_WAVES 5
#endif
#ifndef TH_TRACE3D_LDS
#define TH_TRACE3D_LDS 9
#endif
#ifndef TH_TRACE3D_MAX_A
#define TH_TRACE3D_MAX_A 8
#endif
#ifndef TH_TRACE3D_FLUSH
#define TH_TRACE3D_FLUSH 16  // with at most this many lanes still holding an interior node, a partly filled queue is tested
#endif

#define negx (inv_d.x < 0.0f)
#define negy (inv_d.y < 0.0f)
#define negz (inv_d.z < 0.0f)
template <bool COUNT, bool FULL_ONLY, bool AXIS = false>
__global__ __launch_bounds__(kBlock, TH_TRACE3D_WAVES) void k_trace3d(DeviceScene sc /* prims: the accelerator's order */, WideScene ws /* the accelerator */, CertHot ch,
                                                                                               const CertCold* __restrict__ cold, SegQueue q, const float4* __restrict__ ro,
                                                                                               const float4* __restrict__ rd, const float* __restrict__ tmax_or_null, TraceOut out,
                                                                                               uint32_t* __restrict__ work, uint2* __restrict__ overflow, Counters* ctr) {
    constexpr int kLds = TH_TRACE3D_LDS;
    constexpr uint32_t kQL = 128u;  // leaf-queue entries per wave
    constexpr uint32_t kLeafBit = 1u << 24;  // a node word >= this (and != kRefNone) is a leaf: ref | count << 24
    __shared__ uint32_t s_ref[kLds][kBlock];
    __shared__ float s_tmin[kLds][kBlock];
    // per-lane state that is only touched when a ray is fetched, accepted or finished lives in LDS, not in registers (the walk runs at the 80-VGPR line of six waves per
    // SIMD; a scratch spill costs a trip to memory, an LDS word 64 cycles): the ray's queue index, its state word, the entry distance of the node in hand
    __shared__ uint32_t s_idx[kBlock];
    __shared__ uint32_t s_st[kBlock];
    __shared__ float s_ex[kBlock];
    // the leaf queue (header): per ray, what the lanes that test its leaves write and its owner reads — the incumbent's t and the runner-up's (float bits: both >= 0, so
    // unsigned order is float order), and how many of its leaves are still queued
    __shared__ uint32_t s_best[kBlock];
    __shared__ uint32_t s_second[kBlock];
    __shared__ uint32_t s_pend[kBlock];
    __shared__ uint32_t q_word[kBlock / 64][kQL];
    __shared__ float q_ex[kBlock / 64][kQL];
    __shared__ uint32_t q_own[kBlock / 64][kQL];
    __shared__ SegView sv;
    seg_load(q, sv);
    const uint32_t tid = threadIdx.x;
    const uint32_t gthreads = gridDim.x * kBlock;
    const uint32_t gtid = blockIdx.x * kBlock + tid;
    const uint32_t lane = lane_id(), wv = tid >> 6;
    uint32_t ql_cnt = 0;  // wave-uniform: entries in this wave's leaf queue
    const unsigned long long lt_mask = (1ull << lane) - 1ull;

    bool active = false, exhausted = false, to_fb = false;
    uint32_t wseg = __builtin_amdgcn_readfirstlane((gtid >> 6) % kSeg), dry = 0, pool_next = 0, pool_end = 0;  // wave-uniform
    uint32_t cur = kRefNone;  // kRefNone, an interior node's index (< 2^24), or a leaf word
    int sp = 0;
    f3 o = splat3(0.0f), inv_d = splat3(0.0f);
    float em = 0.0f;
    RayShear shear{0, 0.0f, 0.0f, 0.0f};
    float t_lim = 0.0f;   // t_max + 2 dt, t_max = the t of the last accepted candidate (or the ray's own t_max): what the primitive tests accept up to; a box is culled when the lower
                          // bound of what it holds — its entry distance minus the margin — reaches it
    float mb = 0.0f;      // per ray: the margin of the lower bound: non-flat primitives their kz extent, everything the growth in the entering axis (with AXIS the growth is applied per
                          // axis to the box instead, and is not in here)
    // s_ex[tid]: entry distance of the node in `cur` (the reference's tx_min of its box)
    // s_st[tid]: 0: nothing accepted yet, 1: a candidate is; bits 8..: 1 + the sphere the ray started INSIDE of (header: what the reference tests before that sphere does not count)
    uint32_t nn = 0, np = 0;
    uint32_t n_fb = 0;    // wave-uniform
    unsigned long long n_why[4] = {0ull, 0ull, 0ull, 0ull};  // COUNT: why rays went to the canonical tree — 0 direction / finiteness / cap, 1 a sphere (clipped, inside two), 2 near tie / guard; [3]: rays that start INSIDE a sphere and were certified (the order word)

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
                        s_best[tid] = __float_as_uint(t_own);       // (a NaN or negative t_max: `plain` below / the first candidate's rule sends the ray back)
                        s_second[tid] = __float_as_uint(kInf);
                        s_pend[tid] = 0u;
                        sp = 0;
                        active = true;
                        if (COUNT) nn++;
                        // what the certificate does not cover goes to the reference-order walk at once: a zero or non-finite direction component (0 x Inf = NaN in the slab
                        // products), a non-finite origin or margin, a NaN t_max — and near-axis-parallel rays, whose scalar margin would make the walk overshoot every hit
                        // (kCertCap; with AXIS the margin is per axis: no cap)
                        const bool plain = t_own >= 0.0f && d.x != 0.0f && d.y != 0.0f && d.z != 0.0f && mb < kInf && fabsf(o.x) < kInf && fabsf(o.y) < kInf && fabsf(o.z) < kInf && fabsf(inv_d.x) < kInf &&
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
                                if (st & 1u) {
                                    const float t_s = __hip_atomic_load(recp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    t_lim = t_s + 2.0f * dt;
                                    s_best[tid] = __float_as_uint(t_s);  // the pre-pass's sphere is the incumbent (its own acceptance was checked there)
                                }
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
        // ---- phase A: pops and interior steps.  A lane that reaches a leaf QUEUES it (leaf word, exact entry distance, owner lane) and goes on with its stack ----
#pragma unroll 1
        for (int it = 0; it < TH_TRACE3D_MAX_A; ++it) {
            if (active) {
                const uint32_t stw = s_st[tid];
                if (stw >> 30) {  // a lane that tested one of this ray's leaves flagged it: the reference-order walk decides — drop the rest of the walk (queued leaves still drain)
                    sp = 0;
                    cur = kRefNone;
                }
                t_lim = __uint_as_float(s_best[tid]) + 2.0f * margin_t();  // the incumbent may have been lowered by any lane
            }
            {
                const bool at_leaf = active && cur >= kLeafBit && cur != kRefNone;
                const unsigned long long m = __ballot(at_leaf);
                if (m != 0ull) {
                    if (at_leaf) {
                        const uint32_t at = ql_cnt + (uint32_t)__popcll(m & lt_mask);
                        q_word[wv][at] = cur;
                        q_ex[wv][at] = s_ex[tid];
                        q_own[wv][at] = lane;
                        s_pend[tid] += 1u;  // (only the owner adds; the testing lanes subtract in other instructions of this same wave)
                        cur = kRefNone;
                    }
                    ql_cnt += (uint32_t)__popcll(m);
                }
            }
            bool drained = false;
            const bool wants_pop = active && cur == kRefNone;
            const bool pop_now = (uint32_t)__popcll(__ballot(wants_pop)) >= (uint32_t)TH_TRACE3C_POP_MIN || __ballot(active && cur < kLeafBit) == 0ull;
            if (pop_now && wants_pop) {
                drained = true;
                const float t_pop = t_lim + (AXIS ? __fmaf_rn(growth(), inv_max(), mb) : mb);
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
                    if (tm < t_pop) {
                        cur = enc;
                        s_ex[tid] = tm;
                        drained = false;
                        break;
                    }
                }
            }
            if (drained && s_pend[tid] == 0u) {  // the walk is over and every leaf it queued has been tested: the final state decides (header)
                active = false;
                const uint32_t st = s_st[tid];
                const float best = __uint_as_float(s_best[tid]), second = __uint_as_float(s_second[tid]);
                bool fl = (st >> 30) != 0u;
                uint32_t w = (st >> 4) & 3u;
                if (!fl && !(best <= second - 2.0f * margin_t())) {  // the runner-up (a candidate, or the ray's own t_max) within 2 dt of the incumbent
                    fl = true;
                    w = 2u;
                }
                if (fl) {
                    to_fb = true;
                    if (COUNT) n_why[w]++;
                } else if ((st & 3u) == 0u) {
                    out.hits[s_idx[tid]] = make_float4(kInf, __int_as_float(-1), 0.0f, 0.0f);
                } else if ((st & 3u) == 1u) {
                    reinterpret_cast<float*>(&out.hits[s_idx[tid]])[2] = 0.0f;  // the pre-pass's sphere record: its third lane held the state
                }
            }
            if (active && cur < kLeafBit) {  // interior: one 64-byte burst, both child boxes
                const float4 a0 = ws.wnodes[4 * (size_t)cur], a1 = ws.wnodes[4 * (size_t)cur + 1], a2 = ws.wnodes[4 * (size_t)cur + 2], a3 = ws.wnodes[4 * (size_t)cur + 3];
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
                if (COUNT) nn += 2;
                const uint32_t lenc = __float_as_uint(a3.x), renc = __float_as_uint(a3.y), meta = __float_as_uint(a3.z);
                const float gr = AXIS ? growth() : 0.0f;
                const float t_cull = t_lim + mb;
                float tl, tr, gl, gr_;
                const bool hl = slab_test3<AXIS>(a0.x, a0.z, a1.x, a0.y, a0.w, a1.y, o, inv_d, em, gr, !(meta & 4u), negx, negy, negz, tl, gl);  // (the accelerator's node layout)
                const bool hr = slab_test3<AXIS>(a1.z, a2.x, a2.z, a1.w, a2.y, a2.w, o, inv_d, em, gr, !(meta & 8u), negx, negy, negz, tr, gr_);
                // per child: its exact entry distance travels with it; what is compared with t_cull is that distance (with AXIS: the entry of the grown box); a missed child: +Inf.
                // (Boxes on a sphere's path keep the reference's clauses alone — bits 2 / 3 — but are culled like any other: the spheres themselves were tested at the fetch.)
                const float bl = hl ? (AXIS ? gl : tl) : kInf, br = hr ? (AXIS ? gr_ : tr) : kInf;
                const uint32_t axis = meta & 3u;
                const bool neg = axis == 0 ? negx : (axis == 1 ? negy : negz);
                const float bn = neg ? br : bl, bf = neg ? bl : br;
                const float vn = AXIS ? (neg ? tr : tl) : bn, vf = AXIS ? (neg ? tl : tr) : bf;  // (only read for a child that is entered: there the two are the same number)
                const uint32_t nenc = neg ? renc : lenc, fenc = neg ? lenc : renc;
                const bool go_n = bn < t_cull, go_f = bf < t_cull;
                // t_max never goes up in THIS walk (a ray that could see it raised is flagged and leaves): an entry that fails now fails at pop time
                if (go_n & go_f) {
                    if (sp < kLds) {
                        s_ref[sp][tid] = fenc;
                        s_tmin[sp][tid] = vf;
                    } else if (sp < kStack2Total) {
                        overflow[(size_t)(sp - kLds) * gthreads + gtid] = make_uint2(fenc, __float_as_uint(vf));
                    }
                    sp++;
                }
                const bool any_child = go_n | go_f;
                cur = any_child ? (go_n ? nenc : fenc) : kRefNone;
                float ex_new = go_n ? vn : vf;
                if (!any_child && sp > 0) {  // nothing was pushed in this step: the top read above is still the top
                    sp--;
                    const float t_pop = AXIS ? __fmaf_rn(gr, inv_max(), t_cull) : t_cull;
                    if (top_tm < t_pop && sp < kStack2Total) {
                        cur = top_enc;
                        ex_new = top_tm;
                    }
                }
                s_ex[tid] = ex_new;
            }
            if (ql_cnt > kQL - 64u) break;                                                        // the queue must be drained before another round can add to it
            if (__ballot(active && (cur != kRefNone || sp > 0)) == 0ull) break;                    // nobody has a node or a stack entry left
        }
        // ---- phase B: queued leaves, 64 at a time, each tested by whichever lane takes it: the owner's ray comes over the wave's crossbar, the outcome goes to the
        //      owner's LDS words (incumbent, runner-up, state) ----
        if (ql_cnt >= 64u || (ql_cnt > 0u && (uint32_t)__popcll(__ballot(active && cur != kRefNone && cur < kLeafBit)) <= (uint32_t)TH_TRACE3D_FLUSH)) {
            const uint32_t n_take = min(64u, ql_cnt), base = ql_cnt - n_take;
            const bool has = lane < n_take;
            const uint32_t word = has ? q_word[wv][base + lane] : 0u, own = has ? q_own[wv][base + lane] : lane;
            const float ex = has ? q_ex[wv][base + lane] : 0.0f;
            ql_cnt = base;
            const f3 po = mk3(__shfl(o.x, (int)own), __shfl(o.y, (int)own), __shfl(o.z, (int)own));
            RayShear psh;
            psh.kz = __shfl(shear.kz, (int)own);
            psh.sx = __shfl(shear.sx, (int)own);
            psh.sy = __shfl(shear.sy, (int)own);
            psh.sz = __shfl(shear.sz, (int)own);
            const float pdt = ch.kdt * __shfl(em, (int)own) * fabsf(psh.sz);
            const uint32_t my_sgn = (negx ? 1u : 0u) | (negy ? 2u : 0u) | (negz ? 4u : 0u);
            const uint32_t psgn = (uint32_t)__shfl((int)my_sgn, (int)own);
            const uint32_t otid = (tid & ~63u) | own;
            if (has) {
                const uint32_t leaf_ref = word & 0x00ffffffu, leaf_cnt = word >> 24;
                const uint32_t pst = s_st[otid], pidx = s_idx[otid];
                float t_lim_p = __uint_as_float(s_best[otid]) + 2.0f * pdt;
                for (uint32_t k = 0; k < leaf_cnt; ++k) {
                    const uint32_t slot = leaf_ref + k;
                    const float4 p0 = sc.prims[3 * slot];
                    const float4 p1 = sc.prims[3 * slot + 1], p2 = sc.prims[3 * slot + 2];
                    asm volatile("" ::"v"(p1.x), "v"(p1.y), "v"(p1.z), "v"(p1.w), "v"(p2.x), "v"(p2.y), "v"(p2.z), "v"(p2.w));  // one burst
                    const uint32_t meta = __float_as_uint(p0.w);
                    if (COUNT) np++;
                    TriTest tt;
                    if (!(meta & (PRIM_SPHERE | PRIM_DEGENERATE)) && tri_intersect_sheared<true>(mk3(p0.x, p0.y, p0.z), mk3(p1.x, p1.y, p1.z), mk3(p2.x, p2.y, p2.z), po, psh, t_lim_p, &tt)) {
                        bool counts = true;  // a ray that started inside sphere s: only what the reference tests AFTER s counts (the order word, as k_trace3c)
                        if ((pst >> 8) & 0x3fu) {
                            const uint32_t ow = __float_as_uint(p2.w) >> (3u * (((pst >> 8) & 0x3fu) - 1u));
                            const uint32_t ax = ow & 3u;
                            const bool second_child = (ow & 4u) != 0u;
                            counts = ax == 3u ? second_child : (second_child != (((psgn >> ax) & 1u) != 0u));
                        }
                        if (counts) {
                            if (!(ex <= tt.t + pdt)) {  // the guard: the reference must be inside the leaf's box by t + dt
                                atomicOr(&s_st[otid], (1u << 30) | (2u << 4));
                            } else {
                                const uint32_t tb = __float_as_uint(tt.t);
                                const uint32_t old = atomicMin(&s_best[otid], tb);
                                atomicMin(&s_second[otid], max(old, tb));
                                if (s_best[otid] == tb) {  // (read after every lane's atomic of this instruction: at most one lane per ray sees itself — or an exact tie, which flags)
                                    atomicOr(&s_st[otid], 3u);
                                    out.hits[pidx] = make_float4(out.bary_mode ? tt.bary.z : tt.t, p1.w /* the canonical slot */, tt.bary.x, tt.bary.y);
                                }
                                t_lim_p = fminf(t_lim_p, tt.t + 2.0f * pdt);
                            }
                        }
                    }
                }
                atomicSub(&s_pend[otid], 1u);
            }
        }
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

}  // namespace th
