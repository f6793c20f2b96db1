// th_trace3c4.h — the hybrid mode's certified closest-hit walk on the accelerator laid out FOUR children wide (tu_scene.hip upload_accelerator: every other level of the library's
// binary tree folded into its parent; 128 bytes per node).  Everything of th_trace3c.h's header holds — the certificate speaks of leaves (parts of canonical leaves with bit for
// bit their boxes), of candidates and of lower bounds of what a culled box can hold; it is indifferent to the interior topology and to the visiting order — with these
// differences:
//   * one step = one 128-byte node = four boxes, each tested with slab_test3's clauses (the reference's own, bounds.jl:186-198 as written, plus the two it lost on the
//     box grown by em) evaluated on the reference's slab products in the packed min / max form of th_trace3c.h "The step": a child's entry distance IS the reference's
//     tx_min, a leaf child is entered iff the reference's clauses pass on its (canonical) box — exactly the binary walk's rule, so the certificate's argument carries
//     over unchanged.  (A first version grew every clause by em instead: + 50 % primitive tests — rays that pass a leaf box laterally within em — and 84 ms against the
//     binary walk's 76; with no margin at all, unsound, 66 ms: profiles/r5.)
//   * the four children are visited nearest first (a five-comparator network on the entry distances; unsorted: 129 ms); up to three wait on the stack (64 entries: the
//     commit lays a tree out four wide only when three times its depth fits);
//   * AXIS (launches whose rays start far outside the scene) adds the per-axis form of the lower bound as in k_trace3c.
// Half the dependent node fetches per ray: the binary walk is bound by the latency of one node fetch per step as much as by its instructions (profiles/r5).
#pragma once
#include "th_trace3c.h"

namespace th {

// five waves per SIMD at 87 VGPRs without a spill and 13 stack levels in LDS: 66.9 / 67.4 ms against 67.8 / 68.1 at six waves (80 VGPRs, 13 spilled, 11 levels); 14 levels: 75 (a block
// fewer per CU)
#ifndef TH_TRACE3C4_WAVES
#define TH_TRACE3C4_WAVES 5
#endif
#ifndef TH_TRACE3C4_FETCH_BURST
#define TH_TRACE3C4_FETCH_BURST 1
#endif
// 1: phase B tests ONE primitive per lane and round (a leaf of n primitives takes n rounds; the lanes of a short leaf go back to the node steps instead of idling through the longest
// leaf of the wave): 66.2 against 67.8 ms on S-mesh, 49.5 against 50.2 on S-blob; thresholds 24 / 40, 4 / 16 node steps per round, majority vote: 68.6 / 67.4, 66.6 / 66.2, 66.1
// 1: a refill skips the root box test and takes 1 / d[kz] from the reciprocal direction it already holds (-35 VALU instructions per refill: 67.0 -> 65.6 ms, bit-equal)
#ifndef TH_TRACE3C4_REFILL_LEAN
#define TH_TRACE3C4_REFILL_LEAN 1
#endif
#ifndef TH_TRACE3C4_LEAF_ONE
#define TH_TRACE3C4_LEAF_ONE 1
#endif
#ifndef TH_TRACE3C4_LDS
#define TH_TRACE3C4_LDS 12  // (13 until the top of the tree moved into LDS, TH_TRACE3C4_TOP below: 36 nodes beside 12 levels beat 18 beside 13)
#endif
// CONSERVATIVE steps (round 6; built, exact, SLOWER: compiled out — profiles/r6/r6_trace3c4_experiments.txt).  The certificate needs the reference's exact clauses only where a
// LEAF is decided (th_trace3c.h header: candidates are defined by their canonical leaf's box); which boxes a walk passes through, and in which order, is free as long as (a) every
// box that holds a candidate's leaf is entered and (b) a box is culled only on a lower bound of what it holds.  TH_TRACE3C4_CHEAP = 2 tests all four children — leaf or interior —
// with ONE conservative slab test and moves the exact clauses to the moment a primitive test reports a hit (a ray has one or two of those; it has ~49 box tests):
//   * per ray and axis, c_lo = -(o + em') x (1 / d) and c_hi = -(o - em') x (1 / d): one v_pk_fma_f32 per plane pair gives the slab distances of the box grown by em' per axis,
//     em' = 2^-19 (max |o| + D): what the fused form's rounding differs from the reference's fl(fl(plane - o) x (1 / d)) by (<= 2^-24 (2 |o| + 3 D + 3 em') |1 / d|: a tenth of it);
//   * enter iff the reference's clauses FOLDED pass (x-y entry <= every exit; the tight clause — z entry <= the earlier x-y exit, grown by em on both sides — in place of
//     bounds.jl:194's loose half; z exit > 0 and the grown x-y exit >= 0 in place of :198) and the entry lies below t_lim + mb.  Every clause of slab_test3 implies these, for the
//     box itself and — the slab distances are monotonic in the planes — for every box inside it; the entry distance is a lower bound of the exact one of every box inside;
//   * a primitive test that reports a hit below the relaxed limit is a CANDIDATE only if slab_test3's clauses pass on its leaf's canonical box: verified on the triangle's OWN
//     box (from the vertices at hand; the clauses are monotonic in the box in float arithmetic, the leaf box holds the triangle's), whose exact entry distance — an upper
//     bound of the leaf's — is what the guard (entry <= t + dt) reads; a hit whose own box fails sends the ray to the reference-order walk.  The t-cull of the leaf itself is
//     left to the guard: a hit below t_lim inside a box whose exact entry is >= t_lim + mb cannot be accepted, it flags the ray.
// TH_TRACE3C4_CHEAP = 3: the same on 64-byte QUANTISED nodes (four loads instead of seven; tu_scene.hip writes them when the library is compiled this way): plane = lo + q x scale,
// one byte per plane rounded outwards, a power-of-two scale per axis and node; one v_perm_b32 per plane dword orders each byte pair {near, far} for the ray's signs.
// Measured: 0 film values differ in either form; S-mesh closest-hit + 0.8 % (2) / + 11 % (3), S-blob + 5 % / + 14 %: the step trades VALU issue, L1 requests and primitive tests
// against each other one for one.  0: the exact clauses per child, shipped.
#ifndef TH_TRACE3C4_CHEAP
#define TH_TRACE3C4_CHEAP 0
#endif
// The TOP of the tree in LDS (round 6).  What a step pays for its node is not the bytes but the L1's address rate: each lane reads seven 16-byte pieces of its OWN 128-byte
// line, one tag lookup per lane and instruction — an eighth load from the same line (no new L2 traffic) costs 4.3 % of the kernel (profiles/r6: 64.9 -> 67.7 ms, the step
// 3 756 -> 3 916 cycles, the leaf rounds 3 460 -> 3 634 as well: they share the L1).  The first TH_TRACE3C4_TOP nodes of the array — the commit numbers the tree breadth-first
// down to there: the root, its children, their children — are copied into LDS by every block when it starts; a lane whose node is one of them, and a lane that only pops,
// reads LDS instead (same values, same arithmetic: nothing of the certificate is touched).  LDS is allocated in units of 1 280 bytes: five blocks per CU have 32 000 bytes each
// (32 308 — 21 nodes beside 13 stack levels — ran four blocks per CU: every phase 10 % faster, the kernel 8 % slower).
// S-mesh closest-hit per 64 spp (frames without the second stream): none 66.1, 18 nodes + 13 stack levels 64.2, 36 + 12: 63.4, 54 + 11: 63.7, 72 + 10: 63.5 ms; S-blob 47.9 / 47.3 / 47.5 / 46.9 / 47.1.
#ifndef TH_TRACE3C4_TOP
#define TH_TRACE3C4_TOP 36
#endif

// a x b[H] + c, two per instruction (v_pk_fma_f32; IEEE, one rounding each)
template <int H>
TH_D v2f pk_fma_h(v2f a, v2f b, v2f c) {
    v2f r;
    if (H == 0)
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    else
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

template <bool COUNT, bool FULL_ONLY, bool BIG = false, bool AXIS = false>
__global__ __launch_bounds__(kBlock, TH_TRACE3C4_WAVES) void k_trace3c4(DeviceScene sc /* prims: the accelerator's order */, WideScene ws /* the accelerator */, CertHot ch,
                                                                                               const CertCold* __restrict__ cold, SegQueue q, const float4* __restrict__ ro,
                                                                                               const float4* __restrict__ rd, const float* __restrict__ tmax_or_null, TraceOut out,
                                                                                               uint32_t* __restrict__ work, uint2* __restrict__ overflow, Counters* ctr) {
    constexpr int kLds = TH_TRACE3C4_LDS;

    constexpr uint32_t kLeafBit = 1u << 24;  // a node word >= this (and != kRefNone) is a leaf: ref | count << 24
    __shared__ uint2 s_stk[kLds][kBlock];  // {child word, entry distance}: one 8-byte LDS access per push / pop
    // per-lane state that is only touched when a ray is fetched, accepted or finished lives in LDS, not in registers (the walk runs at the 80-VGPR line of six waves per
    // SIMD; a scratch spill costs a trip to memory, an LDS word 64 cycles): the ray's queue index, its state word, the entry distance of the node in hand
    __shared__ uint32_t s_idx[kBlock];
    __shared__ uint32_t s_st[kBlock];
    __shared__ float s_ex[kBlock];
    __shared__ SegView sv;
#if TH_TRACE3C4_TOP && TH_TRACE3C4_CHEAP == 3
    __shared__ float4 s_top[TH_TRACE3C4_TOP * 4];  // (64-byte quantised records: four 16-byte pieces each)
    const uint32_t n_top = min((uint32_t)TH_TRACE3C4_TOP, ws.n_w4nodes);
    for (uint32_t i = threadIdx.x; i < n_top * 4u; i += kBlock) s_top[i] = ws.w4nodes[i];
#elif TH_TRACE3C4_TOP
    __shared__ float4 s_top[TH_TRACE3C4_TOP * 7];
    const uint32_t n_top = min((uint32_t)TH_TRACE3C4_TOP, ws.n_w4nodes);
    for (uint32_t i = threadIdx.x; i < n_top * 7u; i += kBlock) s_top[i] = ws.w4nodes[8u * (i / 7u) + i % 7u];  // (seg_load's barrier publishes it)
#endif
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
#if TH_TRACE3C4_CHEAP
    v2f cx = v2f{0.0f, 0.0f}, cy = v2f{0.0f, 0.0f}, cz = v2f{0.0f, 0.0f};  // per axis {-(o + em') / d, -(o - em') / d}: the CHEAP step's addends
#if TH_TRACE3C4_CHEAP >= 2
    float gxz = 0.0f, gyz = 0.0f;  // em x (|1 / d.x| + |1 / d.z|), em x (|1 / d.y| + |1 / d.z|): the tight clause's growth, both sides at once
#endif
#if TH_TRACE3C4_CHEAP == 3
    uint32_t sel_xy = 0x03020100u, sel_zx = 0x03020100u, sel_yz = 0x03020100u;  // v_perm_b32 selectors: a dword's two {low plane, high plane} byte pairs in {near, far} order for this ray's signs
#endif
#endif
    // (the direction signs are read off inv_d where they are needed: a ray with a zero component, the one case where sign(1 / d) is not sign(d), never walks here)
#define negx (inv_d.x < 0.0f)
#define negy (inv_d.y < 0.0f)
#define negz (inv_d.z < 0.0f)
    float t_lim = 0.0f;   // t_max + 2 dt, t_max = the t of the last accepted candidate (or the ray's own t_max): what the primitive tests accept up to; a box is culled when the lower
                          // bound of what it holds — its entry distance minus the margin — reaches it
    float mkz = 0.0f;     // per ray: the kz-extent part of mb alone: what the per-axis bound of the step adds to t_lim
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
                        // everything the fetch reads, in ONE round trip: origin, direction, the ray's own t_max and what the chunk's sphere pre-pass left in the hit record (read past
                        // L1: same wave, other lane) — left to itself the compiler sinks the last two behind the root box test: a second dependent trip per refill (2 000 of its 8 000 cycles)
                        const float4 o4 = ro[idx], d4 = rd[idx];
                        const float t_own = tmax_or_null ? tmax_or_null[idx] : kInf;
                        uint32_t st_rec = 0u;
                        float t_rec = 0.0f;
                        if (ch.n_spheres != 0u) {  // (wave-uniform)
                            const float* recp = reinterpret_cast<const float*>(&out.hits[idx]);
                            st_rec = __float_as_uint(__hip_atomic_load(recp + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                            t_rec = __hip_atomic_load(recp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
#if TH_TRACE3C4_FETCH_BURST
                        asm volatile("" ::"v"(o4.x), "v"(d4.x), "v"(t_own), "v"(st_rec), "v"(t_rec));
#endif
                        o = mk3(o4.x, o4.y, o4.z);
                        const f3 d = mk3(d4.x, d4.y, d4.z);
#ifdef TH_DIAG_FAST_REFILL  // DIAGNOSTIC (results NOT exact): what a refill without its arithmetic would cost — approximate reciprocals, the margin from the origin's first coordinate alone
                        inv_d = mk3(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z));
                        em = (fabsf(ws.root_box[3] - o.x) + fabsf(ws.root_box[0] - o.x)) * ws.tight_scale;
#else
                        inv_d = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                        em = slab_margin(ws.root_box, ws.tight_scale, o);
#endif
                        shear = TH_TRACE3C4_REFILL_LEAN ? ray_shear(d, inv_d) : ray_shear(d);
                        const float dt = margin_t();
                        const float mkz_ = (shear.kz == 0 ? ch.mle[0] : (shear.kz == 1 ? ch.mle[1] : ch.mle[2])) * fabsf(shear.sz);
                        mkz = mkz_;
                        mb = __fmaf_rn(growth(), inv_max(), mkz_);  // the scalar form: what a popped entry's exact entry distance is held against
                        t_lim = t_own + 2.0f * dt;
                        sp = 0;
                        active = true;
                        if (COUNT) nn++;
                        // what the certificate does not cover goes to the reference-order walk at once: a zero or non-finite direction component (0 x Inf = NaN in the slab
                        // products), a non-finite origin or margin, a NaN t_max — and near-axis-parallel rays, whose scalar margin would make the walk overshoot every hit
                        // (kCertCap; with AXIS the margin is per axis: no cap)
#if TH_TRACE3C4_CHEAP
                        // the CHEAP step's per-ray addends (header); reach = what a plane x (1 / d) product can be: kept far from overflow (Inf - Inf = NaN would close every box)
                        const float o_max = fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z));
                        const float reach = o_max + em * uniform_load(&cold->inv_tight, 0);
#if TH_TRACE3C4_CHEAP >= 2
                        const float emc = 1.9073486328125e-6f * reach;  // (the fused form's rounding slack alone: the tight clause's own growth is applied where that clause is)
                        gxz = em * fabsf(inv_d.x) + em * fabsf(inv_d.z);
                        gyz = em * fabsf(inv_d.y) + em * fabsf(inv_d.z);
#else
#error "TH_TRACE3C4_CHEAP: 0 (the exact step, shipped), 2 (folded clauses + fused products) or 3 (2 on 64-byte quantised nodes)"
#endif
                        cx = v2f{-(o.x + emc) * inv_d.x, -(o.x - emc) * inv_d.x};
                        cy = v2f{-(o.y + emc) * inv_d.y, -(o.y - emc) * inv_d.y};
                        cz = v2f{-(o.z + emc) * inv_d.z, -(o.z - emc) * inv_d.z};
#if TH_TRACE3C4_CHEAP == 3
                        // QUANTISED nodes: the addends in {near plane, far plane} order (the step brings the plane bytes into that order with one v_perm_b32 per dword: no min / max)
                        if (negx) cx = v2f{cx.y, cx.x};
                        if (negy) cy = v2f{cy.y, cy.x};
                        if (negz) cz = v2f{cz.y, cz.x};
                        sel_xy = (negx ? 0x0001u : 0x0100u) | (negy ? 0x02030000u : 0x03020000u);
                        sel_zx = (negz ? 0x0001u : 0x0100u) | (negx ? 0x02030000u : 0x03020000u);
                        sel_yz = (negy ? 0x0001u : 0x0100u) | (negz ? 0x02030000u : 0x03020000u);
#endif
                        const bool cheap_ok = reach * inv_max() < 1e36f;
#else
                        const bool cheap_ok = true;
#endif
                        const bool plain = cheap_ok && d.x != 0.0f && d.y != 0.0f && d.z != 0.0f && mb < kInf && fabsf(o.x) < kInf && fabsf(o.y) < kInf && fabsf(o.z) < kInf && fabsf(inv_d.x) < kInf &&
                                           fabsf(inv_d.y) < kInf && fabsf(inv_d.z) < kInf && t_own == t_own && (AXIS || mb - mkz_ <= kCertCap * mkz_ + dt);
                        float tmin = 0.0f;
                        if (!plain) {
                            to_fb = true;
                            active = false;
                            if (COUNT) n_why[0]++;
                        } else if (ws.root_ref != kRefNone && (TH_TRACE3C4_REFILL_LEAN || slab_test2(ws.root_box[0], ws.root_box[1], ws.root_box[2], ws.root_box[3], ws.root_box[4], ws.root_box[5], o, inv_d, em, false, negx, negy, negz, tmin))) {
                            // (REFILL_LEAN: the root box (bvh.jl:226) is not tested — the root of a four-wide tree is interior, each of its children's boxes lies inside it and every clause is
                            // monotonic in the box: a ray that fails the root fails all four children in its first step)
                            cur = ws.root_ref | (ws.root_cnt << 24);  // (the root is not culled by t: the reference's clause `tmin < t_max` holds whenever anything inside can be accepted)
                            s_ex[tid] = tmin;
                            // the sphere pre-pass of the chunk left this ray's state in its hit record: an accepted sphere (the incumbent), the sphere it starts inside of, or
                            // "to the reference-order walk"
                            bool flagged = false;
                            if (ch.n_spheres != 0u) {
                                st = st_rec;
                                flagged = (st >> 31) != 0u;
                                st &= 0x7fffffffu;
                                if (st & 1u) t_lim = t_rec + 2.0f * dt;
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
#if TH_TRACE3C4_LEAF_ONE
        // (phase B tests ONE primitive per lane and round: a round goes to the side most lanes wait on)
        const bool run_a = (uint32_t)__popcll(__ballot(active && (cur < kLeafBit || cur == kRefNone))) > (uint32_t)TH_TRACE3C_LEAF_WAIT || __ballot(active && cur >= kLeafBit && cur != kRefNone) == 0ull;
#else
        const bool run_a = true;
#endif
#pragma unroll 1
        for (int it = 0; run_a && it < TH_TRACE3C_MAX_A; ++it) {
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
                // interior: one 128-byte line, four child boxes.  Every lane of the section loads — a lane that only pops reads the root's line (always cached) and ignores it: loads under
                // `if (stepping)` into zero-initialised registers cost 28 v_mov per step (the compiler keeps the zeros alive across the stack-top read between the loads and their use)
#if TH_TRACE3C4_CHEAP == 3
                // QUANTISED node: 64 bytes = four loads {lo.xyz, scale.x | 16 plane bytes | 8 plane bytes, scale.yz | four child words} (tu_scene.hip upload_accelerator)
                float4 a0, a1, a2, a6;
                const uint32_t ncur = stepping ? cur : 0u;
#if TH_TRACE3C4_TOP
                if (ncur < n_top) {
                    const float4* tp = s_top + 4u * ncur;
                    a0 = tp[0], a1 = tp[1], a2 = tp[2], a6 = tp[3];
                } else
#endif
                {
                    const float4* np = ws.w4nodes + 4 * (size_t)ncur;
                    a0 = np[0], a1 = np[1], a2 = np[2], a6 = np[3];
                }
#elif TH_TRACE3C4_TOP
                // a node of the top of the tree, and the placeholder of a lane that only pops, from LDS (header "TOP"); both sides of the branch define all seven values
                float4 a0, a1, a2, a3, a4, a5, a6;
                const uint32_t ncur = stepping ? cur : 0u;
                if (ncur < n_top) {
                    const float4* tp = s_top + 7u * ncur;
                    a0 = tp[0], a1 = tp[1], a2 = tp[2], a3 = tp[3], a4 = tp[4], a5 = tp[5], a6 = tp[6];
                } else {
                    const float4* np = ws.w4nodes + 8 * (size_t)ncur;
                    a0 = np[0], a1 = np[1], a2 = np[2], a3 = np[3], a4 = np[4], a5 = np[5], a6 = np[6];
                }
#else
                const float4* np = ws.w4nodes + 8 * (size_t)(stepping ? cur : 0u);
                const float4 a0 = np[0], a1 = np[1], a2 = np[2], a3 = np[3], a4 = np[4], a5 = np[5], a6 = np[6];
#endif
#if defined(TH_TRACE3C4_DUMMY_LOAD) && !TH_TRACE3C4_TOP  // DIAGNOSTIC: an eighth 16-byte load from the same line (no new L2 traffic): does the step pay for the L1's address rate?
                {
                    const float4 a7 = np[7];
                    asm volatile("" ::"v"(a7.x), "v"(a7.y), "v"(a7.z), "v"(a7.w));
                }
#endif
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
                const float t_pop = t_lim + mb;  // (scalar form, against an exact entry distance)
                uint32_t n_go = 0;
                float ex_new = 0.0f;
                cur = kRefNone;
                if (stepping) {
                    if (COUNT) nn += 4;
                    const float t_push = t_lim + mkz;  // (AXIS: the per-axis form of the bound, against the entry distance of the box grown by growth — launches whose rays start far outside the scene)
#if TH_TRACE3C4_CHEAP == 3
                    const float grow = AXIS ? growth() : 0.0f;
                    const float ax_x = grow * fabsf(inv_d.x), ax_y = grow * fabsf(inv_d.y), ax_z = grow * fabsf(inv_d.z);
                    // plane = lo + q x scale (q: one byte, rounded outwards at commit), so its slab distance is q x (scale / d) + ((lo - o -+ em') / d): per node three products and three
                    // fused pairs, per plane pair two byte conversions and one fused pair — after one v_perm_b32 per dword has put each byte pair into {near, far} order for this ray
                    const v2f i_xy = v2f{inv_d.x, inv_d.y}, i_z = v2f{inv_d.z, inv_d.z};
                    const v2f S_xy = v2f{a0.w * inv_d.x, a2.z * inv_d.y}, S_z = v2f{a2.w * inv_d.z, 0.0f};
                    const v2f Cx = pk_fma_h<0>(v2f{a0.x, a0.x}, i_xy, cx), Cy = pk_fma_h<1>(v2f{a0.y, a0.y}, i_xy, cy), Cz = pk_fma_h<0>(v2f{a0.z, a0.z}, i_z, cz);
                    const uint32_t d0 = __builtin_amdgcn_perm(__float_as_uint(a1.x), __float_as_uint(a1.x), sel_xy), d1 = __builtin_amdgcn_perm(__float_as_uint(a1.y), __float_as_uint(a1.y), sel_zx),
                                   d2 = __builtin_amdgcn_perm(__float_as_uint(a1.z), __float_as_uint(a1.z), sel_yz), d3 = __builtin_amdgcn_perm(__float_as_uint(a1.w), __float_as_uint(a1.w), sel_xy),
                                   d4 = __builtin_amdgcn_perm(__float_as_uint(a2.x), __float_as_uint(a2.x), sel_zx), d5 = __builtin_amdgcn_perm(__float_as_uint(a2.y), __float_as_uint(a2.y), sel_yz);
                    auto lo_pair = [](uint32_t dw) { return v2f{(float)(dw & 0xffu), (float)((dw >> 8) & 0xffu)}; };       // v_cvt_f32_ubyte0 / 1
                    auto hi_pair = [](uint32_t dw) { return v2f{(float)((dw >> 16) & 0xffu), (float)(dw >> 24)}; };          // v_cvt_f32_ubyte2 / 3
                    // one child, leaf or interior: {near, far} slab distances per axis, the folded clauses of CHEAP = 2 (header); an empty slot (child word kRefNone) is never entered
                    auto child = [&](v2f qx, v2f qy, v2f qz, uint32_t word) {
                        const v2f Tx = pk_fma_h<0>(qx, S_xy, Cx), Ty = pk_fma_h<1>(qy, S_xy, Cy), Tz = pk_fma_h<0>(qz, S_z, Cz);
                        const float nx = Tx.x, fx = Tx.y, ny = Ty.x, fy = Ty.y, nz = Tz.x, fz = Tz.y;
                        const float A = amax(nx, ny), t_out = amin3(fx, fy, fz), t_in = amax(A, nz);
                        const float Bp = amin(fx + gxz, fy + gyz);
                        bool enter = (A <= t_out) && (nz <= Bp) && (fz > 0.0f) && (Bp >= 0.0f) && (t_in < t_pop) && (word != kRefNone);
                        if constexpr (AXIS) enter = enter && (amax3(nx - ax_x, ny - ax_y, nz - ax_z) < t_push);
                        return enter ? t_in : kInf;
                    };
                    uint32_t e0 = __float_as_uint(a6.x), e1 = __float_as_uint(a6.y), e2 = __float_as_uint(a6.z), e3 = __float_as_uint(a6.w);
                    float k0 = child(lo_pair(d0), hi_pair(d0), lo_pair(d1), e0);
                    float k1 = child(hi_pair(d1), lo_pair(d2), hi_pair(d2), e1);
                    float k2 = child(lo_pair(d3), hi_pair(d3), lo_pair(d4), e2);
                    float k3 = child(hi_pair(d4), lo_pair(d5), hi_pair(d5), e3);
#else
#if TH_TRACE3C4_CHEAP
                    const float grow = AXIS ? growth() : 0.0f;
                    const float ax_x = grow * fabsf(inv_d.x), ax_y = grow * fabsf(inv_d.y), ax_z = grow * fabsf(inv_d.z);
                    const v2f i_xy = v2f{inv_d.x, inv_d.y}, i_z = v2f{inv_d.z, inv_d.z};
                    // one child, leaf or interior: the slab distances of its box grown by em' (one fused instruction per plane pair), the standard overlap test; the sort key is a lower
                    // bound of the exact entry distance of everything inside (header "CHEAP").  An empty slot's NaN planes fail every comparison.
                    auto child = [&](v2f X, v2f Y, v2f Z) {
                        const v2f Tx = pk_fma_h<0>(X, i_xy, cx), Ty = pk_fma_h<1>(Y, i_xy, cy), Tz = pk_fma_h<0>(Z, i_z, cz);
                        const float nx = amin(Tx.x, Tx.y), fx = amax(Tx.x, Tx.y), ny = amin(Ty.x, Ty.y), fy = amax(Ty.x, Ty.y), nz = amin(Tz.x, Tz.y), fz = amax(Tz.x, Tz.y);
#if TH_TRACE3C4_CHEAP >= 2
                        // the reference's clauses FOLDED (x-y entry <= every exit: bounds.jl:188 and the first half of :194) on the box grown by the rounding slack, the tight
                        // clause (z entry <= the earlier x-y exit, grown by em on both sides) in place of :194's loose half, z exit > 0 and the grown x-y exit >= 0 in place of :198:
                        // each is implied by slab_test3 passing on the box or on any box inside it; visits as the exact clauses' (the lateral growth of CHEAP = 1 cost + 12 % boxes and
                        // + 58 % primitive tests on S-mesh, + 41 % / + 183 % on S-blob: profiles/r6)
                        const float A = amax(nx, ny), t_out = amin3(fx, fy, fz), t_in = amax(A, nz);
                        const float Bp = amin(fx + gxz, fy + gyz);
                        bool enter = (A <= t_out) && (nz <= Bp) && (fz > 0.0f) && (Bp >= 0.0f) && (t_in < t_pop);
#endif
                        if constexpr (AXIS) enter = enter && (amax3(nx - ax_x, ny - ax_y, nz - ax_z) < t_push);
                        return enter ? t_in : kInf;
                    };
#else
                    const v2f p_a = v2f{o.x, o.y}, p_b = v2f{o.z, inv_d.x}, p_c = v2f{inv_d.y, inv_d.z};
                    const float gx = em * fabsf(inv_d.x), gy = em * fabsf(inv_d.y), gz = em * fabsf(inv_d.z);
                    const float grow = AXIS ? growth() : 0.0f;
                    const float ax_x = grow * fabsf(inv_d.x), ax_y = grow * fabsf(inv_d.y), ax_z = grow * fabsf(inv_d.z);
                    // one child: its three {min, max} pairs -> the sort key = its entry distance (the reference's tx_min, bit for bit), +Inf when it is not entered.  The clauses are
                    // slab_test3's — bounds.jl:186-198 as written (the LARGER of the x and y exits, :190), t_max aside, and the two clauses that lost on the box grown by em — on the
                    // reference's own slab products, two per instruction (th_trace3c.h "The step").  Every clause of the reference's is monotonic in the box: a box that fails one holds
                    // no leaf box that passes it, i.e. no candidate — no margin is needed on them, for leaves or interior boxes (header: the set of candidates is the same in every tree)
                    auto child = [&](v2f X, v2f Y, v2f Z) {
                        const v2f Tx = pk_mul_h<1>(pk_sub_h<0>(X, p_a), p_b), Ty = pk_mul_h<0>(pk_sub_h<1>(Y, p_a), p_c), Tz = pk_mul_h<1>(pk_sub_h<0>(Z, p_b), p_c);  // bounds.jl:183-193: (plane - o) x inv_d
                        const float nx = amin(Tx.x, Tx.y), fx = amax(Tx.x, Tx.y), ny = amin(Ty.x, Ty.y), fy = amax(Ty.x, Ty.y), nz = amin(Tz.x, Tz.y), fz = amax(Tz.x, Tz.y);
                        const float a = amax(nx, ny), b = amax(fx, fy);        // :189-190
                        const float t_in = amax(a, nz), t_out = amin(fz, b);   // :196-197
                        const bool ref = !(nx > fy) && !(ny > fx) && !(a > fz) && !(nz > b) && (t_out > 0.0f);  // :188, :194, :198
                        const float exit_xy = amin(fx + gx, fy + gy);
                        const bool tight = !(nz - gz > exit_xy) && !(exit_xy < 0.0f);
                        bool enter = ref && tight && (t_in < t_pop);
                        // (the clauses are the reference's EXACTLY for a leaf child: what it lets in is what the reference tests.  A folded form — a <= min3(fx, fy, fz), :194's loose
                        // half and t_out > 0 left to the grown clauses — is 18 VALU instructions shorter per step, 2 % faster and WRONG: 152 559 film values differ, a leaf let in by the
                        // difference holds candidates the reference never tests)
                        if constexpr (AXIS) enter = enter && (amax3(nx - ax_x, ny - ax_y, nz - ax_z) < t_push);
                        return enter ? t_in : kInf;
                    };
#endif
                    float k0 = child(v2f{a0.x, a0.y}, v2f{a0.z, a0.w}, v2f{a1.x, a1.y});
                    float k1 = child(v2f{a1.z, a1.w}, v2f{a2.x, a2.y}, v2f{a2.z, a2.w});
                    float k2 = child(v2f{a3.x, a3.y}, v2f{a3.z, a3.w}, v2f{a4.x, a4.y});
                    float k3 = child(v2f{a4.z, a4.w}, v2f{a5.x, a5.y}, v2f{a5.z, a5.w});
                    uint32_t e0 = __float_as_uint(a6.x), e1 = __float_as_uint(a6.y), e2 = __float_as_uint(a6.z), e3 = __float_as_uint(a6.w);
#endif  // TH_TRACE3C4_CHEAP == 3
                    // nearest first (a five-comparator network on {key, child word}); a child that is not entered sorts last
#define TH_CSWAP(ka, kb, ea, eb)              \
    {                                         \
        const bool sw = kb < ka;              \
        const float tk = sw ? kb : ka;        \
        const uint32_t te = sw ? eb : ea;     \
        kb = sw ? ka : kb;                    \
        eb = sw ? ea : eb;                    \
        ka = tk;                              \
        ea = te;                              \
    }
                    TH_CSWAP(k0, k1, e0, e1);
                    TH_CSWAP(k2, k3, e2, e3);
                    TH_CSWAP(k0, k2, e0, e2);
                    TH_CSWAP(k1, k3, e1, e3);
                    TH_CSWAP(k1, k2, e1, e2);
#undef TH_CSWAP
                    n_go = (k0 < kInf ? 1u : 0u) + (k1 < kInf ? 1u : 0u) + (k2 < kInf ? 1u : 0u) + (k3 < kInf ? 1u : 0u);
                    // the far ones wait on the stack, the farthest deepest (an entry that fails now fails at pop time: t_lim never goes up in this walk)
                    auto store = [&](int pos, uint32_t enc, float tm) {
                        if (pos < kLds) {
                            s_stk[pos][tid] = make_uint2(enc, __float_as_uint(tm));
                        } else if (pos < kStack2Total) {
                            overflow[(size_t)(pos - kLds) * gthreads + gtid] = make_uint2(enc, __float_as_uint(tm));
                        }
                    };
                    if (n_go > 3u) store(sp + (int)n_go - 4, e3, k3);
                    if (n_go > 2u) store(sp + (int)n_go - 3, e2, k2);
                    if (n_go > 1u) store(sp + (int)n_go - 2, e1, k1);
                    if (n_go > 1u) sp += (int)n_go - 1;
                    if (n_go) {
                        cur = e0;
                        ex_new = k0;
                    }
                }
                if (n_go == 0u && sp > 0) {  // nothing was pushed in this step: the top read above is still the top
                    sp--;
                    if (top_tm < t_pop && sp < kStack2Total) {
                        cur = top_enc;
                        ex_new = top_tm;
                    }
                }
#if !TH_TRACE3C4_CHEAP
                s_ex[tid] = ex_new;
#else
                (void)ex_new;
#endif
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
            if (sp > 0 && (!TH_TRACE3C4_LEAF_ONE || leaf_cnt == 1u)) {
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
            // (requesting primitive k + 1's records before primitive k is tested — one round trip per leaf instead of one per primitive — costs 12 live registers and measured 74.7 against
            // 67.8 ms: the kernel is VALU-bound at 97 % busy, the latency was already hidden)
            for (uint32_t k = 0; k < (TH_TRACE3C4_LEAF_ONE ? 1u : leaf_cnt); ++k) {
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
#if TH_TRACE3C4_CHEAP
                    // the step let this leaf in on the conservative test: the hit is a candidate only if the reference's clauses pass on its leaf's canonical box (header
                    // "CHEAP").  They are evaluated on the TRIANGLE's own box, from the vertices at hand — no load: the leaf box holds it, and every clause of slab_test3 is
                    // monotonic in the box IN FLOAT ARITHMETIC (fl(fl(plane - o) x (1 / d)) is a monotonic function of the plane; so are fl(far + g) and fl(near - g)): passing
                    // on the triangle's box implies passing on the leaf's, and the leaf's exact entry distance is <= the triangle box's, which the guard reads.  A hit whose own
                    // box fails (the ray grazes it within rounding) says nothing about the leaf box: the ray goes to the reference-order walk.
                    float ex_leaf = 0.0f;
                    bool unverified = false;
                    if (counts) {
                        const v2f bx = v2f{amin3(p0.x, p1.x, p2.x), amax3(p0.x, p1.x, p2.x)}, by = v2f{amin3(p0.y, p1.y, p2.y), amax3(p0.y, p1.y, p2.y)}, bz = v2f{amin3(p0.z, p1.z, p2.z), amax3(p0.z, p1.z, p2.z)};
                        const v2f p_a = v2f{o.x, o.y}, p_b = v2f{o.z, inv_d.x}, p_c = v2f{inv_d.y, inv_d.z};
                        const v2f Tx = pk_mul_h<1>(pk_sub_h<0>(bx, p_a), p_b), Ty = pk_mul_h<0>(pk_sub_h<1>(by, p_a), p_c), Tz = pk_mul_h<1>(pk_sub_h<0>(bz, p_b), p_c);
                        const float nx = amin(Tx.x, Tx.y), fx = amax(Tx.x, Tx.y), ny = amin(Ty.x, Ty.y), fy = amax(Ty.x, Ty.y), nz = amin(Tz.x, Tz.y), fz = amax(Tz.x, Tz.y);
                        const float a = amax(nx, ny), b = amax(fx, fy);        // bounds.jl:189-190
                        ex_leaf = amax(a, nz);                                 // :196
                        const float t_out = amin(fz, b);                       // :197
                        const bool ref = !(nx > fy) && !(ny > fx) && !(a > fz) && !(nz > b) && (t_out > 0.0f);  // :188, :194, :198
                        const float exit_xy = amin(fx + em * fabsf(inv_d.x), fy + em * fabsf(inv_d.y));
                        unverified = !(ref && !(nz - em * fabsf(inv_d.z) > exit_xy) && !(exit_xy < 0.0f));
                    }
#else
                    const float ex_leaf = s_ex[tid];
                    const bool unverified = false;
#endif
                    if (counts) {
                        const float dt = margin_t();
                        // accepted iff it lies 2 dt below the incumbent (t_lim - 4 dt; the ray's own t_max at first) AND its leaf box lets the reference in by t + dt (the guard); a NaN fails
                        if (unverified || !(tt.t <= t_lim - 4.0f * dt) || !(ex_leaf <= tt.t + dt)) {
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
            cur = (TH_TRACE3C4_LEAF_ONE && !flagged && leaf_cnt > 1u) ? ((leaf_ref + 1u) | ((leaf_cnt - 1u) << 24)) : kRefNone;  // LEAF_ONE: the rest of the leaf in the next rounds
            if (cur != kRefNone) {
            } else if (flagged) {  // the reference-order walk decides this ray
                active = false;
                to_fb = true;
                sp = 0;
                if (COUNT) n_why[why & 3u]++;
            } else if (sp > 0) {  // the next stack entry against the limit the leaf left
                sp--;
                const float t_pop = t_lim + mb;  // (mb holds the growth term in either form: the step's own t_pop)
                if (top_tm < t_pop && sp < kStack2Total) {
                    cur = top_enc;
#if !TH_TRACE3C4_CHEAP
                    s_ex[tid] = top_tm;
#endif
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

}  // namespace th
