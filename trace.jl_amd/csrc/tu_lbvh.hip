// tu_lbvh.hip — BVHAccel construction on the device (th_lbvh.h).
#include "th_host.h"
#include "th_sppm.h"
#ifdef TRHIP_EXPERIMENTS
#include "th_lbvh.h"
#endif
#include "th_sahb.h"

// BVHAccel on the device (th_lbvh.h): returns TRHIP_ERR_UNSUPPORTED when the tree is deeper than the traversal stack allows
// (the caller then falls back to the host builder).
#ifdef TRHIP_EXPERIMENTS
int build_bvh_device(trhip_ctx* ctx, const std::vector<HostAABB>& pb, FlatBVH& out) {
    const uint32_t n = (uint32_t)pb.size();
    if (n < 2 || n >= (1u << 30)) return TRHIP_ERR_UNSUPPORTED;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    static_assert(sizeof(HostAABB) == 6 * sizeof(float), "HostAABB layout");
    struct Buf {
        void* p = nullptr;
        ~Buf() {
            if (p) (void)hipFree(p);
        }
    };
    Buf d_pb, d_keys, d_keys2, d_sorted, d_sorted2, d_tmp, d_u32, d_ib, d_misc, d_fb, d_fa, d_ff, d_fo;
    const size_t n_int = n - 1, total = 2 * (size_t)n - 1;
    HIP_TRY(ctx, hipMalloc(&d_pb.p, (size_t)n * 6 * sizeof(float)));
    HIP_TRY(ctx, hipMalloc(&d_keys.p, (size_t)n * 8));
    HIP_TRY(ctx, hipMalloc(&d_keys2.p, (size_t)n * 8));
    HIP_TRY(ctx, hipMalloc(&d_sorted.p, (size_t)n * 4));
    HIP_TRY(ctx, hipMalloc(&d_sorted2.p, (size_t)n * 4));
    HIP_TRY(ctx, hipMalloc(&d_u32.p, (6 * n_int + n) * sizeof(uint32_t)));  // left, right, lo, split, parent_int, visits | parent_leaf
    HIP_TRY(ctx, hipMalloc(&d_ib.p, n_int * 6 * sizeof(float)));
    HIP_TRY(ctx, hipMalloc(&d_misc.p, 8 * sizeof(uint32_t)));
    HIP_TRY(ctx, hipMalloc(&d_fb.p, total * 6 * sizeof(float)));
    HIP_TRY(ctx, hipMalloc(&d_fa.p, total * sizeof(uint32_t)));
    HIP_TRY(ctx, hipMalloc(&d_ff.p, total * sizeof(uint32_t)));
    HIP_TRY(ctx, hipMalloc(&d_fo.p, (size_t)n * sizeof(uint32_t)));
    hipStream_t st = ctx->stream;
    HIP_TRY(ctx, hipMemcpyAsync(d_pb.p, pb.data(), (size_t)n * 6 * sizeof(float), hipMemcpyHostToDevice, st));
    uint32_t* u = (uint32_t*)d_u32.p;
    LbvhBuild b{(const float*)d_pb.p, (uint64_t*)d_keys.p, (uint32_t*)d_sorted.p, u, u + n_int, u + 2 * n_int, u + 3 * n_int, u + 4 * n_int, u + 6 * n_int, u + 5 * n_int, (float*)d_ib.p,
                (uint32_t*)d_misc.p, (uint32_t*)d_misc.p + 6, n};
    const uint32_t init[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0u, 0u};  // enc(+Inf) < 0xffffffff and enc(-Inf) > 0: any real value wins
    HIP_TRY(ctx, hipMemcpyAsync(d_misc.p, init, sizeof init, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemsetAsync(b.visits, 0, n_int * sizeof(uint32_t), st));
    const dim3 grid(grid_for(ctx, n, 8)), gridt(grid_for(ctx, total, 8)), blk(kBlock);
    hipLaunchKernelGGL(k_lbvh_centroid_bounds, dim3(ctx->num_cu), blk, 0, st, b);
    hipLaunchKernelGGL(k_lbvh_keys, grid, blk, 0, st, b);
    size_t tmp_bytes = 0;
    HIP_TRY(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, (const uint64_t*)d_keys.p, (uint64_t*)d_keys2.p, (const uint32_t*)d_sorted.p, (uint32_t*)d_sorted2.p, (int)n, 0, 63, st));
    HIP_TRY(ctx, hipMalloc(&d_tmp.p, tmp_bytes));
    HIP_TRY(ctx, hipcub::DeviceRadixSort::SortPairs(d_tmp.p, tmp_bytes, (const uint64_t*)d_keys.p, (uint64_t*)d_keys2.p, (const uint32_t*)d_sorted.p, (uint32_t*)d_sorted2.p, (int)n, 0, 63, st));
    b.keys = (uint64_t*)d_keys2.p;
    b.sorted = (uint32_t*)d_sorted2.p;
    hipLaunchKernelGGL(k_lbvh_hierarchy, grid, blk, 0, st, b);
    hipLaunchKernelGGL(k_lbvh_refit, grid, blk, 0, st, b);
    const LbvhFlat f{(float*)d_fb.p, (uint32_t*)d_fa.p, (uint32_t*)d_ff.p, (uint32_t*)d_fo.p};
    hipLaunchKernelGGL(k_lbvh_flatten, gridt, blk, 0, st, b, f);
    HIP_TRY(ctx, hipGetLastError());
    uint32_t misc[8];
    HIP_TRY(ctx, hipMemcpyAsync(misc, d_misc.p, sizeof misc, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    if (misc[6] > (uint32_t)kStack2Total) return TRHIP_ERR_UNSUPPORTED;
    out.bounds.resize(total * 6);
    out.a.resize(total);
    out.flags.resize(total);
    out.order.resize(n);
    HIP_TRY(ctx, hipMemcpy(out.bounds.data(), d_fb.p, total * 6 * sizeof(float), hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(out.a.data(), d_fa.p, total * sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(out.flags.data(), d_ff.p, total * sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(out.order.data(), d_fo.p, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    out.max_depth = misc[6];
    return 0;
}
#else  // the linear BVH (option bvh_builder = 1: +25-35 % node visits against the SAH builders) is an EXPERIMENTS-build kernel set
int build_bvh_device(trhip_ctx*, const std::vector<HostAABB>&, FlatBVH&) { return TRHIP_ERR_UNSUPPORTED; }
#endif

// BVHAccel on the device with the host builder's binned SAH (th_sahb.h).  TRHIP_ERR_UNSUPPORTED: a scene this builder hands back to the host
// (a large set of coincident centroids, a tree past depth 39 before the nodes get small, a leaf hint above kSahSmall).
int build_bvh_device_sah(trhip_ctx* ctx, const std::vector<HostAABB>& pb, int max_node_prims, bool split_coincident, FlatBVH& out, double* ms_device) {
    const uint32_t n = (uint32_t)pb.size();
    if (n < 2 || n >= (1u << 30)) return TRHIP_ERR_UNSUPPORTED;
    // th_bvh.h keeps a node of n <= max_node_prims primitives a leaf when splitting does not pay (its leaf-cost test); the top phase here splits every node of more
    // than kSahSmall primitives without asking.  With a leaf hint of 65 .. 255 the two would differ: such scenes go to the host builder, whose tree this one promises.
    if (std::min(255, max_node_prims) > (int)kSahSmall) return TRHIP_ERR_UNSUPPORTED;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    struct Buf {
        void* p = nullptr;
        ~Buf() {
            if (p) (void)hipFree(p);
        }
    };
    struct Ev {
        hipEvent_t e = nullptr;
        ~Ev() {
            if (e) (void)hipEventDestroy(e);
        }
    };
    const uint32_t pool = 2 * n, cap_active = n / kSahSmall + 2;
    Buf d_pb, d_cen, d_idx[2], d_pos[2], d_nb, d_nu, d_cnt, d_act, d_lvl, d_bins, d_flag, d_scan, d_tmp, d_fb, d_fa, d_ff;
    HIP_TRY(ctx, hipMalloc(&d_pb.p, (size_t)n * 6 * sizeof(float)));
    HIP_TRY(ctx, hipMalloc(&d_cen.p, (size_t)n * 3 * sizeof(float)));
    for (int k = 0; k < 2; ++k) {
        HIP_TRY(ctx, hipMalloc(&d_idx[k].p, (size_t)n * 4));
        HIP_TRY(ctx, hipMalloc(&d_pos[k].p, (size_t)n * 4));
    }
    HIP_TRY(ctx, hipMalloc(&d_nb.p, (size_t)pool * 6 * sizeof(float)));
    HIP_TRY(ctx, hipMalloc(&d_nu.p, (size_t)pool * 7 * sizeof(uint32_t)));                     // lo, hi, left, right, axis, depth, lefts
    HIP_TRY(ctx, hipMalloc(&d_cnt.p, 8 * sizeof(uint32_t)));
    HIP_TRY(ctx, hipMalloc(&d_act.p, ((size_t)cap_active * 5 + n) * sizeof(uint32_t)));         // act, act_next, split, cslot x 2 | small
    HIP_TRY(ctx, hipMalloc(&d_lvl.p, (size_t)cap_active * 12 * sizeof(uint32_t)));
    HIP_TRY(ctx, hipMalloc(&d_bins.p, (size_t)cap_active * kSahBinWords * sizeof(uint32_t)));
    HIP_TRY(ctx, hipMalloc(&d_flag.p, ((size_t)n + 1) * 4));
    HIP_TRY(ctx, hipMalloc(&d_scan.p, ((size_t)n + 1) * 4));
    size_t tmp_bytes = 0;
    HIP_TRY(ctx, hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, (const uint32_t*)d_flag.p, (uint32_t*)d_scan.p, (int)(n + 1), ctx->stream));
    HIP_TRY(ctx, hipMalloc(&d_tmp.p, tmp_bytes));
    hipStream_t st = ctx->stream;
    Ev e0, e1;
    HIP_TRY(ctx, hipEventCreate(&e0.e));
    HIP_TRY(ctx, hipEventCreate(&e1.e));
    static_assert(sizeof(HostAABB) == 6 * sizeof(float), "HostAABB layout");
    HIP_TRY(ctx, hipMemcpyAsync(d_pb.p, pb.data(), (size_t)n * 6 * sizeof(float), hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipEventRecord(e0.e, st));
    uint32_t* nu = (uint32_t*)d_nu.p;
    uint32_t* au = (uint32_t*)d_act.p;
    SahBuild b{};
    b.pb = (const float*)d_pb.p;
    b.cen = (float*)d_cen.p;
    b.idx_in = (uint32_t*)d_idx[0].p;
    b.idx_out = (uint32_t*)d_idx[1].p;
    b.pos_in = (uint32_t*)d_pos[0].p;
    b.pos_out = (uint32_t*)d_pos[1].p;
    b.nb = (float*)d_nb.p;
    b.n_lo = nu;
    b.n_hi = nu + (size_t)pool;
    b.n_left = nu + 2 * (size_t)pool;
    b.n_right = nu + 3 * (size_t)pool;
    b.n_axis = nu + 4 * (size_t)pool;
    b.n_depth = nu + 5 * (size_t)pool;
    b.n_lefts = nu + 6 * (size_t)pool;
    b.counters = (uint32_t*)d_cnt.p;
    b.pool_cap = pool;
    b.act = au;
    b.act_next = au + cap_active;
    b.split = au + 2 * (size_t)cap_active;
    b.cslot = au + 3 * (size_t)cap_active;
    b.small = au + 5 * (size_t)cap_active;
    b.lvl_b = (uint32_t*)d_lvl.p;
    b.lvl_cb = (uint32_t*)d_lvl.p + 6 * (size_t)cap_active;
    b.bins = (uint32_t*)d_bins.p;
    b.flag = (uint32_t*)d_flag.p;
    b.scan = (uint32_t*)d_scan.p;
    b.n = n;
    b.max_leaf = std::max(1, std::min(255, max_node_prims));
    b.split_coincident = split_coincident ? 1 : 0;
    const dim3 blk(kBlock), grid(grid_for(ctx, n, 8)), grid_chunk((n + kSahChunk - 1) / kSahChunk);
    hipLaunchKernelGGL(k_sah_init, grid, blk, 0, st, b);
    uint32_t cnt[8] = {1, 0, n > kSahSmall ? 0u : 1u, 0, 1, 0, 0, 0};
    b.n_active = n > kSahSmall ? 1u : 0u;
    int rounds = 0;
    while (b.n_active > 0) {
        if (b.n_active > cap_active || ++rounds > 64) return fail(ctx, TRHIP_ERR_HIP, "device SAH build: %u active nodes in round %d", b.n_active, rounds);
        const dim3 grid_slots((b.n_active + kBlock - 1) / kBlock);
        hipLaunchKernelGGL(k_sah_round_init, dim3(grid_for(ctx, (uint64_t)b.n_active * kSahBinWords, 8)), blk, 0, st, b);
        hipLaunchKernelGGL(k_sah_bounds, grid_chunk, blk, 0, st, b);
        hipLaunchKernelGGL(k_sah_bin, grid_chunk, blk, 0, st, b);
        hipLaunchKernelGGL(k_sah_split, grid_slots, blk, 0, st, b);
        hipLaunchKernelGGL(k_sah_flag, grid, blk, 0, st, b);
        HIP_TRY(ctx, hipcub::DeviceScan::ExclusiveSum(d_tmp.p, tmp_bytes, (const uint32_t*)b.flag, b.scan, (int)(n + 1), st));
        hipLaunchKernelGGL(k_sah_children, grid_slots, blk, 0, st, b);
        hipLaunchKernelGGL(k_sah_scatter, grid, blk, 0, st, b);
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipMemcpyAsync(cnt, b.counters, sizeof cnt, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));
        if (cnt[3]) return TRHIP_ERR_UNSUPPORTED;
        std::swap(b.idx_in, b.idx_out);
        std::swap(b.pos_in, b.pos_out);
        std::swap(b.act, b.act_next);
        b.n_active = cnt[1];
    }
    const uint32_t n_small = cnt[2];
    if (n_small) hipLaunchKernelGGL(k_sah_small, dim3((n_small + 63) / 64), dim3(64), 0, st, b, n_small);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(cnt, b.counters, sizeof cnt, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    if (cnt[3]) return TRHIP_ERR_UNSUPPORTED;
    const uint32_t total = cnt[0];
    if (total > pool || (total & 1u) == 0u) return fail(ctx, TRHIP_ERR_HIP, "device SAH build: %u nodes", total);
    if (cnt[4] > (uint32_t)kStack2Total) return TRHIP_ERR_UNSUPPORTED;
    HIP_TRY(ctx, hipMalloc(&d_fb.p, (size_t)total * 6 * sizeof(float)));
    HIP_TRY(ctx, hipMalloc(&d_fa.p, (size_t)total * sizeof(uint32_t)));
    HIP_TRY(ctx, hipMalloc(&d_ff.p, (size_t)total * sizeof(uint32_t)));
    HIP_TRY(ctx, hipMemsetAsync(b.flag, 0, ((size_t)n + 1) * 4, st));
    const dim3 gridt(grid_for(ctx, total, 8));
    hipLaunchKernelGGL(k_sah_mark_leaves, gridt, blk, 0, st, b, total);
    HIP_TRY(ctx, hipcub::DeviceScan::ExclusiveSum(d_tmp.p, tmp_bytes, (const uint32_t*)b.flag, b.scan, (int)(n + 1), st));
    const SahFlat f{(float*)d_fb.p, (uint32_t*)d_fa.p, (uint32_t*)d_ff.p};
    hipLaunchKernelGGL(k_sah_flatten, gridt, blk, 0, st, b, total, f);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(e1.e, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    if (ms_device) {
        float ms = 0.0f;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, e0.e, e1.e));
        *ms_device = ms;
    }
    out.bounds.resize((size_t)total * 6);
    out.a.resize(total);
    out.flags.resize(total);
    out.order.resize(n);
    HIP_TRY(ctx, hipMemcpy(out.bounds.data(), d_fb.p, (size_t)total * 6 * sizeof(float), hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(out.a.data(), d_fa.p, (size_t)total * sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(out.flags.data(), d_ff.p, (size_t)total * sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(out.order.data(), b.idx_in, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    out.max_depth = cnt[4];
    return 0;
}

int trhip_last_bvh_build_ms(trhip_ctx* ctx, double* ms_device) {
    if (!ctx || !ms_device) return TRHIP_ERR_INVALID;
    *ms_device = ctx->bvh_device_ms;
    return TRHIP_OK;
}
