// tu_lbvh.hip — BVHAccel construction on the device (th_lbvh.h).
#include "th_host.h"
#include "th_sppm.h"
#include "th_lbvh.h"

// BVHAccel on the device (th_lbvh.h): returns TRHIP_ERR_UNSUPPORTED when the tree is deeper than the traversal stack allows
// (the caller then falls back to the host builder).
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
