// fetch_calib.hip — what does rocprofv3's FETCH_SIZE report on gfx950 for the access patterns of the traversal kernels?
// MI355X_MICROARCH.md calibrates it for 16 B / lane coalesced streaming reads only (it reports half the bytes) and says "other access widths
// are uncalibrated: calibrate on a known byte count in your own access pattern".  This program reads tables of KNOWN size, every byte exactly
// once per launch, in four patterns; tools/calib/run_fetch_calib.sh runs it under `rocprofv3 --pmc FETCH_SIZE` and prints bytes / FETCH_SIZE:
//   k_stream16    lane i reads 16 B at 16 i                                  (the guide's pattern: expect FETCH_SIZE = bytes / 2)
//   k_gather64    lane i reads the 64-byte record perm(i) as 4 x dwordx4     (k_trace3's node fetch: one 64-byte children-in-parent node per lane)
//   k_gather48    lane i reads the 48-byte record perm(i) as 3 x dwordx4     (a primitive record: 3 x float4, may straddle two 64-byte halves)
//   k_gather128   lane i reads the 128-byte record perm(i) as 8 x dwordx4    (the shading record / an 8-wide node)
// perm(i) = (i * odd) mod 2^k is a bijection of the 2^k records: no record is read twice, so neither L2 nor the 256 MiB Infinity Cache can serve
// anything (tables are 1 GiB and 768 MiB); consecutive lanes land ~2.6 GB·frac apart.  Build: hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                                \
            exit(1);                                                                               \
        }                                                                                          \
    } while (0)

__global__ __launch_bounds__(256) void k_stream16(const float4* __restrict__ t, uint64_t n, float* __restrict__ sink) {
    float acc = 0.0f;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const float4 v = t[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) sink[0] = acc;
}
template <int Q>  // Q float4 per record
__global__ __launch_bounds__(256) void k_gather(const float4* __restrict__ t, uint32_t log2_records, float* __restrict__ sink) {
    const uint64_t n = 1ull << log2_records, mask = n - 1;
    float acc = 0.0f;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const uint64_t r = (i * 0x9E3779B1ull) & mask;  // odd multiplier: a bijection of [0, 2^k)
        const float4* p = t + r * Q;
        float4 v[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) v[q] = p[q];
#pragma unroll
        for (int q = 0; q < Q; ++q) acc += v[q].x + v[q].y + v[q].z + v[q].w;
    }
    if (acc == 12345.678f) sink[0] = acc;
}

int main() {
    const size_t bytes = (size_t)1 << 30;  // 1 GiB
    float4* t = nullptr;
    float* sink = nullptr;
    CHECK(hipMalloc(&t, bytes));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(t, 0, bytes));
    const dim3 grid(256 * 16), blk(256);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    auto timed = [&](const char* name, size_t moved, auto launch) {
        launch();  // warm-up (also under the profiler: two dispatches per pattern, identical)
        CHECK(hipEventRecord(e0));
        launch();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("{\"kernel\": \"%s\", \"bytes\": %zu, \"ms\": %.3f, \"GBps\": %.1f}\n", name, moved, ms, moved / (ms * 1e-3) / 1e9);
    };
    timed("k_stream16", bytes, [&] { hipLaunchKernelGGL(k_stream16, grid, blk, 0, 0, t, bytes / 16, sink); });
    timed("k_gather<4> (64 B records)", bytes, [&] { hipLaunchKernelGGL(k_gather<4>, grid, blk, 0, 0, t, 24u, sink); });               // 2^24 x 64 B = 1 GiB
    timed("k_gather<3> (48 B records)", (size_t)48 << 24, [&] { hipLaunchKernelGGL(k_gather<3>, grid, blk, 0, 0, t, 24u, sink); });    // 2^24 x 48 B = 768 MiB
    timed("k_gather<8> (128 B records)", bytes, [&] { hipLaunchKernelGGL(k_gather<8>, grid, blk, 0, 0, t, 23u, sink); });             // 2^23 x 128 B = 1 GiB
    CHECK(hipDeviceSynchronize());
    return 0;
}
