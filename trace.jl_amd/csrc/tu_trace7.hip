// tu_trace7.hip — the k_trace7 kernel family (closest-hit rays front to back with tie detection, th_trace7.h; option "traversal" = 7).
#include "th_host.h"
#ifdef TRHIP_EXPERIMENTS  // (the default build does not carry this kernel family: __graft_entry__.build_library(extra_flags=["-DTRHIP_EXPERIMENTS"], …))

#define TH_LAUNCH7(CNTV, FULLV, BIGV)                                                                                                                                       \
    do {                                                                                                                                                                \
        if (cheap)                                                                                                                                                      \
            hipLaunchKernelGGL((k_trace7<CNTV, FULLV, BIGV, true>), grid, block, 0, st, sc->dev, wide_view(ctx, sc), q, ro, rd, tmax, out, work_cursors, ov, ctr, fb);  \
        else                                                                                                                                                            \
            hipLaunchKernelGGL((k_trace7<CNTV, FULLV, BIGV, false>), grid, block, 0, st, sc->dev, wide_view(ctx, sc), q, ro, rd, tmax, out, work_cursors, ov, ctr, fb); \
    } while (0)

void launch_trace7(trhip_ctx* ctx, hipStream_t st, const trhip_scene* sc, bool cnt, bool full_only, bool big, const SegQueue& q, const float4* ro, const float4* rd, const float* tmax,
                   const TraceOut& out, uint32_t* work_cursors, uint2* ov, Counters* ctr, const FallbackList& fb) {
    const dim3 grid(trace_grid(ctx)), block(kBlock);
    // the cheap interior test needs leaf boxes that are exactly their triangles' (every tree built here; option "trace7_cheap" 0 keeps the reference's test on every box)
    const bool cheap = ctx->trace7_cheap && sc->wide.leaf_tight != 0u;
    if (cnt) {
        if (full_only) TH_LAUNCH7(true, true, false); else TH_LAUNCH7(true, false, false);
    } else if (big) {
        if (full_only) TH_LAUNCH7(false, true, true); else TH_LAUNCH7(false, false, true);
    } else {
        if (full_only) TH_LAUNCH7(false, true, false); else TH_LAUNCH7(false, false, false);
    }
}
#endif
