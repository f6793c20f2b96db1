// tu_trace8.hip — the k_trace8 kernel family (8-wide quantised nodes in the binary walk's order, th_trace8.h; option "traversal" = 4).
#include "th_host.h"
#ifdef TRHIP_EXPERIMENTS  // (the default build does not carry this kernel family: __graft_entry__.build_library(extra_flags=["-DTRHIP_EXPERIMENTS"], …))

#define TH_LAUNCH8(ANYV, CNTV, FULLV) hipLaunchKernelGGL((k_trace8<ANYV, CNTV, FULLV>), grid, block, 0, st, sc->dev, w8, q, ro, rd, tmax, out, work_cursors, ov8, ctr, fb)

void launch_trace8(trhip_ctx* ctx, hipStream_t st, const trhip_scene* sc, bool any, bool cnt, bool full_only, const Wide8Scene& w8, const SegQueue& q, const float4* ro, const float4* rd,
                   const float* tmax, const TraceOut& out, uint32_t* work_cursors, uint32_t* ov8, Counters* ctr, const FallbackList& fb) {
    const dim3 grid(trace_grid(ctx)), block(kBlock);
    if (any) {
        if (cnt) {
            if (full_only) TH_LAUNCH8(true, true, true); else TH_LAUNCH8(true, true, false);
        } else {
            if (full_only) TH_LAUNCH8(true, false, true); else TH_LAUNCH8(true, false, false);
        }
    } else {
        if (cnt) {
            if (full_only) TH_LAUNCH8(false, true, true); else TH_LAUNCH8(false, true, false);
        } else {
            if (full_only) TH_LAUNCH8(false, false, true); else TH_LAUNCH8(false, false, false);
        }
    }
}
#endif
