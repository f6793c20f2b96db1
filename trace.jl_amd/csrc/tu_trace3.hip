// tu_trace3.hip — the k_trace3 (default traversal of scenes with a hierarchy) and k_trace4 (two rays per lane) kernel families, instantiated
// here and nowhere else; tu_trace.hip's launch_trace picks the variant.
#include "th_host.h"

#define TH_LAUNCH3(ANYV, CNTV, FULLV, BIGV)                                                                                                                                         \
    do {                                                                                                                                                                            \
        if (ANYV && on_accelerator) /* the accelerator's node layout (PAIRS); only any-hit rays walk it with this kernel */                                                        \
            hipLaunchKernelGGL((k_trace3<ANYV, CNTV, FULLV, false, ANYV>), grid, block, 0, st, sc->dev_acc, wide_view_acc(ctx, sc), q, ro, rd, tmax, out, work_cursors, ov, ctr);    \
        else                                                                                                                                                                        \
            hipLaunchKernelGGL((k_trace3<ANYV, CNTV, FULLV, BIGV>), grid, block, 0, st, sc->dev, wide_view(ctx, sc), q, ro, rd, tmax, out, work_cursors, ov, ctr);                   \
    } while (0)
#ifdef TRHIP_EXPERIMENTS
#define TH_LAUNCH4(ANYV, CNTV, FULLV) hipLaunchKernelGGL((k_trace4<ANYV, CNTV, FULLV>), grid, block, 0, st, sc->dev, wide_view(ctx, sc), q, ro, rd, tmax, out, work_cursors, ov, ctr)
#endif

void launch_trace3(trhip_ctx* ctx, hipStream_t st, const trhip_scene* sc, bool any, bool cnt, bool full_only, bool big, const SegQueue& q, const float4* ro, const float4* rd, const float* tmax,
                   const TraceOut& out, uint32_t* work_cursors, uint2* ov, Counters* ctr, bool on_accelerator) {
    const dim3 grid(trace_grid(ctx)), block(kBlock);
    if (any) {
        if (cnt) {
            if (full_only) TH_LAUNCH3(true, true, true, false); else TH_LAUNCH3(true, true, false, false);
        } else {
            if (full_only) TH_LAUNCH3(true, false, true, false); else TH_LAUNCH3(true, false, false, false);
        }
    } else if (cnt) {
        if (full_only) TH_LAUNCH3(false, true, true, false); else TH_LAUNCH3(false, true, false, false);
    } else if (big) {
        if (full_only) TH_LAUNCH3(false, false, true, true); else TH_LAUNCH3(false, false, false, true);
    } else {
        if (full_only) TH_LAUNCH3(false, false, true, false); else TH_LAUNCH3(false, false, false, false);
    }
}

#ifdef TRHIP_EXPERIMENTS
void launch_trace4(trhip_ctx* ctx, hipStream_t st, const trhip_scene* sc, bool any, bool cnt, bool full_only, const SegQueue& q, const float4* ro, const float4* rd, const float* tmax,
                   const TraceOut& out, uint32_t* work_cursors, uint2* ov, Counters* ctr) {
    const dim3 grid(trace_grid(ctx)), block(kBlock);
    if (any) {
        if (cnt) {
            if (full_only) TH_LAUNCH4(true, true, true); else TH_LAUNCH4(true, true, false);
        } else {
            if (full_only) TH_LAUNCH4(true, false, true); else TH_LAUNCH4(true, false, false);
        }
    } else {
        if (cnt) {
            if (full_only) TH_LAUNCH4(false, true, true); else TH_LAUNCH4(false, true, false);
        } else {
            if (full_only) TH_LAUNCH4(false, false, true); else TH_LAUNCH4(false, false, false);
        }
    }
}
#endif

#ifdef TH_DIAG_PHASES
extern "C" __attribute__((visibility("default"))) int trhip_debug_phases(uint64_t* out12, int reset) {  // DIAGNOSTIC build only (tools/phase_probe.py)
    unsigned long long h[16];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_phase), sizeof h) != hipSuccess) return -1;
    for (int i = 0; i < 13; ++i) out12[i] = h[i];
    if (reset) {
        std::memset(h, 0, sizeof h);
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase), h, sizeof h) != hipSuccess) return -1;
    }
    return 0;
}
#endif
