// tu_trace3c.hip — the hybrid mode's kernels (th_trace3c.h): the certified closest-hit walk on the accelerator tree (k_trace3c) and the one-leaf accelerator's
// walks (k_trace_leaf_c, k_any_leaf_c), instantiated here and nowhere else; tu_trace.hip's launch_trace decides when they run and sends their fallback lists
// through k_trace3 on the canonical tree.
#include "th_host.h"
#ifdef TRHIP_EXPERIMENTS
#include "th_trace3d.h"
#endif
#include "th_trace3c4.h"

// Every ray of a scene committed with both trees walks the accelerator when: the option is on, the default traversal is selected (the others are the A/B kernels
// and walk the canonical tree), the tight slab clauses are on (the certificate's margins derive from the same reach D), one pipeline, no diagnostic budget.
bool hybrid_active(const trhip_ctx* ctx, const trhip_scene* sc) {
    return sc->hybrid_ok && ctx->hybrid && ctx->traversal == 3 && ctx->slab_margin_log2 > 0 && ctx->pipelines <= 1 && ctx->debug_trace_budget == 0 && sc->wide_ok;
}

// Why a scene that holds both trees is NOT walked through its accelerator under the context's current options ("" when it is, or when the scene holds one tree): those
// launches walk the reference's tree alone — exact, at about twice the closest-hit time.  trhip_accelerator_note; a frame says it once on stderr (tu_path.hip).
const char* hybrid_idle_reason(const trhip_ctx* ctx, const trhip_scene* sc) {
    if (!sc->hybrid_ok || !sc->wide_ok) return "";
    if (!ctx->hybrid) return "option hybrid = 0: every ray walks the canonical tree";
    if (ctx->traversal != 3) return "option traversal != 3: only the default traversal walks the accelerator, the A/B kernels walk the canonical tree (pair them with bvh_builder = 0 for the library's tree)";
    if (ctx->slab_margin_log2 <= 0) return "option slab_margin_log2 = 0: the certificate's margins derive from the tight slab clauses";
    if (ctx->pipelines > 1) return "option pipelines > 1: the certified walk runs with one pipeline";
    if (ctx->debug_trace_budget != 0) return "option debug_trace_budget: diagnostic walks use the canonical tree";
    return "";
}
extern "C" __attribute__((visibility("default"))) int trhip_accelerator_note(const trhip_ctx* ctx, const trhip_scene* sc, char* buf, size_t n) {
    if (!ctx || !sc || !buf || !n) return TRHIP_ERR_INVALID;
    const char* why = hybrid_idle_reason(ctx, sc);
    if (!why[0] && hybrid_active(ctx, sc) && ctx->last_fallback_share > 0.2)
        std::snprintf(buf, n, "the certified walk handed %.0f %% of the last frame's closest-hit rays back to the reference-order walk (near-ties, grazed leaf boxes): exact, but the accelerator saves little", 100.0 * ctx->last_fallback_share);
    else
        std::snprintf(buf, n, "%s", why);
    return 0;
}

WideScene wide_view_acc(const trhip_ctx* ctx, const trhip_scene* sc) {
    WideScene w = sc->wide_acc;
    w.tight_scale = std::ldexp(1.0f, -ctx->slab_margin_log2);
    w.spec_spheres = 0xffffffffu;
    return w;
}

static CertScene cert_view(const trhip_ctx* ctx, const trhip_scene* sc) {
    CertScene c = sc->cert;
    c.inv_tight = std::ldexp(1.0f, ctx->slab_margin_log2);
    return c;
}

#ifdef TRHIP_EXPERIMENTS
#define TH_LAUNCH3D(CNTV, FULLV)                                                                                                                                       \
    do {                                                                                                                                                           \
        if (out.far_hint)                                                                                                                                          \
            hipLaunchKernelGGL((k_trace3d<CNTV, FULLV, true>), grid, block, 0, st, sc->dev_acc, wide_view_acc(ctx, sc), hot, cold, q, ro, rd, tmax, out, work_cursors, ov, ctr);  \
        else                                                                                                                                                       \
            hipLaunchKernelGGL((k_trace3d<CNTV, FULLV, false>), grid, block, 0, st, sc->dev_acc, wide_view_acc(ctx, sc), hot, cold, q, ro, rd, tmax, out, work_cursors, ov, ctr); \
    } while (0)
#endif
#define TH_LAUNCH3C(CNTV, FULLV, BIGV)                                                                                                                                 \
    do {                                                                                                                                                           \
        if (out.far_hint)                                                                                                                                          \
            hipLaunchKernelGGL((k_trace3c<CNTV, FULLV, BIGV, true>), grid, block, 0, st, sc->dev_acc, wide_view_acc(ctx, sc), hot, cold, q, ro, rd, tmax, out, work_cursors, ov, ctr);  \
        else                                                                                                                                                       \
            hipLaunchKernelGGL((k_trace3c<CNTV, FULLV, BIGV, false>), grid, block, 0, st, sc->dev_acc, wide_view_acc(ctx, sc), hot, cold, q, ro, rd, tmax, out, work_cursors, ov, ctr); \
    } while (0)

void launch_trace3c(trhip_ctx* ctx, hipStream_t st, const trhip_scene* sc, bool cnt, bool full_only, bool big, const SegQueue& q, const float4* ro, const float4* rd, const float* tmax,
                    const TraceOut& out, uint32_t* work_cursors, uint2* ov, Counters* ctr, const FallbackList& fb) {
    const dim3 grid(trace_grid(ctx)), block(kBlock);
    const CertScene cv = cert_view(ctx, sc);
    // what the kernel reads only at a ray's fetch / when it hands a ray back: one struct in HBM, written in stream order by a one-thread kernel (its value travels in that
    // launch's own argument buffer: no host memory has to outlive the call)
    if (ensure(ctx, ctx->cert_cold, sizeof(CertCold)) != 0) return;
    CertCold* cold = (CertCold*)ctx->cert_cold.p;
    CertCold cc{};
    cc.sphere_boxes = cv.sphere_boxes;
    cc.sphere_slots = cv.sphere_slots;
    cc.fb_list = fb.list;
    cc.fb_counts = fb.counts;
    cc.n_spheres = cv.n_spheres;
    cc.fb_cap = fb.cap;
    for (int a = 0; a < 3; ++a) cc.mle_small[a] = cv.mle_small[a];
    cc.sq_flat = cv.sq_flat;
    cc.inv_tight = cv.inv_tight;
    hipLaunchKernelGGL(k_store_cert_cold, dim3(1), dim3(1), 0, st, cold, cc);
    const CertHot hot{kCertDt * cv.inv_tight, kCertGrow * cv.inv_tight, kCertFlat * cv.sq_flat, cv.n_spheres, (const SphereCert*)cv.sphere_cert, {cv.mle_small[0], cv.mle_small[1], cv.mle_small[2]}};
#ifdef TRHIP_EXPERIMENTS
    if (ctx->leaf_queue && !big) {  // option "leaf_queue": the same walk with queued leaves (th_trace3d.h)
        if (cnt) {
            if (full_only) TH_LAUNCH3D(true, true); else TH_LAUNCH3D(true, false);
        } else {
            if (full_only) TH_LAUNCH3D(false, true); else TH_LAUNCH3D(false, false);
        }
        return;
    }
#endif
    const WideScene wv4 = wide_view_acc(ctx, sc);
    if (ctx->wide4 && wv4.w4nodes) {  // the accelerator four children wide (th_trace3c4.h): one form for every launch
#define TH_LAUNCH3C4(CNTV, FULLV, BIGV)                                                                                                                                     \
    do {                                                                                                                                                                \
        if (out.far_hint)                                                                                                                                               \
            hipLaunchKernelGGL((k_trace3c4<CNTV, FULLV, BIGV, true>), grid, block, 0, st, sc->dev_acc, wv4, hot, cold, q, ro, rd, tmax, out, work_cursors, ov, ctr);    \
        else                                                                                                                                                            \
            hipLaunchKernelGGL((k_trace3c4<CNTV, FULLV, BIGV, false>), grid, block, 0, st, sc->dev_acc, wv4, hot, cold, q, ro, rd, tmax, out, work_cursors, ov, ctr);   \
    } while (0)
        if (cnt) {
            if (full_only) TH_LAUNCH3C4(true, true, false); else TH_LAUNCH3C4(true, false, false);
        } else if (big) {
            if (full_only) TH_LAUNCH3C4(false, true, true); else TH_LAUNCH3C4(false, false, true);
        } else {
            if (full_only) TH_LAUNCH3C4(false, true, false); else TH_LAUNCH3C4(false, false, false);
        }
#undef TH_LAUNCH3C4
        return;
    }
    if (cnt) {
        if (full_only) TH_LAUNCH3C(true, true, false); else TH_LAUNCH3C(true, false, false);
    } else if (big) {
        if (full_only) TH_LAUNCH3C(false, true, true); else TH_LAUNCH3C(false, false, true);
    } else {
        if (full_only) TH_LAUNCH3C(false, true, false); else TH_LAUNCH3C(false, false, false);
    }
}

void launch_leaf_c(trhip_ctx* ctx, hipStream_t st, const trhip_scene* sc, bool any, bool cnt, bool full_only, const SegQueue& q, const float4* ro, const float4* rd, const float* tmax,
                   const TraceOut& out, Counters* ctr, const FallbackList& fb) {
    const dim3 lgrid(ctx->num_cu * 8), block(kBlock);
    const WideScene wv = wide_view_acc(ctx, sc);
    const CertScene cv = cert_view(ctx, sc);
#define TH_LEAFC(K, C, F) hipLaunchKernelGGL((K<C, F>), lgrid, block, 0, st, sc->dev, wv, cv, q, ro, rd, tmax, out, ctr, fb)
    if (any) {
        if (cnt) { if (full_only) TH_LEAFC(k_any_leaf_c, true, true); else TH_LEAFC(k_any_leaf_c, true, false); }
        else { if (full_only) TH_LEAFC(k_any_leaf_c, false, true); else TH_LEAFC(k_any_leaf_c, false, false); }
    } else {
        if (cnt) { if (full_only) TH_LEAFC(k_trace_leaf_c, true, true); else TH_LEAFC(k_trace_leaf_c, true, false); }
        else { if (full_only) TH_LEAFC(k_trace_leaf_c, false, true); else TH_LEAFC(k_trace_leaf_c, false, false); }
    }
#undef TH_LEAFC
}

#ifdef TH_DIAG_PHASES
extern "C" __attribute__((visibility("default"))) int trhip_debug_phases_c(uint64_t* out12, int reset) {  // DIAGNOSTIC build only (tools/phase_probe.py): k_trace3c's phases
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
