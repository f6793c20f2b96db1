// tu_trace.hip — which traversal kernel walks a queue (launch_trace) and the kernel-level trace entry points.  The k_trace3 / k_trace4 / k_trace8
// families are instantiated in tu_trace3.hip / tu_trace8.hip.
#include "th_host.h"
#ifdef TRHIP_EXPERIMENTS
#include "th_leaf2.h"
#endif

#ifndef TH_TRACE_BLOCKS_PER_CU
#define TH_TRACE_BLOCKS_PER_CU 6
#endif
int trace_grid(const trhip_ctx* ctx) { return ctx->num_cu * TH_TRACE_BLOCKS_PER_CU; }  // persistent blocks per CU (LDS stack: kStack2Lds x 256 x 8 B each)

// k_trace2 keeps stack levels 16..63 of every resident thread in a global slab laid out [level][thread].
int ensure_overflow(trhip_ctx* ctx) {
    const size_t threads = (size_t)trace_grid(ctx) * kBlock;
    return ensure(ctx, ctx->overflow, threads * (size_t)kStackSlabLevels * sizeof(uint2));
}

// the scene's children-in-parent view with the context's slab margin (option "slab_margin_log2")
WideScene wide_view(const trhip_ctx* ctx, const trhip_scene* sc) {
    WideScene w = sc->wide;
    w.tight_scale = ctx->slab_margin_log2 > 0 ? std::ldexp(1.0f, -ctx->slab_margin_log2) : 0.0f;
    // k_trace3's postponed leaves need a non-increasing t_max: the rays that can see it raised start inside (or on) a sphere (A.18) — inside its world bound
    // grown by a thousandth of its size
    w.spec_spheres = 0xffffffffu;
    if (ctx->trace3_spec && sc->sphere_bounds.size() <= 8) {
        w.spec_spheres = (uint32_t)sc->sphere_bounds.size();
        for (size_t k = 0; k < sc->sphere_bounds.size(); ++k) {
            const HostAABB& b = sc->sphere_bounds[k];
            for (int a = 0; a < 3; ++a) {
                const float grow = 1e-3f * (b.mx[a] - b.mn[a]) + 1e-6f * std::fmax(std::fabs(b.mn[a]), std::fabs(b.mx[a])) + 1e-30f;
                w.spec_box[k][a] = b.mn[a] - grow;
                w.spec_box[k][3 + a] = b.mx[a] + grow;
            }
        }
    }
    return w;
}

// k_trace8 takes the launch (traversal 4) only with the tight slab clauses on and a single pipeline; otherwise k_trace3 walks the 32-byte boxes.
// One place for the rule: launch_trace and the byte model of trhip_stats (traversal_info) must agree.
#ifdef TRHIP_EXPERIMENTS
static bool uses_trace8(const trhip_ctx* ctx, const trhip_scene* sc) { return ctx->traversal == 4 && sc->w8_ok && ctx->slab_margin_log2 > 0 && ctx->pipelines <= 1; }
#else  // traversals 4, 6, 7, leaf_sorted, leaf_queue: kernels of the EXPERIMENTS build (trhip_set_option refuses them here)
static bool uses_trace8(const trhip_ctx*, const trhip_scene*) { return false; }
#endif

// k_trace7 (traversal 7) takes the closest-hit launches of scenes with a hierarchy when the tight slab clauses are on (its margins derive from them)
#ifdef TRHIP_EXPERIMENTS
static bool uses_trace7(const trhip_ctx* ctx, const trhip_scene* sc) { return ctx->traversal == 7 && sc->wide_ok && sc->wide.root_cnt == 0 && ctx->slab_margin_log2 > 0 && ctx->pipelines <= 1; }
#else
static bool uses_trace7(const trhip_ctx*, const trhip_scene*) { return false; }
#endif

#ifdef TRHIP_EXPERIMENTS
static constexpr bool kExperimentsBuild = true;
#else
static constexpr bool kExperimentsBuild = false;
#endif
// which kernel launch_trace picks for this scene, and the bytes one unit of the visit counters stands for (trhip_stats)
void traversal_info(const trhip_ctx* ctx, const trhip_scene* sc, uint32_t* trav, uint32_t* node_bytes) {
    uint32_t t = 1, nb = 32;
    if (hybrid_active(ctx, sc)) {
        *trav = 9;  // the certified walk on the accelerator tree + the reference-order walk of the flagged rays on the canonical tree (th_trace3c.h)
        *node_bytes = sc->wide_acc.root_cnt > 0 ? 0 : 32;
        return;
    }
    if (ctx->traversal >= 2 && sc->wide_ok) {
        if (sc->wide.root_cnt > 0 && ctx->leaf_kernel && ctx->debug_trace_budget == 0) {
            t = 5;
            nb = 0;
        } else if (sc->wide.root_cnt > 0) {
            t = 2;
        } else if (uses_trace8(ctx, sc)) {
            t = 4;
            nb = 96;
        } else if (ctx->traversal == 6) {
            t = 6;
        } else if (uses_trace7(ctx, sc)) {
            t = 7;
        } else {
            t = ctx->traversal >= 3 ? 3 : 2;
        }
    }
    *trav = t;
    *node_bytes = nb;
}

// The name of the kernel a closest-hit launch on this scene runs under the context's CURRENT options (launch_trace's own decisions, restated in one place for the
// measurement side: a profile filter or a roofline must name the kernel that ran, not the one the option string suggests).
extern "C" __attribute__((visibility("default"))) int trhip_closest_kernel_name(const trhip_ctx* ctx, const trhip_scene* sc, char* buf, size_t n) {
    if (!ctx || !sc || !buf || !n) return TRHIP_ERR_INVALID;
    const char* name = "k_trace_closest";
    if (hybrid_active(ctx, sc)) {
        name = sc->wide_acc.root_cnt > 0 ? "k_trace_leaf_c" : ((ctx->wide4 && sc->wide_acc.w4nodes && !(kExperimentsBuild && ctx->leaf_queue)) ? "k_trace3c4" : (kExperimentsBuild && ctx->leaf_queue ? "k_trace3d" : "k_trace3c"));
    } else {
        uint32_t t = 1, nb = 0;
        traversal_info(ctx, sc, &t, &nb);
        static const char* const kNames[] = {"k_trace_closest", "k_trace_closest", "k_trace2", "k_trace3", "k_trace8", "k_trace_leaf", "k_trace4", "k_trace7"};
        name = kNames[t < 8 ? t : 0];
    }
    std::snprintf(buf, n, "%s", name);
    return 0;
}

// One traversal launch over a queue (count in HBM at count_ptr, or n_max when count_ptr is null).
// ctx->traversal == 1: the literal accel/bvh.jl loop (k_trace_closest / k_trace_any); 2: k_trace2 (same results).
void launch_trace(trhip_ctx* ctx, hipStream_t st, const trhip_scene* sc, bool any, SegQueue q, const float4* ro, const float4* rd, const float* tmax, TraceOut out, uint32_t* work_cursors,
                  Counters* ctr, void* overflow_slab) {
    const dim3 grid(trace_grid(ctx)), block(kBlock);
    const bool v2 = ctx->traversal >= 2 && sc->wide_ok;
    const bool cnt = ctx->count_visits;
    const bool full_only = !sc->partial_spheres;  // no clipped sphere in the scene: kernels without the Float64 atan2 path
    // ---- hybrid mode (th_trace3c.h): the scene holds the canonical tree (the reference's construction / the host's own) AND the library's tree as an accelerator.
    //      Closest-hit rays walk the accelerator with the order-independence certificate; the rays it flags come back on fallback lists that k_trace3 walks on the
    //      canonical tree right after.  Any-hit rays of a one-leaf accelerator likewise (only rays with a zero direction component need the canonical tree); any-hit
    //      rays of a hierarchy take the canonical path below (pre-pass on the largest triangles + k_trace3 on the canonical tree).
    if (hybrid_active(ctx, sc) && !q.indirect && (!any || sc->wide_acc.root_cnt > 0)) {
        const int w = any ? 1 : 0;
        const uint32_t fcap = q.counts ? q.cap : (q.n_dense + kSeg - 1) / kSeg;
        const size_t ctr_words = 2 * (size_t)kSeg * kCtrStride;  // counts, then the work cursors of the fallback launch
        if (ensure(ctx, ctx->fb_list[w], (size_t)fcap * kSeg * sizeof(uint32_t)) == 0 && ensure(ctx, ctx->fb_counts[w], ctr_words * sizeof(uint32_t)) == 0) {
            uint32_t* fcounts = (uint32_t*)ctx->fb_counts[w].p;
            (void)hipMemsetAsync(fcounts, 0, ctr_words * sizeof(uint32_t), st);
            const FallbackList fb{(uint32_t*)ctx->fb_list[w].p, fcounts, fcap};
            uint2* ov = (uint2*)(overflow_slab ? overflow_slab : ctx->overflow.p);
            const size_t canon_bytes = (size_t)sc->wide.n_wnodes * 64u + (size_t)sc->dev.n_prims * 48u;
            if (sc->wide_acc.root_cnt > 0) {
                launch_leaf_c(ctx, st, sc, any, cnt, full_only, q, ro, rd, tmax, out, ctr, fb);
            } else {
                const bool big = !cnt && (size_t)sc->wide_acc.n_wnodes * 64u + (size_t)sc->dev.n_prims * 48u > ((size_t)256 << 20);
                launch_trace3c(ctx, st, sc, cnt, full_only, big, q, ro, rd, tmax, out, work_cursors, ov, ctr, fb);
            }
            // the flagged rays, in the reference's order on the canonical tree (already counted: no_total)
            const SegQueue fq{fcounts, fcap, 0u, fb.list, 1u};
            if (!any && ctx->active_timer) ctx->active_timer->mark_fallback(st);
            if (!any && cnt && ctr) hipLaunchKernelGGL(k_hybrid_count_mark, dim3(1), dim3(1), 0, st, ctr, 0);
            launch_trace3(ctx, st, sc, any, cnt, full_only, !any && !cnt && canon_bytes > ((size_t)256 << 20), fq, ro, rd, tmax, out, fcounts + (size_t)kSeg * kCtrStride, ov, ctr);
            if (!any && cnt && ctr) hipLaunchKernelGGL(k_hybrid_count_mark, dim3(1), dim3(1), 0, st, ctr, 1);
            return;
        }
    }
    if (v2 && ctx->traversal >= 3 && sc->wide.root_cnt == 0) {  // k_trace8 / k_trace3; a single-leaf scene has nothing to postpone and runs k_trace_leaf / k_trace2
        uint2* ov = (uint2*)(overflow_slab ? overflow_slab : ctx->overflow.p);
        if (any && sc->n_occluders && ctx->occluder_pretest && ctx->pipelines <= 1 && !q.indirect) {
            // the largest triangles first (k_any_occluders); what they do not stop goes through per-segment survivor lists
            // a survivor list takes the rays of every kSeg-th 64-ray chunk of the padded work space (k_any_occluders): at most (sum of the segment
            // counts + kSeg * (kSegGran - 1)) / kSeg + 64 <= the queue's per-segment capacity + 319 entries, whatever the caller's slack
            const uint32_t scap = (q.counts ? q.cap : q.n_dense) + 1024u;
            const size_t entries = (size_t)scap * (q.counts ? kSeg : 1);
            if (ensure(ctx, ctx->surv_list, entries * sizeof(uint32_t)) == 0 && ensure(ctx, ctx->surv_counts, (size_t)kSeg * kCtrStride * sizeof(uint32_t)) == 0) {
                uint32_t* sl = (uint32_t*)ctx->surv_list.p;
                uint32_t* scn = (uint32_t*)ctx->surv_counts.p;
                (void)hipMemsetAsync(scn, 0, (size_t)kSeg * kCtrStride * sizeof(uint32_t), st);
                const OccluderSet oc{(const uint32_t*)sc->d_occ_slots.p, (const float*)sc->d_occ_boxes.p, sc->n_occluders};
                const dim3 pgrid(ctx->num_cu * 8);
                if (cnt)
                    hipLaunchKernelGGL((k_any_occluders<true>), pgrid, block, 0, st, sc->dev, oc, q, ro, rd, tmax, out, sl, scn, scap, ctr);
                else
                    hipLaunchKernelGGL((k_any_occluders<false>), pgrid, block, 0, st, sc->dev, oc, q, ro, rd, tmax, out, sl, scn, scap, ctr);
                q = SegQueue{scn, scap, 0u, sl, 1u};
            }
        }
#ifdef TRHIP_EXPERIMENTS
        // ---- traversal 4: 8-wide nodes (th_trace8.h); the rays it does not take come back on a fallback list that k_trace3 walks below ----
        if (uses_trace8(ctx, sc)) {
            const int w = any ? 1 : 0;
            const uint32_t fcap = q.counts ? q.cap : q.n_dense;
            const size_t entries = (size_t)fcap * (q.counts ? kSeg : 1);
            const size_t ctr_words = 2 * (size_t)kSeg * kCtrStride;  // counts, then the work cursors of the fallback launch
            const size_t ov8_bytes = (size_t)trace_grid(ctx) * kBlock * (size_t)kStack8Global * 3 * sizeof(uint32_t);
            if (ensure(ctx, ctx->fb_list[w], entries * sizeof(uint32_t)) == 0 && ensure(ctx, ctx->fb_counts[w], ctr_words * sizeof(uint32_t)) == 0 && ensure(ctx, ctx->ov8[w], ov8_bytes) == 0) {
                uint32_t* fcounts = (uint32_t*)ctx->fb_counts[w].p;
                (void)hipMemsetAsync(fcounts, 0, ctr_words * sizeof(uint32_t), st);
                Wide8Scene w8 = sc->w8;
                w8.tight_scale = std::ldexp(1.0f, -ctx->slab_margin_log2);
                const FallbackList fb{(uint32_t*)ctx->fb_list[w].p, fcounts, fcap};
                uint32_t* ov8 = (uint32_t*)ctx->ov8[w].p;
                launch_trace8(ctx, st, sc, any, cnt, full_only, w8, q, ro, rd, tmax, out, work_cursors, ov8, ctr, fb);
                q = SegQueue{fcounts, fcap, 0u, fb.list, 1u};
                work_cursors = fcounts + (size_t)kSeg * kCtrStride;
            }
        }
        // ---- traversal 7: closest-hit rays front to back (th_trace7.h); the rays whose answer depends on the visiting order come back on ONE fallback list
        //      (segment 0 of a SegQueue whose other segments are empty) that k_trace3 walks below ----
        if (!any && uses_trace7(ctx, sc) && !q.indirect) {
            // kSeg lists of `fcap` entries: together as many as the queue holds rays (a wave whose list is full moves on to the next one)
            const uint32_t fcap = q.counts ? q.cap : (q.n_dense + kSeg - 1) / kSeg;
            const size_t ctr_words = 2 * (size_t)kSeg * kCtrStride;  // counts, then the work cursors of the fallback launch
            if (ensure(ctx, ctx->fb_list[0], (size_t)fcap * kSeg * sizeof(uint32_t)) == 0 && ensure(ctx, ctx->fb_counts[0], ctr_words * sizeof(uint32_t)) == 0) {
                uint32_t* fcounts = (uint32_t*)ctx->fb_counts[0].p;
                (void)hipMemsetAsync(fcounts, 0, ctr_words * sizeof(uint32_t), st);
                const FallbackList fb{(uint32_t*)ctx->fb_list[0].p, fcounts, fcap};
                const bool big7 = !cnt && (size_t)sc->wide.n_wnodes * 64u + (size_t)sc->dev.n_prims * 48u > ((size_t)256 << 20);
                launch_trace7(ctx, st, sc, cnt, full_only, big7, q, ro, rd, tmax, out, work_cursors, ov, ctr, fb);
                q = SegQueue{fcounts, fcap, 0u, fb.list, 1u};
                work_cursors = fcounts + (size_t)kSeg * kCtrStride;
            }
        }
        if (ctx->traversal == 6) {  // two rays per lane (th_trace4.h)
            launch_trace4(ctx, st, sc, any, cnt, full_only, q, ro, rd, tmax, out, work_cursors, ov, ctr);
            return;
        }
#endif
        // scenes larger than the last-level cache (256 MB of MALL): one wave per SIMD fewer (k_trace3's BIG variant)
        const bool big = !any && !cnt && (size_t)sc->wide.n_wnodes * 64u + (size_t)sc->dev.n_prims * 48u > ((size_t)256 << 20);
        if (any && hybrid_active(ctx, sc) && (ctx->any_on_accelerator > 0 || (ctx->any_on_accelerator < 0 && out.any_acc_hint)) && sc->wide_acc.root_cnt == 0) {
            // any-hit rays of a two-tree scene (TraceOut::zero_mode): the library's tree for every ray without a zero direction component, the canonical tree for the rest
            const size_t words = 16 + (size_t)kSeg * kCtrStride;  // the flag, then the second launch's work cursors
            if (ensure(ctx, ctx->fb_counts[1], words * sizeof(uint32_t)) == 0) {
                uint32_t* zf = (uint32_t*)ctx->fb_counts[1].p;
                (void)hipMemsetAsync(zf, 0, words * sizeof(uint32_t), st);
                TraceOut o1 = out, o2 = out;
                o1.zero_mode = 1u;
                o2.zero_mode = 2u;
                o1.zero_flag = o2.zero_flag = zf;
                launch_trace3(ctx, st, sc, true, cnt, full_only, false, q, ro, rd, tmax, o1, work_cursors, ov, ctr, true);
                SegQueue q2 = q;
                q2.no_total = 1u;  // (counted by the first launch)
                launch_trace3(ctx, st, sc, true, cnt, full_only, false, q2, ro, rd, tmax, o2, zf + 16, ov, ctr, false);
                return;
            }
        }
        launch_trace3(ctx, st, sc, any, cnt, full_only, big, q, ro, rd, tmax, out, work_cursors, ov, ctr);
        return;
    }
    if (v2) {
        if (sc->wide.root_cnt > 0 && ctx->debug_trace_budget == 0 && ctx->leaf_kernel) {  // one-leaf scene: the dedicated kernel (th_trace2.h, k_trace_leaf)
            const dim3 lgrid(ctx->num_cu * 8);
#ifdef TRHIP_EXPERIMENTS
            if (ctx->leaf_sorted && sc->d_leaf_boxes.p && sc->wide.root_cnt <= 30 && ctx->slab_margin_log2 > 0) {  // rays grouped by what they can hit (th_leaf2.h)
                const float* lb = (const float*)sc->d_leaf_boxes.p;
                const WideScene wsv = wide_view(ctx, sc);
#define TH_LEAF2(A, C, F) hipLaunchKernelGGL((k_leaf_sorted<A, C, F>), lgrid, block, 0, st, sc->dev, wsv, q, ro, rd, tmax, out, ctr, lb)
                if (any) {
                    if (cnt) { if (full_only) TH_LEAF2(true, true, true); else TH_LEAF2(true, true, false); }
                    else { if (full_only) TH_LEAF2(true, false, true); else TH_LEAF2(true, false, false); }
                } else {
                    if (cnt) { if (full_only) TH_LEAF2(false, true, true); else TH_LEAF2(false, true, false); }
                    else { if (full_only) TH_LEAF2(false, false, true); else TH_LEAF2(false, false, false); }
                }
#undef TH_LEAF2
                return;
            }
#endif
            if (any) {
                if (cnt)
                    { if (full_only) hipLaunchKernelGGL((k_any_leaf<true, true>), lgrid, block, 0, st, sc->dev, wide_view(ctx, sc), q, ro, rd, tmax, out, ctr); else hipLaunchKernelGGL((k_any_leaf<true, false>), lgrid, block, 0, st, sc->dev, wide_view(ctx, sc), q, ro, rd, tmax, out, ctr); }
                else
                    { if (full_only) hipLaunchKernelGGL((k_any_leaf<false, true>), lgrid, block, 0, st, sc->dev, wide_view(ctx, sc), q, ro, rd, tmax, out, ctr); else hipLaunchKernelGGL((k_any_leaf<false, false>), lgrid, block, 0, st, sc->dev, wide_view(ctx, sc), q, ro, rd, tmax, out, ctr); }
            } else {
                if (cnt)
                    { if (full_only) hipLaunchKernelGGL((k_trace_leaf<false, true, true>), lgrid, block, 0, st, sc->dev, wide_view(ctx, sc), q, ro, rd, tmax, out, ctr); else hipLaunchKernelGGL((k_trace_leaf<false, true, false>), lgrid, block, 0, st, sc->dev, wide_view(ctx, sc), q, ro, rd, tmax, out, ctr); }
                else
                    { if (full_only) hipLaunchKernelGGL((k_trace_leaf<false, false, true>), lgrid, block, 0, st, sc->dev, wide_view(ctx, sc), q, ro, rd, tmax, out, ctr); else hipLaunchKernelGGL((k_trace_leaf<false, false, false>), lgrid, block, 0, st, sc->dev, wide_view(ctx, sc), q, ro, rd, tmax, out, ctr); }
            }
            return;
        }
        uint2* ov = (uint2*)(overflow_slab ? overflow_slab : ctx->overflow.p);
        if (any) {
            if (cnt)
                hipLaunchKernelGGL((k_trace2<true, true>), grid, block, 0, st, sc->dev, wide_view(ctx, sc), q, ro, rd, tmax, out, work_cursors, ov, ctr, ctx->debug_trace_budget);
            else
                hipLaunchKernelGGL((k_trace2<true, false>), grid, block, 0, st, sc->dev, wide_view(ctx, sc), q, ro, rd, tmax, out, work_cursors, ov, ctr, ctx->debug_trace_budget);
        } else {
            if (cnt)
                hipLaunchKernelGGL((k_trace2<false, true>), grid, block, 0, st, sc->dev, wide_view(ctx, sc), q, ro, rd, tmax, out, work_cursors, ov, ctr, ctx->debug_trace_budget);
            else
                hipLaunchKernelGGL((k_trace2<false, false>), grid, block, 0, st, sc->dev, wide_view(ctx, sc), q, ro, rd, tmax, out, work_cursors, ov, ctr, ctx->debug_trace_budget);
        }
        return;
    }
    if (any) {
        if (cnt)
            hipLaunchKernelGGL(k_trace_any<true>, grid, block, 0, st, sc->dev, q, ro, rd, out.contrib, tmax, out.L, out.occluded, ctr);
        else
            hipLaunchKernelGGL(k_trace_any<false>, grid, block, 0, st, sc->dev, q, ro, rd, out.contrib, tmax, out.L, out.occluded, ctr);
    } else {
        if (cnt)
            hipLaunchKernelGGL(k_trace_closest<true>, grid, block, 0, st, sc->dev, q, ro, rd, tmax, out.hits, ctr);
        else
            hipLaunchKernelGGL(k_trace_closest<false>, grid, block, 0, st, sc->dev, q, ro, rd, tmax, out.hits, ctr);
    }
}

// the streaming wavefront's rounds (tu_path.hip, render_stream_impl): k_trace2 with the suspend / resume logic
void launch_trace2_stream(trhip_ctx* ctx, hipStream_t st, const trhip_scene* sc, bool any, SegQueue q, const float4* ro, const float4* rd, TraceOut out, uint32_t* work_cursors, void* overflow_slab,
                          Counters* ctr, const StreamCtl& sx) {
    const dim3 grid(trace_grid(ctx)), block(kBlock);
    const bool cnt = ctx->count_visits;
    uint2* ov = (uint2*)overflow_slab;
    if (any) {
        if (cnt)
            hipLaunchKernelGGL((k_trace2<true, true, true>), grid, block, 0, st, sc->dev, wide_view(ctx, sc), q, ro, rd, nullptr, out, work_cursors, ov, ctr, 0u, sx);
        else
            hipLaunchKernelGGL((k_trace2<true, false, true>), grid, block, 0, st, sc->dev, wide_view(ctx, sc), q, ro, rd, nullptr, out, work_cursors, ov, ctr, 0u, sx);
    } else {
        if (cnt)
            hipLaunchKernelGGL((k_trace2<false, true, true>), grid, block, 0, st, sc->dev, wide_view(ctx, sc), q, ro, rd, nullptr, out, work_cursors, ov, ctr, 0u, sx);
        else
            hipLaunchKernelGGL((k_trace2<false, false, true>), grid, block, 0, st, sc->dev, wide_view(ctx, sc), q, ro, rd, nullptr, out, work_cursors, ov, ctr, 0u, sx);
    }
}

extern "C" {

// Shared body of the four kernel-level trace entry points: stage rays (host or device, n*8 floats) into SoA, run the
// traversal `repeat` times, time it with HIP events on the library's stream.
static int api_trace(trhip_ctx* ctx, const trhip_scene* sc, bool any, const void* rays, bool rays_on_device, uint64_t n, void* d_out, int repeat, double* avg_ms) {
    if (!sc->committed) return fail(ctx, TRHIP_ERR_INVALID, "scene not committed");
    if (n >= (1ull << 31)) return fail(ctx, TRHIP_ERR_INVALID, "too many rays in one call");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const float* d_rays = (const float*)rays;
    if (!rays_on_device) {
        if (int rc = upload(ctx, ctx->scratch[3], rays, n * 8 * sizeof(float))) return rc;
        d_rays = (const float*)ctx->scratch[3].p;
    }
    for (int j = 0; j < 2; ++j)
        if (int rc = ensure(ctx, ctx->scratch[j], n * sizeof(float4))) return rc;
    if (int rc = ensure(ctx, ctx->scratch[2], n * sizeof(float))) return rc;
    if (int rc = ensure(ctx, ctx->counters, sizeof(Counters))) return rc;
    if (int rc = ensure_overflow(ctx)) return rc;
    if (n) hipLaunchKernelGGL(k_prepare_rays, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_rays, (uint32_t)n, (float4*)ctx->scratch[0].p, (float4*)ctx->scratch[1].p,
                              (float*)ctx->scratch[2].p);
    Counters* ctr = (Counters*)ctx->counters.p;
    HIP_TRY(ctx, hipMemsetAsync(ctr, 0, sizeof(Counters), ctx->stream));
    hipEvent_t e0, e1;
    HIP_TRY(ctx, hipEventCreate(&e0));
    HIP_TRY(ctx, hipEventCreate(&e1));
    repeat = std::max(1, std::min(repeat, kMaxDepth + 1));
    TraceOut out{any ? nullptr : (float4*)d_out, nullptr, nullptr, any ? (uint8_t*)d_out : nullptr};
    HIP_TRY(ctx, hipEventRecord(e0, ctx->stream));
    if (n)
        for (int r = 0; r < repeat; ++r)  // every repetition uses its own (zeroed) work cursor
            launch_trace(ctx, ctx->stream, sc, any, SegQueue{nullptr, (uint32_t)n, (uint32_t)n}, (const float4*)ctx->scratch[0].p, (const float4*)ctx->scratch[1].p, (const float*)ctx->scratch[2].p,
                         out, any ? ctr->work_shadow[r] : ctr->work_closest[r], ctr);
    HIP_TRY(ctx, hipEventRecord(e1, ctx->stream));
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (avg_ms) *avg_ms = ms / repeat;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return 0;
}
int trhip_trace_closest(trhip_ctx* ctx, const trhip_scene* sc, const float* rays, uint64_t n, trhip_hit* out) {
    if (!ctx || !sc || !rays || !out) return fail(ctx, TRHIP_ERR_INVALID, "null argument");
    static_assert(sizeof(trhip_hit) == sizeof(float4), "trhip_hit layout");
    if (int rc = ensure(ctx, ctx->hits, n * sizeof(float4))) return rc;
    if (int rc = api_trace(ctx, sc, false, rays, false, n, ctx->hits.p, 1, nullptr)) return rc;
    HIP_TRY(ctx, hipMemcpy(out, ctx->hits.p, n * sizeof(float4), hipMemcpyDeviceToHost));
    return 0;
}
int trhip_trace_any(trhip_ctx* ctx, const trhip_scene* sc, const float* rays, uint64_t n, uint8_t* occluded) {
    if (!ctx || !sc || !rays || !occluded) return fail(ctx, TRHIP_ERR_INVALID, "null argument");
    if (int rc = ensure(ctx, ctx->hits, n)) return rc;
    if (int rc = api_trace(ctx, sc, true, rays, false, n, ctx->hits.p, 1, nullptr)) return rc;
    HIP_TRY(ctx, hipMemcpy(occluded, ctx->hits.p, n, hipMemcpyDeviceToHost));
    return 0;
}
// d_rays: n*8 floats on the device (same layout as the host entry points); d_hits: n trhip_hit / d_occ: n bytes
int trhip_trace_closest_device(trhip_ctx* ctx, const trhip_scene* sc, const void* d_rays, uint64_t n, void* d_hits, int repeat, double* avg_ms) {
    if (!ctx || !sc || !d_rays || !d_hits) return fail(ctx, TRHIP_ERR_INVALID, "null argument");
    return api_trace(ctx, sc, false, d_rays, true, n, d_hits, repeat, avg_ms);
}
int trhip_trace_any_device(trhip_ctx* ctx, const trhip_scene* sc, const void* d_rays, uint64_t n, void* d_occ, int repeat, double* avg_ms) {
    if (!ctx || !sc || !d_rays || !d_occ) return fail(ctx, TRHIP_ERR_INVALID, "null argument");
    return api_trace(ctx, sc, true, d_rays, true, n, d_occ, repeat, avg_ms);
}
// visit counters of the last *_device trace call (when "count_visits" is on): nodes, prims for closest then shadow
int trhip_last_visit_counts(trhip_ctx* ctx, uint64_t* out4) {
    if (!ctx || !out4 || !ctx->counters.p) return fail(ctx, TRHIP_ERR_INVALID, "no counters");
    Counters h;
    HIP_TRY(ctx, hipMemcpy(&h, ctx->counters.p, sizeof h, hipMemcpyDeviceToHost));
    out4[0] = h.nodes_closest;
    out4[1] = h.prims_closest;
    out4[2] = h.nodes_shadow;
    out4[3] = h.prims_shadow;
    return 0;
}
int trhip_last_fallback_counts(trhip_ctx* ctx, uint64_t* out2) {
    if (!ctx || !out2 || !ctx->counters.p) return fail(ctx, TRHIP_ERR_INVALID, "no counters");
    Counters h;
    HIP_TRY(ctx, hipMemcpy(&h, ctx->counters.p, sizeof h, hipMemcpyDeviceToHost));
    out2[0] = h.closest_total;
    out2[1] = h.fallback_total;
    return 0;
}

int trhip_hit_geometry(trhip_ctx* ctx, const trhip_scene* sc, const float* rays, uint64_t n, float* out15) {
    if (!ctx || !sc || !rays || !out15) return fail(ctx, TRHIP_ERR_INVALID, "null argument");
    if (int rc = ensure(ctx, ctx->hits, n * sizeof(float4))) return rc;
    if (int rc = ensure(ctx, ctx->film, n * 15 * sizeof(float))) return rc;
    if (int rc = api_trace(ctx, sc, false, rays, false, n, ctx->hits.p, 1, nullptr)) return rc;
    if (n) hipLaunchKernelGGL(k_hit_geometry, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, sc->dev, (const float4*)ctx->scratch[0].p, (const float4*)ctx->scratch[1].p,
                              (const float4*)ctx->hits.p, (uint32_t)n, (float*)ctx->film.p);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(out15, ctx->film.p, n * 15 * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

}  // extern "C"
