// tu_whitted.hip — WhittedIntegrator: the ray tree built level by level and folded bottom-up (th_whitted.h).
#include "th_host.h"
#include "th_whitted.h"

// WhittedIntegrator: ray tree built level by level, folded bottom-up (th_whitted.h).
int render_whitted_impl(trhip_ctx* ctx, const trhip_scene* scene, const DeviceSensor& ds, const trhip_sensor* sensor, uint32_t spp, int max_depth, uint64_t seed, uint32_t sample_offset,
                        void* d_film, trhip_stats* stats, double* ms_total) {
    const uint64_t npix = (uint64_t)ds.sb_w * ds.sb_h;
    const uint64_t total_slots = npix * spp;
    const uint32_t n_lights = std::max<uint32_t>(1u, scene->dev.n_lights);
    // bytes per camera ray of a batch: tree pool (growth factor 2 per level budgeted) + queues
    const double per_ray = 2.0 * (max_depth * 44.0 + 2 * 2 * 16.0 + 16.0 + n_lights * (48.0 + 1.0));
    uint64_t batch = ctx->batch_paths;
    if (batch == 0) {
        size_t free_b = 0, total_b = 0;
        HIP_TRY(ctx, hipMemGetInfo(&free_b, &total_b));
        const double avail = 0.6 * (double)free_b - (double)total_slots * 24.0;
        batch = avail > 0 ? (uint64_t)(avail / per_ray) : npix;
    }
    uint64_t spp_batch = std::min<uint64_t>(std::max<uint64_t>(1, batch / npix), spp);
    auto phys_of = [&](uint64_t n1) { return (uint64_t)(((2 * n1 + kSeg - 1) / kSeg + 2 * kSegGran + kSegGran - 1) / kSegGran * kSegGran) * kSeg; };
    while (spp_batch > 1 && phys_of(npix * spp_batch) * (uint64_t)max_depth >= (1ull << 32)) spp_batch = (spp_batch + 1) / 2;  // node ids are 32-bit
    const uint64_t n1 = npix * spp_batch;
    const uint64_t Pphys = phys_of(n1);
    if (Pphys * (uint64_t)max_depth >= (1ull << 32)) return fail(ctx, TRHIP_ERR_UNSUPPORTED, "Whitted ray tree does not fit 32-bit node ids at this resolution / depth");
    const uint32_t cap = (uint32_t)(Pphys / kSeg);
    const uint32_t cap_shadow = cap * n_lights;
    const uint64_t Sphys = (uint64_t)cap_shadow * kSeg;
    const uint64_t pool_n = Pphys * (uint64_t)max_depth;
    for (int k = 0; k < 2; ++k)
        for (int j = 0; j < 3; ++j)
            if (int rc = ensure(ctx, ctx->q[k][j], Pphys * sizeof(float4))) return rc;
    for (int j = 0; j < 3; ++j)
        if (int rc = ensure(ctx, ctx->sq[j], Sphys * sizeof(float4))) return rc;
    if (int rc = ensure(ctx, ctx->hits, Pphys * sizeof(float4))) return rc;
    if (int rc = ensure(ctx, ctx->occl, Sphys)) return rc;
    if (int rc = ensure(ctx, ctx->wh_L, pool_n * sizeof(float4))) return rc;
    if (int rc = ensure(ctx, ctx->wh_parent, pool_n * sizeof(uint32_t))) return rc;
    if (int rc = ensure(ctx, ctx->wh_coef, pool_n * sizeof(float4))) return rc;
    if (int rc = ensure(ctx, ctx->wh_pdf, pool_n * sizeof(float2))) return rc;
    if (int rc = ensure(ctx, ctx->wh_flags, sizeof(WhittedFlags))) return rc;
    if (int rc = ensure_overflow(ctx)) return rc;
    hipStream_t st = ctx->stream;
    Counters* ctr = (Counters*)ctx->counters.p;
    const DeviceSensor* dsp = (const DeviceSensor*)ctx->sensor.p;
    PathQueue pq[2];
    for (int k = 0; k < 2; ++k) pq[k] = PathQueue{(float4*)ctx->q[k][0].p, (float4*)ctx->q[k][1].p, (float4*)ctx->q[k][2].p};
    ShadowQueue sq{(float4*)ctx->sq[0].p, (float4*)ctx->sq[1].p, (float4*)ctx->sq[2].p};
    WhittedPool pool{(float4*)ctx->wh_L.p, (uint32_t*)ctx->wh_parent.p, (float4*)ctx->wh_coef.p, (float2*)ctx->wh_pdf.p};
    WhittedFlags* flags = (WhittedFlags*)ctx->wh_flags.p;
    float4* L = (float4*)ctx->Lbuf.p;
    float4* hits = (float4*)ctx->hits.p;
    Timer tm(ctx, ctx->timing && stats);
    hipEvent_t e0, e1;
    HIP_TRY(ctx, hipEventCreate(&e0));
    HIP_TRY(ctx, hipEventCreate(&e1));
    HIP_TRY(ctx, hipEventRecord(e0, st));
    HIP_TRY(ctx, hipMemsetAsync(ctr, 0, sizeof(Counters), st));
    HIP_TRY(ctx, hipMemsetAsync(flags, 0, sizeof(WhittedFlags), st));
    HIP_TRY(ctx, hipMemsetAsync(L, 0, total_slots * sizeof(float4), st));
    const int g_shade = ctx->num_cu * 8;
    const dim3 gsmall(ctx->num_cu * 8), blk(kBlock);
    uint32_t n_batches = 0;
    for (uint64_t s0 = 0; s0 < spp; s0 += spp_batch) {
        const uint64_t nb = std::min<uint64_t>(spp_batch, spp - s0) * npix;
        n_batches++;
        HIP_TRY(ctx, hipMemsetAsync(ctr, 0, offsetof(Counters, closest_total), st));
        HIP_TRY(ctx, hipMemsetAsync(pool.L, 0, pool_n * sizeof(float4), st));
        tm.begin(0, st);
        hipLaunchKernelGGL(k_raygen, dim3(grid_for(ctx, nb, 8)), blk, 0, st, dsp, (uint32_t)(s0 * npix), (uint32_t)nb, seed, sample_offset, pq[0], cap, ctr);
        tm.end(0, st);
        int cur = 0;
        for (int depth = 1; depth <= max_depth; ++depth) {
            const uint32_t base_in = (uint32_t)((uint64_t)(depth - 1) * Pphys), base_out = (uint32_t)((uint64_t)depth * Pphys);
            tm.begin(1, st);
            launch_trace(ctx, st, scene, false, SegQueue{ctr->n_queue[depth - 1], cap, 0u}, pq[cur].o, pq[cur].d, nullptr,
                         TraceOut{hits, nullptr, nullptr, nullptr, 0u, depth == 1 && far_camera(scene, sensor) ? 1u : 0u}, ctr->work_closest[depth - 1], ctr);
            tm.end(1, st);
            tm.begin(2, st);
            hipLaunchKernelGGL(k_shade_whitted, dim3(g_shade), blk, 0, st, scene->dev, pq[cur], pq[cur ^ 1], sq, cap, cap_shadow, hits, pool, base_in, base_out, ctr, flags, depth, max_depth);
            tm.end(2, st);
            tm.begin(3, st);
            launch_trace(ctx, st, scene, true, SegQueue{ctr->n_shadow[depth - 1], cap_shadow, 0u}, sq.o, sq.d, nullptr, TraceOut{nullptr, nullptr, nullptr, (uint8_t*)ctx->occl.p},
                         ctr->work_shadow[depth - 1], ctr);
            tm.end(3, st);
            tm.begin(2, st);
            for (uint32_t l = 0; l < scene->dev.n_lights; ++l)
                hipLaunchKernelGGL(k_whitted_direct, gsmall, blk, 0, st, SegQueue{ctr->n_shadow[depth - 1], cap_shadow, 0u}, sq, (const uint8_t*)ctx->occl.p, l, pool.L);
            tm.end(2, st);
            cur ^= 1;
        }
        tm.begin(2, st);
        for (int depth = max_depth; depth >= 2; --depth)
            for (uint32_t branch = 0; branch < 2; ++branch)
                hipLaunchKernelGGL(k_whitted_resolve, gsmall, blk, 0, st, SegQueue{ctr->n_queue[depth - 1], cap, 0u}, pool, (uint32_t)((uint64_t)(depth - 1) * Pphys), branch);
        hipLaunchKernelGGL(k_whitted_finish, gsmall, blk, 0, st, SegQueue{ctr->n_queue[0], cap, 0u}, (const uint32_t*)pool.parent, (const float4*)pool.L, L);
        tm.end(2, st);
    }
    tm.begin(4, st);
    launch_film(ctx, st, ds, dsp, L, total_slots, spp, seed, sample_offset, (float4*)d_film, false);
    tm.end(4, st);
    HIP_TRY(ctx, hipEventRecord(e1, st));
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(st));
    WhittedFlags hf;
    HIP_TRY(ctx, hipMemcpy(&hf, flags, sizeof hf, hipMemcpyDeviceToHost));
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    *ms_total = ms;
    if (stats) {
        stats->ms_raygen = tm.total(0, &stats->launches_raygen);
        stats->ms_trace_closest = tm.total(1, &stats->launches_trace_closest);
        stats->ms_fallback = tm.fallback_total(&stats->launches_fallback);
        stats->ms_shade = tm.total(2, &stats->launches_shade);
        stats->ms_trace_any = tm.total(3, &stats->launches_trace_any);
        stats->ms_film = tm.total(4, &stats->launches_film);
        stats->n_batches = n_batches;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (hf.overflow) return fail(ctx, TRHIP_ERR_UNSUPPORTED, "Whitted ray tree outgrew its queues (more than 2 rays per camera ray at some depth): lower \"batch_paths\"");
    return 0;
}
