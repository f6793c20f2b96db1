// tu_path.hip — PathIntegrator frames (classic per-depth wavefront, streaming wavefront, bands), the film pass, and the kernel-level entry
// points that use the shading / film kernels.
#include "th_host.h"

// Film / sample-grid geometry derived from the sensor (film.jl:68-73, integrators/sampler.jl:13-20)
void derive_sensor(const trhip_sensor* sn, DeviceSensor& d) {
    std::memcpy(d.raster_to_camera, sn->raster_to_camera, sizeof d.raster_to_camera);
    std::memcpy(d.camera_to_world, sn->camera_to_world, sizeof d.camera_to_world);
    d.lens_radius = sn->lens_radius;
    d.focal_distance = sn->focal_distance;
    d.shutter_open = sn->shutter_open;
    d.shutter_close = sn->shutter_close;
    for (int i = 0; i < 2; ++i) {
        d.crop_min[i] = sn->crop_min[i];
        d.crop_max[i] = sn->crop_max[i];
        d.filter_radius[i] = sn->filter_radius[i];
        d.sb_min[i] = (int)std::floor(sn->crop_min[i] + 0.5f - sn->filter_radius[i]);
        d.sb_max[i] = (int)std::ceil(sn->crop_max[i] - 0.5f + sn->filter_radius[i]);
    }
    d.scale = sn->scale;
    d.sb_w = d.sb_max[0] - d.sb_min[0] + 1;
    d.sb_h = d.sb_max[1] - d.sb_min[1] + 1;
    d.film_w = (int)std::fabs(sn->crop_max[0] - (sn->crop_min[0] - 1.0f));  // inclusive_sides bounds.jl:100-102
    d.film_h = (int)std::fabs(sn->crop_max[1] - (sn->crop_min[1] - 1.0f));
    d.tiles_x = (int)std::floor(((float)(d.sb_max[0] - d.sb_min[0]) + 16.0f) / 16.0f);
    d.tiles_y = (int)std::floor(((float)(d.sb_max[1] - d.sb_min[1]) + 16.0f) / 16.0f);
    d.band_y0 = d.sb_min[1];  // one band: the whole frame
    d.band_rows = d.sb_h;
    d.band_ty0 = 0;
    d.band_ty1 = d.tiles_y - 1;
    d.accumulate = 0;
}

// Film accumulation: positions, then the LDS-tiled gather (falls back to the per-pixel gather when a 16x16 film tile is reached
// by more than two sample tiles per axis, i.e. very wide filters).
#ifndef TH_FILM_BX
#define TH_FILM_BX 1  // film_block = 2; measured at 1024^2, 256 spp, 4 samples in flight per thread: 1x1 67 ms, 2x2 40, 1x4 32, 1x6 47, 1x8 42, 2x4 39
#define TH_FILM_BY 4
#endif
// film_block = 3: the gather reads one 16-byte splat descriptor per sample (k_film_descriptors) instead of recomputing the sample's pixel range
// and table indices in every thread it reaches; needs a filter radius <= 3 (<= 8 columns / rows per sample)
bool film_uses_desc(const trhip_ctx* ctx, const DeviceSensor& ds) {
    return ctx->film_block == 3 && !ctx->film_tiled && !ctx->film_transpose && std::fmax(ds.filter_radius[0], ds.filter_radius[1]) <= 3.0f && ds.film_w < 32000 && ds.film_h < 32000;
}
// film_block >= 4: the gather reads a 32-bit descriptor from the .w lane of the radiance records (k_film_pack_w / k_film_gather_packed); needs a filter radius <= 1
// (<= 4 columns / rows per sample) — every scene of the reference uses LanczosSincFilter(Point2f(1f0), 3f0)
bool film_uses_packed(const trhip_ctx* ctx, const DeviceSensor& ds) {
    return ctx->film_block >= 4 && !ctx->film_tiled && !ctx->film_transpose && std::fmax(ds.filter_radius[0], ds.filter_radius[1]) <= 1.0f && ds.filter_radius[0] > 0.0f && ds.filter_radius[1] > 0.0f;
}
// the per-sample buffer the film pass needs next to the radiance: nothing (packed), descriptors (16 B) or film positions (8 B)
int ensure_film_samples(trhip_ctx* ctx, const DeviceSensor& ds, uint64_t total_slots) {
    if (film_uses_packed(ctx, ds)) return 0;
    if (film_uses_desc(ctx, ds)) return ensure(ctx, ctx->fdesc, total_slots * sizeof(uint4));
    return ensure(ctx, ctx->pfilm, total_slots * sizeof(float2));
}
// can k_raygen write the radiance records in the film pass's layout, descriptors included (th_kernels.h, k_raygen's Lf)?  Whole sample passes of one band, 32-bit indices.
bool film_fused(const trhip_ctx* ctx, const DeviceSensor& ds, uint32_t spp) {
    const uint64_t npix_b = (uint64_t)ds.sb_w * ds.band_rows;
    return ctx->film_fused && ctx->film_relayout && film_uses_packed(ctx, ds) && ((npix_b + 63u) / 64u) * 64u * spp < (1ull << 32);
}
FilmSideTable film_side_table(trhip_ctx* ctx, hipStream_t st, uint64_t total_slots, bool* ok) {
    // side table for the descriptors that do not fit 30 bits (one sample in ~4000 at 1024^2): entry 0 of film_side is the counter, the descriptors follow
    const uint32_t side_cap = (uint32_t)std::min<uint64_t>(total_slots / 16 + 65536, 0x7ffffff0ull);
    *ok = ensure(ctx, ctx->film_side, ((size_t)side_cap + 1) * sizeof(uint4)) == 0;
    if (!*ok) return FilmSideTable{nullptr, nullptr, 0};
    (void)hipMemsetAsync(ctx->film_side.p, 0, sizeof(uint4), st);
    return FilmSideTable{(uint4*)ctx->film_side.p + 1, (uint32_t*)ctx->film_side.p, side_cap};
}
void launch_film(trhip_ctx* ctx, hipStream_t st, const DeviceSensor& ds, const DeviceSensor* dsp, const float4* L, uint64_t total_slots, uint32_t spp, uint64_t seed, uint32_t sample_offset,
                 float4* d_film, bool fused) {
    if (film_uses_packed(ctx, ds)) {
        // the descriptors go into the records' .w lanes — on the way into a pixel-group-major copy when there is room for one (option "film_relayout", default on)
        const uint32_t npix_b = (uint32_t)(ds.sb_w * ds.band_rows);
        const uint64_t padded = (uint64_t)((npix_b + 63u) / 64u) * 64u * spp;
        uint32_t layout = 0;
        FilmSideTable side{ctx->film_side.p ? (uint4*)ctx->film_side.p + 1 : nullptr, (uint32_t*)ctx->film_side.p, 0};
        if (fused) {
            layout = 1;  // k_raygen wrote the records re-laid and with their descriptors (its side table is ctx->film_side, reset before the frame's first launch)
        } else {
            bool ok = false;
            side = film_side_table(ctx, st, total_slots, &ok);
            if (!ok) return;
        }
        if (fused) {
        } else if (ctx->film_relayout && total_slots == (uint64_t)npix_b * spp && ensure(ctx, ctx->film_Lt, padded * sizeof(float4)) == 0) {
            layout = 1;
            hipLaunchKernelGGL(k_film_pack_transpose, dim3(grid_for(ctx, padded, 8)), dim3(kBlock), 0, st, dsp, npix_b, spp, seed, sample_offset, L, (float4*)ctx->film_Lt.p, side);
            L = (const float4*)ctx->film_Lt.p;
        } else {
            hipLaunchKernelGGL(k_film_pack_w, dim3(grid_for(ctx, total_slots, 8)), dim3(kBlock), 0, st, dsp, total_slots, seed, sample_offset, const_cast<float4*>(L), side);
        }
        const float* tb = (const float*)ctx->table.p;
        auto threads = [&](int bx, int by) { return (uint64_t)((ds.film_w + bx - 1) / bx) * (uint64_t)((ds.film_h + by - 1) / by); };
#define TH_FILM_PACKED(BXV, BYV) hipLaunchKernelGGL((k_film_gather_packed<BXV, BYV>), dim3(grid_for(ctx, threads(BXV, BYV), 8)), dim3(kBlock), 0, st, dsp, tb, L, spp, seed, sample_offset, layout, (const uint4*)side.desc, d_film, ctx->film_swizzle ? 1u : 0u)
        switch (ctx->film_block) {
        case 4: TH_FILM_PACKED(1, 4); break;
        case 7: TH_FILM_PACKED(2, 2); break;
        case 8: TH_FILM_PACKED(4, 2); break;
        case 9: TH_FILM_PACKED(8, 4); break;
        case 10: TH_FILM_PACKED(1, 8); break;
        case 11: TH_FILM_PACKED(1, 16); break;
        case 12: TH_FILM_PACKED(1, 2); break;
        case 13: TH_FILM_PACKED(1, 1); break;
        case 6: TH_FILM_PACKED(4, 4); break;
        default: TH_FILM_PACKED(2, 4); break;  // 5
        }
#undef TH_FILM_PACKED
        return;
    }
    if (film_uses_desc(ctx, ds) && ctx->fdesc.bytes >= total_slots * sizeof(uint4)) {
        hipLaunchKernelGGL(k_film_descriptors, dim3(grid_for(ctx, total_slots, 8)), dim3(kBlock), 0, st, dsp, total_slots, seed, sample_offset, (uint4*)ctx->fdesc.p);
        const uint64_t nthreads = (uint64_t)ds.film_w * ((ds.film_h + 3) / 4);
        hipLaunchKernelGGL((k_film_gather_desc<4>), dim3(grid_for(ctx, nthreads, 8)), dim3(kBlock), 0, st, dsp, (const float*)ctx->table.p, L, (const uint4*)ctx->fdesc.p, spp, d_film);
        return;
    }
    // pixel-group-major inputs for the gather (th_kernels.h, film_index): p_film is written that way, L is re-laid into a second buffer
    // (the frame's radiance, 16 B per sample, once more); without room for it the gather reads the sample-major arrays as before
    const uint32_t npix = (uint32_t)(ds.sb_w * ds.band_rows);
    const uint64_t padded = (uint64_t)((npix + 63u) / 64u) * 64u * spp;
    uint32_t layout = 0;
    const bool whole = ds.band_rows == ds.sb_h;
    if (whole && ctx->film_transpose && spp > 1 && total_slots == (uint64_t)npix * spp && ensure(ctx, ctx->pfilm, padded * sizeof(float2)) == 0 && ensure(ctx, ctx->film_Lt, padded * sizeof(float4)) == 0) {
        layout = 1;
        hipLaunchKernelGGL(k_film_transpose, dim3(grid_for(ctx, padded, 8)), dim3(kBlock), 0, st, L, npix, spp, (float4*)ctx->film_Lt.p);
        L = (const float4*)ctx->film_Lt.p;
    }
    hipLaunchKernelGGL(k_film_positions, dim3(grid_for(ctx, total_slots, 8)), dim3(kBlock), 0, st, dsp, total_slots, seed, sample_offset, (float2*)ctx->pfilm.p, layout, spp);
    const float rmax = std::fmax(ds.filter_radius[0], ds.filter_radius[1]);
    if (whole && ctx->film_tiled && rmax <= 6.0f) {  // reach of a pixel = 2r + 3 sample pixels <= 16: at most 2 x 2 sample tiles
        const uint32_t budget = 24 * 1024 / 20;  // staged {p_film, L} elements in 24 KiB of LDS: ~6 blocks per CU
        const uint32_t nc_max = 16 + 2 * (uint32_t)std::ceil(rmax) + 4;
        uint32_t cols, ns;
        if (spp <= budget) {
            ns = spp;
            cols = std::max(1u, std::min(nc_max, budget / spp));
        } else {
            cols = 1;
            ns = budget;
        }
        const dim3 grid((ds.film_w + 15) / 16, (ds.film_h + 15) / 16);
        hipLaunchKernelGGL(k_film_gather_tiled, grid, dim3(kBlock), (size_t)cols * ns * 20 + 16, st, dsp, (const float*)ctx->table.p, L, (const float2*)ctx->pfilm.p, spp, layout, cols, ns, d_film);
    } else {
        const uint64_t npx = (uint64_t)ds.film_w * ds.film_h;
        const int fblock = ctx->film_block >= 4 ? 2 : ctx->film_block;  // a filter too wide for the packed descriptor: the 1 x 4 block gather
        if (fblock == 1)
            hipLaunchKernelGGL((k_film_gather_block<2, 2>), dim3(grid_for(ctx, (npx + 3) / 4, 8)), dim3(kBlock), 0, st, dsp, (const float*)ctx->table.p, L, (const float2*)ctx->pfilm.p, spp, layout, d_film);
        else if (fblock == 2)
            hipLaunchKernelGGL((k_film_gather_block<TH_FILM_BX, TH_FILM_BY>), dim3(grid_for(ctx, (npx + TH_FILM_BX * TH_FILM_BY - 1) / (TH_FILM_BX * TH_FILM_BY), 8)), dim3(kBlock), 0, st, dsp, (const float*)ctx->table.p, L, (const float2*)ctx->pfilm.p, spp, layout, d_film);
        else
            hipLaunchKernelGGL(k_film_gather, dim3(grid_for(ctx, npx, 8)), dim3(kBlock), 0, st, dsp, (const float*)ctx->table.p, L, (const float2*)ctx->pfilm.p, spp, layout, d_film);
    }
}

namespace {
// PathIntegrator as a STREAMING wavefront (th_trace2.h "streaming wavefront"): rounds instead of depths.  A round
// traces every queued ray with a fetch budget, resumes the rays suspended in the round before, shades what finished (entries
// carry their own depth), and traces the shadow rays the same way.  max_depth + 16 budgeted rounds, then max_depth rounds
// without a budget, which complete whatever is left.  Radiance terms go to per-depth slots and are folded in depth order, so
// the per-sample radiance (and the film) is bit-identical to the classic per-depth wavefront.
int render_stream_impl(trhip_ctx* ctx, const trhip_scene* scene, const trhip_sensor* sensor, const DeviceSensor& ds, uint32_t spp, int max_depth, uint64_t seed, uint32_t sample_offset, void* out,
                       bool out_is_device, trhip_stats* stats, bool* declined) {
    *declined = true;
    const uint64_t npix = (uint64_t)ds.sb_w * ds.sb_h;
    const uint64_t total_slots = npix * spp;
    const int R_b = max_depth + 16, R = R_b + max_depth;
    if (R + 1 > kMaxDepth + 1) return 0;
    if ((uint64_t)max_depth * total_slots >= (1ull << 32)) return 0;
    if (total_slots >= (1ull << 31)) return 0;  // one batch: queue indices are 32-bit
    const uint64_t P = total_slots;
    const uint32_t list_cap = ctx->stream_list_cap ? ctx->stream_list_cap : (uint32_t)std::max<uint64_t>(65536, P / 128);
    const uint32_t cap = (uint32_t)(((P + kSeg - 1) / kSeg + 2 * kSegGran + list_cap / kSeg + 64 + kSegGran - 1) / kSegGran * kSegGran);
    const uint64_t Pphys = (uint64_t)cap * kSeg;
    const size_t list_bytes = (size_t)list_cap * (4 * 16 + 16 + 4 + (size_t)kStack2Total * 8);
    const size_t terms_bytes = (size_t)max_depth * total_slots * sizeof(float4);
    const size_t need = terms_bytes + Pphys * (10 * 16 + 2 * 4) + 4 * list_bytes + total_slots * 24 + (3ull << 30);
    {
        size_t free_b = 0, total_b = 0;
        HIP_TRY(ctx, hipMemGetInfo(&free_b, &total_b));
        size_t held = ctx->Lbuf.bytes + ctx->pfilm.bytes + ctx->st_terms.bytes + ctx->st_tags[0].bytes + ctx->st_tags[1].bytes;
        Pipe& p0 = ctx->pipes[0];
        held += p0.hits.bytes;
        for (auto& a : p0.q)
            for (auto& b : a) held += b.bytes;
        for (auto& b : p0.sq) held += b.bytes;
        for (auto& a : ctx->st_list)
            for (auto& b : a)
                for (auto& c : b) held += c.bytes;
        if ((double)need > 0.9 * (double)(free_b + held)) return 0;  // does not fit as one batch: the classic path cuts the frame into batches
    }
    *declined = false;
    Pipe& pp = ctx->pipes[0];
    if (!pp.st) {
        HIP_TRY(ctx, hipStreamCreate(&pp.st));
        HIP_TRY(ctx, hipStreamCreate(&pp.st2));
        HIP_TRY(ctx, hipEventCreateWithFlags(&pp.ev_shade, hipEventDisableTiming));
        HIP_TRY(ctx, hipEventCreateWithFlags(&pp.ev_any, hipEventDisableTiming));
        HIP_TRY(ctx, hipEventCreateWithFlags(&pp.ev_done, hipEventDisableTiming));
    }
    if (int rc = upload(ctx, ctx->sensor, &ds, sizeof ds)) return rc;
    if (int rc = upload(ctx, ctx->table, sensor->filter_table, 256 * sizeof(float))) return rc;
    for (int k = 0; k < 2; ++k)
        for (int j = 0; j < 3; ++j)
            if (int rc = ensure(ctx, pp.q[k][j], Pphys * sizeof(float4))) return rc;
    for (int j = 0; j < 3; ++j)
        if (int rc = ensure(ctx, pp.sq[j], Pphys * sizeof(float4))) return rc;
    if (int rc = ensure(ctx, pp.hits, Pphys * sizeof(float4))) return rc;
    if (int rc = ensure(ctx, pp.counters, sizeof(Counters))) return rc;
    const size_t slab_bytes = (size_t)trace_grid(ctx) * kBlock * (size_t)kStackSlabLevels * sizeof(uint2);
    for (int k = 0; k < 2; ++k)
        if (int rc = ensure(ctx, pp.overflow[k], slab_bytes)) return rc;
    if (int rc = ensure(ctx, ctx->Lbuf, total_slots * sizeof(float4))) return rc;
    if (int rc = ensure_film_samples(ctx, ds, total_slots)) return rc;
    if (int rc = ensure(ctx, ctx->st_terms, terms_bytes)) return rc;
    for (int k = 0; k < 2; ++k)
        if (int rc = ensure(ctx, ctx->st_tags[k], Pphys * sizeof(uint32_t))) return rc;
    const size_t row_bytes = (size_t)kSeg * kCtrStride * sizeof(uint32_t);
    if (int rc = ensure(ctx, ctx->st_frozen, row_bytes)) return rc;
    if (int rc = ensure(ctx, ctx->st_counts, 16 * sizeof(uint32_t))) return rc;
    const size_t field_bytes[7] = {16, 16, 16, 16, 16, 4, (size_t)kStack2Total * 8};
    for (int kind = 0; kind < 2; ++kind)
        for (int pg = 0; pg < 2; ++pg)
            for (int f = 0; f < 7; ++f)
                if (int rc = ensure(ctx, ctx->st_list[kind][pg][f], (size_t)list_cap * field_bytes[f])) return rc;
    const size_t film_bytes = (size_t)ds.film_w * ds.film_h * sizeof(float4);
    void* d_film = out;
    if (!out_is_device) {
        if (int rc = ensure(ctx, ctx->film, film_bytes)) return rc;
        d_film = ctx->film.p;
    }
    auto list_of = [&](int kind, int pg) {
        DevBuf* b = ctx->st_list[kind][pg];
        return SuspendList{(float4*)b[0].p, (float4*)b[1].p, (float4*)b[2].p, (float4*)b[3].p, (uint4*)b[4].p, (uint32_t*)b[5].p, (uint2*)b[6].p, list_cap};
    };
    uint32_t* lc = (uint32_t*)ctx->st_counts.p;  // [0..1] closest list counts (ping-pong), [2] closest cursor, [4..5] any counts, [6] any cursor
    hipStream_t st = ctx->stream, ps = pp.st, ps2 = ctx->overlap ? pp.st2 : pp.st;
    const DeviceSensor* dsp = (const DeviceSensor*)ctx->sensor.p;
    float4* L = (float4*)ctx->Lbuf.p;
    float4* terms = (float4*)ctx->st_terms.p;
    Counters* ctr = (Counters*)pp.counters.p;
    PathQueue pq[2];
    for (int k = 0; k < 2; ++k) pq[k] = PathQueue{(float4*)pp.q[k][0].p, (float4*)pp.q[k][1].p, (float4*)pp.q[k][2].p};
    uint32_t* tags[2] = {(uint32_t*)ctx->st_tags[0].p, (uint32_t*)ctx->st_tags[1].p};
    ShadowQueue sq{(float4*)pp.sq[0].p, (float4*)pp.sq[1].p, (float4*)pp.sq[2].p};
    float4* hits = (float4*)pp.hits.p;
    uint32_t* frozen = (uint32_t*)ctx->st_frozen.p;

    Timer tm(ctx, ctx->timing && stats);
    hipEvent_t e0, e1, ev_start;
    HIP_TRY(ctx, hipEventCreate(&e0));
    HIP_TRY(ctx, hipEventCreate(&e1));
    HIP_TRY(ctx, hipEventCreateWithFlags(&ev_start, hipEventDisableTiming));
    HIP_TRY(ctx, hipEventRecord(e0, st));
    HIP_TRY(ctx, hipMemsetAsync(terms, 0, terms_bytes, st));
    HIP_TRY(ctx, hipEventRecord(ev_start, st));
    HIP_TRY(ctx, hipStreamWaitEvent(ps, ev_start, 0));
    HIP_TRY(ctx, hipMemsetAsync(ctr, 0, sizeof(Counters), ps));
    HIP_TRY(ctx, hipMemsetAsync(lc, 0, 16 * sizeof(uint32_t), ps));
    tm.begin(0, ps);
    hipLaunchKernelGGL(k_raygen, dim3(grid_for(ctx, P, 8)), dim3(kBlock), 0, ps, dsp, 0u, (uint32_t)P, seed, sample_offset, pq[0], cap, ctr);
    hipLaunchKernelGGL(k_fill_u32, dim3(grid_for(ctx, Pphys, 8)), dim3(kBlock), 0, ps, tags[0], Pphys, 1u);
    tm.end(0, ps);
    const int g_shade = ctx->num_cu * 8;
    int cur = 0;
    for (int r = 0; r < R; ++r) {
        const uint32_t budget_min = r < R_b ? ctx->stream_budget_min : 0u;  // the last max_depth rounds run every ray to its end
        const int in_pg = r & 1, out_pg = (r + 1) & 1;
        // ---- closest hits: fresh rays through frozen counts (finished resumed rays are appended to the live queue) ----
        HIP_TRY(ctx, hipMemcpyAsync(frozen, ctr->n_queue[r], row_bytes, hipMemcpyDeviceToDevice, ps));
        HIP_TRY(ctx, hipMemsetAsync(&lc[out_pg], 0, sizeof(uint32_t), ps));
        HIP_TRY(ctx, hipMemsetAsync(&lc[2], 0, sizeof(uint32_t), ps));
        StreamCtl sc_c{list_of(0, in_pg), list_of(0, out_pg), &lc[in_pg], &lc[2], &lc[out_pg], budget_min, ctx->stream_budget_shift, pq[cur].beta, tags[cur], pq[cur].o, pq[cur].d, pq[cur].beta, hits, tags[cur],
                       ctr->n_queue[r], cap};
        const SegQueue qc{frozen, cap, 0u};
        const TraceOut oc{hits, nullptr, nullptr, nullptr, 1u};
        tm.begin(1, ps);
        launch_trace2_stream(ctx, ps, scene, false, qc, pq[cur].o, pq[cur].d, oc, ctr->work_closest[r], pp.overflow[0].p, ctr, sc_c);
        tm.end(1, ps);
        if (ps2 != ps && r > 0) HIP_TRY(ctx, hipStreamWaitEvent(ps, pp.ev_any, 0));  // shade(r) reuses the shadow queue
        tm.begin(2, ps);
        if (scene->dev.tri_tan)
            hipLaunchKernelGGL(k_shade_path<true>, dim3(g_shade), dim3(kBlock), 0, ps, scene->dev, dsp, pq[cur], pq[cur ^ 1], sq, cap, hits, terms, ctr, r, 0, max_depth, 1u,
                           ShadeStream{tags[cur], tags[cur ^ 1], (uint32_t)total_slots});
        else
            hipLaunchKernelGGL((k_shade_path<true, false>), dim3(g_shade), dim3(kBlock), 0, ps, scene->dev, dsp, pq[cur], pq[cur ^ 1], sq, cap, hits, terms, ctr, r, 0, max_depth, 1u,
                           ShadeStream{tags[cur], tags[cur ^ 1], (uint32_t)total_slots});
        tm.end(2, ps);
        if (ps2 != ps) {
            HIP_TRY(ctx, hipEventRecord(pp.ev_shade, ps));
            HIP_TRY(ctx, hipStreamWaitEvent(ps2, pp.ev_shade, 0));
        }
        // ---- shadow rays: unoccluded ones add their contribution to the term slot ----
        HIP_TRY(ctx, hipMemsetAsync(&lc[4 + out_pg], 0, sizeof(uint32_t), ps2));
        HIP_TRY(ctx, hipMemsetAsync(&lc[6], 0, sizeof(uint32_t), ps2));
        StreamCtl sc_a{list_of(1, in_pg), list_of(1, out_pg), &lc[4 + in_pg], &lc[6], &lc[4 + out_pg], budget_min, ctx->stream_budget_shift, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0u};
        const SegQueue qa{ctr->n_shadow[r], cap, 0u};
        const TraceOut oa{nullptr, terms, sq.c, nullptr, 0u};
        tm.begin(3, ps2);
        launch_trace2_stream(ctx, ps2, scene, true, qa, sq.o, sq.d, oa, ctr->work_shadow[r], pp.overflow[1].p, ctr, sc_a);
        tm.end(3, ps2);
        if (ps2 != ps) HIP_TRY(ctx, hipEventRecord(pp.ev_any, ps2));
        cur ^= 1;
    }
    if (ps2 != ps) HIP_TRY(ctx, hipStreamWaitEvent(ps, pp.ev_any, 0));
    tm.begin(2, ps);
    hipLaunchKernelGGL(k_fold_terms, dim3(grid_for(ctx, total_slots, 8)), dim3(kBlock), 0, ps, (const float4*)terms, total_slots, (uint32_t)max_depth, L);
    tm.end(2, ps);
    HIP_TRY(ctx, hipEventRecord(pp.ev_done, ps));
    HIP_TRY(ctx, hipStreamWaitEvent(st, pp.ev_done, 0));
    tm.begin(4, st);
    // (no k_apply_poison here: the streaming shade kernel writes its terms — NaN included — straight into the per-depth slots and never notes
    //  poison; ctx->poison belongs to the classic path's two-stream mode alone)
    launch_film(ctx, st, ds, dsp, L, total_slots, spp, seed, sample_offset, (float4*)d_film, false);
    tm.end(4, st);
    HIP_TRY(ctx, hipEventRecord(e1, st));
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(st));
    ctx->last_L_count = total_slots;
    ctx->last_L_layout = 0;
    if (!out_is_device) HIP_TRY(ctx, hipMemcpy(out, d_film, film_bytes, hipMemcpyDeviceToHost));
    uint32_t left[8];
    HIP_TRY(ctx, hipMemcpy(left, lc, sizeof left, hipMemcpyDeviceToHost));
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        stats->camera_samples = total_slots;
        Counters h;
        HIP_TRY(ctx, hipMemcpy(&h, ctr, sizeof h, hipMemcpyDeviceToHost));
        stats->closest_rays = h.closest_total;
        stats->shadow_rays = h.shadow_total;
        stats->nodes_visited = h.nodes_closest;
        stats->prims_tested = h.prims_closest;
        stats->nodes_visited_shadow = h.nodes_shadow;
        stats->prims_tested_shadow = h.prims_shadow;
        stats->fallback_rays = h.fallback_total;
        stats->nodes_visited_fallback = h.nodes_fallback;
        stats->prims_tested_fallback = h.prims_fallback;
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        stats->ms_total = ms;
        stats->ms_raygen = tm.total(0, &stats->launches_raygen);
        stats->ms_trace_closest = tm.total(1, &stats->launches_trace_closest);
        stats->ms_fallback = tm.fallback_total(&stats->launches_fallback);
        stats->ms_shade = tm.total(2, &stats->launches_shade);
        stats->ms_trace_any = tm.total(3, &stats->launches_trace_any);
        stats->ms_film = tm.total(4, &stats->launches_film);
        stats->n_batches = 1;
        stats->max_depth_reached = (uint32_t)max_depth;
        traversal_info(ctx, scene, &stats->traversal, &stats->node_bytes);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipEventDestroy(ev_start);
    if (left[R & 1] || left[4 + (R & 1)]) return fail(ctx, TRHIP_ERR_HIP, "streaming wavefront: %u + %u rays still suspended after the drain rounds", left[R & 1], left[4 + (R & 1)]);
    return 0;
}

// One band of a frame (band == nullptr: the whole frame, the normal case).  A band is a range of whole tile rows; its DeviceSensor carries
// the range and whether the film gather starts from zero or adds onto the bands before (th_scene.h).
int render_impl_band(trhip_ctx* ctx, const trhip_scene* scene, const trhip_sensor* sensor, int integrator, uint32_t spp, int max_depth, uint64_t seed, uint32_t sample_offset, void* out,
                     bool out_is_device, trhip_stats* stats, const DeviceSensor* band) {
    HostClock hclk;
    if (!ctx || !scene || !sensor || !out) return fail(ctx, TRHIP_ERR_INVALID, "null argument");
    if (!scene->committed) return fail(ctx, TRHIP_ERR_INVALID, "scene not committed");
    if (integrator != 0 && integrator != 1) return fail(ctx, TRHIP_ERR_INVALID, "unknown integrator %d", integrator);
    // A GeometricPrimitive without a material makes the reference re-spawn the ray behind the hit without counting a bounce
    // (sppm.jl:219-222; Whitted calls a method that does not exist, sampler.jl:77-80).  The wavefront does not model that.
    if (scene->has_materialless_prim)  // found once, at commit: walking a million host records here cost 2.8 ms of every frame
            return fail(ctx, TRHIP_ERR_UNSUPPORTED, "rendering a scene with a material-less primitive is not supported (the trace entry points accept it)");
    if (spp == 0 || max_depth < 1 || max_depth > kMaxDepth) return fail(ctx, TRHIP_ERR_INVALID, "spp must be >= 1 and max_depth in 1..%d", kMaxDepth);
    if (!ctx->warned_idle_accelerator && ctx->hybrid) {  // a two-tree scene rendered with options that leave its accelerator idle: said once per context (trhip_accelerator_note has it too)
        const char* why = hybrid_idle_reason(ctx, scene);
        if (why[0]) {
            ctx->warned_idle_accelerator = true;
            std::fprintf(stderr, "[tracehip] this scene holds the reference's tree and an accelerator, but the accelerator is idle: %s\n", why);
        }
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DeviceSensor ds;
    if (band)
        ds = *band;
    else
        derive_sensor(sensor, ds);
    if (ds.film_w <= 0 || ds.film_h <= 0 || ds.sb_w <= 0 || ds.sb_h <= 0) return fail(ctx, TRHIP_ERR_INVALID, "empty film");
    const uint64_t npix = (uint64_t)ds.sb_w * ds.band_rows;
    const uint64_t total_slots = npix * spp;
    if (total_slots >= (1ull << 32)) return fail(ctx, TRHIP_ERR_UNSUPPORTED, "more than 2^32 camera samples in one band of a frame");
    if (integrator == 0) {
        if (int rc = upload(ctx, ctx->sensor, &ds, sizeof ds)) return rc;
        if (int rc = upload(ctx, ctx->table, sensor->filter_table, 256 * sizeof(float))) return rc;
        if (int rc = ensure(ctx, ctx->Lbuf, total_slots * sizeof(float4))) return rc;
        if (int rc = ensure_film_samples(ctx, ds, total_slots)) return rc;
        if (int rc = ensure(ctx, ctx->counters, sizeof(Counters))) return rc;
        const size_t fb = (size_t)ds.film_w * ds.film_h * sizeof(float4);
        void* df = out;
        if (!out_is_device) {
            if (int rc = ensure(ctx, ctx->film, fb)) return rc;
            df = ctx->film.p;
        }
        if (stats) std::memset(stats, 0, sizeof *stats);
        double ms = 0;
        if (int rc = render_whitted_impl(ctx, scene, ds, sensor, spp, max_depth, seed, sample_offset, df, stats, &ms)) return rc;
        ctx->last_L_count = total_slots;
        ctx->last_L_layout = 0;
        if (!out_is_device) HIP_TRY(ctx, hipMemcpy(out, df, fb, hipMemcpyDeviceToHost));
        if (stats) {
            Counters h;
            HIP_TRY(ctx, hipMemcpy(&h, ctx->counters.p, sizeof h, hipMemcpyDeviceToHost));
            stats->camera_samples = total_slots;
            stats->closest_rays = h.closest_total;
            stats->shadow_rays = h.shadow_total;
            stats->nodes_visited = h.nodes_closest;
            stats->prims_tested = h.prims_closest;
            stats->nodes_visited_shadow = h.nodes_shadow;
            stats->prims_tested_shadow = h.prims_shadow;
            stats->fallback_rays = h.fallback_total;
            stats->ms_total = ms;
            stats->max_depth_reached = (uint32_t)max_depth;
        traversal_info(ctx, scene, &stats->traversal, &stats->node_bytes);
        }
        return 0;
    }
    if (!band && (ctx->streaming == 1 || (ctx->streaming < 0 && total_slots <= 96ull * scene->prims.size())) && ctx->traversal >= 2 && scene->wide_ok && scene->wide.root_cnt == 0 && scene->wide.root_ref != kRefNone && ctx->batch_paths == 0 && ctx->pipelines <= 1) {
        bool declined = false;
        const int rc = render_stream_impl(ctx, scene, sensor, ds, spp, max_depth, seed, sample_offset, out, out_is_device, stats, &declined);
        if (!declined) return rc;
    }
    // wavefront batch = whole sample passes; per path in flight: 2 x 3 queue float4 + 3 shadow float4 + 1 hit float4 = 160 B
    uint64_t batch_paths = ctx->batch_paths;
    if (batch_paths == 0) {
        size_t free_b = 0, total_b = 0;
        HIP_TRY(ctx, hipMemGetInfo(&free_b, &total_b));
        size_t held = ctx->Lbuf.bytes + ctx->pfilm.bytes;  // reused below, so it counts as available
        for (auto& pp : ctx->pipes) {
            held += pp.hits.bytes;
            for (auto& a : pp.q)
                for (auto& b : a) held += b.bytes;
            for (auto& b : pp.sq) held += b.bytes;
        }
        const double avail = 0.85 * (double)(free_b + held) - (double)total_slots * (sizeof(float4) + sizeof(float2)) - 2.5e9;
        batch_paths = avail > 0 ? (uint64_t)(avail / 212.0) : npix;  // per path in flight: 2 x 3 queue float4 + 2 x 3 shadow float4 + 1 hit float4 + counters
    }
    uint64_t spp_batch = std::max<uint64_t>(1, batch_paths / npix);
    spp_batch = std::min<uint64_t>(spp_batch, spp);
    // Several batches run concurrently (one per pipeline): the memory budget is shared and the frame is cut into at
    // least `pipelines` batches when it has that many sample passes.
    const int want_pipes = std::max(1, std::min(ctx->pipelines, kMaxPipes));
    spp_batch = std::max<uint64_t>(1, std::min<uint64_t>(spp_batch / want_pipes, (spp + want_pipes - 1) / want_pipes));
    while (npix * spp_batch >= (1ull << 31)) spp_batch = (spp_batch + 1) / 2;  // queue indices are 32-bit
    const uint64_t n_batches_total = (spp + spp_batch - 1) / spp_batch;
    const int NP = (int)std::min<uint64_t>(want_pipes, n_batches_total);
    const uint64_t P = npix * spp_batch;
    // physical queue layout: kSeg segments of `cap` entries (th_kernels.h "SegQueue"); a segment receives at most
    // P/kSeg + O(kSegGran) entries per bounce by construction
    const uint32_t cap = (uint32_t)(((P + kSeg - 1) / kSeg + 2 * kSegGran + kSegGran - 1) / kSegGran * kSegGran);
    const uint64_t Pphys = (uint64_t)cap * kSeg;
    if (int rc = upload(ctx, ctx->sensor, &ds, sizeof ds)) return rc;
    if (int rc = upload(ctx, ctx->table, sensor->filter_table, 256 * sizeof(float))) return rc;
    const size_t slab_bytes = (size_t)trace_grid(ctx) * kBlock * (size_t)kStackSlabLevels * sizeof(uint2);
    for (int pi = 0; pi < NP; ++pi) {
        Pipe& pp = ctx->pipes[pi];
        if (!pp.st) {
            // the shadow-ray stream gets its own priority level: streams of one level can share a hardware queue (then any(d) and
            // closest(d+1) run one after the other: measured in the first context of a process), streams of different levels cannot
            int prio_lo = 0, prio_hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
            HIP_TRY(ctx, hipStreamCreate(&pp.st));
            if (ctx->stream2_priority && prio_hi != prio_lo)
                HIP_TRY(ctx, hipStreamCreateWithPriority(&pp.st2, hipStreamDefault, ctx->stream2_priority > 0 ? prio_hi : prio_lo));
            else
                HIP_TRY(ctx, hipStreamCreate(&pp.st2));
            HIP_TRY(ctx, hipEventCreateWithFlags(&pp.ev_shade, hipEventDisableTiming));
            HIP_TRY(ctx, hipEventCreateWithFlags(&pp.ev_any, hipEventDisableTiming));
            HIP_TRY(ctx, hipEventCreateWithFlags(&pp.ev_done, hipEventDisableTiming));
        }
        for (int k = 0; k < 2; ++k)
            for (int j = 0; j < 3; ++j)
                if (int rc = ensure(ctx, pp.q[k][j], Pphys * sizeof(float4))) return rc;
        for (int j = 0; j < 3; ++j)
            if (int rc = ensure(ctx, pp.sq[j], Pphys * sizeof(float4))) return rc;
        if (ctx->overlap)
            for (int j = 0; j < 3; ++j)
                if (int rc = ensure(ctx, pp.sq2[j], Pphys * sizeof(float4))) return rc;
        if (!pp.ev_any2) HIP_TRY(ctx, hipEventCreateWithFlags(&pp.ev_any2, hipEventDisableTiming));
        if (int rc = ensure(ctx, pp.hits, Pphys * sizeof(float4))) return rc;
        if (int rc = ensure(ctx, pp.counters, sizeof(Counters))) return rc;
        for (int k = 0; k < 2; ++k)
            if (int rc = ensure(ctx, pp.overflow[k], slab_bytes)) return rc;
    }
    // the radiance records: sample-major, or — when k_raygen can initialise them for the film pass (film_fused) — pixel-group-major, padded to whole groups of 64 pixels
    const bool fused = film_fused(ctx, ds, spp);
    const uint64_t l_slots = fused ? ((npix + 63u) / 64u) * 64u * (uint64_t)spp : total_slots;
    if (int rc = ensure(ctx, ctx->poison, l_slots)) return rc;
    if (int rc = ensure(ctx, ctx->Lbuf, l_slots * sizeof(float4))) return rc;
    if (int rc = ensure_film_samples(ctx, ds, total_slots)) return rc;
    const size_t film_bytes = (size_t)ds.film_w * ds.film_h * sizeof(float4);
    void* d_film = out;
    if (!out_is_device) {
        if (int rc = ensure(ctx, ctx->film, film_bytes)) return rc;
        d_film = ctx->film.p;
    }
    hipStream_t st = ctx->stream;
    const DeviceSensor* dsp = (const DeviceSensor*)ctx->sensor.p;
    float4* L = (float4*)ctx->Lbuf.p;

    hclk.tick("setup (budget, buffers)");
    Timer tm(ctx, ctx->timing && stats);
    hipEvent_t e0, e1, ev_start;
    HIP_TRY(ctx, hipEventCreate(&e0));
    HIP_TRY(ctx, hipEventCreate(&e1));
    HIP_TRY(ctx, hipEventCreateWithFlags(&ev_start, hipEventDisableTiming));
    HIP_TRY(ctx, hipEventRecord(e0, st));
    FilmSideTable fside{nullptr, nullptr, 0};
    if (fused) {
        bool ok = false;
        fside = film_side_table(ctx, st, total_slots, &ok);
        if (!ok) return TRHIP_ERR_HIP;
    } else {
        HIP_TRY(ctx, hipMemsetAsync(L, 0, total_slots * sizeof(float4), st));
    }
    HIP_TRY(ctx, hipMemsetAsync(ctx->poison.p, 0, l_slots, st));
    HIP_TRY(ctx, hipEventRecord(ev_start, st));
    const int g_shade = ctx->num_cu * 8;
    const uint32_t bary_mode = (ctx->traversal >= 2 && scene->wide_ok) ? 1u : 0u;  // k_trace2 hands the barycentrics to the shading kernel
    uint32_t n_batches = 0;
    for (int pi = 0; pi < NP; ++pi) {
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->pipes[pi].st, ev_start, 0));
        HIP_TRY(ctx, hipMemsetAsync(ctx->pipes[pi].counters.p, 0, sizeof(Counters), ctx->pipes[pi].st));
    }
    for (uint64_t s0 = 0; s0 < spp; s0 += spp_batch) {
        Pipe& pp = ctx->pipes[n_batches % NP];
        n_batches++;
        const uint64_t nb = std::min<uint64_t>(spp_batch, spp - s0) * npix;
        // Within a batch, shadow rays of depth d (any-hit + accumulate) and closest-hit rays of depth d+1 are independent: two streams.
        hipStream_t ps = pp.st, ps2 = ctx->overlap ? pp.st2 : pp.st;
        Counters* ctr = (Counters*)pp.counters.p;
        PathQueue pq[2];
        for (int k = 0; k < 2; ++k) pq[k] = PathQueue{(float4*)pp.q[k][0].p, (float4*)pp.q[k][1].p, (float4*)pp.q[k][2].p};
        // Two shadow queues, by parity of the depth: the shadow rays of depth d (low-priority stream, starved while the closest-hit rays of
        // depth d + 1 run) may go on while shade(d + 1) fills the other queue; shade(d + 2) waits for them.  They add into L, shade(d + 1)
        // only NOTES a non-finite beta in ctx->poison (k_apply_poison) — no two writers of one L entry at a time.  (One queue made every
        // shade launch wait for the shadow rays of the depth before: 2-6 ms each, 26 ms of the 436 ms S-mesh frame.)
        const bool two = ps2 != ps;
        const ShadowQueue sqs[2] = {ShadowQueue{(float4*)pp.sq[0].p, (float4*)pp.sq[1].p, (float4*)pp.sq[2].p},
                                    two ? ShadowQueue{(float4*)pp.sq2[0].p, (float4*)pp.sq2[1].p, (float4*)pp.sq2[2].p} : ShadowQueue{(float4*)pp.sq[0].p, (float4*)pp.sq[1].p, (float4*)pp.sq[2].p}};
        hipEvent_t ev_anys[2] = {pp.ev_any, pp.ev_any2};
        float4* hits = (float4*)pp.hits.p;
        HIP_TRY(ctx, hipMemsetAsync(ctr, 0, offsetof(Counters, closest_total), ps));  // queue sizes + work cursors of this batch
        tm.begin(0, ps);
        hipLaunchKernelGGL(k_raygen, dim3(grid_for(ctx, nb, 8)), dim3(kBlock), 0, ps, dsp, (uint32_t)(s0 * npix), (uint32_t)nb, seed, sample_offset, pq[0], cap, ctr, fused ? L : nullptr, spp, fside);
        tm.end(0, ps);
        int cur = 0;
        for (int depth = 1; depth <= max_depth; ++depth) {
            const ShadowQueue& sq = sqs[depth & 1];
            tm.begin(1, ps);
            launch_trace(ctx, ps, scene, false, SegQueue{ctr->n_queue[depth - 1], cap, 0u}, pq[cur].o, pq[cur].d, nullptr,
                         TraceOut{hits, nullptr, nullptr, nullptr, bary_mode, depth == 1 && far_camera(scene, sensor) ? 1u : 0u}, ctr->work_closest[depth - 1], ctr, pp.overflow[0].p);
            tm.end(1, ps);
            if (two && depth > 2) HIP_TRY(ctx, hipStreamWaitEvent(ps, ev_anys[depth & 1], 0));  // shade(d) refills the queue the shadow rays of depth d - 2 read
            tm.begin(2, ps);
            const ShadeStream sst{nullptr, nullptr, 0u, two ? (uint8_t*)ctx->poison.p : nullptr};
            if (scene->dev.tri_tan) {
                hipLaunchKernelGGL(k_shade_path<false>, dim3(g_shade), dim3(kBlock), 0, ps, scene->dev, dsp, pq[cur], pq[cur ^ 1], sq, cap, hits, L, ctr, depth - 1, depth, max_depth, bary_mode, sst);
            } else {
                hipLaunchKernelGGL((k_shade_path<false, false>), dim3(g_shade), dim3(kBlock), 0, ps, scene->dev, dsp, pq[cur], pq[cur ^ 1], sq, cap, hits, L, ctr, depth - 1, depth, max_depth, bary_mode, sst);
            }
            tm.end(2, ps);
            if (two) {
                HIP_TRY(ctx, hipEventRecord(pp.ev_shade, ps));
                HIP_TRY(ctx, hipStreamWaitEvent(ps2, pp.ev_shade, 0));
            }
            tm.begin(3, ps2);
            launch_trace(ctx, ps2, scene, true, SegQueue{ctr->n_shadow[depth - 1], cap, 0u}, sq.o, sq.d, nullptr, TraceOut{nullptr, L, sq.c, nullptr}, ctr->work_shadow[depth - 1], ctr, pp.overflow[1].p);
            tm.end(3, ps2);
            if (two) HIP_TRY(ctx, hipEventRecord(ev_anys[depth & 1], ps2));
            cur ^= 1;
        }
        if (two) {  // the pipeline's next batch (or the film gather) needs every shadow ray resolved
            HIP_TRY(ctx, hipStreamWaitEvent(ps, ev_anys[max_depth & 1], 0));                       // the last depth's shadow rays
            if (max_depth >= 2) HIP_TRY(ctx, hipStreamWaitEvent(ps, ev_anys[(max_depth - 1) & 1], 0));  // and the depth's before
        }
    }
    for (int pi = 0; pi < NP; ++pi) {
        HIP_TRY(ctx, hipEventRecord(ctx->pipes[pi].ev_done, ctx->pipes[pi].st));
        HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->pipes[pi].ev_done, 0));
    }
    tm.begin(4, st);
    if (ctx->overlap) hipLaunchKernelGGL(k_apply_poison, dim3(grid_for(ctx, l_slots, 8)), dim3(kBlock), 0, st, L, (const uint8_t*)ctx->poison.p, l_slots);
    launch_film(ctx, st, ds, dsp, L, total_slots, spp, seed, sample_offset, (float4*)d_film, fused);
    tm.end(4, st);
    ctx->last_L_layout = fused ? 1u : 0u;
    ctx->last_L_npix = (uint32_t)npix;
    ctx->last_L_spp = spp;
    HIP_TRY(ctx, hipEventRecord(e1, st));
    HIP_TRY(ctx, hipGetLastError());
    hclk.tick("enqueue");
    HIP_TRY(ctx, hipStreamSynchronize(st));
    hclk.tick("stream sync");
    ctx->last_L_count = total_slots;
    if (!out_is_device) HIP_TRY(ctx, hipMemcpy(out, d_film, film_bytes, hipMemcpyDeviceToHost));
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        stats->camera_samples = total_slots;
        for (int pi = 0; pi < NP; ++pi) {
            Counters h;
            HIP_TRY(ctx, hipMemcpy(&h, ctx->pipes[pi].counters.p, sizeof h, hipMemcpyDeviceToHost));
            stats->closest_rays += h.closest_total;
            stats->shadow_rays += h.shadow_total;
            stats->nodes_visited += h.nodes_closest;
            stats->prims_tested += h.prims_closest;
            stats->nodes_visited_shadow += h.nodes_shadow;
            stats->prims_tested_shadow += h.prims_shadow;
            stats->fallback_rays += h.fallback_total;
            stats->nodes_visited_fallback += h.nodes_fallback;
            stats->prims_tested_fallback += h.prims_fallback;
            for (int k = 0; k < 4; ++k) stats->count_sub[k] += h.fallback_why[k];
        }
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        stats->ms_total = ms;
        stats->ms_raygen = tm.total(0, &stats->launches_raygen);
        stats->ms_trace_closest = tm.total(1, &stats->launches_trace_closest);
        stats->ms_fallback = tm.fallback_total(&stats->launches_fallback);
        stats->ms_shade = tm.total(2, &stats->launches_shade);
        stats->ms_trace_any = tm.total(3, &stats->launches_trace_any);
        stats->ms_film = tm.total(4, &stats->launches_film);
        stats->n_batches = n_batches;
        stats->max_depth_reached = (uint32_t)max_depth;
        traversal_info(ctx, scene, &stats->traversal, &stats->node_bytes);
        // The certified walk's cliff: scenes made of near-ties (tiny coplanar triangles, rays in a wall's plane — tests/attack_scenes.py) send up to half of their rays to the
        // reference-order walk: exact, at about twice the closest-hit time.  Said once per context, and kept for trhip_accelerator_note.
        ctx->last_fallback_share = stats->closest_rays ? (double)stats->fallback_rays / (double)stats->closest_rays : 0.0;
        if (stats->traversal == 9 && ctx->last_fallback_share > 0.2 && !ctx->warned_fallback_cliff) {
            ctx->warned_fallback_cliff = true;
            std::fprintf(stderr, "[tracehip] the certified walk handed %.0f %% of this frame's closest-hit rays back to the reference-order walk (near-ties, grazed leaf boxes, rays in a "
                                 "primitive's plane): the frame is exact but the accelerator saves little on this scene\n", 100.0 * ctx->last_fallback_share);
        }
    }
    hclk.tick("counters + event times");
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipEventDestroy(ev_start);
    return 0;
}

// Bands of a frame whose per-sample buffers do not fit (render_impl below): whole rows of 16 x 16 sample tiles, as many per band as `budget` bytes hold at `bytes_per_sample`
// (and fewer than 2^32 sample slots: 32-bit indices).  Host arithmetic only: trhip_plan_bands exposes it, tests/test_sharding_gloo.py checks C5's numbers on CPU.
int plan_rows_per_band(const DeviceSensor& ds, uint32_t spp, double budget_bytes, double bytes_per_sample) {
    const double row_bytes = (double)ds.sb_w * 16.0 * (double)spp * bytes_per_sample;
    int rows = (int)std::max(1.0, std::min((double)ds.tiles_y, std::floor(budget_bytes / row_bytes)));
    while (rows > 1 && (uint64_t)ds.sb_w * 16ull * (uint64_t)rows * spp >= (1ull << 32)) --rows;
    return rows;
}
DeviceSensor band_of(const DeviceSensor& ds, int t0, int rows_per_band) {
    DeviceSensor b = ds;
    b.band_ty0 = t0;
    b.band_ty1 = std::min(ds.tiles_y, t0 + rows_per_band) - 1;
    b.band_y0 = ds.sb_min[1] + 16 * t0;
    b.band_rows = std::min(ds.sb_max[1], ds.sb_min[1] + 16 * b.band_ty1 + 15) - b.band_y0 + 1;
    b.accumulate = t0 > 0 ? 1 : 0;
    return b;
}
// A frame: one band when its per-sample buffers (radiance 16 B + film position 8 B per camera sample) fit in HBM next to the queues — every
// BASELINE configuration up to 1024^2 x 256 spp does — otherwise bands of whole tile rows, rendered one after the other into the same film.
// The film is the sequential tile loop's bit for bit either way: a film pixel receives its tiles in k order (integrators/sampler.jl:24-52,
// film.jl:182-193), and bands are ranges of k.  (4096^2 x 1024 spp, BASELINE configs[4], is 412 GB of samples: 3 bands on one MI355X.)
int render_impl(trhip_ctx* ctx, const trhip_scene* scene, const trhip_sensor* sensor, int integrator, uint32_t spp, int max_depth, uint64_t seed, uint32_t sample_offset, void* out,
                bool out_is_device, trhip_stats* stats) {
    if (!ctx || !scene || !sensor || !out) return fail(ctx, TRHIP_ERR_INVALID, "null argument");
    DeviceSensor ds;
    derive_sensor(sensor, ds);
    int rows_per_band = ds.tiles_y;  // in tile rows
    if (integrator == 1 && ds.sb_w > 0 && ds.sb_h > 0 && spp > 0) {
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        if (ctx->band_tile_rows > 0) {
            rows_per_band = std::min<int>(ds.tiles_y, ctx->band_tile_rows);
        } else {
            size_t free_b = 0, total_b = 0;
            HIP_TRY(ctx, hipMemGetInfo(&free_b, &total_b));
            size_t held = ctx->Lbuf.bytes + ctx->pfilm.bytes;
            for (auto& pp : ctx->pipes) {
                held += pp.hits.bytes;
                for (auto& a : pp.q)
                    for (auto& b : a) held += b.bytes;
                for (auto& b : pp.sq) held += b.bytes;
            }
            // half of what is free for the per-sample buffers, the rest for the wavefront queues (164 B per path in flight); 32-bit slot indices
            const double budget = std::min(0.5 * (double)(free_b + held), 4.0e9 * 24.0);
            rows_per_band = plan_rows_per_band(ds, spp, budget, film_uses_packed(ctx, ds) ? 17.0 : film_uses_desc(ctx, ds) ? 33.0 : 25.0);  // radiance (+ descriptor / film position) + poison byte
        }
    }
    if (rows_per_band >= ds.tiles_y) return render_impl_band(ctx, scene, sensor, integrator, spp, max_depth, seed, sample_offset, out, out_is_device, stats, nullptr);
    const size_t film_bytes = (size_t)ds.film_w * ds.film_h * sizeof(float4);
    void* d_film = out;
    if (!out_is_device) {
        if (int rc = ensure(ctx, ctx->film, film_bytes)) return rc;
        d_film = ctx->film.p;
    }
    trhip_stats sum;
    std::memset(&sum, 0, sizeof sum);
    uint32_t n_bands = 0;
    for (int t0 = 0; t0 < ds.tiles_y; t0 += rows_per_band, ++n_bands) {
        const DeviceSensor b = band_of(ds, t0, rows_per_band);
        trhip_stats st;
        if (int rc = render_impl_band(ctx, scene, sensor, integrator, spp, max_depth, seed, sample_offset, d_film, true, &st, &b)) return rc;
        sum.camera_samples += st.camera_samples;
        sum.closest_rays += st.closest_rays;
        sum.shadow_rays += st.shadow_rays;
        sum.nodes_visited += st.nodes_visited;
        sum.prims_tested += st.prims_tested;
        sum.nodes_visited_shadow += st.nodes_visited_shadow;
        sum.prims_tested_shadow += st.prims_tested_shadow;
        sum.fallback_rays += st.fallback_rays;
        sum.nodes_visited_fallback += st.nodes_visited_fallback;
        sum.prims_tested_fallback += st.prims_tested_fallback;
        sum.ms_fallback += st.ms_fallback;
        sum.launches_fallback += st.launches_fallback;
        for (int k = 0; k < 4; ++k) sum.count_sub[k] += st.count_sub[k];
        sum.ms_total += st.ms_total;
        sum.ms_raygen += st.ms_raygen;
        sum.ms_trace_closest += st.ms_trace_closest;
        sum.ms_shade += st.ms_shade;
        sum.ms_trace_any += st.ms_trace_any;
        sum.ms_film += st.ms_film;
        sum.launches_raygen += st.launches_raygen;
        sum.launches_trace_closest += st.launches_trace_closest;
        sum.launches_shade += st.launches_shade;
        sum.launches_trace_any += st.launches_trace_any;
        sum.launches_film += st.launches_film;
        sum.n_batches += st.n_batches;
        sum.max_depth_reached = st.max_depth_reached;
        sum.traversal = st.traversal;
        sum.node_bytes = st.node_bytes;
    }
    ctx->last_L_count = 0;  // trhip_last_sample_radiance describes whole frames only
    ctx->last_L_layout = 0;
    if (!out_is_device) HIP_TRY(ctx, hipMemcpy(out, d_film, film_bytes, hipMemcpyDeviceToHost));
    if (stats) *stats = sum;
    return 0;
}
}  // namespace

extern "C" {

int trhip_plan_bands(const trhip_sensor* sensor, uint32_t spp, uint64_t budget_bytes, uint32_t bytes_per_sample, uint32_t cap, uint32_t* n_bands, int32_t* first_row, int32_t* n_rows) {
    if (!sensor || !n_bands || spp == 0 || bytes_per_sample == 0) return TRHIP_ERR_INVALID;
    DeviceSensor ds;
    derive_sensor(sensor, ds);
    if (ds.sb_w <= 0 || ds.sb_h <= 0) return TRHIP_ERR_INVALID;
    const int rows = plan_rows_per_band(ds, spp, (double)budget_bytes, (double)bytes_per_sample);
    uint32_t n = 0;
    for (int t0 = 0; t0 < ds.tiles_y; t0 += rows, ++n) {
        const DeviceSensor b = band_of(ds, t0, rows);
        if (n < cap) {
            if (first_row) first_row[n] = b.band_y0;
            if (n_rows) n_rows[n] = b.band_rows;
        }
    }
    *n_bands = n;
    return 0;
}


int trhip_render_path(trhip_ctx* ctx, const trhip_scene* sc, const trhip_sensor* sn, uint32_t spp, int max_depth, uint64_t seed, uint32_t off, float* out, trhip_stats* st) {
    return render_impl(ctx, sc, sn, 1, spp, max_depth, seed, off, out, false, st);
}
int trhip_render_path_device(trhip_ctx* ctx, const trhip_scene* sc, const trhip_sensor* sn, uint32_t spp, int max_depth, uint64_t seed, uint32_t off, void* d_out, trhip_stats* st) {
    return render_impl(ctx, sc, sn, 1, spp, max_depth, seed, off, d_out, true, st);
}
int trhip_render_whitted(trhip_ctx* ctx, const trhip_scene* sc, const trhip_sensor* sn, uint32_t spp, int max_depth, uint64_t seed, uint32_t off, float* out, trhip_stats* st) {
    return render_impl(ctx, sc, sn, 0, spp, max_depth, seed, off, out, false, st);
}
int trhip_render_whitted_device(trhip_ctx* ctx, const trhip_scene* sc, const trhip_sensor* sn, uint32_t spp, int max_depth, uint64_t seed, uint32_t off, void* d_out, trhip_stats* st) {
    return render_impl(ctx, sc, sn, 0, spp, max_depth, seed, off, d_out, true, st);
}
int trhip_last_sample_radiance(trhip_ctx* ctx, float* out, uint64_t n_floats) {
    if (!ctx || !out) return fail(ctx, TRHIP_ERR_INVALID, "null argument");
    if (n_floats != ctx->last_L_count * 3) return fail(ctx, TRHIP_ERR_INVALID, "expected %llu floats", (unsigned long long)(ctx->last_L_count * 3));
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (int rc = ensure(ctx, ctx->scratch[0], n_floats * sizeof(float))) return rc;
    const uint64_t n = ctx->last_L_count;
    if (n) hipLaunchKernelGGL(k_export_L, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const float4*)ctx->Lbuf.p, n, (float*)ctx->scratch[0].p, ctx->last_L_layout, ctx->last_L_npix, ctx->last_L_spp);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(out, ctx->scratch[0].p, n_floats * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}
int trhip_film_to_rgb(trhip_ctx* ctx, const float* xyzw, uint32_t w, uint32_t h, float scale, float* out) {
    if (!ctx || !xyzw || !out) return fail(ctx, TRHIP_ERR_INVALID, "null argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint32_t n = w * h;
    if (int rc = upload(ctx, ctx->scratch[0], xyzw, (size_t)n * 4 * sizeof(float))) return rc;
    if (int rc = ensure(ctx, ctx->scratch[1], (size_t)n * 3 * sizeof(float))) return rc;
    if (n) hipLaunchKernelGGL(k_film_to_rgb, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, (const float4*)ctx->scratch[0].p, n, scale, (float*)ctx->scratch[1].p);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(out, ctx->scratch[1].p, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}
int trhip_generate_rays(trhip_ctx* ctx, const trhip_sensor* sn, const float* samples5, uint64_t n, float* out8) {
    if (!ctx || !sn || !samples5 || !out8) return fail(ctx, TRHIP_ERR_INVALID, "null argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DeviceSensor ds;
    derive_sensor(sn, ds);
    if (int rc = upload(ctx, ctx->sensor, &ds, sizeof ds)) return rc;
    if (int rc = upload(ctx, ctx->scratch[0], samples5, n * 5 * sizeof(float))) return rc;
    if (int rc = ensure(ctx, ctx->scratch[1], n * 8 * sizeof(float))) return rc;
    if (n) hipLaunchKernelGGL(k_generate_rays, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const DeviceSensor*)ctx->sensor.p, (const float*)ctx->scratch[0].p, (uint32_t)n,
                              (float*)ctx->scratch[1].p);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(out8, ctx->scratch[1].p, n * 8 * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}
int trhip_bsdf_query(trhip_ctx* ctx, const trhip_scene* sc, uint32_t material, int multi, int mode, int flags, const float* frame9, const float* dirs6, uint64_t n, float* out8) {
    if (!ctx || !sc || !frame9 || !dirs6 || !out8) return fail(ctx, TRHIP_ERR_INVALID, "null argument");
    if (!sc->committed) return fail(ctx, TRHIP_ERR_INVALID, "scene not committed");
    if (material >= sc->materials.size()) return fail(ctx, TRHIP_ERR_INVALID, "material %u not defined", material);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (int rc = upload(ctx, ctx->scratch[0], frame9, n * 9 * sizeof(float))) return rc;
    if (int rc = upload(ctx, ctx->scratch[1], dirs6, n * 6 * sizeof(float))) return rc;
    if (int rc = ensure(ctx, ctx->scratch[2], n * 8 * sizeof(float))) return rc;
    if (n) hipLaunchKernelGGL(k_bsdf_query, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, sc->dev, material, multi, mode, flags, (const float*)ctx->scratch[0].p,
                              (const float*)ctx->scratch[1].p, (uint32_t)n, (float*)ctx->scratch[2].p);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(out8, ctx->scratch[2].p, n * 8 * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}
int trhip_film_accumulate(trhip_ctx* ctx, const trhip_sensor* sn, uint32_t spp, uint64_t seed, uint32_t sample_offset, const float* sample_L, float* out_xyzw) {
    if (!ctx || !sn || !sample_L || !out_xyzw) return fail(ctx, TRHIP_ERR_INVALID, "null argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DeviceSensor ds;
    derive_sensor(sn, ds);
    const uint64_t n = (uint64_t)ds.sb_w * ds.sb_h * spp;
    if (int rc = upload(ctx, ctx->sensor, &ds, sizeof ds)) return rc;
    if (int rc = upload(ctx, ctx->table, sn->filter_table, 256 * sizeof(float))) return rc;
    if (int rc = upload(ctx, ctx->scratch[0], sample_L, n * 3 * sizeof(float))) return rc;
    if (int rc = ensure(ctx, ctx->Lbuf, n * sizeof(float4))) return rc;
    const size_t film_bytes = (size_t)ds.film_w * ds.film_h * sizeof(float4);
    if (int rc = ensure(ctx, ctx->film, film_bytes)) return rc;
    if (int rc = ensure_film_samples(ctx, ds, n)) return rc;
    if (n) hipLaunchKernelGGL(k_import_L, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const float*)ctx->scratch[0].p, n, (float4*)ctx->Lbuf.p);
    launch_film(ctx, ctx->stream, ds, (const DeviceSensor*)ctx->sensor.p, (const float4*)ctx->Lbuf.p, n, spp, seed, sample_offset, (float4*)ctx->film.p, false);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->last_L_count = n;
    ctx->last_L_layout = 0;
    HIP_TRY(ctx, hipMemcpy(out_xyzw, ctx->film.p, film_bytes, hipMemcpyDeviceToHost));
    return 0;
}

}  // extern "C"
