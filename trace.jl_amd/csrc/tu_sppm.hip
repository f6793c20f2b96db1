// tu_sppm.hip — SPPMIntegrator (th_sppm.h): camera pass, hash grid, photon pass, pixel update.
#include "th_host.h"
#include "th_sppm.h"

namespace {
// SPPMIntegrator (integrators/sppm.jl:132-173): n_iterations x {camera pass, grid, photon pass, pixel update}, then
// _sppm_to_image + set_image!.  Everything runs on one stream; queue sizes stay in HBM, the host only enqueues.
int render_sppm_impl(trhip_ctx* ctx, const trhip_scene* scene, const trhip_sensor* sensor, float initial_radius, int max_depth, uint32_t n_iterations, int64_t photons_per_iteration,
                     uint64_t seed, float* out_xyzw, trhip_stats* stats, uint32_t write_frequency, trhip_sppm_write_fn write_cb, void* write_user) {
    if (!ctx || !scene || !sensor || !out_xyzw) return fail(ctx, TRHIP_ERR_INVALID, "null argument");
    if (!scene->committed) return fail(ctx, TRHIP_ERR_INVALID, "scene not committed");
    if (n_iterations == 0 || max_depth < 1 || max_depth > kMaxDepth) return fail(ctx, TRHIP_ERR_INVALID, "n_iterations must be >= 1 and max_depth in 1..%d", kMaxDepth);
    if (!(initial_radius > 0.0f)) return fail(ctx, TRHIP_ERR_INVALID, "initial_search_radius must be positive");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DeviceSensor ds;
    derive_sensor(sensor, ds);
    if (ds.film_w <= 0 || ds.film_h <= 0) return fail(ctx, TRHIP_ERR_INVALID, "empty film");
    if (ds.crop_min[0] != 1.0f || ds.crop_min[1] != 1.0f)
        return fail(ctx, TRHIP_ERR_UNSUPPORTED, "SPPM needs a film whose crop starts at pixel (1, 1): sppm.jl:203 indexes pixels[y, x] with raster coordinates");
    if (scene->has_materialless_prim) return fail(ctx, TRHIP_ERR_UNSUPPORTED, "SPPM: primitives without a material are not supported on the device");
    const uint32_t W = (uint32_t)ds.film_w, H = (uint32_t)ds.film_h;
    const uint64_t n64 = (uint64_t)W * H;
    if (n64 >= (1ull << 26)) return fail(ctx, TRHIP_ERR_UNSUPPORTED, "SPPM: more than 2^26 film pixels");
    const uint32_t n = (uint32_t)n64;
    const int64_t P = photons_per_iteration > 0 ? photons_per_iteration : (int64_t)((ds.crop_max[0] - ds.crop_min[0]) * (ds.crop_max[1] - ds.crop_min[1]));  // area(crop_bounds) :121-124
    if (P >= (1ll << 31)) return fail(ctx, TRHIP_ERR_UNSUPPORTED, "SPPM: more than 2^31 photons per iteration");
    const uint32_t n_lights = scene->dev.n_lights;
    // light power distribution (sampling.jl:3-31, sppm.jl:564-569) on the host
    std::vector<float> ld_host;
    float func_int = 0.0f;
    if (n_lights) {
        std::vector<float> func(n_lights), cdf(n_lights + 1);
        for (uint32_t l = 0; l < n_lights; ++l) {
            const LightRec& lr = scene->lights[l];
            const f3 I = mk3(lr.I[0], lr.I[1], lr.I[2]);
            const f3 power = lr.kind == 0 ? 4.0f * kPi * I : I * 2.0f * kPi * (1.0f - 0.5f * (lr.cos_falloff_start + lr.cos_total_width));  // point.jl:74-76, spot.jl:42-44
            func[l] = to_Y(power);
        }
        cdf[0] = 0.0f;
        for (uint32_t i = 1; i <= n_lights; ++i) cdf[i] = cdf[i - 1] + func[i - 1] / (float)n_lights;
        func_int = cdf[n_lights];
        for (uint32_t i = 1; i <= n_lights; ++i) cdf[i] = func_int == 0.0f ? (float)((double)(i + 1) / (double)n_lights) : cdf[i] / func_int;
        ld_host = func;
        ld_host.insert(ld_host.end(), cdf.begin(), cdf.end());
        if (int rc = upload(ctx, ctx->sp_ldist, ld_host.data(), ld_host.size() * sizeof(float))) return rc;
    }
    const LightDistribution ldist{(const float*)ctx->sp_ldist.p, (const float*)ctx->sp_ldist.p + n_lights, func_int, (int32_t)n_lights};
    // Iterations are processed in batches of B: the camera paths and the photon paths of different iterations are
    // independent of the pixel statistics, so B iterations share every traversal / shading launch (B x the rays per
    // launch, 1/B the launches and traversal tails); only grid -> deposit -> update runs once per iteration, in order.
    const uint64_t Qit = std::max<uint64_t>(n, (uint64_t)P);
    const int ndep = std::max(1, max_depth - 1);
    const double per_iter = (double)n * (7 * 16.0 + max_depth * 16.0) + (double)P * ndep * 49.0 + (double)Qit * 10 * 16.0;
    uint64_t B = ctx->sppm_batch;
    if (B == 0) {
        size_t free_b = 0, total_b = 0;
        HIP_TRY(ctx, hipMemGetInfo(&free_b, &total_b));
        B = (uint64_t)std::max(1.0, 0.5 * (double)free_b / per_iter);
        B = std::min<uint64_t>(B, 128);  // measured on C4: 32 iterations per batch 906 ms, 50: 769 ms, 100: 701 ms
    }
    B = std::min<uint64_t>(B, n_iterations);
    while (B > 1 && (B * Qit >= (1ull << 31) || B * (uint64_t)max_depth * n >= (1ull << 32) || B * (uint64_t)P * ndep >= (1ull << 32))) B = (B + 1) / 2;
    if ((uint64_t)P * ndep >= (1ull << 32)) return fail(ctx, TRHIP_ERR_UNSUPPORTED, "SPPM: photons_per_iteration x (max_depth - 1) must stay below 2^32");
    const uint64_t Q = B * Qit;
    const uint32_t cap = (uint32_t)(((Q + kSeg - 1) / kSeg + 2 * kSegGran + kSegGran - 1) / kSegGran * kSegGran);
    const uint64_t Pphys = (uint64_t)cap * kSeg;
    Pipe& pp = ctx->pipes[0];
    if (!pp.st) {
        HIP_TRY(ctx, hipStreamCreate(&pp.st));
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
    if (int rc = ensure(ctx, pp.hits, Pphys * sizeof(float4))) return rc;
    if (int rc = ensure(ctx, pp.counters, sizeof(Counters))) return rc;
    const size_t slab_bytes = (size_t)trace_grid(ctx) * kBlock * (size_t)kStackSlabLevels * sizeof(uint2);
    if (int rc = ensure(ctx, pp.overflow[0], slab_bytes)) return rc;
    for (auto& b : ctx->sp_vp)
        if (int rc = ensure(ctx, b, (size_t)B * n * sizeof(float4))) return rc;
    const size_t n_terms = (size_t)B * max_depth * n;
    const size_t n_rec = (size_t)B * (size_t)P * ndep;
    if (int rc = ensure(ctx, ctx->sp_terms, n_terms * sizeof(float4))) return rc;
    for (auto& b : ctx->sp_rec)
        if (int rc = ensure(ctx, b, n_rec * sizeof(float4))) return rc;
    if (int rc = ensure(ctx, ctx->sp_rec_valid, n_rec)) return rc;
    const size_t entry_cap = (size_t)P * ndep;  // photon hits of one iteration, sorted by bucket
    if (int rc = ensure(ctx, ctx->sp_Ld, (size_t)n * sizeof(float4))) return rc;
    if (int rc = ensure(ctx, ctx->sp_tau, (size_t)n * sizeof(float4))) return rc;
    if (int rc = ensure(ctx, ctx->sp_radius, (size_t)n * sizeof(float))) return rc;
    if (int rc = ensure(ctx, ctx->sp_N, (size_t)n * sizeof(double))) return rc;
    if (int rc = ensure(ctx, ctx->sp_phi, (size_t)n * 3 * sizeof(float))) return rc;
    if (int rc = ensure(ctx, ctx->sp_M, (size_t)n * sizeof(uint32_t))) return rc;
    if (int rc = ensure(ctx, ctx->sp_counts, (size_t)n * sizeof(uint32_t))) return rc;
    if (int rc = ensure(ctx, ctx->sp_starts, ((size_t)n + 1) * sizeof(uint32_t))) return rc;
    if (int rc = ensure(ctx, ctx->sp_entries, (size_t)entry_cap * sizeof(float4))) return rc;
    if (int rc = ensure(ctx, ctx->sp_grid, sizeof(GridInfo))) return rc;
    if (int rc = ensure(ctx, ctx->sp_snap_M, (size_t)n * sizeof(uint32_t))) return rc;
    if (int rc = ensure(ctx, ctx->sp_snap_phi, (size_t)n * 3 * sizeof(float))) return rc;
    if (int rc = ensure(ctx, ctx->sp_snap_p, (size_t)n * sizeof(float4))) return rc;
    if (int rc = ensure(ctx, ctx->sp_snap_beta, (size_t)n * sizeof(float4))) return rc;
    if (int rc = ensure(ctx, ctx->film, (size_t)n * sizeof(float4))) return rc;
    if (int rc = upload(ctx, ctx->sensor, &ds, sizeof ds)) return rc;
    ctx->sp_pixels = n;
    ctx->sp_photons = P;
    hipStream_t st = pp.st;
    const DeviceSensor* dsp = (const DeviceSensor*)ctx->sensor.p;
    Counters* ctr = (Counters*)pp.counters.p;
    GridInfo* grid = (GridInfo*)ctx->sp_grid.p;
    PathQueue pq[2];
    for (int k = 0; k < 2; ++k) pq[k] = PathQueue{(float4*)pp.q[k][0].p, (float4*)pp.q[k][1].p, (float4*)pp.q[k][2].p};
    ShadowQueue sq{(float4*)pp.sq[0].p, (float4*)pp.sq[1].p, (float4*)pp.sq[2].p};
    float4* hits = (float4*)pp.hits.p;
    const VisiblePoints vp_all{(float4*)ctx->sp_vp[0].p, (float4*)ctx->sp_vp[1].p, (float4*)ctx->sp_vp[2].p, (float4*)ctx->sp_vp[3].p, (float4*)ctx->sp_vp[4].p,
                               (float4*)ctx->sp_vp[5].p, (float4*)ctx->sp_vp[6].p};
    auto vp_slice = [&](uint64_t j) {
        const size_t o = (size_t)j * n;
        return VisiblePoints{vp_all.p_mat + o, vp_all.wo + o, vp_all.beta + o, vp_all.ng + o, vp_all.ns + o, vp_all.ss + o, vp_all.ts + o};
    };
    PixelStats px{(float4*)ctx->sp_Ld.p, (float4*)ctx->sp_tau.p, (float*)ctx->sp_radius.p, (double*)ctx->sp_N.p, (float*)ctx->sp_phi.p, (uint32_t*)ctx->sp_M.p};
    float4* terms = (float4*)ctx->sp_terms.p;
    const PhotonRecords rec{(float4*)ctx->sp_rec[0].p, (float4*)ctx->sp_rec[1].p, (float4*)ctx->sp_rec[2].p, (uint8_t*)ctx->sp_rec_valid.p};
    uint32_t* counts = (uint32_t*)ctx->sp_counts.p;
    uint32_t* starts = (uint32_t*)ctx->sp_starts.p;
    float4* entries = (float4*)ctx->sp_entries.p;
    const uint32_t n_tiles = (n + kScanTile - 1) / kScanTile;
    if (int rc = ensure(ctx, ctx->scratch[0], (size_t)n_tiles * sizeof(uint32_t))) return rc;
    if (int rc = ensure(ctx, ctx->scratch[1], ((size_t)n_tiles + 1) * sizeof(uint32_t))) return rc;
    uint32_t* tile_sums = (uint32_t*)ctx->scratch[0].p;
    uint32_t* tile_offsets = (uint32_t*)ctx->scratch[1].p;
    if (int rc = ensure(ctx, ctx->scratch[2], (size_t)n * sizeof(uint32_t))) return rc;
    uint32_t* hot_list = (uint32_t*)ctx->scratch[2].p;

    Timer tm(ctx, ctx->timing && stats);
    hipEvent_t e0, e1;
    HIP_TRY(ctx, hipEventCreate(&e0));
    HIP_TRY(ctx, hipEventCreate(&e1));
    HIP_TRY(ctx, hipEventRecord(e0, st));
    // pixels = [SPPMPixel(radius = initial_search_radius) …] (:136-139)
    HIP_TRY(ctx, hipMemsetAsync(counts, 0, (size_t)n * sizeof(uint32_t), st));  // once per call: every iteration leaves them at zero again
    HIP_TRY(ctx, hipMemsetAsync(px.Ld, 0, (size_t)n * sizeof(float4), st));
    HIP_TRY(ctx, hipMemsetAsync(px.tau, 0, (size_t)n * sizeof(float4), st));
    HIP_TRY(ctx, hipMemsetAsync(px.N, 0, (size_t)n * sizeof(double), st));
    HIP_TRY(ctx, hipMemsetAsync(px.phi, 0, (size_t)n * 3 * sizeof(float), st));
    HIP_TRY(ctx, hipMemsetAsync(px.M, 0, (size_t)n * sizeof(uint32_t), st));
    HIP_TRY(ctx, hipMemsetAsync(ctr, 0, sizeof(Counters), st));
    HIP_TRY(ctx, hipMemsetAsync(grid, 0, sizeof(GridInfo), st));
    {
        std::vector<float> r0(n, initial_radius);
        HIP_TRY(ctx, hipMemcpyAsync(px.radius, r0.data(), (size_t)n * sizeof(float), hipMemcpyHostToDevice, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));
    }
    const dim3 blk(kBlock), g_pix(grid_for(ctx, n, 8)), g_shade(ctx->num_cu * 8);
    const float gamma = 2.0f / 3.0f;
    // multi-GPU job: this rank's slice of every iteration's photons (all of them without a communicator)
    const uint64_t n_ranks = ctx->comm.comm ? (uint64_t)ctx->comm.n_ranks : 1u, my_rank = ctx->comm.comm ? (uint64_t)ctx->comm.rank : 0u;
    const uint32_t p_lo = (uint32_t)((uint64_t)P * my_rank / n_ranks), p_hi = (uint32_t)((uint64_t)P * (my_rank + 1) / n_ranks);
    uint32_t n_batches = 0;
    // ray totals after every batch's camera pass and photon pass: what of closest_rays + shadow_rays every rank of a multi-GPU job repeats
    const uint32_t max_batches = (uint32_t)((n_iterations + B - 1) / B) + (write_cb && write_frequency ? n_iterations / write_frequency + 1u : 0u);  // (a write iteration ends its batch)
    if (int rc = ensure(ctx, ctx->sp_raysnap, (size_t)max_batches * 4 * sizeof(unsigned long long))) return rc;
    unsigned long long* raysnap = (unsigned long long*)ctx->sp_raysnap.p;
    // sppm.jl:166-171 stores and saves the image whenever `iteration % write_frequency == 0`: with a write callback no batch runs past such an iteration (the image of
    // iteration k is built from the pixels as they are after k iterations — Ld is folded per batch), and the host gets the image there
    const bool periodic = write_cb != nullptr && write_frequency > 0;
    uint32_t nb = 0;
    for (uint32_t it0 = 1; it0 <= n_iterations; it0 += nb) {
        nb = (uint32_t)std::min<uint64_t>(B, n_iterations - it0 + 1);
        if (periodic) {
            const uint32_t next_write = (uint32_t)std::min<uint64_t>(n_iterations, ((uint64_t)(it0 + write_frequency - 1) / write_frequency) * write_frequency);
            nb = std::min(nb, next_write - it0 + 1);
        }
        n_batches++;
        // ---- camera pass of iterations it0 .. it0 + nb - 1 (:175-270) ----
        for (auto& b : ctx->sp_vp) HIP_TRY(ctx, hipMemsetAsync(b.p, 0, (size_t)nb * n * sizeof(float4), st));  // vp.β = 0: no visible point
        HIP_TRY(ctx, hipMemsetAsync(terms, 0, (size_t)nb * max_depth * n * sizeof(float4), st));
        HIP_TRY(ctx, hipMemsetAsync(ctr, 0, offsetof(Counters, closest_total), st));
        tm.begin(0, st);
        hipLaunchKernelGGL(k_sppm_raygen, dim3(grid_for(ctx, (uint64_t)nb * n, 8)), blk, 0, st, dsp, nb * n, n, W, seed, it0, pq[0], cap, ctr);
        tm.end(0, st);
        int cur = 0;
        for (int depth = 1; depth <= max_depth; ++depth) {
            tm.begin(1, st);
            launch_trace(ctx, st, scene, false, SegQueue{ctr->n_queue[depth - 1], cap, 0u}, pq[cur].o, pq[cur].d, nullptr,
                         TraceOut{hits, nullptr, nullptr, nullptr, 0u, depth == 1 && far_camera(scene, sensor) ? 1u : 0u}, ctr->work_closest[depth - 1], ctr, pp.overflow[0].p);
            tm.end(1, st);
            tm.begin(2, st);
            if (scene->dev.tri_tan)
                hipLaunchKernelGGL(k_shade_sppm<true>, g_shade, blk, 0, st, scene->dev, pq[cur], pq[cur ^ 1], sq, cap, hits, vp_all, terms, ctr, depth, max_depth, seed, it0, n, W);
            else
                hipLaunchKernelGGL(k_shade_sppm<false>, g_shade, blk, 0, st, scene->dev, pq[cur], pq[cur ^ 1], sq, cap, hits, vp_all, terms, ctr, depth, max_depth, seed, it0, n, W);
            tm.end(2, st);
            tm.begin(3, st);
            launch_trace(ctx, st, scene, true, SegQueue{ctr->n_shadow[depth - 1], cap, 0u}, sq.o, sq.d, nullptr, TraceOut{nullptr, terms, sq.c, nullptr, 0u, 0u, 0u, nullptr, 1u}, ctr->work_shadow[depth - 1], ctr,
                         pp.overflow[0].p);
            tm.end(3, st);
            cur ^= 1;
        }
        tm.begin(7, st);
        hipLaunchKernelGGL(k_sppm_fold_ld, g_pix, blk, 0, st, n, nb, (uint32_t)max_depth, (const float4*)terms, px.Ld);
        tm.end(7, st);
        HIP_TRY(ctx, hipMemcpyAsync(raysnap + 4 * (size_t)(n_batches - 1), &ctr->closest_total, 2 * sizeof(unsigned long long), hipMemcpyDeviceToDevice, st));
        // ---- photon paths of the same iterations (:320-365, 393-418): Halton indices (it0 - 1) * P .. (it0 - 1 + nb) * P - 1 ----
        const uint32_t NP = nb * (uint32_t)P;
        if (n_lights) {
            const uint64_t halton_base = (uint64_t)(it0 - 1) * (uint64_t)P;
            HIP_TRY(ctx, hipMemsetAsync(rec.valid, 0, (size_t)NP * ndep, st));
            HIP_TRY(ctx, hipMemsetAsync(ctr, 0, offsetof(Counters, closest_total), st));
            tm.begin(0, st);
            hipLaunchKernelGGL(k_photon_gen, dim3(grid_for(ctx, NP, 8)), blk, 0, st, scene->dev, ldist, NP, halton_base, pq[0], cap, ctr, (uint32_t)P, p_lo, p_hi);
            tm.end(0, st);
            cur = 0;
            for (int depth = 1; depth <= max_depth; ++depth) {
                tm.begin(1, st);
                launch_trace(ctx, st, scene, false, SegQueue{ctr->n_queue[depth - 1], cap, 0u}, pq[cur].o, pq[cur].d, nullptr, TraceOut{hits, nullptr, nullptr, nullptr}, ctr->work_closest[depth - 1],
                             ctr, pp.overflow[0].p);
                tm.end(1, st);
                tm.begin(2, st);
                if (scene->dev.tri_tan)
                    hipLaunchKernelGGL(k_shade_photon<true>, g_shade, blk, 0, st, scene->dev, pq[cur], pq[cur ^ 1], cap, hits, rec, NP, ctr, depth, max_depth, halton_base);
                else
                    hipLaunchKernelGGL(k_shade_photon<false>, g_shade, blk, 0, st, scene->dev, pq[cur], pq[cur ^ 1], cap, hits, rec, NP, ctr, depth, max_depth, halton_base);
                tm.end(2, st);
                cur ^= 1;
            }
        }
        HIP_TRY(ctx, hipMemcpyAsync(raysnap + 4 * (size_t)(n_batches - 1) + 2, &ctr->closest_total, 2 * sizeof(unsigned long long), hipMemcpyDeviceToDevice, st));
        // ---- per iteration, in order: grid (:272-318), photon contributions (:366-391), _update_pixels! (:438-459) ----
        for (uint32_t j = 0; j < nb; ++j) {
            const VisiblePoints vp = vp_slice(j);
            tm.begin(6, st);
            // (the bucket counters are zero here: k_sppm_hit_bin's fill pass counts every bucket back down to 0)
            hipLaunchKernelGGL(k_sppm_grid_reset, dim3(1), blk, 0, st, grid);
            hipLaunchKernelGGL(k_sppm_grid_bounds, dim3(ctx->num_cu), blk, 0, st, vp, (const float*)px.radius, n, grid);  // few waves: 7 same-address atomics each
            hipLaunchKernelGGL(k_sppm_grid_setup, dim3(1), dim3(64), 0, st, grid);
            const dim3 g_rec(grid_for(ctx, (uint64_t)P * ndep, 8));
            if (n_lights)
                hipLaunchKernelGGL(k_sppm_hit_bin, g_rec, blk, 0, st, (const float4*)rec.p, (const uint8_t*)rec.valid, NP, j * (uint32_t)P, (uint32_t)P, (uint32_t)(max_depth - 1), n, grid, counts,
                                   (const uint32_t*)starts, entries, 0);
            hipLaunchKernelGGL(k_sppm_scan_tiles, dim3(n_tiles), blk, 0, st, (const uint32_t*)counts, starts, n, tile_sums);
            hipLaunchKernelGGL(k_sppm_scan, dim3(1), dim3(1024), 0, st, (const uint32_t*)tile_sums, tile_offsets, n_tiles, grid);
            hipLaunchKernelGGL(k_sppm_scan_add, g_pix, blk, 0, st, starts, n, (const uint32_t*)tile_offsets, n_tiles);
            if (n_lights)
                hipLaunchKernelGGL(k_sppm_hit_bin, g_rec, blk, 0, st, (const float4*)rec.p, (const uint8_t*)rec.valid, NP, j * (uint32_t)P, (uint32_t)P, (uint32_t)(max_depth - 1), n, grid, counts,
                                   (const uint32_t*)starts, entries, 1);
            tm.end(6, st);
            tm.begin(5, st);
            hipLaunchKernelGGL(k_sppm_gather, g_pix, blk, 0, st, scene->dev, rec, vp, px, n, grid, (const uint32_t*)starts, (const float4*)entries, n, hot_list, it0 + j == n_iterations ? 1u : 0u, ctx->count_visits ? 1u : 0u);
            hipLaunchKernelGGL(k_sppm_gather_hot, g_shade, blk, 0, st, scene->dev, rec, vp, px, grid, (const uint32_t*)starts, (const float4*)entries, n, (const uint32_t*)hot_list, ctx->count_visits ? 1u : 0u);
            tm.end(5, st);
            if (ctx->comm.comm && ctx->comm.n_ranks > 1) {
                // the one exchange of an iteration: every rank traced its slice of the photons, ϕ and M are the sums over all of
                // them (the reference adds them with Threads.Atomic, sppm.jl:398-399) — then _update_pixels! runs identically everywhere
                RcclApi* api = rccl_api();
                NCCL_TRY(ctx, api->GroupStart());
                NCCL_TRY(ctx, api->AllReduce(px.phi, px.phi, (size_t)n * 3, ncclFloat32, ncclSum, ctx->comm.comm, st));
                NCCL_TRY(ctx, api->AllReduce(px.M, px.M, (size_t)n, ncclUint32, ncclSum, ctx->comm.comm, st));
                NCCL_TRY(ctx, api->GroupEnd());
            }
            if (it0 + j == n_iterations) {  // snapshot for trhip_sppm_state: the last iteration's M, ϕ and visible points
                HIP_TRY(ctx, hipMemcpyAsync(ctx->sp_snap_M.p, px.M, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
                HIP_TRY(ctx, hipMemcpyAsync(ctx->sp_snap_phi.p, px.phi, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToDevice, st));
                HIP_TRY(ctx, hipMemcpyAsync(ctx->sp_snap_p.p, vp.p_mat, (size_t)n * sizeof(float4), hipMemcpyDeviceToDevice, st));
                HIP_TRY(ctx, hipMemcpyAsync(ctx->sp_snap_beta.p, vp.beta, (size_t)n * sizeof(float4), hipMemcpyDeviceToDevice, st));
            }
            tm.begin(7, st);
            hipLaunchKernelGGL(k_sppm_update, g_pix, blk, 0, st, n, gamma, px, vp);
            tm.end(7, st);
        }
        const uint32_t it_done = it0 + nb - 1;
        if (periodic && it_done < n_iterations && it_done % write_frequency == 0) {  // (the last iteration's image is the call's result: the caller stores that one)
            tm.begin(4, st);
            hipLaunchKernelGGL(k_sppm_image, g_pix, blk, 0, st, n, it_done, (uint64_t)P, px, (float4*)ctx->film.p);  // _sppm_to_image(i, pixels, iteration)
            tm.end(4, st);
            HIP_TRY(ctx, hipStreamSynchronize(st));
            HIP_TRY(ctx, hipMemcpy(out_xyzw, ctx->film.p, (size_t)n * sizeof(float4), hipMemcpyDeviceToHost));
            int rc_cb = write_cb(write_user, it_done, out_xyzw);
            if (ctx->comm.comm && ctx->comm.n_ranks > 1) {
                // every rank leaves the loop together, or none does: a rank that returned here alone would leave the others inside the next iteration's all-reduce
                if (int rc = ensure(ctx, ctx->cb_rc, sizeof(int32_t))) return rc;
                const int32_t mine = rc_cb != 0 ? 1 : 0;
                int32_t all = 0;
                HIP_TRY(ctx, hipMemcpyAsync(ctx->cb_rc.p, &mine, sizeof mine, hipMemcpyHostToDevice, st));
                NCCL_TRY(ctx, rccl_api()->AllReduce(ctx->cb_rc.p, ctx->cb_rc.p, 1, ncclInt32, ncclMax, ctx->comm.comm, st));
                HIP_TRY(ctx, hipMemcpyAsync(&all, ctx->cb_rc.p, sizeof all, hipMemcpyDeviceToHost, st));
                HIP_TRY(ctx, hipStreamSynchronize(st));
                if (all && !rc_cb) rc_cb = -1;  // (another rank's callback failed)
            }
            if (rc_cb) return fail(ctx, TRHIP_ERR_INVALID, "the SPPM write callback returned %d at iteration %u%s", rc_cb, it_done, rc_cb == -1 ? " (on another rank of the job)" : "");
        }
    }
    tm.begin(4, st);
    hipLaunchKernelGGL(k_sppm_image, g_pix, blk, 0, st, n, n_iterations, (uint64_t)P, px, (float4*)ctx->film.p);
    tm.end(4, st);
    HIP_TRY(ctx, hipEventRecord(e1, st));
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(st));
    HIP_TRY(ctx, hipMemcpy(out_xyzw, ctx->film.p, (size_t)n * sizeof(float4), hipMemcpyDeviceToHost));
    GridInfo gi;
    HIP_TRY(ctx, hipMemcpy(&gi, grid, sizeof gi, hipMemcpyDeviceToHost));
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        Counters h;
        HIP_TRY(ctx, hipMemcpy(&h, ctr, sizeof h, hipMemcpyDeviceToHost));
        stats->camera_samples = (uint64_t)n * n_iterations;
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
        stats->ms_sub[0] = tm.total(5, &stats->launches_sub[0]);  // photon gather
        stats->ms_sub[1] = tm.total(2, &stats->launches_sub[1]);  // camera / photon shading
        stats->ms_sub[2] = tm.total(6, &stats->launches_sub[2]);  // grid bounds, hit binning, scans
        stats->ms_sub[3] = tm.total(7, &stats->launches_sub[3]);  // Ld fold, pixel update
        stats->ms_shade = stats->ms_sub[0] + stats->ms_sub[1] + stats->ms_sub[2] + stats->ms_sub[3];
        stats->launches_shade = stats->launches_sub[0] + stats->launches_sub[1] + stats->launches_sub[2] + stats->launches_sub[3];
        stats->ms_trace_any = tm.total(3, &stats->launches_trace_any);
        stats->ms_film = tm.total(4, &stats->launches_film);
        stats->n_batches = n_batches;
        stats->max_depth_reached = (uint32_t)max_depth;
        traversal_info(ctx, scene, &stats->traversal, &stats->node_bytes);
        if (ctx->count_visits) {
            stats->count_sub[0] = gi.stat_candidates;
            stats->count_sub[1] = gi.stat_accepted;
            stats->count_sub[2] = gi.photon_hits;
            stats->count_sub[3] = gi.stat_visible_points;
        }
        // the camera pass (closest-hit + shadow rays) is traced by every rank of a job; only the photons are sharded
        std::vector<unsigned long long> snap((size_t)n_batches * 4);
        HIP_TRY(ctx, hipMemcpy(snap.data(), raysnap, snap.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        unsigned long long prev_c = 0, prev_s = 0;
        for (uint32_t b = 0; b < n_batches; ++b) {
            stats->replicated_rays += (snap[4 * b] - prev_c) + (snap[4 * b + 1] - prev_s);
            prev_c = snap[4 * b + 2];
            prev_s = snap[4 * b + 3];
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return 0;
}
}  // namespace

extern "C" {

int trhip_render_sppm_ex(trhip_ctx* ctx, const trhip_scene* sc, const trhip_sensor* sn, float initial_search_radius, int max_depth, uint32_t n_iterations, int64_t photons_per_iteration,
                         uint64_t seed, float* out_xyzw, trhip_stats* st, uint32_t write_frequency, trhip_sppm_write_fn write_cb, void* user) {
    return render_sppm_impl(ctx, sc, sn, initial_search_radius, max_depth, n_iterations, photons_per_iteration, seed, out_xyzw, st, write_frequency, write_cb, user);
}
int trhip_render_sppm(trhip_ctx* ctx, const trhip_scene* sc, const trhip_sensor* sn, float initial_search_radius, int max_depth, uint32_t n_iterations, int64_t photons_per_iteration,
                      uint64_t seed, float* out_xyzw, trhip_stats* st) {
    return render_sppm_impl(ctx, sc, sn, initial_search_radius, max_depth, n_iterations, photons_per_iteration, seed, out_xyzw, st, 0u, nullptr, nullptr);
}
int trhip_sppm_state(trhip_ctx* ctx, float* Ld3, float* tau3, float* radius, double* N, int64_t* M, float* phi3, float* vp_p3, float* vp_beta3, int64_t* info6) {
    if (!ctx) return TRHIP_ERR_INVALID;
    const uint32_t n = ctx->sp_pixels;
    if (n == 0) return fail(ctx, TRHIP_ERR_INVALID, "no SPPM render on this context yet");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::vector<float> f4((size_t)n * 4);
    auto unpack3 = [&](const DevBuf& b, float* out) -> int {
        if (!out) return 0;
        HIP_TRY(ctx, hipMemcpy(f4.data(), b.p, (size_t)n * sizeof(float4), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; ++i)
            for (int c = 0; c < 3; ++c) out[3 * i + c] = f4[4 * i + c];
        return 0;
    };
    if (int rc = unpack3(ctx->sp_Ld, Ld3)) return rc;
    if (int rc = unpack3(ctx->sp_tau, tau3)) return rc;
    if (int rc = unpack3(ctx->sp_snap_p, vp_p3)) return rc;
    if (int rc = unpack3(ctx->sp_snap_beta, vp_beta3)) return rc;
    if (radius) HIP_TRY(ctx, hipMemcpy(radius, ctx->sp_radius.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    if (N) HIP_TRY(ctx, hipMemcpy(N, ctx->sp_N.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    if (phi3) HIP_TRY(ctx, hipMemcpy(phi3, ctx->sp_snap_phi.p, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost));
    if (M) {
        std::vector<uint32_t> m(n);
        HIP_TRY(ctx, hipMemcpy(m.data(), ctx->sp_snap_M.p, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; ++i) M[i] = (int64_t)m[i];
    }
    if (info6) {
        GridInfo gi;
        HIP_TRY(ctx, hipMemcpy(&gi, ctx->sp_grid.p, sizeof gi, hipMemcpyDeviceToHost));
        info6[0] = gi.res[0], info6[1] = gi.res[1], info6[2] = gi.res[2];
        info6[3] = (int64_t)gi.registrations;
        info6[4] = (int64_t)gi.photon_hits;
        info6[5] = ctx->sp_photons;
    }
    return 0;
}

}  // extern "C"
