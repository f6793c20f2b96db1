// tu_api.hip — context, options, the job communicator (RCCL), host utilities of the C ABI (include/tracehip.h).  gfx950 only; there is no CPU
// fallback: every entry point that computes needs a GPU.
#include "th_host.h"

thread_local std::string g_init_error;

extern "C" {

int trhip_version(void) { return 3001; }

int trhip_init(trhip_ctx** out, int device_id) {
    if (!out) return fail(nullptr, TRHIP_ERR_INVALID, "ctx out pointer is null");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) return fail(nullptr, TRHIP_ERR_HIP, "no HIP device available (%s): libtracehip has no CPU fallback", hipGetErrorString(e));
    if (device_id < 0 || device_id >= n) return fail(nullptr, TRHIP_ERR_INVALID, "device %d out of range (%d devices)", device_id, n);
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess) return fail(nullptr, TRHIP_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return fail(nullptr, TRHIP_ERR_HIP, "device %d is %s; this library is built for gfx950 only", device_id, prop.gcnArchName);
    auto ctx = new trhip_ctx();
    ctx->device = device_id;
    ctx->num_cu = prop.multiProcessorCount;
    if ((e = hipSetDevice(device_id)) != hipSuccess || (e = hipStreamCreate(&ctx->stream)) != hipSuccess || (e = hipStreamCreate(&ctx->stream2)) != hipSuccess) {
        delete ctx;
        return fail(nullptr, TRHIP_ERR_HIP, "stream creation failed: %s", hipGetErrorString(e));
    }
    *out = ctx;
    return 0;
}
void trhip_shutdown(trhip_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->comm.comm) (void)rccl_api()->CommDestroy(ctx->comm.comm);
    for (auto& a : ctx->q)
        for (auto& b : a) release(b);
    for (auto& b : ctx->sq) release(b);
    for (auto& b : ctx->scratch) release(b);
    release(ctx->hits);
    release(ctx->Lbuf);
    release(ctx->counters);
    release(ctx->sensor);
    release(ctx->table);
    release(ctx->film);
    release(ctx->overflow);
    release(ctx->pfilm);
    release(ctx->fdesc);
    release(ctx->film_side);
    release(ctx->cert_cold);
    release(ctx->cb_rc);
    for (auto& pp : ctx->pipes) {
        for (auto& a : pp.q)
            for (auto& b : a) release(b);
        for (auto& b : pp.sq) release(b);
        for (auto& b : pp.sq2) release(b);
        if (pp.ev_any2) (void)hipEventDestroy(pp.ev_any2);
        release(pp.hits);
        release(pp.counters);
        release(pp.overflow[0]);
        release(pp.overflow[1]);
        if (pp.ev_shade) (void)hipEventDestroy(pp.ev_shade);
        if (pp.ev_any) (void)hipEventDestroy(pp.ev_any);
        if (pp.ev_done) (void)hipEventDestroy(pp.ev_done);
        if (pp.st) (void)hipStreamDestroy(pp.st);
        if (pp.st2) (void)hipStreamDestroy(pp.st2);
    }
    release(ctx->wh_L);
    release(ctx->wh_parent);
    release(ctx->wh_coef);
    release(ctx->wh_pdf);
    release(ctx->wh_flags);
    release(ctx->occl);
    release(ctx->film_Lt);
    release(ctx->surv_list);
    release(ctx->surv_counts);
    release(ctx->poison);
    for (int k = 0; k < 2; ++k) {
        release(ctx->ov8[k]);
        release(ctx->fb_list[k]);
        release(ctx->fb_counts[k]);
    }
    for (auto& b : ctx->sp_vp) release(b);
    release(ctx->st_terms);
    release(ctx->st_tags[0]);
    release(ctx->st_tags[1]);
    release(ctx->st_frozen);
    release(ctx->st_counts);
    for (auto& a : ctx->st_list)
        for (auto& b : a)
            for (auto& c : b) release(c);
    for (DevBuf* b : {&ctx->sp_Ld, &ctx->sp_tau, &ctx->sp_radius, &ctx->sp_N, &ctx->sp_phi, &ctx->sp_M, &ctx->sp_counts, &ctx->sp_starts, &ctx->sp_entries, &ctx->sp_grid, &ctx->sp_ldist,
                      &ctx->sp_snap_M, &ctx->sp_snap_phi, &ctx->sp_snap_p, &ctx->sp_snap_beta, &ctx->sp_terms, &ctx->sp_rec[0], &ctx->sp_rec[1], &ctx->sp_rec[2],
                      &ctx->sp_rec_valid, &ctx->sp_raysnap})
        release(*b);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
    delete ctx;
}
const char* trhip_last_error(const trhip_ctx* ctx) { return ctx ? ctx->err.c_str() : g_init_error.c_str(); }

// ---- multi-GPU: RCCL over xGMI, one process per GPU (th_comm.h) ---------------------------------------------------------------------

int trhip_comm_unique_id(uint8_t* out_id128) {
    if (!out_id128) return fail(nullptr, TRHIP_ERR_INVALID, "null argument");
    static_assert(sizeof(ncclUniqueId) == TRHIP_UNIQUE_ID_BYTES, "ncclUniqueId size");
    RcclApi* api = rccl_api();
    if (!api->error.empty()) return fail(nullptr, TRHIP_ERR_UNSUPPORTED, "%s", api->error.c_str());
    ncclUniqueId id;
    NCCL_TRY(nullptr, api->GetUniqueId(&id));
    std::memcpy(out_id128, &id, sizeof id);
    return 0;
}
int trhip_comm_init(trhip_ctx* ctx, const uint8_t* id128, int rank, int n_ranks) {
    if (!ctx || !id128) return fail(ctx, TRHIP_ERR_INVALID, "null argument");
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(ctx, TRHIP_ERR_INVALID, "rank %d of %d", rank, n_ranks);
    if (ctx->comm.comm) return fail(ctx, TRHIP_ERR_INVALID, "the context already has a communicator (trhip_comm_destroy first)");
    RcclApi* api = rccl_api();
    if (!api->error.empty()) return fail(ctx, TRHIP_ERR_UNSUPPORTED, "%s", api->error.c_str());
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    NCCL_TRY(ctx, api->CommInitRank(&ctx->comm.comm, n_ranks, id, rank));
    ctx->comm.rank = rank;
    ctx->comm.n_ranks = n_ranks;
    return 0;
}
int trhip_comm_destroy(trhip_ctx* ctx) {
    if (!ctx) return fail(ctx, TRHIP_ERR_INVALID, "null argument");
    if (ctx->comm.comm) {
        (void)hipSetDevice(ctx->device);
        NCCL_TRY(ctx, rccl_api()->CommDestroy(ctx->comm.comm));
    }
    ctx->comm = Comm{};
    return 0;
}
int trhip_comm_rank(const trhip_ctx* ctx, int* rank, int* n_ranks) {
    if (!ctx) return TRHIP_ERR_INVALID;
    if (rank) *rank = ctx->comm.rank;
    if (n_ranks) *n_ranks = ctx->comm.n_ranks;
    return 0;
}
static int film_collective(trhip_ctx* ctx, void* d_xyzw, uint64_t n_pixels, int root, bool all) {
    if (!ctx || !d_xyzw) return fail(ctx, TRHIP_ERR_INVALID, "null argument");
    // Without a communicator the context is a single-process job and its film already is the sum: a no-op, by design (hosts call the
    // reduce unconditionally).  A host that runs SEVERAL processes must check trhip_comm_rank's n_ranks against its own world size before it
    // trusts the film — parallel.Job.reduce_film and bench.py do; the library cannot know about processes that never called trhip_comm_init.
    if (!ctx->comm.comm) return 0;
    if (!all && (root < 0 || root >= ctx->comm.n_ranks)) return fail(ctx, TRHIP_ERR_INVALID, "root %d of %d ranks", root, ctx->comm.n_ranks);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    RcclApi* api = rccl_api();
    // Film.pixels are additive: xyz sums and filter_weight_sum (film.jl:161-162, 190-191) — the sum over ranks is what
    // merge_film_tile! (film.jl:182-193) would have produced from all tiles of all samples, up to Float32 summation order
    if (all)
        NCCL_TRY(ctx, api->AllReduce(d_xyzw, d_xyzw, (size_t)n_pixels * 4, ncclFloat32, ncclSum, ctx->comm.comm, ctx->stream));
    else
        NCCL_TRY(ctx, api->Reduce(d_xyzw, d_xyzw, (size_t)n_pixels * 4, ncclFloat32, ncclSum, root, ctx->comm.comm, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}
int trhip_film_reduce(trhip_ctx* ctx, void* d_xyzw, uint64_t n_pixels, int root) { return film_collective(ctx, d_xyzw, n_pixels, root, false); }
int trhip_film_allreduce(trhip_ctx* ctx, void* d_xyzw, uint64_t n_pixels) { return film_collective(ctx, d_xyzw, n_pixels, 0, true); }

// Kernel families that lost their measurements (traversals 4 / 6 / 7, the leaf queue, the sorted one-leaf walk, the linear BVH builder) are compiled only with
// -DTRHIP_EXPERIMENTS (__graft_entry__.build_library(extra_flags=["-DTRHIP_EXPERIMENTS"], out_name="libtracehip_experiments.so")); the default library refuses their options.
#ifdef TRHIP_EXPERIMENTS
static constexpr bool kExperiments = true;
#else
static constexpr bool kExperiments = false;
#endif
static const char* const kNeedsExperiments = "needs the EXPERIMENTS build of the library (-DTRHIP_EXPERIMENTS): not in the default binary";
static bool experiments_only(const char* name, int64_t value) {
    if (!std::strcmp(name, "traversal")) return value == 4 || value == 6 || value == 7;
    if (!std::strcmp(name, "bvh_builder")) return value == 1;
    if (!std::strcmp(name, "leaf_sorted") || !std::strcmp(name, "leaf_queue")) return value != 0;
    return false;
}
int trhip_option_in_build(const char* name, int64_t value) { return (name && (kExperiments || !experiments_only(name, value))) ? 1 : 0; }
int trhip_set_option(trhip_ctx* ctx, const char* name, int64_t value) {
    if (!ctx || !name) return fail(ctx, TRHIP_ERR_INVALID, "null argument");
    if (!std::strcmp(name, "count_visits"))
        ctx->count_visits = value != 0;
    else if (!std::strcmp(name, "timing"))
        ctx->timing = value != 0;
    else if (!std::strcmp(name, "debug_trace_budget"))
        ctx->debug_trace_budget = (uint32_t)value;
    else if (!std::strcmp(name, "streaming"))
        ctx->streaming = value < 0 ? -1 : (value != 0 ? 1 : 0);
    else if (!std::strcmp(name, "stream_budget_shift"))
        ctx->stream_budget_shift = (uint32_t)std::max<int64_t>(0, std::min<int64_t>(31, value));
    else if (!std::strcmp(name, "stream_list_cap"))
        ctx->stream_list_cap = (uint32_t)std::max<int64_t>(0, value);
    else if (!std::strcmp(name, "stream_budget_min"))
        ctx->stream_budget_min = (uint32_t)std::max<int64_t>(1, value);
    else if (!std::strcmp(name, "sppm_batch"))
        ctx->sppm_batch = (uint64_t)std::max<int64_t>(0, value);
    else if (!std::strcmp(name, "bvh_builder")) {
        if (value == 1 && !kExperiments) return fail(ctx, TRHIP_ERR_UNSUPPORTED, "bvh_builder = 1 (the linear BVH) %s", kNeedsExperiments);
        ctx->bvh_builder = value < 0 ? -1 : (value > 4 ? 4 : (int)value);
    }
    else if (!std::strcmp(name, "hybrid"))
        ctx->hybrid = value != 0;
    else if (!std::strcmp(name, "film_transpose"))
        ctx->film_transpose = value != 0;
    else if (!std::strcmp(name, "band_tile_rows"))
        ctx->band_tile_rows = (int)std::max<int64_t>(0, value);
    else if (!std::strcmp(name, "compose_spheres"))
        ctx->compose_spheres = value < 0 ? -1 : (value != 0 ? 1 : 0);
    else if (!std::strcmp(name, "occluder_pretest"))
        ctx->occluder_pretest = value != 0;
    else if (!std::strcmp(name, "stream2_priority"))
        ctx->stream2_priority = (int)value;
    else if (!std::strcmp(name, "leaf_sorted")) {
        if (value != 0 && !kExperiments) return fail(ctx, TRHIP_ERR_UNSUPPORTED, "leaf_sorted %s", kNeedsExperiments);
        ctx->leaf_sorted = value != 0;
    }
    else if (!std::strcmp(name, "film_fused"))
        ctx->film_fused = value != 0;
    else if (!std::strcmp(name, "trace3_spec"))
        ctx->trace3_spec = value != 0;
    else if (!std::strcmp(name, "trace7_cheap"))
        ctx->trace7_cheap = value != 0;
    else if (!std::strcmp(name, "leaf_kernel"))
        ctx->leaf_kernel = value != 0;
    else if (!std::strcmp(name, "slab_margin_log2"))
        ctx->slab_margin_log2 = (int)std::max<int64_t>(0, std::min<int64_t>(20, value));
    else if (!std::strcmp(name, "tiny_scene_prims"))
        ctx->tiny_scene_prims = (uint32_t)std::max<int64_t>(0, std::min<int64_t>(255, value));
    else if (!std::strcmp(name, "film_block"))
        ctx->film_block = (int)std::max<int64_t>(0, std::min<int64_t>(13, value));
    else if (!std::strcmp(name, "film_relayout"))
        ctx->film_relayout = value != 0;
    else if (!std::strcmp(name, "any_on_accelerator"))
        ctx->any_on_accelerator = value < 0 ? -1 : (value != 0 ? 1 : 0);
    else if (!std::strcmp(name, "leaf_queue")) {
        if (value != 0 && !kExperiments) return fail(ctx, TRHIP_ERR_UNSUPPORTED, "leaf_queue %s", kNeedsExperiments);
        ctx->leaf_queue = value != 0;
    }
    else if (!std::strcmp(name, "wide4"))
        ctx->wide4 = value != 0;
    else if (!std::strcmp(name, "node_layout")) {
        if (value < 0 || value > 1) return fail(ctx, TRHIP_ERR_INVALID, "node_layout: 0 (depth-first) or 1 (sibling pairs per 128-byte line)");
        ctx->node_layout = (int)value;
    } else if (!std::strcmp(name, "film_swizzle"))
        ctx->film_swizzle = value != 0;
    else if (!std::strcmp(name, "film_tiled"))
        ctx->film_tiled = value != 0;
    else if (!std::strcmp(name, "pipelines")) {
        if (value < 1 || value > kMaxPipes) return fail(ctx, TRHIP_ERR_INVALID, "pipelines must be in 1..%d", kMaxPipes);
        ctx->pipelines = (int)value;
    } else if (!std::strcmp(name, "overlap"))
        ctx->overlap = value != 0;
    else if (!std::strcmp(name, "traversal")) {
        if (value < 1 || value > 7 || value == 5) return fail(ctx, TRHIP_ERR_INVALID, "traversal must be 1, 2, 3, 4, 6 or 7");
        if ((value == 4 || value == 6 || value == 7) && !kExperiments) return fail(ctx, TRHIP_ERR_UNSUPPORTED, "traversal %d %s", (int)value, kNeedsExperiments);
        ctx->traversal = (int)value;
    } else if (!std::strcmp(name, "batch_paths")) {
        if (value < 0) return fail(ctx, TRHIP_ERR_INVALID, "batch_paths must be >= 0 (0 = auto)");
        ctx->batch_paths = (uint64_t)value;
    } else
        return fail(ctx, TRHIP_ERR_INVALID, "unknown option %s", name);
    return 0;
}

}  // extern "C"

// Host utility: the deterministic elementary functions of include/trace_detmath.h for hosts that cannot include a C
// header (the Python mirror needs tan() for perspective(), transformations.jl:128).  fn: 0 sin 1 cos 2 tan 3 atan2(y,x)
// 4 acos 5 log.  This is specification math evaluated on the host, not a fallback of any device path.
namespace {
TH_HD float detmath_eval(int fn, float x, float y) {
    switch (fn) {
    case 0: return tm_sinf(x);
    case 1: return tm_cosf(x);
    case 2: return tm_tanf(x);
    case 3: return tm_atan2f(y, x);
    case 4: return tm_acosf(x);
    case 5: return tm_logf(x);
    default: {  // 6 / 7: tm_sincosf, sin part / cos part
        float sn, cs;
        tm_sincosf(x, &sn, &cs);
        return fn == 6 ? sn : cs;
    }
    }
}
__global__ void k_detmath(int fn, const float* __restrict__ x, const float* __restrict__ y, uint64_t n, float* __restrict__ out) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) out[i] = detmath_eval(fn, x[i], y ? y[i] : 0.0f);
}
}  // namespace
// The same functions evaluated BY THE KERNELS' COMPILER on the GPU: the parity contract needs device == host bit for bit
// (tests/test_gpu_edge_cases.py compares this with trhip_detmath_f32).
extern "C" int trhip_detmath_f32_device(trhip_ctx* ctx, int fn, const float* x, const float* y, uint64_t n, float* out) {
    if (!ctx || !x || !out || (fn == 3 && !y) || fn < 0 || fn > 7) return fail(ctx, TRHIP_ERR_INVALID, "bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (int rc = upload(ctx, ctx->scratch[0], x, n * sizeof(float))) return rc;
    if (y)
        if (int rc = upload(ctx, ctx->scratch[1], y, n * sizeof(float))) return rc;
    if (int rc = ensure(ctx, ctx->scratch[2], n * sizeof(float))) return rc;
    if (n) hipLaunchKernelGGL(k_detmath, dim3(grid_for(ctx, n, 4)), dim3(kBlock), 0, ctx->stream, fn, (const float*)ctx->scratch[0].p, y ? (const float*)ctx->scratch[1].p : nullptr, n, (float*)ctx->scratch[2].p);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (n) HIP_TRY(ctx, hipMemcpy(out, ctx->scratch[2].p, n * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}
extern "C" int trhip_detmath_f32(int fn, const float* x, const float* y, uint64_t n, float* out) {
    if (!x || !out || (fn == 3 && !y)) return TRHIP_ERR_INVALID;
    for (uint64_t i = 0; i < n; ++i) {
        switch (fn) {
        case 0: out[i] = tm_sinf(x[i]); break;
        case 1: out[i] = tm_cosf(x[i]); break;
        case 2: out[i] = tm_tanf(x[i]); break;
        case 3: out[i] = tm_atan2f(y[i], x[i]); break;
        case 4: out[i] = tm_acosf(x[i]); break;
        case 5: out[i] = tm_logf(x[i]); break;
        case 6:
        case 7: {  // tm_sincosf: sin part / cos part
            float sn, cs;
            tm_sincosf(x[i], &sn, &cs);
            out[i] = fn == 6 ? sn : cs;
            break;
        }
        default: return TRHIP_ERR_INVALID;
        }
    }
    return 0;
}
