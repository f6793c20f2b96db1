// th_host.h — what the translation units of libtracehip.so share on the host side: the context and scene objects behind the opaque
// handles of include/tracehip.h, error / buffer helpers, and the functions one unit calls in another.  Units (each compiled on its own
// and linked into the one shared object): tu_api.hip (context, options, communicator), tu_scene.hip (scene flattening, commit, upload),
// tu_lbvh.hip (BVH build on the device), tu_trace.hip / tu_trace3.hip / tu_trace8.hip (traversal launches and entry points),
// tu_path.hip (PathIntegrator frames, film), tu_whitted.hip, tu_sppm.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#pragma GCC visibility push(default)
#include "../../include/tracehip.h"
#pragma GCC visibility pop
#include "th_bvh.h"
#include "th_kernels.h"
#include "th_trace2.h"
#include "th_trace8.h"  // (FallbackList, the 8-wide view's types: the kernels themselves are only instantiated by an EXPERIMENTS build)
#ifdef TRHIP_EXPERIMENTS
#include "th_trace4.h"
#include "th_trace7.h"
#endif
#include "th_trace3c.h"
#include "th_comm.h"

using namespace th;

// ---- context ------------------------------------------------------------------------------------------------------------------
extern thread_local std::string g_init_error;  // tu_api.hip: the message of a failed trhip_init

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

constexpr int kMaxPipes = 8;
// One wavefront pipeline: its own queues, counters and stream pair.  Several batches of one frame run concurrently on
// different pipelines so that the long single-ray tail of one batch's traversal launch overlaps the bulk of another's.
struct Pipe {
    hipStream_t st = nullptr, st2 = nullptr;
    hipEvent_t ev_shade = nullptr, ev_any = nullptr, ev_any2 = nullptr, ev_done = nullptr;
    DevBuf q[2][3], sq[3], sq2[3], hits, counters, overflow[2];  // sq / sq2: the shadow queues of odd / even depths (any(d) may still run while shade(d+1) fills the other)
};

struct Timer;
struct trhip_ctx {
    int device = 0;
    Timer* active_timer = nullptr;  // the Timer of the render call in progress (hybrid launches mark the hand-over between their two walks in it)
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;  // shadow rays of depth d overlap with the closest-hit rays of depth d+1
    std::string err;
    int num_cu = 256;
    // options
    bool count_visits = false;
    bool timing = true;
    uint64_t batch_paths = 0;  // 0 = as many whole sample passes as fit in free HBM (fewer launches, fewer traversal tails)
    int pipelines = 1;    // concurrent wavefront batches (each on its own stream pair); measured: no gain, every batch pays every tail
    Pipe pipes[kMaxPipes];
    uint32_t debug_trace_budget = 0;  // DIAGNOSTIC: k_trace2 abandons rays after this many node fetches (results wrong; measures bulk vs tail)
    double bvh_device_ms = 0.0;  // device time of the last bvh_builder 3 build
    int bvh_builder = -1;  // BVHAccel construction: 0 = binned SAH on the host (th_bvh.h), 1 = linear BVH on the device (th_lbvh.h), 3 = the host builder's
                           // binned SAH on the device (th_sahb.h: the same tree), 2 = the reference's own construction node for node (th_bvh_ref.h:
                           // the tree Trace.jl builds, hence its tie-breaks), -1 = automatic: builder 3 from 64 Ki primitives on, builder 0 below and
                           // for the scenes builder 3 hands back.  Measured (r3): building the tree 18 ms (1 M triangles) / 53 ms (10 M) on the device
                           // against ~0.17 s / ~1.5 s on the host, commit 0.39 -> 0.26 s and 3.67 -> 2.37 s; the LBVH builds a 25-35 % costlier tree
    bool film_transpose = false;     // film pass on pixel-group-major copies of p_film / L (option "film_transpose"; launch_film)
    bool occluder_pretest = true;    // any-hit rays test the scene's largest triangles before the walk (option "occluder_pretest")
    int stream2_priority = -1;       // shadow-ray stream: 1 highest priority, -1 lowest, 0 the default level (option "stream2_priority", read when the streams are created)
    bool leaf_kernel = true;         // one-leaf scenes run k_trace_leaf instead of k_trace2 (option "leaf_kernel", for A/B)
    int slab_margin_log2 = 14;       // k_trace2 / k_trace3 add the slab clauses the reference's box test lost, on boxes grown by 2^-this x the ray's reach
                                     // (th_trace2.h, slab_test2); 0 = the reference's loose test alone (its exact visit set)
    int band_tile_rows = 0;          // DIAGNOSTIC / tests: render frames in bands of this many tile rows (0 = one band unless the samples do not fit in HBM)
    uint32_t tiny_scene_prims = 16;  // scenes of at most this many primitives get a single-leaf BVH (th_bvh.h); 0 = always build the hierarchy
    int film_block = 5;  // film gather: >= 4 (filter radius <= 1; wider filters run 2): from a 32-bit splat descriptor in the radiance record's .w lane, a thread owning
                         // 4: 1 x 4, 5 (default): 2 x 4, 6: 4 x 4, 7: 2 x 2, 8: 4 x 2, 9: 8 x 4, 10: 1 x 8, 11: 1 x 16, 12: 1 x 2, 13: 1 x 1 film pixels (th_kernels.h,
                         // k_film_gather_packed; measured at 1024^2 x 256 spp: 19.0 / 17.1 / 24.2 / 25.9 / 30.8 / 40.4 / 17.1 / - / 22.5 / 33.7 ms against 25.0 for 2);
                         // 0 = one film pixel per thread, 1 = 2 x 2 pixels per thread, 2 = TH_FILM_BX x TH_FILM_BY = 1 x 4, all three recomputing a
                         // sample's pixel range and table indices per thread; 3 = 1 x 4 from per-sample splat descriptors (k_film_descriptors): measured SLOWER
                         // (1024^2, 256 spp: 29.0 ms against 24.3 ms: the 16-byte descriptor doubles the gather's loads and the arithmetic it saves was hidden)
    bool film_relayout = true;  // packed film pass: gather from a pixel-group-major copy of the radiance records (k_film_pack_transpose) instead of the integrators' sample-major order
    bool warned_idle_accelerator = false;  // (tu_path.hip: the one-time stderr note)
    bool warned_fallback_cliff = false;    // … and the one about a frame whose certified walk handed back more than a fifth of its rays
    double last_fallback_share = 0.0;      // fallback rays / closest-hit rays of the last frame that reported statistics (trhip_accelerator_note)
    int any_on_accelerator = -1;  // hybrid mode: any-hit rays without a zero direction component walk the library's tree (TraceOut::zero_mode): 1 always, 0 never, -1 where the
                                  // integrator asks for it (TraceOut::any_acc_hint: SPPM).  Option "any_on_accelerator"
    bool wide4 = true;  // hybrid mode: the accelerator is also laid out four children wide and the certified walk runs on that (th_trace3c4.h); option "wide4", read at commit and at launch
    bool leaf_queue = false;  // hybrid mode: the certified walk queues the leaves it reaches and tests them 64 at a time with whichever lanes (th_trace3d.h, option "leaf_queue")
    int node_layout = 0;  // children-in-parent nodes: 0 depth-first, 1 the two interior children of a node in one aligned 128-byte line (option "node_layout", read at commit; tu_scene.hip)
    bool film_swizzle = false;  // packed film gather: XCD x owns the x-th contiguous eighth of the workgroups (option "film_swizzle"; measured: no effect, th_kernels.h)
    bool film_tiled = false;  // LDS-staged film gather (k_film_gather_tiled): bit-identical, measured 2.7x SLOWER than k_film_gather (11 % lane use), kept as an option
    bool overlap = true;   // shadow rays of depth d on a second stream beside the closest-hit rays of depth d+1 (option "overlap").  On again since round 5: with the four-wide
                           // certified walk no longer saturating VALU issue the shadow kernels fill its gaps and tails — 256 spp, off / on: S-mesh 325.5 / 314.5 ms, S-blob 234.8 / 222.3,
                           // S-cornell 160.4 / 157.1, 10.5 M triangles 376.4 / 367.7.  (Rounds 2-4 measured it neutral — S-cornell 158.2 / 157.9, S-mesh 399 / 393 — and kept it off.)
    int compose_spheres = -1;  // commit: spheres as a chain of leaves above the triangles' subtree, what k_trace8 needs of a scene with spheres
                               // (option "compose_spheres": 1 / 0 = one SAH tree over everything / -1 = when "traversal" is 4 at commit time)
    int traversal = 3;  // 1 = literal accel/bvh.jl loop, 2 = children-in-parent nodes + per-lane ray replacement, 3 = 2 with leaves postponed (while-while),
                        // 4 = 8-wide quantised nodes in the binary walk's order (th_trace8.h; scenes / rays it cannot take run 3), 6 = 3 with two rays per lane (th_trace4.h),
                        // 7 = closest-hit rays front to back with tie detection, flagged rays re-traced by 3 (th_trace7.h); any-hit rays as 3
    bool leaf_sorted = false;  // one-leaf scenes: rays grouped by the primitives they can hit before the leaf is walked (th_leaf2.h, option "leaf_sorted"); exact, measured SLOWER
                               // (S-cornell closest-hit 49.2 -> 60.2 ms, any-hit 15.9 -> 30.8: twelve candidate tests cost as much as the exact tests they save), off
    bool film_fused = true;    // the path integrator's k_raygen writes the radiance records in the film pass's layout with their splat descriptors (no memset, no pack pass; option "film_fused")
    uint32_t last_L_layout = 0, last_L_npix = 1, last_L_spp = 1;  // how Lbuf is laid out after the last render (trhip_last_sample_radiance)
    bool trace3_spec = true;   // k_trace3 (closest-hit): lanes park the leaf they reach and go on descending (th_trace2.h, TH_TRACE3_SPEC); 0 = wait for the leaf phase
    bool hybrid = true;        // scenes committed with both trees (bvh_builder 4 / -1): closest-hit rays walk the ACCELERATOR with the order-independence certificate of
                               // th_trace3c.h and only the flagged ones the canonical tree (option "hybrid"; 0 = every ray walks the canonical tree: same answers, slower)
    bool trace7_cheap = true;  // k_trace7: conservative fma slab test on interior boxes, the reference's exact test once per leaf (th_trace7.h); 0 = exact test on every box
    // workspace (grown on demand, reused across calls)
    DevBuf q[2][3], sq[3], hits, Lbuf, pfilm, counters, sensor, table, film, scratch[4], overflow, wh_L, wh_parent, wh_coef, wh_pdf, wh_flags, occl, film_Lt, surv_list, surv_counts;
    uint64_t last_L_count = 0;  // float4 entries valid in Lbuf
    // SPPM state (th_sppm.h): per film pixel, kept after trhip_render_sppm for trhip_sppm_state
    DevBuf sp_vp[7], sp_Ld, sp_tau, sp_radius, sp_N, sp_phi, sp_M, sp_counts, sp_starts, sp_entries, sp_grid, sp_ldist, sp_snap_M, sp_snap_phi, sp_snap_p, sp_snap_beta;
    DevBuf sp_terms, sp_rec[3], sp_rec_valid, sp_raysnap;  // sp_raysnap: {closest_total, shadow_total} after every batch's camera pass and photon pass
    // streaming wavefront (render_stream_impl)
    DevBuf st_terms, st_tags[2], st_frozen, st_counts, st_list[2][2][7];  // [closest|any][ping-pong][o, d, b, trav, st, depth, stack]
    int streaming = 0;             // PathIntegrator on scenes with a real hierarchy: suspend / resume stragglers.  1 = always, 0 = never (classic
                                   // per-depth launches), -1 = automatic: when the frame has at most 96 camera samples per primitive, which is where the
                                   // traversal tails dominate (measured, 1 M triangles: 16 spp 1170 -> 716 ms, 64 spp 1700 -> 1443, 128 spp 2288 vs
                                   // 2412, 256 spp 3742 vs 3823; 10 M triangles, depth 16: 32 spp 6075 -> 2465 ms, 128 spp 8364 -> 4913)
    uint32_t stream_budget_shift = 12;  // budget = max(stream_budget_min, fresh rays of the round >> shift)
    uint32_t stream_list_cap = 0;       // suspended-ray list capacity (0 = max(65536, paths / 128)); tests shrink it
    uint32_t stream_budget_min = 2048;  // interior fetches before a ray may be suspended (tests lower it to force suspensions)
    uint64_t sppm_batch = 0;  // SPPM iterations per wavefront batch (0 = from free HBM, at most 128)
    uint32_t sp_pixels = 0;
    int64_t sp_photons = 0;
    DevBuf film_side;  // packed film pass: counter + the full descriptors of the samples whose range does not fit 30 bits (th_kernels.h, FilmSideTable)
    DevBuf fdesc;   // film_block 3: one SplatDesc (16 B) per camera sample of the band (th_kernels.h, k_film_descriptors)
    DevBuf poison;  // one byte per camera sample of the band: ShadeStream::poison
    DevBuf cert_cold;  // k_trace3c's CertCold (th_trace3c.h)
    DevBuf cb_rc;      // one word: a host callback's return code, max-reduced over the ranks of a job (tu_sppm.hip)
    DevBuf ov8[2], fb_list[2], fb_counts[2];  // k_trace8: global stack levels, fallback lists + their counters / work cursors ([closest | any])
    Comm comm;  // multi-GPU job this context belongs to (trhip_comm_init); n_ranks == 1 without one
};

struct HostPrim {
    uint32_t kind;       // 0 triangle, 1 sphere
    float v[9];          // triangle vertices (world)
    float n[9];          // vertex normals
    uint32_t meta;       // material | flags
    uint32_t sphere_id;  // for spheres
};

struct trhip_scene {
    trhip_ctx* ctx = nullptr;
    std::vector<MaterialRec> materials;
    std::vector<HostPrim> prims;  // caller order
    // the two optional mesh arrays (shapes/triangle_mesh.jl:11-14), beside the primitives and only when some mesh carries them (no scene of the reference does):
    // prim_tan[9 i ..] = primitive i's vertex tangents (PRIM_HAS_TANGENTS), prim_uv[7 i ..] = its corner (u, v)s and a "has" flag; empty = none
    std::vector<float> prim_tan, prim_uv;
    bool has_materialless_prim = false;  // set at commit: some GeometricPrimitive has no material (the integrators refuse such a scene; the trace entry points accept it)
    std::vector<SphereRec> spheres;
    std::vector<HostAABB> sphere_bounds;
    std::vector<LightRec> lights;
    FlatBVH bvh;
    bool committed = false;
    DevBuf d_leaf_boxes;  // one-leaf scenes: the boxes of the leaf's triangles in slot order (th_leaf2.h)
    DevBuf d_nodes, d_prims, d_nrm, d_tan, d_shade, d_spheres, d_materials, d_lights, d_wnodes;
    DeviceScene dev{};
    WideScene wide{};
    DevBuf d_occ_slots, d_occ_boxes, d_w8nodes, d_w8tris, d_leaf_order;
    Wide8Scene w8{};              // the 8-wide view of the triangles' subtree (th_wide8.h / th_trace8.h)
    uint32_t w8_nodes = 0, w8_depth = 0;
    uint32_t n_occluders = 0;     // the scene's largest triangles, tested first by any-hit rays (th_trace2.h, k_any_occluders)
    bool partial_spheres = false;  // some sphere is clipped (z range or ϕ_max): traversal kernels with the general sphere test
    bool wide_ok = false;
    bool w8_ok = false;            // the 8-wide view exists (th_trace8.h)
    bool literal_only = false;     // a caller-supplied BVH whose boxes do not nest (trhip_scene_set_bvh): literal kernels only
    // ---- hybrid mode (th_trace3c.h): `bvh` above is the CANONICAL tree (the reference's construction, or the host's own tree) — slots, shading records, the inspection API
    // and the answers are its; `acc` is the library's tree over the same primitives, which most rays walk instead
    FlatBVH acc;                   // order[k] = caller primitive of accelerator slot k; empty without an accelerator
    bool hybrid_ok = false;        // the accelerator exists and its leaves carry the canonical leaves' boxes bit for bit
    std::string bvh_note;          // why a default commit ended with one tree (trhip_scene_bvh_note)
    int bvh_mode = 0;              // what trhip_scene_commit / trhip_scene_set_bvh built: 0 the library's tree alone, 1 the canonical (reference / host) tree alone, 2 both
    DevBuf d_acc_w4nodes;  // the accelerator four children wide (th_trace3c4.h)
    DevBuf d_acc_wnodes, d_acc_prims, d_slot_boxes, d_sphere_boxes, d_sphere_slots, d_sphere_cert, d_acc_leaf_order;
    WideScene wide_acc{};
    DeviceScene dev_acc{};         // dev with the accelerator's primitive records
    CertScene cert{};
};

inline int fail(trhip_ctx* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx)
        ctx->err = buf;
    else
        g_init_error = buf;
    return code;
}
#define HIP_TRY(ctx, expr)                                                                                       \
    do {                                                                                                         \
        hipError_t e_ = (expr);                                                                                  \
        if (e_ != hipSuccess) return fail(ctx, TRHIP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

#define NCCL_TRY(ctx, expr)                                                                                                  \
    do {                                                                                                                     \
        ncclResult_t r_ = (expr);                                                                                            \
        if (r_ != ncclSuccess) return fail(ctx, TRHIP_ERR_HIP, "%s failed: %s", #expr, rccl_api()->GetErrorString ? rccl_api()->GetErrorString(r_) : "RCCL error"); \
    } while (0)

inline int ensure(trhip_ctx* ctx, DevBuf& b, size_t bytes) {
    if (b.bytes >= bytes && b.p) return 0;
    if (b.p) HIP_TRY(ctx, hipFree(b.p));
    b.p = nullptr;
    b.bytes = 0;
    if (bytes == 0) bytes = 16;
    HIP_TRY(ctx, hipMalloc(&b.p, bytes));
    b.bytes = bytes;
    return 0;
}
inline void release(DevBuf& b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
}
inline int upload(trhip_ctx* ctx, DevBuf& b, const void* src, size_t bytes) {
    if (int rc = ensure(ctx, b, bytes)) return rc;
    if (bytes) HIP_TRY(ctx, hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
    return 0;
}
// a staging array that is NOT zero-filled on allocation (std::vector value-initialises: 1.5 GB of single-threaded memset for a 10 M-triangle scene's records, all of which the
// fill loops overwrite)
template <class T>
struct RawArray {
    std::unique_ptr<T[]> p;
    size_t n = 0;
    explicit RawArray(size_t count) : p(new T[count ? count : 1]), n(count) {}
    T* data() { return p.get(); }
    const T* data() const { return p.get(); }
    size_t size() const { return n; }
    T& operator[](size_t i) { return p[i]; }
    const T& operator[](size_t i) const { return p[i]; }
};
// f(begin, end) over [0, n) on the host's cores (scene commit's loops over millions of nodes / primitives); ranges are disjoint and contiguous
template <class F>
inline void parallel_for(size_t n, F&& f, size_t grain = size_t(1) << 15) {
    const size_t want = n / std::max<size_t>(1, grain);
    const size_t nt = std::min<size_t>({want, (size_t)std::max(1u, std::thread::hardware_concurrency()), (size_t)32});
    if (nt <= 1) {
        f((size_t)0, n);
        return;
    }
    std::vector<std::thread> th;
    th.reserve(nt - 1);
    const size_t step = (n + nt - 1) / nt;
    for (size_t t = 1; t < nt; ++t) th.emplace_back([&f, t, step, n] { f(std::min(n, t * step), std::min(n, (t + 1) * step)); });
    f((size_t)0, std::min(n, step));
    for (auto& x : th) x.join();
}
// do camera rays start far outside the scene (more than 4 scene extents away)?  Then the hybrid walk's launch over them takes the per-axis form of its cull bound
// (TraceOut::far_hint, th_trace3c.h): the scalar margin scales with the reach D of the rays' arithmetic, which is what such a distance inflates
inline bool far_camera(const trhip_scene* sc, const trhip_sensor* sn) {
    if (!sc->hybrid_ok || !sn) return false;
    const float* rb = sc->wide_acc.root_box;
    const float o[3] = {sn->camera_to_world[3], sn->camera_to_world[7], sn->camera_to_world[11]};
    float reach = 0.0f, extent = 0.0f;
    for (int a = 0; a < 3; ++a) {
        reach = std::fmax(reach, std::fmax(std::fabs(rb[a] - o[a]), std::fabs(rb[3 + a] - o[a])));
        extent = std::fmax(extent, rb[3 + a] - rb[a]);
    }
    return reach > 4.0f * extent;
}
inline int grid_for(const trhip_ctx* ctx, uint64_t n, int blocks_per_cu) {
    const uint64_t need = (n + kBlock - 1) / kBlock;
    const uint64_t cap = (uint64_t)ctx->num_cu * blocks_per_cu;
    return (int)std::max<uint64_t>(1, std::min(need, cap));
}

// TRHIP_FRAME_TIMING=1: host-side time between the stages of a render call, one line per stage on stderr
struct HostClock {
    bool on = std::getenv("TRHIP_FRAME_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void tick(const char* what) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[frame] %-32s %8.1f us\n", what, std::chrono::duration<double, std::micro>(now - t).count());
        t = now;
    }
};
struct Timer {
    trhip_ctx* ctx;
    bool on;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[8];  // 0 raygen, 1 closest, 2 shade, 3 any, 4 film; 5-7: sub-classes of an integrator (tu_sppm.hip)
    std::vector<std::pair<size_t, hipEvent_t>> marks;       // hybrid mode: (index into ev[1], event between the certified walk and the fallback walk of that closest-hit launch)
    explicit Timer(trhip_ctx* c, bool enable) : ctx(c), on(enable) { c->active_timer = this; }
    ~Timer() {
        if (ctx->active_timer == this) ctx->active_timer = nullptr;
        for (auto& v : ev)
            for (auto& p : v) {
                (void)hipEventDestroy(p.first);
                (void)hipEventDestroy(p.second);
            }
        for (auto& m : marks) (void)hipEventDestroy(m.second);
    }
    void mark_fallback(hipStream_t st) {  // called by launch_trace between the two walks of a hybrid closest-hit launch (inside begin(1) .. end(1))
        if (!on || ev[1].empty()) return;
        hipEvent_t e;
        (void)hipEventCreate(&e);
        (void)hipEventRecord(e, st);
        marks.push_back({ev[1].size() - 1, e});
    }
    double fallback_total(uint32_t* launches) {  // time from each mark to the end of its launch
        double ms = 0;
        for (auto& m : marks) {
            float t = 0;
            if (m.first < ev[1].size()) (void)hipEventElapsedTime(&t, m.second, ev[1][m.first].second);
            ms += t;
        }
        *launches = (uint32_t)marks.size();
        return ms;
    }
    void begin(int cls, hipStream_t st) {
        if (!on) return;
        hipEvent_t a, b;
        (void)hipEventCreate(&a);
        (void)hipEventCreate(&b);
        (void)hipEventRecord(a, st);
        ev[cls].push_back({a, b});
    }
    void end(int cls, hipStream_t st) {
        if (!on) return;
        (void)hipEventRecord(ev[cls].back().second, st);
    }
    double total(int cls, uint32_t* launches) {
        double ms = 0;
        for (auto& p : ev[cls]) {
            float t = 0;
            (void)hipEventElapsedTime(&t, p.first, p.second);
            ms += t;
        }
        *launches = (uint32_t)ev[cls].size();
        return ms;
    }
};

// ---- functions one unit calls in another ---------------------------------------------------------------------------------------------
// tu_scene.hip
int upload_scene(trhip_scene* s);
int upload_accelerator(trhip_scene* s);
// tu_lbvh.hip
int build_bvh_device(trhip_ctx* ctx, const std::vector<HostAABB>& pb, FlatBVH& out);
int build_bvh_device_sah(trhip_ctx* ctx, const std::vector<HostAABB>& pb, int max_node_prims, bool split_coincident, FlatBVH& out, double* ms_device);
// tu_trace.hip
int trace_grid(const trhip_ctx* ctx);
int ensure_overflow(trhip_ctx* ctx);
WideScene wide_view(const trhip_ctx* ctx, const trhip_scene* sc);
void traversal_info(const trhip_ctx* ctx, const trhip_scene* sc, uint32_t* trav, uint32_t* node_bytes);
void launch_trace(trhip_ctx* ctx, hipStream_t st, const trhip_scene* sc, bool any, SegQueue q, const float4* ro, const float4* rd, const float* tmax, TraceOut out, uint32_t* work_cursors,
                  Counters* ctr, void* overflow_slab = nullptr);
void launch_trace2_stream(trhip_ctx* ctx, hipStream_t st, const trhip_scene* sc, bool any, SegQueue q, const float4* ro, const float4* rd, TraceOut out, uint32_t* work_cursors, void* overflow_slab,
                          Counters* ctr, const StreamCtl& sx);
// tu_trace3.hip / tu_trace8.hip: the kernel families launch_trace picks from
void launch_trace3(trhip_ctx* ctx, hipStream_t st, const trhip_scene* sc, bool any, bool cnt, bool full_only, bool big, const SegQueue& q, const float4* ro, const float4* rd, const float* tmax,
                   const TraceOut& out, uint32_t* work_cursors, uint2* ov, Counters* ctr, bool on_accelerator = false);
void launch_trace4(trhip_ctx* ctx, hipStream_t st, const trhip_scene* sc, bool any, bool cnt, bool full_only, const SegQueue& q, const float4* ro, const float4* rd, const float* tmax,
                   const TraceOut& out, uint32_t* work_cursors, uint2* ov, Counters* ctr);
// tu_trace3c.hip: the hybrid mode's certified walks on the accelerator tree (th_trace3c.h)
bool hybrid_active(const trhip_ctx* ctx, const trhip_scene* sc);
const char* hybrid_idle_reason(const trhip_ctx* ctx, const trhip_scene* sc);
WideScene wide_view_acc(const trhip_ctx* ctx, const trhip_scene* sc);
void launch_trace3c(trhip_ctx* ctx, hipStream_t st, const trhip_scene* sc, bool cnt, bool full_only, bool big, const SegQueue& q, const float4* ro, const float4* rd, const float* tmax,
                    const TraceOut& out, uint32_t* work_cursors, uint2* ov, Counters* ctr, const FallbackList& fb);
void launch_leaf_c(trhip_ctx* ctx, hipStream_t st, const trhip_scene* sc, bool any, bool cnt, bool full_only, const SegQueue& q, const float4* ro, const float4* rd, const float* tmax,
                   const TraceOut& out, Counters* ctr, const FallbackList& fb);
void launch_trace7(trhip_ctx* ctx, hipStream_t st, const trhip_scene* sc, bool cnt, bool full_only, bool big, const SegQueue& q, const float4* ro, const float4* rd, const float* tmax,
                   const TraceOut& out, uint32_t* work_cursors, uint2* ov, Counters* ctr, const FallbackList& fb);
void launch_trace8(trhip_ctx* ctx, hipStream_t st, const trhip_scene* sc, bool any, bool cnt, bool full_only, const Wide8Scene& w8, const SegQueue& q, const float4* ro, const float4* rd,
                   const float* tmax, const TraceOut& out, uint32_t* work_cursors, uint32_t* ov8, Counters* ctr, const FallbackList& fb);
// tu_path.hip
void derive_sensor(const trhip_sensor* sn, DeviceSensor& d);
bool film_uses_desc(const trhip_ctx* ctx, const DeviceSensor& ds);
bool film_uses_packed(const trhip_ctx* ctx, const DeviceSensor& ds);
int ensure_film_samples(trhip_ctx* ctx, const DeviceSensor& ds, uint64_t total_slots);
void launch_film(trhip_ctx* ctx, hipStream_t st, const DeviceSensor& ds, const DeviceSensor* dsp, const float4* L, uint64_t total_slots, uint32_t spp, uint64_t seed, uint32_t sample_offset,
                 float4* d_film, bool fused);
// tu_whitted.hip
int render_whitted_impl(trhip_ctx* ctx, const trhip_scene* scene, const DeviceSensor& ds, const trhip_sensor* sensor, uint32_t spp, int max_depth, uint64_t seed, uint32_t sample_offset,
                        void* d_film, trhip_stats* stats, double* ms_total);
