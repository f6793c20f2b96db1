// tracehip.hip — host side of libtracehip.so: the C ABI of include/tracehip.h, scene flattening into HBM, BVH build,
// and the wavefront launch loop.  gfx950 only; there is no CPU fallback: every entry point that computes needs a GPU.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/tracehip.h"
#include "th_bvh.h"
#include "th_kernels.h"
#include "th_trace2.h"
#include "th_trace8.h"
#include "th_trace4.h"
#include "th_whitted.h"
#include "th_sppm.h"
#include "th_lbvh.h"
#include "th_comm.h"

using namespace th;

// ---- context ------------------------------------------------------------------------------------------------------------------
namespace {
thread_local std::string g_init_error;

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};
}  // namespace

constexpr int kMaxPipes = 8;
// One wavefront pipeline: its own queues, counters and stream pair.  Several batches of one frame run concurrently on
// different pipelines so that the long single-ray tail of one batch's traversal launch overlaps the bulk of another's.
struct Pipe {
    hipStream_t st = nullptr, st2 = nullptr;
    hipEvent_t ev_shade = nullptr, ev_any = nullptr, ev_any2 = nullptr, ev_done = nullptr;
    DevBuf q[2][3], sq[3], sq2[3], hits, counters, overflow[2];  // sq / sq2: the shadow queues of odd / even depths (any(d) may still run while shade(d+1) fills the other)
};

struct trhip_ctx {
    int device = 0;
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
    int bvh_builder = -1;  // BVHAccel construction: 0 = binned SAH on the host (th_bvh.h), 1 = linear BVH on the device (th_lbvh.h),
                           // -1 = automatic: the device builder above 16 Mi primitives.  Measured: commit 0.72 -> 0.13 s (1 M triangles), 8.0 -> 1.5 s
                           // (10 M); the LBVH costs 25-35 % more node visits per ray (frame +4 % at 1 M / 64 spp, +37 % at 10 M / 16 spp)
    bool film_transpose = false;     // film pass on pixel-group-major copies of p_film / L (option "film_transpose"; launch_film)
    bool occluder_pretest = true;    // any-hit rays test the scene's largest triangles before the walk (option "occluder_pretest")
    int stream2_priority = -1;       // shadow-ray stream: 1 highest priority, -1 lowest, 0 the default level (option "stream2_priority", read when the streams are created)
    bool leaf_kernel = true;         // one-leaf scenes run k_trace_leaf instead of k_trace2 (option "leaf_kernel", for A/B)
    int slab_margin_log2 = 14;       // k_trace2 / k_trace3 add the slab clauses the reference's box test lost, on boxes grown by 2^-this x the ray's reach
                                     // (th_trace2.h, slab_test2); 0 = the reference's loose test alone (its exact visit set)
    int band_tile_rows = 0;          // DIAGNOSTIC / tests: render frames in bands of this many tile rows (0 = one band unless the samples do not fit in HBM)
    uint32_t tiny_scene_prims = 16;  // scenes of at most this many primitives get a single-leaf BVH (th_bvh.h); 0 = always build the hierarchy
    int film_block = 2;  // film gather: 0 = one film pixel per thread, 1 = 2 x 2 pixels per thread, 2 (default) = TH_FILM_BX x TH_FILM_BY = 1 x 4, all three recomputing a
                         // sample's pixel range and table indices per thread; 3 = 1 x 4 from per-sample splat descriptors (k_film_descriptors): measured SLOWER
                         // (1024^2, 256 spp: 29.0 ms against 24.3 ms: the 16-byte descriptor doubles the gather's loads and the arithmetic it saves was hidden)
    bool film_tiled = false;  // LDS-staged film gather (k_film_gather_tiled): bit-identical, measured 2.7x SLOWER than k_film_gather (11 % lane use), kept as an option
    bool overlap = false;  // shadow rays of depth d on a second stream beside the closest-hit rays of depth d+1 (option "overlap").  Off since the two-stage
                           // any-hit kernels (k_any_occluders, k_any_leaf) halved the shadow rays' cost: 256 spp, on / off: S-cornell 158.2 / 157.9 ms, S-mesh 399 / 393,
                           // 10 M triangles 485 / 480 (it was worth 5 ms of S-cornell's 172 before)
    int compose_spheres = -1;  // commit: spheres as a chain of leaves above the triangles' subtree, what k_trace8 needs of a scene with spheres
                               // (option "compose_spheres": 1 / 0 = one SAH tree over everything / -1 = when "traversal" is 4 at commit time)
    int traversal = 3;  // 1 = literal accel/bvh.jl loop, 2 = children-in-parent nodes + per-lane ray replacement, 3 = 2 with leaves postponed (while-while),
                        // 4 = 8-wide quantised nodes in the binary walk's order (th_trace8.h; scenes / rays it cannot take run 3), 6 = 3 with two rays per lane (th_trace4.h)
    // workspace (grown on demand, reused across calls)
    DevBuf q[2][3], sq[3], hits, Lbuf, pfilm, counters, sensor, table, film, scratch[4], overflow, wh_L, wh_parent, wh_coef, wh_pdf, wh_flags, occl, film_Lt, surv_list, surv_counts;
    uint64_t last_L_count = 0;  // float4 entries valid in Lbuf
    // SPPM state (th_sppm.h): per film pixel, kept after trhip_render_sppm for trhip_sppm_state
    DevBuf sp_vp[7], sp_Ld, sp_tau, sp_radius, sp_N, sp_phi, sp_M, sp_counts, sp_starts, sp_entries, sp_grid, sp_ldist, sp_snap_M, sp_snap_phi, sp_snap_p, sp_snap_beta;
    DevBuf sp_terms, sp_rec[3], sp_rec_valid;
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
    DevBuf fdesc;   // film_block 3: one SplatDesc (16 B) per camera sample of the band (th_kernels.h, k_film_descriptors)
    DevBuf poison;  // one byte per camera sample of the band: ShadeStream::poison
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
    std::vector<SphereRec> spheres;
    std::vector<HostAABB> sphere_bounds;
    std::vector<LightRec> lights;
    FlatBVH bvh;
    bool committed = false;
    DevBuf d_nodes, d_prims, d_nrm, d_shade, d_spheres, d_materials, d_lights, d_wnodes;
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
};

namespace {

int fail(trhip_ctx* ctx, int code, const char* fmt, ...) {
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

int ensure(trhip_ctx* ctx, DevBuf& b, size_t bytes) {
    if (b.bytes >= bytes && b.p) return 0;
    if (b.p) HIP_TRY(ctx, hipFree(b.p));
    b.p = nullptr;
    b.bytes = 0;
    if (bytes == 0) bytes = 16;
    HIP_TRY(ctx, hipMalloc(&b.p, bytes));
    b.bytes = bytes;
    return 0;
}
void release(DevBuf& b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
}
int upload(trhip_ctx* ctx, DevBuf& b, const void* src, size_t bytes) {
    if (int rc = ensure(ctx, b, bytes)) return rc;
    if (bytes) HIP_TRY(ctx, hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
    return 0;
}
int grid_for(const trhip_ctx* ctx, uint64_t n, int blocks_per_cu) {
    const uint64_t need = (n + kBlock - 1) / kBlock;
    const uint64_t cap = (uint64_t)ctx->num_cu * blocks_per_cu;
    return (int)std::max<uint64_t>(1, std::min(need, cap));
}

// ---- materials: the lobes each Material adds (materials/material.jl), precomputed per material ---------------------------------
float roughness_to_alpha(float roughness) {  // microfacet.jl:82-87
    roughness = jmax(1e-3f, roughness);
    const float x = tm_logf(roughness);
    return 1.62142f + 0.819955f * x + 0.1734f * (x * x) + 0.0171201f * (x * x * x) + 0.000640711f * pow4(x);
}
void clamp_rgb(const float* in, float* out) {  // clamp(spectrum) spectrum.jl:34-38
    for (int i = 0; i < 3; ++i) out[i] = jclamp(in[i], 0.0f, kInf);
}
bool black(const float* c) { return c[0] == 0.0f && c[1] == 0.0f && c[2] == 0.0f; }
Lobe base_lobe(int kind, int type) {
    Lobe l;
    std::memset(&l, 0, sizeof l);
    l.kind = kind;
    l.type = type;
    l.fresnel = FRESNEL_NOOP;
    l.eta_a = l.eta_b = l.fr_eta_i = l.fr_eta_t = 1.0f;
    return l;
}
void set_rgb(float* dst, const float* src) {
    dst[0] = src[0];
    dst[1] = src[1];
    dst[2] = src[2];
}
Lobe microfacet_lobe(int kind, int type, const float* rgb, float ax, float ay) {
    Lobe l = base_lobe(kind, type);
    set_rgb(l.r, rgb);
    l.a = jmax(1e-3f, ax);  // TrowbridgeReitzDistribution ctor microfacet.jl:61-65
    l.b = jmax(1e-3f, ay);
    return l;
}
int build_material(int kind, const float* p, int n, MaterialRec& m) {
    std::memset(&m, 0, sizeof m);
    for (int multi = 0; multi < 2; ++multi) {
        LobeSet& s = m.set[multi];
        s.n = 0;
        s.eta = 1.0f;
        switch (kind) {
        case TRHIP_MATTE: {  // material.jl:16-31
            if (n != 4) return -1;
            float r[3];
            clamp_rgb(p, r);
            if (black(r)) break;
            const float sigma = jclamp(p[3], 0.0f, 90.0f);
            if (sigma == 0.0f) {
                Lobe l = base_lobe(LOBE_LAMBERT_R, BSDF_DIFFUSE | BSDF_REFLECTION);
                set_rgb(l.r, r);
                s.lobe[s.n++] = l;
            } else {  // OrenNayar ctor microfacet.jl:12-19
                Lobe l = base_lobe(LOBE_OREN_NAYAR, BSDF_DIFFUSE | BSDF_REFLECTION);
                set_rgb(l.r, r);
                const float sg = deg2rad(sigma);
                const float s2 = sg * sg;
                l.a = 1.0f - (s2 / (2.0f * (s2 + 0.33f)));
                l.b = 0.45f * s2 / (s2 + 0.09f);
                s.lobe[s.n++] = l;
            }
            break;
        }
        case TRHIP_MIRROR: {  // material.jl:39-46
            if (n != 3) return -1;
            float r[3];
            clamp_rgb(p, r);
            if (black(r)) break;
            Lobe l = base_lobe(LOBE_SPECULAR_R, BSDF_SPECULAR | BSDF_REFLECTION);
            set_rgb(l.r, r);
            s.lobe[s.n++] = l;
            break;
        }
        case TRHIP_GLASS: {  // material.jl:75-116
            if (n != 10) return -1;
            const float eta = p[8];
            float ur = p[6], vr = p[7];
            const bool remap = p[9] != 0.0f;
            s.eta = eta;
            float r[3], t[3];
            clamp_rgb(p, r);
            clamp_rgb(p + 3, t);
            if (black(r) && black(t)) break;
            const bool is_specular = ur == 0.0f && vr == 0.0f;
            if (is_specular && multi) {
                Lobe l = base_lobe(LOBE_FRESNEL_SPECULAR, BSDF_SPECULAR | BSDF_TRANSMISSION | BSDF_REFLECTION);
                set_rgb(l.r, r);
                set_rgb(l.t, t);
                l.eta_a = 1.0f;
                l.eta_b = eta;
                s.lobe[s.n++] = l;
                break;
            }
            if (remap) {
                ur = roughness_to_alpha(ur);
                vr = roughness_to_alpha(vr);
            }
            if (!black(r)) {
                Lobe l = is_specular ? base_lobe(LOBE_SPECULAR_R, BSDF_SPECULAR | BSDF_REFLECTION) : microfacet_lobe(LOBE_MICROFACET_R, BSDF_REFLECTION | BSDF_GLOSSY, r, ur, vr);
                set_rgb(l.r, r);
                l.fresnel = FRESNEL_DIELECTRIC;
                l.fr_eta_i = 1.0f;
                l.fr_eta_t = eta;
                s.lobe[s.n++] = l;
            }
            if (!black(t)) {
                Lobe l = is_specular ? base_lobe(LOBE_SPECULAR_T, BSDF_SPECULAR | BSDF_TRANSMISSION) : microfacet_lobe(LOBE_MICROFACET_T, BSDF_TRANSMISSION | BSDF_GLOSSY, t, ur, vr);
                set_rgb(l.r, t);
                l.eta_a = 1.0f;
                l.eta_b = eta;
                l.fresnel = FRESNEL_DIELECTRIC;  // FresnelDielectric(η_a, η_b) specular.jl:60, microfacet.jl:275
                l.fr_eta_i = 1.0f;
                l.fr_eta_t = eta;
                s.lobe[s.n++] = l;
            }
            break;
        }
        case TRHIP_PLASTIC: {  // material.jl:135-151
            if (n != 8) return -1;
            float kd[3], ks[3];
            clamp_rgb(p, kd);
            if (!black(kd)) {
                Lobe l = base_lobe(LOBE_LAMBERT_R, BSDF_DIFFUSE | BSDF_REFLECTION);
                set_rgb(l.r, kd);
                s.lobe[s.n++] = l;
            }
            clamp_rgb(p + 3, ks);
            if (black(ks)) break;
            float rough = p[6];
            if (p[7] != 0.0f) rough = roughness_to_alpha(rough);
            Lobe l = microfacet_lobe(LOBE_MICROFACET_R, BSDF_REFLECTION | BSDF_GLOSSY, ks, rough, rough);
            l.fresnel = FRESNEL_DIELECTRIC;
            l.fr_eta_i = 1.5f;
            l.fr_eta_t = 1.0f;
            s.lobe[s.n++] = l;
            break;
        }
        default: return -1;
        }
    }
    return 0;
}

// world_bound(sphere) = object_to_world(object_bound) (Shape.jl:17-19, transformations.jl:141-143)
HostAABB sphere_world_bound(const SphereRec& s) {
    HostAABB b;
    b.reset();
    const float lo[3] = {-s.radius, -s.radius, s.z_min}, hi[3] = {s.radius, s.radius, s.z_max};
    for (int c = 0; c < 8; ++c) {
        const f3 p = xf_point(s.o2w, mk3((c & 1) ? hi[0] : lo[0], (c & 2) ? hi[1] : lo[1], (c & 4) ? hi[2] : lo[2]));
        const float q[3] = {p.x, p.y, p.z};
        b.grow_point(q);
    }
    return b;
}
float det3(const float* m) {  // rows of the upper-left 3x3 of a row-major 4x4
    return m[0] * (m[5] * m[10] - m[6] * m[9]) - m[1] * (m[4] * m[10] - m[6] * m[8]) + m[2] * (m[4] * m[9] - m[5] * m[8]);
}

// does the subtree rooted at flat node `root` hold a sphere?  (depth-first layout: the subtree is a contiguous index range)
bool has_sphere_subtree(const trhip_scene* s, uint32_t root) {
    const uint32_t n_nodes = (uint32_t)s->bvh.a.size(), n_prims = (uint32_t)s->bvh.order.size();
    if (root >= n_nodes) return true;
    // end of the subtree: follow second children until a leaf
    uint32_t end = root;
    while ((s->bvh.flags[end] & 3u) != 3u) end = s->bvh.a[end];
    for (uint32_t i = root; i <= end && i < n_nodes; ++i)
        if ((s->bvh.flags[i] & 3u) == 3u)
            for (uint32_t k = s->bvh.a[i]; k < s->bvh.a[i] + (s->bvh.flags[i] >> 2) && k < n_prims; ++k)
                if (s->prims[s->bvh.order[k]].kind == 1) return true;
    return false;
}

int upload_scene(trhip_scene* s) {
    trhip_ctx* ctx = s->ctx;
    const uint32_t n_nodes = (uint32_t)s->bvh.a.size(), n_prims = (uint32_t)s->bvh.order.size();
    std::vector<float4> nodes((size_t)n_nodes * 2), prims((size_t)n_prims * 3), nrm((size_t)n_prims * 3);
    for (uint32_t i = 0; i < n_nodes; ++i) {
        const float* b = &s->bvh.bounds[6 * (size_t)i];
        nodes[2 * (size_t)i] = make_float4(b[0], b[1], b[2], __builtin_bit_cast(float, s->bvh.a[i]));
        nodes[2 * (size_t)i + 1] = make_float4(b[3], b[4], b[5], __builtin_bit_cast(float, s->bvh.flags[i]));
    }
    for (uint32_t k = 0; k < n_prims; ++k) {
        const HostPrim& p = s->prims[s->bvh.order[k]];
        if (p.kind == 1) {
            prims[3 * (size_t)k] = make_float4(__builtin_bit_cast(float, p.sphere_id), 0, 0, __builtin_bit_cast(float, p.meta));
            prims[3 * (size_t)k + 1] = prims[3 * (size_t)k + 2] = make_float4(0, 0, 0, 0);
            nrm[3 * (size_t)k] = nrm[3 * (size_t)k + 1] = nrm[3 * (size_t)k + 2] = make_float4(0, 0, 0, 0);
        } else {
            prims[3 * (size_t)k] = make_float4(p.v[0], p.v[1], p.v[2], __builtin_bit_cast(float, p.meta));
            prims[3 * (size_t)k + 1] = make_float4(p.v[3], p.v[4], p.v[5], 0);
            prims[3 * (size_t)k + 2] = make_float4(p.v[6], p.v[7], p.v[8], 0);
            const uint32_t mat = p.meta & PRIM_MATERIAL_MASK;
            const bool fast = mat != PRIM_NO_MATERIAL && mat < s->materials.size() && s->materials[mat].set[1].n == 1 && s->materials[mat].set[1].lobe[0].kind == LOBE_LAMBERT_R;
            if (fast) prims[3 * (size_t)k].w = __builtin_bit_cast(float, p.meta | PRIM_FAST);
            for (int j = 0; j < 3; ++j) nrm[3 * (size_t)k + j] = make_float4(p.n[3 * j], p.n[3 * j + 1], p.n[3 * j + 2], fast ? s->materials[mat].set[1].lobe[0].r[j] : 0.0f);
        }
    }
    if (int rc = upload(ctx, s->d_nodes, nodes.data(), nodes.size() * sizeof(float4))) return rc;
    if (int rc = upload(ctx, s->d_prims, prims.data(), prims.size() * sizeof(float4))) return rc;
    if (int rc = upload(ctx, s->d_nrm, nrm.data(), nrm.size() * sizeof(float4))) return rc;
    {  // the shading kernels' interleaved view (th_scene.h): one 128-byte line per slot
        std::vector<float4> rec((size_t)n_prims * 8, make_float4(0, 0, 0, 0));
        for (uint32_t k = 0; k < n_prims; ++k)
            for (int j = 0; j < 3; ++j) {
                rec[8 * (size_t)k + j] = prims[3 * (size_t)k + j];
                rec[8 * (size_t)k + 3 + j] = nrm[3 * (size_t)k + j];
            }
        if (int rc = upload(ctx, s->d_shade, rec.data(), rec.size() * sizeof(float4))) return rc;
        // records 6 / 7: what a triangle's interaction derives from its vertices alone, computed by the code the shading kernels would run
        hipLaunchKernelGGL(k_shade_constants, dim3(std::max(1u, std::min((n_prims + kBlock - 1) / kBlock, 4096u))), dim3(kBlock), 0, ctx->stream, (float4*)s->d_shade.p, n_prims);
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (int rc = upload(ctx, s->d_spheres, s->spheres.data(), s->spheres.size() * sizeof(SphereRec))) return rc;
    if (int rc = upload(ctx, s->d_materials, s->materials.data(), s->materials.size() * sizeof(MaterialRec))) return rc;
    if (int rc = upload(ctx, s->d_lights, s->lights.data(), s->lights.size() * sizeof(LightRec))) return rc;
    s->dev.nodes = (const float4*)s->d_nodes.p;
    s->dev.prims = (const float4*)s->d_prims.p;
    s->dev.tri_nrm = (const float4*)s->d_nrm.p;
    s->dev.shade = (const float4*)s->d_shade.p;
    s->dev.spheres = (const SphereRec*)s->d_spheres.p;
    s->dev.materials = (const MaterialRec*)s->d_materials.p;
    s->dev.lights = (const LightRec*)s->d_lights.p;
    s->dev.n_nodes = n_nodes;
    s->dev.n_prims = n_prims;
    s->dev.n_spheres = (uint32_t)s->spheres.size();
    s->dev.n_materials = (uint32_t)s->materials.size();
    s->dev.n_lights = (uint32_t)s->lights.size();
    // ---- children-in-parent nodes for k_trace2 (th_trace2.h) ----
    s->wide_ok = false;
    std::memset(&s->wide, 0, sizeof s->wide);
    s->wide.root_ref = kRefNone;
    if (n_nodes > 0 && n_prims < (1u << 24) && !s->literal_only) {
        std::vector<uint32_t> widx(n_nodes, 0);
        uint32_t n_int = 0;
        for (uint32_t i = 0; i < n_nodes; ++i)
            if ((s->bvh.flags[i] & 3u) != 3u) widx[i] = n_int++;
        bool ok = n_int < (1u << 24);
        std::vector<float4> wn((size_t)n_int * 4);
        // subtrees that hold a sphere keep the reference's loose slab test (th_trace2.h, slab_test2): the fp32 sphere quadratic
        // (sphere.jl:120-150) accepts rays that pass the sphere at a distance far beyond the tight test's margin
        std::vector<uint8_t> has_sphere(n_nodes, 0);
        for (uint32_t i = n_nodes; i-- > 0;) {
            if ((s->bvh.flags[i] & 3u) == 3u) {
                const uint32_t first = s->bvh.a[i], cnt = s->bvh.flags[i] >> 2;
                for (uint32_t k = first; k < first + cnt && k < n_prims; ++k) has_sphere[i] |= s->prims[s->bvh.order[k]].kind == 1;
            } else {
                has_sphere[i] = has_sphere[i + 1] | (s->bvh.a[i] < n_nodes ? has_sphere[s->bvh.a[i]] : 1);
            }
        }
        for (uint32_t i = 0; i < n_nodes && ok; ++i) {
            if ((s->bvh.flags[i] & 3u) == 3u) continue;
            const uint32_t c[2] = {i + 1, s->bvh.a[i]};
            uint32_t ref[2], cnt[2];
            for (int k = 0; k < 2; ++k) {
                if ((s->bvh.flags[c[k]] & 3u) == 3u) {
                    ref[k] = s->bvh.a[c[k]];
                    cnt[k] = s->bvh.flags[c[k]] >> 2;
                    if (cnt[k] == 0 || cnt[k] > 255) ok = false;  // empty / oversized leaves only come from foreign BVHs: use the literal kernel
                } else {
                    ref[k] = widx[c[k]];
                    cnt[k] = 0;
                }
            }
            const float* l = &s->bvh.bounds[6 * (size_t)c[0]];
            const float* r = &s->bvh.bounds[6 * (size_t)c[1]];
            float4* w = &wn[4 * (size_t)widx[i]];
            w[0] = make_float4(l[0], l[1], l[2], l[3]);
            w[1] = make_float4(l[4], l[5], r[0], r[1]);
            w[2] = make_float4(r[2], r[3], r[4], r[5]);
            // child word = ref | count << 24 (the stack entry format); meta = split axis | "subtree holds a sphere" bits 2 (first) / 3 (second)
            w[3] = make_float4(__builtin_bit_cast(float, ref[0] | (cnt[0] << 24)), __builtin_bit_cast(float, ref[1] | (cnt[1] << 24)),
                               __builtin_bit_cast(float, (s->bvh.flags[i] & 3u) | ((uint32_t)has_sphere[c[0]] << 2) | ((uint32_t)has_sphere[c[1]] << 3)), 0.0f);
        }
        if (ok) {
            if (int rc = upload(ctx, s->d_wnodes, wn.data(), wn.size() * sizeof(float4))) return rc;
            s->wide.wnodes = (const float4*)s->d_wnodes.p;
            s->wide.n_wnodes = n_int;
            std::memcpy(s->wide.root_box, &s->bvh.bounds[0], 6 * sizeof(float));
            if ((s->bvh.flags[0] & 3u) == 3u) {
                s->wide.root_ref = s->bvh.a[0];
                s->wide.root_cnt = s->bvh.flags[0] >> 2;
                ok = s->wide.root_cnt > 0 && s->wide.root_cnt <= 255;
            } else {
                s->wide.root_ref = 0;
                s->wide.root_cnt = 0;
            }
            s->wide_ok = ok;
        }
    }
    // ---- 8-wide nodes over the triangles' subtree for k_trace8 (th_wide8.h) ----
    s->w8_ok = false;
    std::memset(&s->w8, 0, sizeof s->w8);
    if (s->wide_ok && s->wide.root_cnt == 0 && n_nodes >= 3) {
        // root of the triangles' subtree: the whole tree when the scene has no sphere; with spheres the commit composed
        // root -> {leaf of all spheres (flat node 1), triangles (flat node 2)} (compose_bvh)
        uint32_t sub_root = 0, n_sph = 0;
        bool shape_ok = true;
        if (!s->spheres.empty()) {
            n_sph = (uint32_t)s->spheres.size();
            sub_root = 2 * n_sph;
            shape_ok = n_sph <= (uint32_t)kW8MaxSpheres && sub_root < n_nodes;
            for (uint32_t i = 0; i < n_sph && shape_ok; ++i)
                shape_ok = (s->bvh.flags[2 * i] & 3u) != 3u && (s->bvh.flags[2 * i] & 3u) == (s->bvh.flags[0] & 3u) && s->bvh.a[2 * i] == 2 * i + 2 &&
                           s->bvh.flags[2 * i + 1] == ((1u << 2) | 3u) && s->bvh.a[2 * i + 1] == i && s->prims[s->bvh.order[i]].kind == 1;
            shape_ok = shape_ok && !has_sphere_subtree(s, sub_root);
        }
        if (shape_ok) {
            Wide8Host wh = build_wide8(s->bvh, sub_root, [&](uint32_t slot, float* v, uint32_t& meta) {
                if (slot >= n_prims) return false;
                const HostPrim& p = s->prims[s->bvh.order[slot]];
                if (p.kind != 0) return false;
                std::memcpy(v, p.v, 9 * sizeof(float));
                meta = p.meta;
                return true;
            });
            if (wh.ok) {
                if (int rc = upload(ctx, s->d_w8nodes, wh.nodes.data(), wh.nodes.size() * sizeof(uint32_t))) return rc;
                if (int rc = upload(ctx, s->d_w8tris, wh.tris.data(), wh.tris.size() * sizeof(float))) return rc;
                s->w8.nodes = (const uint4*)s->d_w8nodes.p;
                s->w8.tris = (const float4*)s->d_w8tris.p;
                std::memcpy(s->w8.root_box, &s->bvh.bounds[0], 6 * sizeof(float));
                std::memcpy(s->w8.tri_box, &s->bvh.bounds[6 * (size_t)sub_root], 6 * sizeof(float));
                for (uint32_t i = 0; i < n_sph; ++i) std::memcpy(s->w8.sph_box[i], &s->bvh.bounds[6 * (size_t)(2 * i + 1)], 6 * sizeof(float));
                s->w8.n_sph = n_sph;
                s->w8.chain_axis = s->bvh.flags[0] & 3u;
                s->w8_nodes = (uint32_t)(wh.nodes.size() / kW8NodeDwords);
                s->w8_depth = wh.depth;
                s->w8_ok = true;
            }
        }
    }
    // ---- one-leaf scenes: the order in which any-hit rays try the leaf's primitives (th_trace2.h, k_any_leaf) ----
    // A shadow ray runs from the surface THROUGH the light (t_max = Inf): what stops it at the latest is what the light sees, so the
    // primitives subtending the largest solid angle at the lights come first (triangles: Van Oosterom & Strackee; spheres: the cap of
    // their bounding sphere).  Any order gives the same boolean.
    s->wide.leaf_order = nullptr;
    if (s->wide_ok && s->wide.root_cnt > 1 && !s->lights.empty()) {
        const uint32_t first = s->wide.root_ref, cnt = s->wide.root_cnt;
        std::vector<std::pair<double, uint32_t>> ord;
        for (uint32_t k = 0; k < cnt; ++k) {
            const HostPrim& p = s->prims[s->bvh.order[first + k]];
            double w = 0.0;
            for (const LightRec& l : s->lights) {
                const float* lp = l.position;
                if (p.kind == 1) {
                    const HostAABB& b = s->sphere_bounds[p.sphere_id];
                    double c[3], r = 0.0, d2 = 0.0;
                    for (int a = 0; a < 3; ++a) {
                        c[a] = 0.5 * ((double)b.mn[a] + b.mx[a]);
                        r = std::max(r, 0.5 * ((double)b.mx[a] - b.mn[a]));
                        d2 += (c[a] - lp[a]) * (c[a] - lp[a]);
                    }
                    w += d2 <= r * r ? 4.0 * 3.14159265358979 : 2.0 * 3.14159265358979 * (1.0 - std::sqrt(std::max(0.0, 1.0 - r * r / d2)));
                } else {
                    double r[3][3], len[3];
                    for (int v = 0; v < 3; ++v) {
                        for (int c = 0; c < 3; ++c) r[v][c] = (double)p.v[3 * v + c] - lp[c];
                        len[v] = std::sqrt(r[v][0] * r[v][0] + r[v][1] * r[v][1] + r[v][2] * r[v][2]);
                    }
                    const double det = r[0][0] * (r[1][1] * r[2][2] - r[1][2] * r[2][1]) - r[0][1] * (r[1][0] * r[2][2] - r[1][2] * r[2][0]) + r[0][2] * (r[1][0] * r[2][1] - r[1][1] * r[2][0]);
                    auto dot3 = [&](int a, int b) { return r[a][0] * r[b][0] + r[a][1] * r[b][1] + r[a][2] * r[b][2]; };
                    const double den = len[0] * len[1] * len[2] + dot3(0, 1) * len[2] + dot3(0, 2) * len[1] + dot3(1, 2) * len[0];
                    w += 2.0 * std::fabs(std::atan2(det, den));
                }
            }
            ord.push_back({w, k});
        }
        std::stable_sort(ord.begin(), ord.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
        std::vector<uint32_t> order(cnt);
        for (uint32_t k = 0; k < cnt; ++k) order[k] = ord[k].second;
        if (int rc = upload(ctx, s->d_leaf_order, order.data(), order.size() * sizeof(uint32_t))) return rc;
        s->wide.leaf_order = (const uint32_t*)s->d_leaf_order.p;
    }
    // ---- largest triangles: the any-hit pre-pass (th_trace2.h, k_any_occluders) ----
    s->n_occluders = 0;
    if (s->wide_ok && s->wide.root_cnt == 0) {
        const float* rb = &s->bvh.bounds[0];
        const double ex = (double)rb[3] - rb[0], ey = (double)rb[4] - rb[1], ez = (double)rb[5] - rb[2];
        const double face = std::max(ex * ey, std::max(ex * ez, ey * ez));
        std::vector<std::pair<double, uint32_t>> big;  // (area, ordered slot)
        for (uint32_t k = 0; k < n_prims; ++k) {
            const HostPrim& p = s->prims[s->bvh.order[k]];
            if (p.kind != 0 || (p.meta & PRIM_DEGENERATE)) continue;
            const double ax = (double)p.v[3] - p.v[0], ay = (double)p.v[4] - p.v[1], az = (double)p.v[5] - p.v[2];
            const double bx = (double)p.v[6] - p.v[0], by = (double)p.v[7] - p.v[1], bz = (double)p.v[8] - p.v[2];
            const double cx = ay * bz - az * by, cy = az * bx - ax * bz, cz = ax * by - ay * bx;
            const double area = 0.5 * std::sqrt(cx * cx + cy * cy + cz * cz);
            if (area >= 0.02 * face) big.push_back({area, k});
        }
        if (!big.empty() && big.size() * 8 <= (size_t)n_prims) {  // a few walls around much else; not a scene that consists of large triangles
            std::sort(big.begin(), big.end(), [](const auto& a, const auto& b) { return a.first > b.first || (a.first == b.first && a.second < b.second); });
            if (big.size() > 16) big.resize(16);
            // test order: a shadow ray runs from the surface THROUGH the light (t_max = Inf) — what stops it at the latest is what the light
            // sees, so the triangles subtending the largest solid angle at the lights come first (Van Oosterom & Strackee)
            auto solid_angle = [&](uint32_t k, const float* lp) {
                const HostPrim& p = s->prims[s->bvh.order[k]];
                double r[3][3], len[3];
                for (int v = 0; v < 3; ++v) {
                    for (int c = 0; c < 3; ++c) r[v][c] = (double)p.v[3 * v + c] - lp[c];
                    len[v] = std::sqrt(r[v][0] * r[v][0] + r[v][1] * r[v][1] + r[v][2] * r[v][2]);
                }
                const double det = r[0][0] * (r[1][1] * r[2][2] - r[1][2] * r[2][1]) - r[0][1] * (r[1][0] * r[2][2] - r[1][2] * r[2][0]) + r[0][2] * (r[1][0] * r[2][1] - r[1][1] * r[2][0]);
                auto dot3 = [&](int a, int b) { return r[a][0] * r[b][0] + r[a][1] * r[b][1] + r[a][2] * r[b][2]; };
                const double den = len[0] * len[1] * len[2] + dot3(0, 1) * len[2] + dot3(0, 2) * len[1] + dot3(1, 2) * len[0];
                return 2.0 * std::fabs(std::atan2(det, den));
            };
            if (!s->lights.empty()) {
                for (auto& b : big) {
                    double w = 0.0;
                    for (const LightRec& l : s->lights) w += solid_angle(b.second, l.position);
                    b.first = w;
                }
                std::stable_sort(big.begin(), big.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
            }
            std::vector<uint32_t> leaf_of(n_prims, 0xffffffffu);
            for (uint32_t i = 0; i < n_nodes; ++i)
                if ((s->bvh.flags[i] & 3u) == 3u)
                    for (uint32_t k = s->bvh.a[i]; k < s->bvh.a[i] + (s->bvh.flags[i] >> 2) && k < n_prims; ++k) leaf_of[k] = i;
            std::vector<uint32_t> slots;
            std::vector<float> boxes;
            for (const auto& b : big) {
                if (leaf_of[b.second] == 0xffffffffu) continue;
                slots.push_back(b.second);
                for (int a = 0; a < 6; ++a) boxes.push_back(s->bvh.bounds[6 * (size_t)leaf_of[b.second] + a]);
            }
            if (slots.size() >= 6) {  // an enclosure (three quads or more); a lone floor stops few shadow rays and the pre-pass only costs (S-caustic)
                if (int rc = upload(ctx, s->d_occ_slots, slots.data(), slots.size() * sizeof(uint32_t))) return rc;
                if (int rc = upload(ctx, s->d_occ_boxes, boxes.data(), boxes.size() * sizeof(float))) return rc;
                s->n_occluders = (uint32_t)slots.size();
            }
        }
    }
    s->committed = true;
    return 0;
}

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
// the per-sample buffer the film pass needs next to the radiance: descriptors (16 B) or film positions (8 B)
int ensure_film_samples(trhip_ctx* ctx, const DeviceSensor& ds, uint64_t total_slots) {
    if (film_uses_desc(ctx, ds)) return ensure(ctx, ctx->fdesc, total_slots * sizeof(uint4));
    return ensure(ctx, ctx->pfilm, total_slots * sizeof(float2));
}
void launch_film(trhip_ctx* ctx, hipStream_t st, const DeviceSensor& ds, const DeviceSensor* dsp, const float4* L, uint64_t total_slots, uint32_t spp, uint64_t seed, uint32_t sample_offset,
                 float4* d_film) {
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
        if (ctx->film_block == 1)
            hipLaunchKernelGGL((k_film_gather_block<2, 2>), dim3(grid_for(ctx, (npx + 3) / 4, 8)), dim3(kBlock), 0, st, dsp, (const float*)ctx->table.p, L, (const float2*)ctx->pfilm.p, spp, layout, d_film);
        else if (ctx->film_block == 2)
            hipLaunchKernelGGL((k_film_gather_block<TH_FILM_BX, TH_FILM_BY>), dim3(grid_for(ctx, (npx + TH_FILM_BX * TH_FILM_BY - 1) / (TH_FILM_BX * TH_FILM_BY), 8)), dim3(kBlock), 0, st, dsp, (const float*)ctx->table.p, L, (const float2*)ctx->pfilm.p, spp, layout, d_film);
        else
            hipLaunchKernelGGL(k_film_gather, dim3(grid_for(ctx, npx, 8)), dim3(kBlock), 0, st, dsp, (const float*)ctx->table.p, L, (const float2*)ctx->pfilm.p, spp, layout, d_film);
    }
}

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
    return w;
}

// which kernel launch_trace picks for this scene, and the bytes one unit of the visit counters stands for (trhip_stats)
void traversal_info(const trhip_ctx* ctx, const trhip_scene* sc, uint32_t* trav, uint32_t* node_bytes) {
    uint32_t t = 1, nb = 32;
    if (ctx->traversal >= 2 && sc->wide_ok) {
        if (sc->wide.root_cnt > 0 && ctx->leaf_kernel && ctx->debug_trace_budget == 0) {
            t = 5;
            nb = 0;
        } else if (sc->wide.root_cnt > 0) {
            t = 2;
        } else if (ctx->traversal == 4 && sc->w8_ok) {
            t = 4;
            nb = 96;
        } else if (ctx->traversal == 6) {
            t = 6;
        } else {
            t = ctx->traversal >= 3 ? 3 : 2;
        }
    }
    *trav = t;
    *node_bytes = nb;
}

// One traversal launch over a queue (count in HBM at count_ptr, or n_max when count_ptr is null).
// ctx->traversal == 1: the literal accel/bvh.jl loop (k_trace_closest / k_trace_any); 2: k_trace2 (same results).
void launch_trace(trhip_ctx* ctx, hipStream_t st, const trhip_scene* sc, bool any, SegQueue q, const float4* ro, const float4* rd, const float* tmax, TraceOut out, uint32_t* work_cursors,
                  Counters* ctr, void* overflow_slab = nullptr) {
    const dim3 grid(trace_grid(ctx)), block(kBlock);
    const bool v2 = ctx->traversal >= 2 && sc->wide_ok;
    const bool cnt = ctx->count_visits;
    const bool full_only = !sc->partial_spheres;  // no clipped sphere in the scene: kernels without the Float64 atan2 path
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
        // ---- traversal 4: 8-wide nodes (th_trace8.h); the rays it does not take come back on a fallback list that k_trace3 walks below ----
        if (ctx->traversal == 4 && sc->w8_ok && ctx->slab_margin_log2 > 0 && ctx->pipelines <= 1) {
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
                if (any) {
                    if (cnt)
                        { if (full_only) hipLaunchKernelGGL((k_trace8<true, true, true>), grid, block, 0, st, sc->dev, w8, q, ro, rd, tmax, out, work_cursors, ov8, ctr, fb); else hipLaunchKernelGGL((k_trace8<true, true, false>), grid, block, 0, st, sc->dev, w8, q, ro, rd, tmax, out, work_cursors, ov8, ctr, fb); }
                    else
                        { if (full_only) hipLaunchKernelGGL((k_trace8<true, false, true>), grid, block, 0, st, sc->dev, w8, q, ro, rd, tmax, out, work_cursors, ov8, ctr, fb); else hipLaunchKernelGGL((k_trace8<true, false, false>), grid, block, 0, st, sc->dev, w8, q, ro, rd, tmax, out, work_cursors, ov8, ctr, fb); }
                } else {
                    if (cnt)
                        { if (full_only) hipLaunchKernelGGL((k_trace8<false, true, true>), grid, block, 0, st, sc->dev, w8, q, ro, rd, tmax, out, work_cursors, ov8, ctr, fb); else hipLaunchKernelGGL((k_trace8<false, true, false>), grid, block, 0, st, sc->dev, w8, q, ro, rd, tmax, out, work_cursors, ov8, ctr, fb); }
                    else
                        { if (full_only) hipLaunchKernelGGL((k_trace8<false, false, true>), grid, block, 0, st, sc->dev, w8, q, ro, rd, tmax, out, work_cursors, ov8, ctr, fb); else hipLaunchKernelGGL((k_trace8<false, false, false>), grid, block, 0, st, sc->dev, w8, q, ro, rd, tmax, out, work_cursors, ov8, ctr, fb); }
                }
                q = SegQueue{fcounts, fcap, 0u, fb.list, 1u};
                work_cursors = fcounts + (size_t)kSeg * kCtrStride;
            }
        }
        if (ctx->traversal == 6) {  // two rays per lane (th_trace4.h)
            if (any) {
                if (cnt)
                    { if (full_only) hipLaunchKernelGGL((k_trace4<true, true, true>), grid, block, 0, st, sc->dev, wide_view(ctx, sc), q, ro, rd, tmax, out, work_cursors, ov, ctr); else hipLaunchKernelGGL((k_trace4<true, true, false>), grid, block, 0, st, sc->dev, wide_view(ctx, sc), q, ro, rd, tmax, out, work_cursors, ov, ctr); }
                else
                    { if (full_only) hipLaunchKernelGGL((k_trace4<true, false, true>), grid, block, 0, st, sc->dev, wide_view(ctx, sc), q, ro, rd, tmax, out, work_cursors, ov, ctr); else hipLaunchKernelGGL((k_trace4<true, false, false>), grid, block, 0, st, sc->dev, wide_view(ctx, sc), q, ro, rd, tmax, out, work_cursors, ov, ctr); }
            } else {
                if (cnt)
                    { if (full_only) hipLaunchKernelGGL((k_trace4<false, true, true>), grid, block, 0, st, sc->dev, wide_view(ctx, sc), q, ro, rd, tmax, out, work_cursors, ov, ctr); else hipLaunchKernelGGL((k_trace4<false, true, false>), grid, block, 0, st, sc->dev, wide_view(ctx, sc), q, ro, rd, tmax, out, work_cursors, ov, ctr); }
                else
                    { if (full_only) hipLaunchKernelGGL((k_trace4<false, false, true>), grid, block, 0, st, sc->dev, wide_view(ctx, sc), q, ro, rd, tmax, out, work_cursors, ov, ctr); else hipLaunchKernelGGL((k_trace4<false, false, false>), grid, block, 0, st, sc->dev, wide_view(ctx, sc), q, ro, rd, tmax, out, work_cursors, ov, ctr); }
            }
            return;
        }
        if (any) {
            if (cnt)
                { if (full_only) hipLaunchKernelGGL((k_trace3<true, true, true>), grid, block, 0, st, sc->dev, wide_view(ctx, sc), q, ro, rd, tmax, out, work_cursors, ov, ctr); else hipLaunchKernelGGL((k_trace3<true, true, false>), grid, block, 0, st, sc->dev, wide_view(ctx, sc), q, ro, rd, tmax, out, work_cursors, ov, ctr); }
            else
                { if (full_only) hipLaunchKernelGGL((k_trace3<true, false, true>), grid, block, 0, st, sc->dev, wide_view(ctx, sc), q, ro, rd, tmax, out, work_cursors, ov, ctr); else hipLaunchKernelGGL((k_trace3<true, false, false>), grid, block, 0, st, sc->dev, wide_view(ctx, sc), q, ro, rd, tmax, out, work_cursors, ov, ctr); }
        } else {
            // scenes larger than the last-level cache (256 MB of MALL): one wave per SIMD fewer (k_trace3's BIG variant)
            const bool big = (size_t)sc->wide.n_wnodes * 64u + (size_t)sc->dev.n_prims * 48u > ((size_t)256 << 20);
            if (cnt)
                { if (full_only) hipLaunchKernelGGL((k_trace3<false, true, true>), grid, block, 0, st, sc->dev, wide_view(ctx, sc), q, ro, rd, tmax, out, work_cursors, ov, ctr); else hipLaunchKernelGGL((k_trace3<false, true, false>), grid, block, 0, st, sc->dev, wide_view(ctx, sc), q, ro, rd, tmax, out, work_cursors, ov, ctr); }
            else if (big)
                { if (full_only) hipLaunchKernelGGL((k_trace3<false, false, true, true>), grid, block, 0, st, sc->dev, wide_view(ctx, sc), q, ro, rd, tmax, out, work_cursors, ov, ctr); else hipLaunchKernelGGL((k_trace3<false, false, false, true>), grid, block, 0, st, sc->dev, wide_view(ctx, sc), q, ro, rd, tmax, out, work_cursors, ov, ctr); }
            else
                { if (full_only) hipLaunchKernelGGL((k_trace3<false, false, true>), grid, block, 0, st, sc->dev, wide_view(ctx, sc), q, ro, rd, tmax, out, work_cursors, ov, ctr); else hipLaunchKernelGGL((k_trace3<false, false, false>), grid, block, 0, st, sc->dev, wide_view(ctx, sc), q, ro, rd, tmax, out, work_cursors, ov, ctr); }
        }
        return;
    }
    if (v2) {
        if (sc->wide.root_cnt > 0 && ctx->debug_trace_budget == 0 && ctx->leaf_kernel) {  // one-leaf scene: the dedicated kernel (th_trace2.h, k_trace_leaf)
            const dim3 lgrid(ctx->num_cu * 8);
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

struct Timer {
    trhip_ctx* ctx;
    bool on;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[5];
    explicit Timer(trhip_ctx* c, bool enable) : ctx(c), on(enable) {}
    ~Timer() {
        for (auto& v : ev)
            for (auto& p : v) {
                (void)hipEventDestroy(p.first);
                (void)hipEventDestroy(p.second);
            }
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
            launch_trace(ctx, st, scene, false, SegQueue{ctr->n_queue[depth - 1], cap, 0u}, pq[cur].o, pq[cur].d, nullptr, TraceOut{hits, nullptr, nullptr, nullptr}, ctr->work_closest[depth - 1], ctr);
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
    launch_film(ctx, st, ds, dsp, L, total_slots, spp, seed, sample_offset, (float4*)d_film);
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

// PathIntegrator as a STREAMING wavefront (th_trace2.h "streaming wavefront"): rounds instead of depths.  A round
// traces every queued ray with a fetch budget, resumes the rays suspended in the round before, shades what finished (entries
// carry their own depth), and traces the shadow rays the same way.  max_depth + 16 budgeted rounds, then max_depth rounds
// without a budget, which complete whatever is left.  Radiance terms go to per-depth slots and are folded in depth order, so
// the per-sample radiance (and the film) is bit-identical to the classic per-depth wavefront.
// BVHAccel on the device (th_lbvh.h): returns TRHIP_ERR_UNSUPPORTED when the tree is deeper than the traversal stack allows
// (the caller then falls back to the host builder).
int build_bvh_device(trhip_ctx* ctx, const std::vector<HostAABB>& pb, FlatBVH& out) {
    const uint32_t n = (uint32_t)pb.size();
    if (n < 2 || n >= (1u << 30)) return TRHIP_ERR_UNSUPPORTED;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    static_assert(sizeof(HostAABB) == 6 * sizeof(float), "HostAABB layout");
    struct Buf {
        void* p = nullptr;
        ~Buf() {
            if (p) (void)hipFree(p);
        }
    };
    Buf d_pb, d_keys, d_keys2, d_sorted, d_sorted2, d_tmp, d_u32, d_ib, d_misc, d_fb, d_fa, d_ff, d_fo;
    const size_t n_int = n - 1, total = 2 * (size_t)n - 1;
    HIP_TRY(ctx, hipMalloc(&d_pb.p, (size_t)n * 6 * sizeof(float)));
    HIP_TRY(ctx, hipMalloc(&d_keys.p, (size_t)n * 8));
    HIP_TRY(ctx, hipMalloc(&d_keys2.p, (size_t)n * 8));
    HIP_TRY(ctx, hipMalloc(&d_sorted.p, (size_t)n * 4));
    HIP_TRY(ctx, hipMalloc(&d_sorted2.p, (size_t)n * 4));
    HIP_TRY(ctx, hipMalloc(&d_u32.p, (6 * n_int + n) * sizeof(uint32_t)));  // left, right, lo, split, parent_int, visits | parent_leaf
    HIP_TRY(ctx, hipMalloc(&d_ib.p, n_int * 6 * sizeof(float)));
    HIP_TRY(ctx, hipMalloc(&d_misc.p, 8 * sizeof(uint32_t)));
    HIP_TRY(ctx, hipMalloc(&d_fb.p, total * 6 * sizeof(float)));
    HIP_TRY(ctx, hipMalloc(&d_fa.p, total * sizeof(uint32_t)));
    HIP_TRY(ctx, hipMalloc(&d_ff.p, total * sizeof(uint32_t)));
    HIP_TRY(ctx, hipMalloc(&d_fo.p, (size_t)n * sizeof(uint32_t)));
    hipStream_t st = ctx->stream;
    HIP_TRY(ctx, hipMemcpyAsync(d_pb.p, pb.data(), (size_t)n * 6 * sizeof(float), hipMemcpyHostToDevice, st));
    uint32_t* u = (uint32_t*)d_u32.p;
    LbvhBuild b{(const float*)d_pb.p, (uint64_t*)d_keys.p, (uint32_t*)d_sorted.p, u, u + n_int, u + 2 * n_int, u + 3 * n_int, u + 4 * n_int, u + 6 * n_int, u + 5 * n_int, (float*)d_ib.p,
                (uint32_t*)d_misc.p, (uint32_t*)d_misc.p + 6, n};
    const uint32_t init[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0u, 0u};  // enc(+Inf) < 0xffffffff and enc(-Inf) > 0: any real value wins
    HIP_TRY(ctx, hipMemcpyAsync(d_misc.p, init, sizeof init, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemsetAsync(b.visits, 0, n_int * sizeof(uint32_t), st));
    const dim3 grid(grid_for(ctx, n, 8)), gridt(grid_for(ctx, total, 8)), blk(kBlock);
    hipLaunchKernelGGL(k_lbvh_centroid_bounds, dim3(ctx->num_cu), blk, 0, st, b);
    hipLaunchKernelGGL(k_lbvh_keys, grid, blk, 0, st, b);
    size_t tmp_bytes = 0;
    HIP_TRY(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, (const uint64_t*)d_keys.p, (uint64_t*)d_keys2.p, (const uint32_t*)d_sorted.p, (uint32_t*)d_sorted2.p, (int)n, 0, 63, st));
    HIP_TRY(ctx, hipMalloc(&d_tmp.p, tmp_bytes));
    HIP_TRY(ctx, hipcub::DeviceRadixSort::SortPairs(d_tmp.p, tmp_bytes, (const uint64_t*)d_keys.p, (uint64_t*)d_keys2.p, (const uint32_t*)d_sorted.p, (uint32_t*)d_sorted2.p, (int)n, 0, 63, st));
    b.keys = (uint64_t*)d_keys2.p;
    b.sorted = (uint32_t*)d_sorted2.p;
    hipLaunchKernelGGL(k_lbvh_hierarchy, grid, blk, 0, st, b);
    hipLaunchKernelGGL(k_lbvh_refit, grid, blk, 0, st, b);
    const LbvhFlat f{(float*)d_fb.p, (uint32_t*)d_fa.p, (uint32_t*)d_ff.p, (uint32_t*)d_fo.p};
    hipLaunchKernelGGL(k_lbvh_flatten, gridt, blk, 0, st, b, f);
    HIP_TRY(ctx, hipGetLastError());
    uint32_t misc[8];
    HIP_TRY(ctx, hipMemcpyAsync(misc, d_misc.p, sizeof misc, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    if (misc[6] > (uint32_t)kStack2Total) return TRHIP_ERR_UNSUPPORTED;
    out.bounds.resize(total * 6);
    out.a.resize(total);
    out.flags.resize(total);
    out.order.resize(n);
    HIP_TRY(ctx, hipMemcpy(out.bounds.data(), d_fb.p, total * 6 * sizeof(float), hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(out.a.data(), d_fa.p, total * sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(out.flags.data(), d_ff.p, total * sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(out.order.data(), d_fo.p, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    out.max_depth = misc[6];
    return 0;
}

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
    const bool cnt = ctx->count_visits;

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
    const dim3 grid(trace_grid(ctx)), block(kBlock);
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
        if (cnt)
            hipLaunchKernelGGL((k_trace2<false, true, true>), grid, block, 0, ps, scene->dev, wide_view(ctx, scene), qc, pq[cur].o, pq[cur].d, nullptr, oc, ctr->work_closest[r], (uint2*)pp.overflow[0].p, ctr, 0u, sc_c);
        else
            hipLaunchKernelGGL((k_trace2<false, false, true>), grid, block, 0, ps, scene->dev, wide_view(ctx, scene), qc, pq[cur].o, pq[cur].d, nullptr, oc, ctr->work_closest[r], (uint2*)pp.overflow[0].p, ctr, 0u, sc_c);
        tm.end(1, ps);
        if (ps2 != ps && r > 0) HIP_TRY(ctx, hipStreamWaitEvent(ps, pp.ev_any, 0));  // shade(r) reuses the shadow queue
        tm.begin(2, ps);
        hipLaunchKernelGGL(k_shade_path<true>, dim3(g_shade), dim3(kBlock), 0, ps, scene->dev, dsp, pq[cur], pq[cur ^ 1], sq, cap, hits, terms, ctr, r, 0, max_depth, 1u,
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
        if (cnt)
            hipLaunchKernelGGL((k_trace2<true, true, true>), grid, block, 0, ps2, scene->dev, wide_view(ctx, scene), qa, sq.o, sq.d, nullptr, oa, ctr->work_shadow[r], (uint2*)pp.overflow[1].p, ctr, 0u, sc_a);
        else
            hipLaunchKernelGGL((k_trace2<true, false, true>), grid, block, 0, ps2, scene->dev, wide_view(ctx, scene), qa, sq.o, sq.d, nullptr, oa, ctr->work_shadow[r], (uint2*)pp.overflow[1].p, ctr, 0u, sc_a);
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
    if (ctx->overlap) hipLaunchKernelGGL(k_apply_poison, dim3(grid_for(ctx, total_slots, 8)), dim3(kBlock), 0, st, L, (const uint8_t*)ctx->poison.p, total_slots);
    launch_film(ctx, st, ds, dsp, L, total_slots, spp, seed, sample_offset, (float4*)d_film);
    tm.end(4, st);
    HIP_TRY(ctx, hipEventRecord(e1, st));
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(st));
    ctx->last_L_count = total_slots;
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
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        stats->ms_total = ms;
        stats->ms_raygen = tm.total(0, &stats->launches_raygen);
        stats->ms_trace_closest = tm.total(1, &stats->launches_trace_closest);
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
    if (!ctx || !scene || !sensor || !out) return fail(ctx, TRHIP_ERR_INVALID, "null argument");
    if (!scene->committed) return fail(ctx, TRHIP_ERR_INVALID, "scene not committed");
    if (integrator != 0 && integrator != 1) return fail(ctx, TRHIP_ERR_INVALID, "unknown integrator %d", integrator);
    // A GeometricPrimitive without a material makes the reference re-spawn the ray behind the hit without counting a bounce
    // (sppm.jl:219-222; Whitted calls a method that does not exist, sampler.jl:77-80).  The wavefront does not model that.
    for (const HostPrim& hp : scene->prims)
        if ((hp.meta & PRIM_MATERIAL_MASK) == PRIM_NO_MATERIAL)
            return fail(ctx, TRHIP_ERR_UNSUPPORTED, "rendering a scene with a material-less primitive is not supported (the trace entry points accept it)");
    if (spp == 0 || max_depth < 1 || max_depth > kMaxDepth) return fail(ctx, TRHIP_ERR_INVALID, "spp must be >= 1 and max_depth in 1..%d", kMaxDepth);
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
    if (int rc = ensure(ctx, ctx->poison, total_slots)) return rc;
    if (int rc = ensure(ctx, ctx->Lbuf, total_slots * sizeof(float4))) return rc;
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

    Timer tm(ctx, ctx->timing && stats);
    hipEvent_t e0, e1, ev_start;
    HIP_TRY(ctx, hipEventCreate(&e0));
    HIP_TRY(ctx, hipEventCreate(&e1));
    HIP_TRY(ctx, hipEventCreateWithFlags(&ev_start, hipEventDisableTiming));
    HIP_TRY(ctx, hipEventRecord(e0, st));
    HIP_TRY(ctx, hipMemsetAsync(L, 0, total_slots * sizeof(float4), st));
    HIP_TRY(ctx, hipMemsetAsync(ctx->poison.p, 0, total_slots, st));
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
        hipLaunchKernelGGL(k_raygen, dim3(grid_for(ctx, nb, 8)), dim3(kBlock), 0, ps, dsp, (uint32_t)(s0 * npix), (uint32_t)nb, seed, sample_offset, pq[0], cap, ctr);
        tm.end(0, ps);
        int cur = 0;
        for (int depth = 1; depth <= max_depth; ++depth) {
            const ShadowQueue& sq = sqs[depth & 1];
            tm.begin(1, ps);
            launch_trace(ctx, ps, scene, false, SegQueue{ctr->n_queue[depth - 1], cap, 0u}, pq[cur].o, pq[cur].d, nullptr, TraceOut{hits, nullptr, nullptr, nullptr, bary_mode}, ctr->work_closest[depth - 1], ctr,
                         pp.overflow[0].p);
            tm.end(1, ps);
            if (two && depth > 2) HIP_TRY(ctx, hipStreamWaitEvent(ps, ev_anys[depth & 1], 0));  // shade(d) refills the queue the shadow rays of depth d - 2 read
            tm.begin(2, ps);
            hipLaunchKernelGGL(k_shade_path<false>, dim3(g_shade), dim3(kBlock), 0, ps, scene->dev, dsp, pq[cur], pq[cur ^ 1], sq, cap, hits, L, ctr, depth - 1, depth, max_depth, bary_mode,
                               ShadeStream{nullptr, nullptr, 0u, two ? (uint8_t*)ctx->poison.p : nullptr});
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
    if (ctx->overlap) hipLaunchKernelGGL(k_apply_poison, dim3(grid_for(ctx, total_slots, 8)), dim3(kBlock), 0, st, L, (const uint8_t*)ctx->poison.p, total_slots);
    launch_film(ctx, st, ds, dsp, L, total_slots, spp, seed, sample_offset, (float4*)d_film);
    tm.end(4, st);
    HIP_TRY(ctx, hipEventRecord(e1, st));
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(st));
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
        }
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        stats->ms_total = ms;
        stats->ms_raygen = tm.total(0, &stats->launches_raygen);
        stats->ms_trace_closest = tm.total(1, &stats->launches_trace_closest);
        stats->ms_shade = tm.total(2, &stats->launches_shade);
        stats->ms_trace_any = tm.total(3, &stats->launches_trace_any);
        stats->ms_film = tm.total(4, &stats->launches_film);
        stats->n_batches = n_batches;
        stats->max_depth_reached = (uint32_t)max_depth;
        traversal_info(ctx, scene, &stats->traversal, &stats->node_bytes);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipEventDestroy(ev_start);
    return 0;
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
            const double row_bytes = (double)ds.sb_w * 16.0 * (double)spp * (film_uses_desc(ctx, ds) ? 33.0 : 25.0);  // radiance + descriptor / film position + poison byte
            rows_per_band = (int)std::max(1.0, std::min((double)ds.tiles_y, std::floor(budget / row_bytes)));
            while (rows_per_band > 1 && (uint64_t)ds.sb_w * 16ull * (uint64_t)rows_per_band * spp >= (1ull << 32)) --rows_per_band;
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
        DeviceSensor b = ds;
        b.band_ty0 = t0;
        b.band_ty1 = std::min(ds.tiles_y, t0 + rows_per_band) - 1;
        b.band_y0 = ds.sb_min[1] + 16 * t0;
        b.band_rows = std::min(ds.sb_max[1], ds.sb_min[1] + 16 * b.band_ty1 + 15) - b.band_y0 + 1;
        b.accumulate = t0 > 0 ? 1 : 0;
        trhip_stats st;
        if (int rc = render_impl_band(ctx, scene, sensor, integrator, spp, max_depth, seed, sample_offset, d_film, true, &st, &b)) return rc;
        sum.camera_samples += st.camera_samples;
        sum.closest_rays += st.closest_rays;
        sum.shadow_rays += st.shadow_rays;
        sum.nodes_visited += st.nodes_visited;
        sum.prims_tested += st.prims_tested;
        sum.nodes_visited_shadow += st.nodes_visited_shadow;
        sum.prims_tested_shadow += st.prims_tested_shadow;
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
    if (!out_is_device) HIP_TRY(ctx, hipMemcpy(out, d_film, film_bytes, hipMemcpyDeviceToHost));
    if (stats) *stats = sum;
    return 0;
}

// SPPMIntegrator (integrators/sppm.jl:132-173): n_iterations x {camera pass, grid, photon pass, pixel update}, then
// _sppm_to_image + set_image!.  Everything runs on one stream; queue sizes stay in HBM, the host only enqueues.
int render_sppm_impl(trhip_ctx* ctx, const trhip_scene* scene, const trhip_sensor* sensor, float initial_radius, int max_depth, uint32_t n_iterations, int64_t photons_per_iteration,
                     uint64_t seed, float* out_xyzw, trhip_stats* stats) {
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
    for (const HostPrim& hp : scene->prims)
        if ((hp.meta & PRIM_MATERIAL_MASK) == PRIM_NO_MATERIAL) return fail(ctx, TRHIP_ERR_UNSUPPORTED, "SPPM: primitives without a material are not supported on the device");
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
    for (uint32_t it0 = 1; it0 <= n_iterations; it0 += (uint32_t)B) {
        const uint32_t nb = (uint32_t)std::min<uint64_t>(B, n_iterations - it0 + 1);
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
            launch_trace(ctx, st, scene, false, SegQueue{ctr->n_queue[depth - 1], cap, 0u}, pq[cur].o, pq[cur].d, nullptr, TraceOut{hits, nullptr, nullptr, nullptr}, ctr->work_closest[depth - 1], ctr,
                         pp.overflow[0].p);
            tm.end(1, st);
            tm.begin(2, st);
            hipLaunchKernelGGL(k_shade_sppm, g_shade, blk, 0, st, scene->dev, pq[cur], pq[cur ^ 1], sq, cap, hits, vp_all, terms, ctr, depth, max_depth, seed, it0, n, W);
            tm.end(2, st);
            tm.begin(3, st);
            launch_trace(ctx, st, scene, true, SegQueue{ctr->n_shadow[depth - 1], cap, 0u}, sq.o, sq.d, nullptr, TraceOut{nullptr, terms, sq.c, nullptr}, ctr->work_shadow[depth - 1], ctr,
                         pp.overflow[0].p);
            tm.end(3, st);
            cur ^= 1;
        }
        tm.begin(2, st);
        hipLaunchKernelGGL(k_sppm_fold_ld, g_pix, blk, 0, st, n, nb, (uint32_t)max_depth, (const float4*)terms, px.Ld);
        tm.end(2, st);
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
                hipLaunchKernelGGL(k_shade_photon, g_shade, blk, 0, st, scene->dev, pq[cur], pq[cur ^ 1], cap, hits, rec, NP, ctr, depth, max_depth, halton_base);
                tm.end(2, st);
                cur ^= 1;
            }
        }
        // ---- per iteration, in order: grid (:272-318), photon contributions (:366-391), _update_pixels! (:438-459) ----
        for (uint32_t j = 0; j < nb; ++j) {
            const VisiblePoints vp = vp_slice(j);
            tm.begin(2, st);
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
            hipLaunchKernelGGL(k_sppm_gather, g_pix, blk, 0, st, scene->dev, rec, vp, px, n, grid, (const uint32_t*)starts, (const float4*)entries, n, hot_list, it0 + j == n_iterations ? 1u : 0u);
            hipLaunchKernelGGL(k_sppm_gather_hot, g_shade, blk, 0, st, scene->dev, rec, vp, px, grid, (const uint32_t*)starts, (const float4*)entries, n, (const uint32_t*)hot_list);
            tm.end(2, st);
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
            tm.begin(2, st);
            hipLaunchKernelGGL(k_sppm_update, g_pix, blk, 0, st, n, gamma, px, vp);
            tm.end(2, st);
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
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        stats->ms_total = ms;
        stats->ms_raygen = tm.total(0, &stats->launches_raygen);
        stats->ms_trace_closest = tm.total(1, &stats->launches_trace_closest);
        stats->ms_shade = tm.total(2, &stats->launches_shade);
        stats->ms_trace_any = tm.total(3, &stats->launches_trace_any);
        stats->ms_film = tm.total(4, &stats->launches_film);
        stats->n_batches = n_batches;
        stats->max_depth_reached = (uint32_t)max_depth;
        traversal_info(ctx, scene, &stats->traversal, &stats->node_bytes);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return 0;
}

}  // namespace

// ---- C ABI ------------------------------------------------------------------------------------------------------------------------
extern "C" {

int trhip_version(void) { return 2000; }

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
                      &ctx->sp_rec_valid})
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
    if (!ctx->comm.comm) {
        if (ctx->comm.n_ranks == 1) return 0;  // a single-process job: the film already is the sum
        return fail(ctx, TRHIP_ERR_INVALID, "no communicator: call trhip_comm_init first");
    }
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
    else if (!std::strcmp(name, "bvh_builder"))
        ctx->bvh_builder = value < 0 ? -1 : (value != 0 ? 1 : 0);
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
    else if (!std::strcmp(name, "leaf_kernel"))
        ctx->leaf_kernel = value != 0;
    else if (!std::strcmp(name, "slab_margin_log2"))
        ctx->slab_margin_log2 = (int)std::max<int64_t>(0, std::min<int64_t>(20, value));
    else if (!std::strcmp(name, "tiny_scene_prims"))
        ctx->tiny_scene_prims = (uint32_t)std::max<int64_t>(0, std::min<int64_t>(255, value));
    else if (!std::strcmp(name, "film_block"))
        ctx->film_block = (int)std::max<int64_t>(0, std::min<int64_t>(3, value));
    else if (!std::strcmp(name, "film_tiled"))
        ctx->film_tiled = value != 0;
    else if (!std::strcmp(name, "pipelines")) {
        if (value < 1 || value > kMaxPipes) return fail(ctx, TRHIP_ERR_INVALID, "pipelines must be in 1..%d", kMaxPipes);
        ctx->pipelines = (int)value;
    } else if (!std::strcmp(name, "overlap"))
        ctx->overlap = value != 0;
    else if (!std::strcmp(name, "traversal")) {
        if (value < 1 || value > 6 || value == 5) return fail(ctx, TRHIP_ERR_INVALID, "traversal must be 1, 2, 3, 4 or 6");
        ctx->traversal = (int)value;
    } else if (!std::strcmp(name, "batch_paths")) {
        if (value < 0) return fail(ctx, TRHIP_ERR_INVALID, "batch_paths must be >= 0 (0 = auto)");
        ctx->batch_paths = (uint64_t)value;
    } else
        return fail(ctx, TRHIP_ERR_INVALID, "unknown option %s", name);
    return 0;
}

int trhip_scene_new(trhip_ctx* ctx, trhip_scene** out) {
    if (!ctx || !out) return fail(ctx, TRHIP_ERR_INVALID, "null argument");
    auto s = new trhip_scene();
    s->ctx = ctx;
    *out = s;
    return 0;
}
void trhip_scene_free(trhip_scene* s) {
    if (!s) return;
    (void)hipSetDevice(s->ctx->device);
    release(s->d_nodes);
    release(s->d_prims);
    release(s->d_nrm);
    release(s->d_shade);
    release(s->d_leaf_order);
    release(s->d_spheres);
    release(s->d_materials);
    release(s->d_lights);
    release(s->d_wnodes);
    release(s->d_occ_slots);
    release(s->d_occ_boxes);
    release(s->d_w8nodes);
    release(s->d_w8tris);
    delete s;
}
int trhip_scene_add_material(trhip_scene* s, int kind, const float* params, int n_params, uint32_t* id_out) {
    if (!s || !params) return fail(s ? s->ctx : nullptr, TRHIP_ERR_INVALID, "null argument");
    MaterialRec m;
    if (build_material(kind, params, n_params, m)) return fail(s->ctx, TRHIP_ERR_INVALID, "bad material kind %d / parameter count %d", kind, n_params);
    if (s->materials.size() >= PRIM_NO_MATERIAL) return fail(s->ctx, TRHIP_ERR_INVALID, "too many materials");
    s->materials.push_back(m);
    if (id_out) *id_out = (uint32_t)s->materials.size() - 1;
    s->committed = false;
    return 0;
}
int trhip_scene_add_triangles(trhip_scene* s, const float* xyz, uint32_t n_verts, const uint32_t* idx, uint32_t n_tris, const float* normals, const uint32_t* mat, int flip,
                              uint32_t* first_out) {
    if (!s || !xyz || !idx) return fail(s ? s->ctx : nullptr, TRHIP_ERR_INVALID, "null argument");
    const uint32_t first = (uint32_t)s->prims.size();
    s->prims.reserve(s->prims.size() + n_tris);
    for (uint32_t k = 0; k < n_tris; ++k) {
        HostPrim p;
        std::memset(&p, 0, sizeof p);
        p.kind = 0;
        for (int j = 0; j < 3; ++j) {
            const uint32_t vi = idx[3 * (size_t)k + j];
            if (vi < 1 || vi > n_verts) return fail(s->ctx, TRHIP_ERR_INVALID, "triangle %u: index %u outside 1..%u (indices are 1-based)", k, vi, n_verts);
            std::memcpy(&p.v[3 * j], &xyz[3 * (size_t)(vi - 1)], 3 * sizeof(float));
            if (normals) std::memcpy(&p.n[3 * j], &normals[3 * (size_t)(vi - 1)], 3 * sizeof(float));
        }
        uint32_t m = mat ? mat[k] : PRIM_NO_MATERIAL;
        if (mat && m != PRIM_NO_MATERIAL && m >= s->materials.size()) return fail(s->ctx, TRHIP_ERR_INVALID, "triangle %u: material %u not defined", k, m);
        // is_degenerate (triangle_mesh.jl:65-68) depends on the triangle alone: evaluated here, once, in the kernels' arithmetic
        const f3 tv0 = mk3(p.v[0], p.v[1], p.v[2]), tv1 = mk3(p.v[3], p.v[4], p.v[5]), tv2 = mk3(p.v[6], p.v[7], p.v[8]);
        const f3 tn = cross(tv2 - tv0, tv1 - tv0);
        const bool degenerate = dot(tn, tn) == 0.0f;
        p.meta = (m & PRIM_MATERIAL_MASK) | (normals ? PRIM_HAS_NORMALS : 0u) | (flip ? PRIM_FLIP : 0u) | (degenerate ? PRIM_DEGENERATE : 0u);
        s->prims.push_back(p);
    }
    if (first_out) *first_out = first;
    s->committed = false;
    return 0;
}
static int add_sphere_rec(trhip_scene* s, const float* o2w, const float* o2w_inv, int reverse, SphereRec r, uint32_t material, uint32_t* prim_out) {
    if (material != PRIM_NO_MATERIAL && material >= s->materials.size()) return fail(s->ctx, TRHIP_ERR_INVALID, "material %u not defined", material);
    std::memcpy(r.o2w, o2w, sizeof r.o2w);
    std::memcpy(r.o2w_inv, o2w_inv, sizeof r.o2w_inv);
    const bool swaps = det3(o2w) < 0.0f;  // transformations.jl:161-163
    r.flip = ((reverse != 0) != swaps) ? 1u : 0u;
    r.never_clipped = (!(r.z_min > -r.radius) && !(r.z_max < r.radius) && r.phi_max >= 2.0f * kPi) ? 1u : 0u;
    if (!r.never_clipped) s->partial_spheres = true;
    HostPrim p;
    std::memset(&p, 0, sizeof p);
    p.kind = 1;
    p.sphere_id = (uint32_t)s->spheres.size();
    p.meta = (material & PRIM_MATERIAL_MASK) | PRIM_SPHERE;
    s->spheres.push_back(r);
    s->sphere_bounds.push_back(sphere_world_bound(r));
    if (prim_out) *prim_out = (uint32_t)s->prims.size();
    s->prims.push_back(p);
    s->committed = false;
    return 0;
}
int trhip_scene_add_sphere(trhip_scene* s, const float* o2w, const float* o2w_inv, int reverse, float radius, float z_min, float z_max, float phi_max_deg, uint32_t material,
                           uint32_t* prim_out) {
    if (!s || !o2w || !o2w_inv) return fail(s ? s->ctx : nullptr, TRHIP_ERR_INVALID, "null argument");
    SphereRec r;
    std::memset(&r, 0, sizeof r);
    r.radius = radius;  // Sphere ctor sphere.jl:13-26
    r.z_min = jclamp(jmin(z_min, z_max), -radius, radius);
    r.z_max = jclamp(jmax(z_min, z_max), -radius, radius);
    r.theta_min = tm_acosf(jclamp(jmin(z_min, z_max) / radius, -1.0f, 1.0f));
    r.theta_max = tm_acosf(jclamp(jmax(z_min, z_max) / radius, -1.0f, 1.0f));
    r.phi_max = deg2rad(jclamp(phi_max_deg, 0.0f, 360.0f));
    return add_sphere_rec(s, o2w, o2w_inv, reverse, r, material, prim_out);
}
int trhip_scene_add_sphere_fields(trhip_scene* s, const float* o2w, const float* o2w_inv, int reverse, float radius, float z_min, float z_max, float theta_min, float theta_max,
                                  float phi_max_rad, uint32_t material, uint32_t* prim_out) {
    if (!s || !o2w || !o2w_inv) return fail(s ? s->ctx : nullptr, TRHIP_ERR_INVALID, "null argument");
    SphereRec r;
    std::memset(&r, 0, sizeof r);
    r.radius = radius;
    r.z_min = z_min;
    r.z_max = z_max;
    r.theta_min = theta_min;
    r.theta_max = theta_max;
    r.phi_max = phi_max_rad;
    return add_sphere_rec(s, o2w, o2w_inv, reverse, r, material, prim_out);
}
static int add_light(trhip_scene* s, int kind, const float* l2w, const float* l2w_inv, const float* I, float total_deg, float falloff_deg, bool fields = false) {
    if (!s || !l2w || !l2w_inv || !I) return fail(s ? s->ctx : nullptr, TRHIP_ERR_INVALID, "null argument");
    LightRec l;
    std::memset(&l, 0, sizeof l);
    l.kind = kind;
    const f3 pos = xf_point(l2w, splat3(0.0f));  // light_to_world(Point3f(0)) point.jl:23, spot.jl:16
    l.position[0] = pos.x;
    l.position[1] = pos.y;
    l.position[2] = pos.z;
    std::memcpy(l.I, I, 3 * sizeof(float));
    if (kind == 1) {
        l.cos_total_width = fields ? total_deg : tm_cosf(deg2rad(total_deg));  // spot.jl:17
        l.cos_falloff_start = fields ? falloff_deg : tm_cosf(deg2rad(falloff_deg));
    }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            l.w2l[3 * r + c] = l2w_inv[4 * r + c];  // world_to_light = inv(light_to_world): .m = inv_m
            l.l2w[3 * r + c] = l2w[4 * r + c];
        }
    s->lights.push_back(l);
    s->committed = false;
    return 0;
}
int trhip_scene_add_point_light(trhip_scene* s, const float* l2w, const float* l2w_inv, const float* I) { return add_light(s, 0, l2w, l2w_inv, I, 0, 0); }
int trhip_scene_add_spot_light(trhip_scene* s, const float* l2w, const float* l2w_inv, const float* I, float total_deg, float falloff_deg) {
    return add_light(s, 1, l2w, l2w_inv, I, total_deg, falloff_deg);
}

int trhip_scene_add_spot_light_fields(trhip_scene* s, const float* l2w, const float* l2w_inv, const float* I, float cos_total, float cos_falloff) {
    return add_light(s, 1, l2w, l2w_inv, I, cos_total, cos_falloff, true);
}

int trhip_scene_commit(trhip_scene* s, int max_node_primitives) {
    if (!s) return fail(nullptr, TRHIP_ERR_INVALID, "null scene");
    HIP_TRY(s->ctx, hipSetDevice(s->ctx->device));
    std::vector<HostAABB> pb(s->prims.size());
    for (size_t i = 0; i < s->prims.size(); ++i) {
        const HostPrim& p = s->prims[i];
        if (p.kind == 1) {
            pb[i] = s->sphere_bounds[p.sphere_id];
        } else {  // world_bound(triangle) triangle_mesh.jl:97
            pb[i].reset();
            for (int j = 0; j < 3; ++j) pb[i].grow_point(&p.v[3 * j]);
        }
    }
    // Scenes with a few spheres beside a mesh: a chain root -> {sphere 1, {sphere 2, ... {sphere k, the triangles' subtree}}}.  Any BVH2 is a valid
    // BVHAccel (results depend on the topology only through exact-t ties); this one keeps the spheres — whose fp32
    // quadratic accepts rays far outside their box and can raise t_max (A.18) — out of the triangles' subtree, which the 8-wide
    // kernel then walks with conservative interior boxes (th_wide8.h).  The leaf-size hint is a hint (bvh.jl:159-165 decides by cost).
    std::vector<uint32_t> sph_ids, tri_ids;
    for (size_t i = 0; i < s->prims.size(); ++i) (s->prims[i].kind == 1 ? sph_ids : tri_ids).push_back((uint32_t)i);
    const bool want_chain = s->ctx->compose_spheres > 0 || (s->ctx->compose_spheres < 0 && s->ctx->traversal == 4);
    const bool compose = want_chain && !sph_ids.empty() && sph_ids.size() <= (size_t)kW8MaxSpheres && tri_ids.size() >= 2 && pb.size() > s->ctx->tiny_scene_prims;
    std::vector<HostAABB> pb_sub;
    if (compose) {
        pb_sub.reserve(tri_ids.size());
        for (uint32_t id : tri_ids) pb_sub.push_back(pb[id]);
    }
    const std::vector<HostAABB>& pb_build = compose ? pb_sub : pb;
    bool built = false;
    const int mode = s->ctx->bvh_builder;
    if ((mode == 1 || (mode < 0 && pb_build.size() > (16u << 20))) && pb_build.size() > s->ctx->tiny_scene_prims) {
        FlatBVH dev;
        const int rc = build_bvh_device(s->ctx, pb_build, dev);
        if (rc == 0) {
            s->bvh = std::move(dev);
            built = true;
        } else if (rc != TRHIP_ERR_UNSUPPORTED) {
            return rc;
        }
    }
    if (!built) {
        BVHBuilder builder(pb_build, max_node_primitives, s->ctx->tiny_scene_prims, want_chain);  // traversal 4 wants one primitive per leaf
        s->bvh = builder.build();
    }
    if (compose) {
        // flat layout (bvh.jl:187-206): chain node i at 2 i = interior {leaf of sphere i at 2 i + 1, rest at 2 i + 2}; the triangles' subtree at 2 n_sph
        FlatBVH sub = std::move(s->bvh), out;
        const uint32_t n_sph = (uint32_t)sph_ids.size(), n_sub = (uint32_t)sub.a.size();
        std::vector<HostAABB> rest(n_sph + 1);
        std::memcpy(rest[n_sph].mn, &sub.bounds[0], 3 * sizeof(float));
        std::memcpy(rest[n_sph].mx, &sub.bounds[3], 3 * sizeof(float));
        for (uint32_t i = n_sph; i-- > 0;) {
            rest[i] = rest[i + 1];
            rest[i].grow(pb[sph_ids[i]]);
        }
        // one split axis for every chain node (it only decides whether a ray meets the sphere leaves before or after the triangles): where the
        // spheres' centre and the triangles' lie furthest apart
        HostAABB sall;
        sall.reset();
        for (uint32_t id : sph_ids) sall.grow(pb[id]);
        uint32_t axis = 0;
        float best = -1.0f;
        for (int a = 0; a < 3; ++a) {
            const float dc = std::fabs((0.5f * sall.mn[a] + 0.5f * sall.mx[a]) - (0.5f * rest[n_sph].mn[a] + 0.5f * rest[n_sph].mx[a]));
            if (dc > best) {
                best = dc;
                axis = (uint32_t)a;
            }
        }
        for (uint32_t i = 0; i < n_sph; ++i) {
            const HostAABB& sbx = pb[sph_ids[i]];
            out.bounds.insert(out.bounds.end(), {rest[i].mn[0], rest[i].mn[1], rest[i].mn[2], rest[i].mx[0], rest[i].mx[1], rest[i].mx[2]});
            out.a.push_back(2 * i + 2);
            out.flags.push_back(axis);
            out.bounds.insert(out.bounds.end(), {sbx.mn[0], sbx.mn[1], sbx.mn[2], sbx.mx[0], sbx.mx[1], sbx.mx[2]});
            out.a.push_back(i);
            out.flags.push_back((1u << 2) | 3u);
        }
        out.bounds.insert(out.bounds.end(), sub.bounds.begin(), sub.bounds.end());
        out.a.reserve(n_sub + 2 * n_sph);
        out.flags.reserve(n_sub + 2 * n_sph);
        for (uint32_t i = 0; i < n_sub; ++i) {
            out.a.push_back(sub.a[i] + ((sub.flags[i] & 3u) == 3u ? n_sph : 2 * n_sph));
            out.flags.push_back(sub.flags[i]);
        }
        out.order = sph_ids;
        out.order.reserve(pb.size());
        for (uint32_t k : sub.order) out.order.push_back(tri_ids[k]);
        out.max_depth = sub.max_depth + n_sph;
        s->bvh = std::move(out);
    }
    if (s->bvh.max_depth > (uint32_t)(kStackLds + kStackSpill))
        return fail(s->ctx, TRHIP_ERR_UNSUPPORTED, "BVH depth %u exceeds the 64-entry traversal stack (bvh.jl:222)", s->bvh.max_depth);
    s->literal_only = false;
    return upload_scene(s);
}
int trhip_scene_bvh_size(const trhip_scene* s, uint32_t* n_nodes, uint32_t* n_prims) {
    if (!s) return TRHIP_ERR_INVALID;
    if (n_nodes) *n_nodes = (uint32_t)s->bvh.a.size();
    if (n_prims) *n_prims = (uint32_t)s->bvh.order.size();
    return 0;
}
int trhip_scene_get_bvh(const trhip_scene* s, float* bounds, uint32_t* a, uint32_t* flags, uint32_t* order) {
    if (!s) return TRHIP_ERR_INVALID;
    if (bounds) std::memcpy(bounds, s->bvh.bounds.data(), s->bvh.bounds.size() * sizeof(float));
    if (a) std::memcpy(a, s->bvh.a.data(), s->bvh.a.size() * sizeof(uint32_t));
    if (flags) std::memcpy(flags, s->bvh.flags.data(), s->bvh.flags.size() * sizeof(uint32_t));
    if (order) std::memcpy(order, s->bvh.order.data(), s->bvh.order.size() * sizeof(uint32_t));
    return 0;
}
int trhip_scene_set_bvh(trhip_scene* s, const float* bounds, const uint32_t* a, const uint32_t* flags, uint32_t n_nodes, const uint32_t* order, uint32_t n_prims) {
    if (!s || !bounds || !a || !flags || !order) return fail(s ? s->ctx : nullptr, TRHIP_ERR_INVALID, "null argument");
    if (n_nodes == 0) return fail(s->ctx, TRHIP_ERR_INVALID, "empty node array");
    for (uint32_t i = 0; i < n_prims; ++i)
        if (order[i] >= s->prims.size()) return fail(s->ctx, TRHIP_ERR_INVALID, "prim_order[%u] = %u out of range", i, order[i]);
    // The array must be ONE tree in the reference's depth-first layout (bvh.jl:187-206): the subtree of node i is the index range
    // [i, end): first child i + 1 .. a[i] - 1, second child a[i] .. end - 1.  Anything else (a[i] <= i + 1 closes a cycle: the
    // traversal kernels would never end) is rejected here; so is a tree deeper than the 64-entry stack, where the reference
    // throws a BoundsError (bvh.jl:222).
    struct Span {
        uint32_t node, end, depth;
    };
    std::vector<Span> todo;
    todo.push_back({0u, n_nodes, 1u});
    uint32_t max_depth = 0;
    bool nested = true;  // every child box inside its parent's, every primitive's bound inside its leaf box
    auto inside = [&](const float* in, const float* out) {
        return in[0] >= out[0] && in[1] >= out[1] && in[2] >= out[2] && in[3] <= out[3] && in[4] <= out[4] && in[5] <= out[5];
    };
    while (!todo.empty()) {
        const Span sp = todo.back();
        todo.pop_back();
        const uint32_t i = sp.node;
        max_depth = std::max(max_depth, sp.depth);
        if ((flags[i] & 3u) == 3u) {
            if (sp.end != i + 1) return fail(s->ctx, TRHIP_ERR_INVALID, "leaf %u is followed by nodes that belong to no subtree (not a depth-first layout)", i);
            const uint32_t cnt = flags[i] >> 2;
            if ((uint64_t)a[i] + cnt > n_prims) return fail(s->ctx, TRHIP_ERR_INVALID, "leaf %u references primitives outside the list", i);
            for (uint32_t k = a[i]; k < a[i] + cnt && nested; ++k) {
                const HostPrim& p = s->prims[order[k]];
                HostAABB pb;
                if (p.kind == 1) {
                    pb = s->sphere_bounds[p.sphere_id];
                } else {
                    pb.reset();
                    for (int j = 0; j < 3; ++j) pb.grow_point(&p.v[3 * j]);
                }
                const float pbox[6] = {pb.mn[0], pb.mn[1], pb.mn[2], pb.mx[0], pb.mx[1], pb.mx[2]};
                nested = inside(pbox, &bounds[6 * (size_t)i]);
            }
            continue;
        }
        if (a[i] <= i + 1 || a[i] >= sp.end)
            return fail(s->ctx, TRHIP_ERR_INVALID, "interior node %u: second child %u outside (%u, %u) — not the depth-first layout of bvh.jl:187-206", i, a[i], i + 1, sp.end);
        if ((flags[i] & 3u) > 2u) return fail(s->ctx, TRHIP_ERR_INVALID, "interior node %u: split axis %u", i, flags[i] & 3u);
        nested = nested && inside(&bounds[6 * (size_t)(i + 1)], &bounds[6 * (size_t)i]) && inside(&bounds[6 * (size_t)a[i]], &bounds[6 * (size_t)i]);
        todo.push_back({a[i], sp.end, sp.depth + 1});
        todo.push_back({i + 1, a[i], sp.depth + 1});
    }
    if (max_depth > (uint32_t)(kStackLds + kStackSpill))
        return fail(s->ctx, TRHIP_ERR_UNSUPPORTED, "BVH depth %u exceeds the 64-entry traversal stack (bvh.jl:222 throws a BoundsError there)", max_depth);
    s->bvh.bounds.assign(bounds, bounds + 6 * (size_t)n_nodes);
    s->bvh.a.assign(a, a + n_nodes);
    s->bvh.flags.assign(flags, flags + n_nodes);
    s->bvh.order.assign(order, order + n_prims);
    s->bvh.max_depth = max_depth;
    // The default kernels' shortcuts (tight slab clauses, largest-triangle pre-pass, wide nodes: th_trace2.h, th_trace8.h) are exact
    // only when boxes nest; a foreign tree that does not is walked by the literal kernels (the reference's loop, op for op).
    s->literal_only = !nested;
    HIP_TRY(s->ctx, hipSetDevice(s->ctx->device));
    return upload_scene(s);
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
int trhip_render_sppm(trhip_ctx* ctx, const trhip_scene* sc, const trhip_sensor* sn, float initial_search_radius, int max_depth, uint32_t n_iterations, int64_t photons_per_iteration,
                      uint64_t seed, float* out_xyzw, trhip_stats* st) {
    return render_sppm_impl(ctx, sc, sn, initial_search_radius, max_depth, n_iterations, photons_per_iteration, seed, out_xyzw, st);
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
int trhip_last_sample_radiance(trhip_ctx* ctx, float* out, uint64_t n_floats) {
    if (!ctx || !out) return fail(ctx, TRHIP_ERR_INVALID, "null argument");
    if (n_floats != ctx->last_L_count * 3) return fail(ctx, TRHIP_ERR_INVALID, "expected %llu floats", (unsigned long long)(ctx->last_L_count * 3));
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (int rc = ensure(ctx, ctx->scratch[0], n_floats * sizeof(float))) return rc;
    const uint64_t n = ctx->last_L_count;
    if (n) hipLaunchKernelGGL(k_export_L, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const float4*)ctx->Lbuf.p, n, (float*)ctx->scratch[0].p);
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
#ifdef TH_DIAG_PHASES
extern "C" int trhip_debug_phases(uint64_t* out12, int reset) {  // DIAGNOSTIC build only (tools/phase_probe.py)
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
    launch_film(ctx, ctx->stream, ds, (const DeviceSensor*)ctx->sensor.p, (const float4*)ctx->Lbuf.p, n, spp, seed, sample_offset, (float4*)ctx->film.p);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->last_L_count = n;
    HIP_TRY(ctx, hipMemcpy(out_xyzw, ctx->film.p, film_bytes, hipMemcpyDeviceToHost));
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
