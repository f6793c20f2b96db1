/* tracehip.h — C ABI of libtracehip.so, the MI355X (gfx950) wavefront ray/path-tracing engine behind Trace.jl's
 * Integrator / Sampler / Film surface.
 *
 * The reference (pxl-th/Trace.jl @ 2024_10_08) has no FFI layer: its "operator API" is Julia dispatch on a handful of
 * types.  Each entry point below names the reference interface (file:line under /root/reference/src)
 * it replaces.  A Julia shim (trace.jl_amd/julia/TraceHIP.jl) `ccall`s these after walking
 * Scene -> BVHAccel -> GeometricPrimitive; the tested host in this repo is the Python mirror in trace.jl_amd/.
 *
 * Conventions: every function returns 0 on success and a negative trhip_status on failure; the message is available
 * from trhip_last_error().  Nothing throws across the boundary.  Host pointers are caller-owned and only borrowed for
 * the duration of the call.  Calls are blocking (internal HIP streams are synchronised before returning).  One
 * trhip_ctx owns one GPU; use one process per GPU.  Several processes form one job through trhip_comm_init (RCCL over xGMI):
 * trhip_film_reduce sums the per-rank film accumulators, trhip_render_sppm shards its photons (multi-GPU section below).
 * Matrices are 16 floats, row-major: m[4*row + col] (Julia's Mat4f is column-major: pass transpose / permutedims).
 * All arithmetic is IEEE Float32 without FMA contraction; transcendental functions and the sampler are the ones
 * specified in trace_detmath.h / trace_sampler.h, so results are reproducible bit-for-bit on any conforming host.
 */
#ifndef TRACEHIP_H
#define TRACEHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct trhip_ctx trhip_ctx;
typedef struct trhip_scene trhip_scene;

typedef enum {
    TRHIP_OK = 0,
    TRHIP_ERR_INVALID = -1,    /* bad argument / call order */
    TRHIP_ERR_HIP = -2,        /* HIP runtime error (no device, out of memory, launch failure) */
    TRHIP_ERR_UNSUPPORTED = -3 /* feature of the reference that this build does not accelerate */
} trhip_status;

/* ---- context ------------------------------------------------------------------------------------------------------- */
/* Fails (TRHIP_ERR_HIP) when no gfx950 device is visible: there is no CPU fallback. */
int trhip_init(trhip_ctx** ctx, int device_id);
void trhip_shutdown(trhip_ctx* ctx);
/* ctx may be NULL to read the message of a failed trhip_init. */
const char* trhip_last_error(const trhip_ctx* ctx);
/* ABI version of this header: major*1000 + minor. */
int trhip_version(void);  /* 3001 */

/* ---- scene flattening (replaces the Scene / BVHAccel / GeometricPrimitive object graph) --------------------------- */
/* Scene(lights, aggregate)  Trace.jl:176-187 */
int trhip_scene_new(trhip_ctx* ctx, trhip_scene** out);
void trhip_scene_free(trhip_scene* scene);

/* Materials with ConstantTexture arguments (textures/basic.jl:4-10), materials/material.jl:
 *   TRHIP_MATTE   (:1-31)    params = Kd.rgb, sigma                                           (4)
 *   TRHIP_MIRROR  (:34-46)   params = Kr.rgb                                                  (3)
 *   TRHIP_GLASS   (:49-116)  params = Kr.rgb, Kt.rgb, u_roughness, v_roughness, index, remap  (10)
 *   TRHIP_PLASTIC (:119-151) params = Kd.rgb, Ks.rgb, roughness, remap                        (8) */
enum { TRHIP_MATTE = 0, TRHIP_MIRROR = 1, TRHIP_GLASS = 2, TRHIP_PLASTIC = 3 };
int trhip_scene_add_material(trhip_scene* scene, int kind, const float* params, int n_params, uint32_t* id_out);

/* create_triangle_mesh + one GeometricPrimitive per triangle  (shapes/triangle_mesh.jl:45-58, primitive.jl:5-9).
 * world_xyz: vertices already moved to world space by the host exactly as TriangleMesh does (triangle_mesh.jl:23);
 * normals are passed untransformed (triangle_mesh.jl:23-28) or NULL; indices are 1-based as in the reference;
 * flip_orientation = reverse_orientation XOR transform_swaps_handedness of the ShapeCore (shapes/Shape.jl:7-14);
 * material_id_per_tri may be NULL only for geometry-only scenes (kernel-level trace entry points).
 * first_prim_out receives the index of the first created primitive in caller order. */
int trhip_scene_add_triangles(trhip_scene* scene, const float* world_xyz, uint32_t n_verts, const uint32_t* indices_1based, uint32_t n_tris,
                              const float* normals_or_null, const uint32_t* material_id_per_tri, int flip_orientation, uint32_t* first_prim_out);

/* The same with the mesh's two optional per-vertex arrays (shapes/triangle_mesh.jl:1-30, create_triangle_mesh's last two arguments):
 * tangents: n_verts x 3, untransformed like the normals, read through the indices (:76-78) — the shading tangent of a hit is their barycentric
 *   mix (:172-176) instead of ∂p∂u; a mesh with tangents and no normals still gets shading geometry (:165);
 * uv: the reference reads `mesh.uv[t.i + j]` (:82) — indexed by the triangle's CORNER position 3k + j in the index list, not through the indices —
 *   so the array holds 3 x n_tris points (2 floats each), corner-major; ∂p∂u / ∂p∂v (:125-141) and interaction.uv follow from it.
 * Either pointer may be NULL (both NULL = trhip_scene_add_triangles). */
int trhip_scene_add_triangles_ex(trhip_scene* scene, const float* world_xyz, uint32_t n_verts, const uint32_t* indices_1based, uint32_t n_tris,
                                 const float* normals_or_null, const float* tangents_or_null, const float* uv_corners_or_null,
                                 const uint32_t* material_id_per_tri, int flip_orientation, uint32_t* first_prim_out);

/* Sphere(core, radius, z_min, z_max, ϕ_max°) + GeometricPrimitive  (shapes/sphere.jl:1-30).  Both matrices of
 * core.object_to_world (m and inv_m, transformations.jl:1-4) are passed because the reference keeps them separately
 * (and multiplies inverses in a non-standard order, transformations.jl:20-22). */
int trhip_scene_add_sphere(trhip_scene* scene, const float obj2world_m[16], const float obj2world_inv_m[16], int reverse_orientation, float radius,
                           float z_min, float z_max, float phi_max_deg, uint32_t material_id, uint32_t* prim_out);

/* Same, for hosts that hold an already constructed Trace.Sphere (the Julia shim): the fields exactly as the reference's
 * constructor derived them (sphere.jl:13-26: clamped z_min/z_max, θ_min, θ_max, ϕ_max in radians), so that no
 * elementary function is re-evaluated on this side of the boundary. */
int trhip_scene_add_sphere_fields(trhip_scene* scene, const float obj2world_m[16], const float obj2world_inv_m[16], int reverse_orientation, float radius,
                                  float z_min, float z_max, float theta_min, float theta_max, float phi_max_rad, uint32_t material_id, uint32_t* prim_out);

/* PointLight(light_to_world, I)  lights/point.jl:19-24 ;  SpotLight(light_to_world, I, total°, falloff_start°)  lights/spot.jl:10-19 */
int trhip_scene_add_point_light(trhip_scene* scene, const float light2world_m[16], const float light2world_inv_m[16], const float I[3]);
int trhip_scene_add_spot_light(trhip_scene* scene, const float light2world_m[16], const float light2world_inv_m[16], const float I[3],
                               float total_width_deg, float falloff_start_deg);

/* SpotLight from its constructed fields (cos_total_width, cos_falloff_start; lights/spot.jl:1-8). */
int trhip_scene_add_spot_light_fields(trhip_scene* scene, const float light2world_m[16], const float light2world_inv_m[16], const float I[3],
                                      float cos_total_width, float cos_falloff_start);

/* BVHAccel(primitives, max_node_primitives)  accel/bvh.jl:55-79.  Builds a binned-SAH BVH2 on the host (results of
 * traversal do not depend on the topology except where two primitives are accepted at (nearly) the same t), flattens it
 * in the reference's depth-first layout (first child = i+1, bvh.jl:187-206) and uploads everything to HBM.  Option "bvh_builder" = 2
 * builds the reference's OWN tree instead (its 12-bucket construction, quirks included): same tie-breaks as Trace.jl; a host that
 * already holds Trace.jl's BVHAccel hands its nodes over with trhip_scene_set_bvh (what TraceHIP.jl does). */
int trhip_scene_commit(trhip_scene* scene, int max_node_primitives);

/* BVHAccel construction alone, on the host (no GPU needed): builder 0 = the library's binned SAH, 2 = the REFERENCE's construction node for node
 * (accel/bvh.jl:87-206 + partition! Trace.jl:128-137, quirks kept — SURVEY.md A.6; what trhip_scene_commit builds under option "bvh_builder" = 2).
 * prim_bounds: n_prims * 6 (world_bound of each primitive: min xyz, max xyz).  Outputs in the layout of trhip_scene_get_bvh; *n_nodes_inout = capacity
 * of the node arrays on entry, the tree's node count on return (all output pointers NULL: size query).  The reference's builder may emit leaves of 0
 * primitives with bounds (+Inf, -Inf): up to ~3 n nodes.  TRHIP_ERR_UNSUPPORTED where the reference's recursion would not end. */
int trhip_build_bvh_host(int builder, const float* prim_bounds, uint32_t n_prims, int max_node_primitives, float* node_bounds, uint32_t* node_a, uint32_t* node_flags,
                         uint32_t* n_nodes_inout, uint32_t* prim_order, uint32_t* max_depth_out);

/* Inspection of the committed BVH (tests feed the same topology to the CPU oracle so that parity is bit-exact).
 * node_bounds: n_nodes*6 (min xyz, max xyz).  Leaf: (flags & 3) == 3, a = first ordered-primitive slot, n = flags >> 2.
 * Interior: a = index of the second child, flags & 3 = split axis (0..2), first child = i + 1.
 * prim_order[slot] = caller primitive index.  Any output pointer may be NULL. */
int trhip_scene_bvh_size(const trhip_scene* scene, uint32_t* n_nodes, uint32_t* n_prims);
int trhip_scene_get_bvh(const trhip_scene* scene, float* node_bounds, uint32_t* node_a, uint32_t* node_flags, uint32_t* prim_order);
/* Replace the BVH by a caller-supplied one in the same layout (e.g. the reference's own builder run elsewhere). */
int trhip_scene_set_bvh(trhip_scene* scene, const float* node_bounds, const uint32_t* node_a, const uint32_t* node_flags, uint32_t n_nodes,
                        const uint32_t* prim_order, uint32_t n_prims);

/* Hybrid mode (option "bvh_builder" = 4, and the default): the scene holds TWO trees over the same primitives.  The CANONICAL one — the reference's own construction
 * (accel/bvh.jl:55-206, or the tree a host handed to trhip_scene_set_bvh) — defines the answers: slots, trhip_scene_get_bvh, shading records are its.  The library's
 * binned-SAH tree rides along as the ACCELERATOR: closest-hit rays walk it and carry a certificate that every valid tree over the same leaves returns the same hit;
 * rays without the certificate (a sphere entered from inside, sphere.jl:137-138; two acceptable primitives within 512 ulps of the ray's reach; a grazed leaf box;
 * a zero direction component) are re-walked on the canonical tree in the reference's order (trhip_stats.fallback_rays).  Results equal a walk of the canonical tree
 * alone bit for bit (option "hybrid" = 0 runs exactly that walk, for A/B).
 * trhip_scene_bvh_mode: *mode = 0 the library's tree alone (bvh_builder 0 / 1 / 3), 1 the canonical tree alone (bvh_builder 2; or a scene whose accelerator could not be
 *   certified: leaf boxes that differ between the trees), 2 both, 3 the library's tree as the canonical tree AND, four children wide, as its own accelerator (a default
 *   commit on a scene where the reference's construction fails: trhip_scene_bvh_note says why); *accel_nodes / *accel_depth describe the accelerator (0 without one).
 * trhip_scene_get_accelerator: the accelerator in the layout of trhip_scene_get_bvh (prim_order[accelerator slot] = caller primitive index); size it with
 *   trhip_scene_bvh_mode.  Any output pointer may be NULL. */
int trhip_scene_bvh_mode(const trhip_scene* scene, int* mode, uint32_t* accel_nodes, uint32_t* accel_depth);
/* The name of the kernel a closest-hit launch on this scene runs under the context's current options ("k_trace3c4": the certified walk on the accelerator four children wide,
 * "k_trace3c": the same on the binary accelerator — wide4 = 0, or an accelerator whose four-wide form does not fit the stack —, "k_trace_leaf_c": a one-leaf accelerator, "k_trace3",
 * "k_trace_leaf", …): for profile filters and rooflines, which must name the kernel that ran. */
int trhip_closest_kernel_name(const trhip_ctx* ctx, const trhip_scene* scene, char* buf, size_t n);
/* Why a scene committed with default options holds ONE tree (mode 0 or 1) instead of two: a NUL-terminated sentence copied into buf (at most n bytes; "" for mode 2 and for
 * explicit builders).  E.g. "the reference's construction: BVH depth 71 exceeds the 64-entry traversal stack (bvh.jl:222 throws a BoundsError there)" -> the library's tree alone. */
int trhip_scene_bvh_note(const trhip_scene* scene, char* buf, size_t n);
/* Why a scene that holds both trees is walked WITHOUT its accelerator under the context's current options ("" when it is used, or the scene holds one tree): such frames
 * are exact but walk the reference's tree alone, at about twice the closest-hit time.  The first such frame of a context also says so on stderr.  Also non-empty when the
 * accelerator IS used but the last frame that reported statistics handed more than a fifth of its closest-hit rays back to the reference-order walk (trhip_stats.fallback_rays /
 * closest_rays: scenes of near-ties — tiny coplanar triangles, rays in a primitive's plane): exact, at little gain; that frame says so once on stderr as well. */
int trhip_accelerator_note(const trhip_ctx* ctx, const trhip_scene* scene, char* buf, size_t n);
int trhip_scene_get_accelerator(const trhip_scene* scene, float* node_bounds, uint32_t* node_a, uint32_t* node_flags, uint32_t* prim_order);

/* ---- sensor: PerspectiveCamera + Film + filter (camera/perspective.jl:58-80, film.jl:34-61, filter.jl) ------------- */
typedef struct {
    float raster_to_camera[16]; /* camera.core.raster_to_camera.m — composed by the host constructors, bugs included (A.3, A.4) */
    float camera_to_world[16];  /* camera.core.core.camera_to_world.m */
    float lens_radius, focal_distance, shutter_open, shutter_close;
    float crop_min[2], crop_max[2]; /* Film.crop_bounds: 1-based inclusive pixel bounds (film.jl:41-44) */
    float filter_radius[2];         /* Film.filter.radius */
    float filter_table[256];        /* Film.filter_table, (y, x) order: table[16*y + x] (film.jl:38-40, 55-59) */
    float scale;                    /* Film.scale */
} trhip_sensor;

typedef struct {
    uint64_t camera_samples; /* camera samples completed */
    uint64_t closest_rays;   /* closest-hit rays traced, all bounces */
    uint64_t shadow_rays;    /* any-hit (shadow) rays traced, all bounces */
    uint64_t nodes_visited;  /* filled only when instrumentation is on (trhip_set_option "count_visits") */
    uint64_t prims_tested;
    uint64_t nodes_visited_shadow; /* any-hit rays: box and primitive RECORDS fetched — a record one scalar fetch brings to all 64 rays of */
    uint64_t prims_tested_shadow;  /* a wave (the any-hit pre-pass kernels of th_trace2.h) counts once                                   */
    double ms_total;         /* wall time of the render call's device work (HIP events) */
    double ms_raygen, ms_trace_closest, ms_shade, ms_trace_any, ms_film;
    uint32_t launches_raygen, launches_trace_closest, launches_shade, launches_trace_any, launches_film;
    uint32_t n_batches, max_depth_reached;
    uint32_t traversal;      /* traversal kernel that ran: 1 literal, 2, 3 binary children-in-parent walk, 4 8-wide nodes, 5 one-leaf scene, 9 hybrid: the certified
                                walk on the accelerator tree + the reference-order walk of the rays it hands back (trhip_scene_bvh_mode) */
    uint32_t node_bytes;     /* bytes fetched per unit of nodes_visited: 32 (a node box of the binary kernels: 64-byte node = 2 boxes),
                                96 (one 8-wide node: six 16-byte loads from one 128-byte line), 0 (one-leaf scene: scalar loads) */
    /* ---- since ABI 3000 ---- */
    uint64_t replicated_rays; /* of closest_rays + shadow_rays: rays that EVERY rank of a multi-GPU job traces identically (the camera pass of
                                 trhip_render_sppm, which only shards its photons); a job's ray total counts them once */
    uint64_t fallback_rays;   /* closest-hit rays the order-free walk (traversal 7) flagged — a second candidate within the tie margin, a box it
                                 could not decide, an origin inside a sphere — and handed to the reference-order walk (k_trace3) */
    double ms_sub[4];         /* parts of ms_shade.  trhip_render_sppm: [0] photon gather (k_sppm_gather + k_sppm_gather_hot), [1] camera / photon
                                 shading, [2] grid bounds + hit binning + scans, [3] pixel update + fold.  Path / Whitted: zeros */
    uint32_t launches_sub[4];
    uint64_t count_sub[4];    /* trhip_render_sppm with "count_visits": [0] (pixel, photon) candidates distance-tested by the gather, [1] pairs accepted
                                 (BSDF evaluated), [2] photon hits binned, [3] visible points.  trhip_render_path with traversal 7 and "count_visits": why rays went to
                                 the reference-order walk — [0] zero / non-finite direction, [1] a sphere (origin inside, limb, clipped), [2] a candidate within the gap of
                                 the ray's own t_max, [3] a second candidate within the gap of the nearest.  Hybrid mode (traversal 9) with "count_visits": why rays went to the
                                 canonical tree — [0] a zero / non-finite direction component or a near-axis-parallel direction, [1] a sphere (clipped; inside two at once),
                                 [2] a candidate within 2 dt of the incumbent or before its own leaf box's entry; and, NOT a fallback, [3] rays that start inside a sphere and were
                                 certified on the accelerator (sphere.jl:137-138 handled through the order word).  Otherwise zeros */
    /* ---- since ABI 3001: hybrid mode (traversal 9).  Of ms_trace_closest / nodes_visited / prims_tested, the part of the FALLBACK walks (k_trace3 over the rays the
       certified walk handed back, on the canonical tree); the certified walk on the accelerator tree (k_trace3c) is the difference ---- */
    double ms_fallback;
    uint32_t launches_fallback, reserved0;
    uint64_t nodes_visited_fallback, prims_tested_fallback;
} trhip_stats;

/* How trhip_render_path splits a frame whose per-sample buffers do not fit into `budget_bytes` (host arithmetic, no GPU): bands of whole rows of 16 x 16 sample tiles
 * (integrators/sampler.jl:15-24: tiles in k order — bands are ranges of k, so the film is the sequential loop's bit for bit).  bytes_per_sample: 17 (radiance record + poison byte: the
 * default film pass), 25 or 33 with the other film passes.  Writes up to `cap` bands: first sample row (film coordinates) and number of sample rows; *n_bands = how many there are. */
int trhip_plan_bands(const trhip_sensor* sensor, uint32_t spp, uint64_t budget_bytes, uint32_t bytes_per_sample, uint32_t cap, uint32_t* n_bands, int32_t* first_row, int32_t* n_rows);

/* ---- integrators (replace `integrator(scene)`, integrators/sampler.jl:12-56) ---------------------------------------
 * out_xyzw: (crop height) * (crop width) * 4 floats in film.pixels (y, x) order = Pixel.xyz sums + filter_weight_sum
 * (film.jl:7-11), i.e. exactly the state `save(film)` (film.jl:204-222) starts from.  NaN radiance samples are zeroed
 * (integrators/sampler.jl:46).  The sampler is the seeded counter-based sampler of trace_sampler.h with
 * `samples_per_pixel = spp`; `sample_offset` shifts the global sample indices (rank r of an N-GPU job renders indices
 * [r*spp, (r+1)*spp) and the host sum-reduces the films).
 * The *_device variants write to a DEVICE pointer (e.g. a torch tensor's data_ptr) instead of host memory. */
int trhip_render_whitted(trhip_ctx* ctx, const trhip_scene* scene, const trhip_sensor* sensor, uint32_t spp, int max_depth, uint64_t seed,
                         uint32_t sample_offset, float* out_xyzw, trhip_stats* stats);
/* PathIntegrator: not in the reference; defined in DESIGN.md from integrators/sppm.jl:208-266, 503-554. */
int trhip_render_path(trhip_ctx* ctx, const trhip_scene* scene, const trhip_sensor* sensor, uint32_t spp, int max_depth, uint64_t seed,
                      uint32_t sample_offset, float* out_xyzw, trhip_stats* stats);
int trhip_render_path_device(trhip_ctx* ctx, const trhip_scene* scene, const trhip_sensor* sensor, uint32_t spp, int max_depth, uint64_t seed,
                             uint32_t sample_offset, void* d_out_xyzw, trhip_stats* stats);
int trhip_render_whitted_device(trhip_ctx* ctx, const trhip_scene* scene, const trhip_sensor* sensor, uint32_t spp, int max_depth, uint64_t seed,
                                uint32_t sample_offset, void* d_out_xyzw, trhip_stats* stats);
/* Per-sample radiance of the last render call (after the NaN rule), n_sample_pixels * spp * 3 floats indexed
 * [(s * n_sample_pixels + (y - sb.min.y) * sb_width + (x - sb.min.x)) * 3 + c]; sample bounds sb = get_sample_bounds(film)
 * (film.jl:68-73).  Parity tests compare this with the oracle bit-for-bit. */
int trhip_last_sample_radiance(trhip_ctx* ctx, float* out_rgb, uint64_t n_floats);

/* SPPMIntegrator(camera, initial_search_radius, max_depth, n_iterations, photons_per_iteration)(scene)
 * (integrators/sppm.jl:108-173): per iteration a camera pass to the first diffuse vertex, a hash grid over the visible
 * points, a photon pass (Halton / radical_inverse, sampler/sampling.jl:43-60) and the Float64 pixel update; afterwards
 * _sppm_to_image + set_image! (film.jl:195-202).  out_xyzw: film_h * film_w * 4 (xyz, filter_weight_sum = 1).
 * photons_per_iteration <= 0: area(crop_bounds) like sppm.jl:121-124.  The film's crop must start at pixel (1, 1).
 * The camera pass of iteration k draws from the seeded stream (seed, pixel, sample k-1).  Photon contributions are added
 * with Float32 atomics as in the reference (sppm.jl:398-399): M, radius, N, Ld and the visible points are reproducible
 * bit for bit, τ (and the image) up to the summation order of ϕ. */
int trhip_render_sppm(trhip_ctx* ctx, const trhip_scene* scene, const trhip_sensor* sensor, float initial_search_radius, int max_depth, uint32_t n_iterations,
                      int64_t photons_per_iteration, uint64_t seed, float* out_xyzw, trhip_stats* stats);
/* … with the reference's periodic image (integrators/sppm.jl:166-171: `iteration % write_frequency == 0 || iteration == n_iterations` -> _sppm_to_image,
 * set_image!, save): after every iteration k < n_iterations that write_frequency divides, `write` receives the image of the first k iterations (in out_xyzw, which the
 * call owns until it returns: film_h * film_w * 4) and returns 0 to go on; the last iteration's image is the call's result, as above.  The batches of iterations that
 * share traversal launches end at those iterations, so write_frequency = 1 (the reference's default) runs one iteration per batch — about 3x the time of a call without
 * a callback.  write == NULL or write_frequency == 0: trhip_render_sppm.  In a multi-GPU job every rank's callback runs (each holds the whole image after the iteration's
 * all-reduce; hosts let rank 0 write) and the return codes are max-reduced over the ranks before anyone acts on them: a failure on one rank ends the call on all. */
typedef int (*trhip_sppm_write_fn)(void* user, uint32_t iteration, const float* xyzw);
int trhip_render_sppm_ex(trhip_ctx* ctx, const trhip_scene* scene, const trhip_sensor* sensor, float initial_search_radius, int max_depth, uint32_t n_iterations,
                         int64_t photons_per_iteration, uint64_t seed, float* out_xyzw, trhip_stats* stats, uint32_t write_frequency, trhip_sppm_write_fn write, void* user);
/* SPPMPixel fields (sppm.jl:65-95) after the last trhip_render_sppm on this context, (film_h, film_w[, 3]) row-major; any
 * pointer may be NULL.  M, phi, vp_p, vp_beta: the last iteration's values before _update_pixels! cleared them.
 * info6 = grid resolution x y z, grid entries, photon hits inside the grid (all iterations), photons per iteration. */
int trhip_sppm_state(trhip_ctx* ctx, float* Ld3, float* tau3, float* radius, double* N, int64_t* M, float* phi3, float* vp_p3, float* vp_beta3, int64_t* info6);

/* save(film) minus the PNG encoder (film.jl:204-222): xyzw -> linear RGB in [0,1], H*W*3, rows not flipped. */
int trhip_film_to_rgb(trhip_ctx* ctx, const float* xyzw, uint32_t width, uint32_t height, float scale, float* out_rgb);

/* ---- kernel-level entry points (parity tests and micro-benchmarks) --------------------------------------------------
 * rays: n*8 floats (o.xyz, t_max, d.xyz, time) = Ray (ray.jl:1-6).
 * trhip_hit: t = ray.t_max after intersect!(bvh, ray) (accel/bvh.jl:212-258; +Inf on a miss), prim = ordered-primitive
 * slot of the hit (-1 on a miss), b1/b2 = first two barycentrics of a triangle hit (triangle_mesh.jl:217-218). */
typedef struct {
    float t;
    int32_t prim;
    float b1, b2;
} trhip_hit;
int trhip_trace_closest(trhip_ctx* ctx, const trhip_scene* scene, const float* rays, uint64_t n, trhip_hit* out);
/* intersect_p(bvh, ray)  accel/bvh.jl:260-299 */
int trhip_trace_any(trhip_ctx* ctx, const trhip_scene* scene, const float* rays, uint64_t n, uint8_t* occluded);
/* Same on device-resident buffers, `repeat` launches back to back; returns average kernel ms (HIP events) — used by bench.py. */
int trhip_trace_closest_device(trhip_ctx* ctx, const trhip_scene* scene, const void* d_rays, uint64_t n, void* d_hits, int repeat, double* avg_ms);
int trhip_trace_any_device(trhip_ctx* ctx, const trhip_scene* scene, const void* d_rays, uint64_t n, void* d_occluded, int repeat, double* avg_ms);

/* Visit counters of the last *_device trace call when "count_visits" is on: nodes, prims (closest) then nodes, prims (any-hit). */
int trhip_last_visit_counts(trhip_ctx* ctx, uint64_t* out4);
/* Of the last trace call of the kernel-level entry points — closest-hit OR any-hit: every one of them zeroes the counters first, an any-hit call leaves 0 handed back —
 * (trhip_trace_closest, trhip_trace_any, trhip_hit_geometry, *_device): rays traced, and how many of them the hybrid mode's certified walk handed to the
 * reference-order walk on the canonical tree (0 when the scene holds one tree).  The frame entry points report the same through trhip_stats.fallback_rays. */
int trhip_last_fallback_counts(trhip_ctx* ctx, uint64_t* out2);

/* Device time (ms, HIP events: upload of the primitive bounds to the last flatten kernel) of the last BVHAccel built by the device
 * SAH builder ("bvh_builder" = 3, accel/bvh.jl:55-206 replaced); 0 when the last commit used another builder. */
int trhip_last_bvh_build_ms(trhip_ctx* ctx, double* ms_device);

/* Geometry of a closest hit as the shading kernel rebuilds it (SurfaceInteraction, surface_interaction.jl:51-88,154-181;
 * BSDF frame materials/bsdf.jl:41-50): per ray 15 floats p(3) n(3) ns(3) wo(3) ss(3); zeros on a miss. */
int trhip_hit_geometry(trhip_ctx* ctx, const trhip_scene* scene, const float* rays, uint64_t n, float* out_geom15);

/* generate_ray(camera, sample) camera/perspective.jl:85-114 for n camera samples (film.xy, lens.xy, time) -> n*8 rays. */
int trhip_generate_rays(trhip_ctx* ctx, const trhip_sensor* sensor, const float* samples5, uint64_t n, float* out_rays8);

/* BSDF through a material (materials/material.jl + materials/bsdf.jl:79-193) for n shading frames frame9 = ng ns ss(=normalize(shading.∂p∂u)):
 *   mode 0: dirs6 = wo(3) wi(3)      -> out8 = f(3) pdf 0 0 0 0          (b(wo,wi,flags), compute_pdf)
 *   mode 1: dirs6 = wo(3) u(2) 0     -> out8 = wi(3) f(3) pdf type       (sample_f) */
int trhip_bsdf_query(trhip_ctx* ctx, const trhip_scene* scene, uint32_t material, int allow_multiple_lobes, int mode, int flags,
                     const float* frame9, const float* dirs6, uint64_t n, float* out8);

/* Film splat of caller-provided samples: add_sample! + merge_film_tile! in the reference's tile order
 * (film.jl:134-193, integrators/sampler.jl:24-52).  sample_L: n_sample_pixels*spp*3 as in trhip_last_sample_radiance;
 * p_film positions are regenerated from (seed, sample_offset). */
int trhip_film_accumulate(trhip_ctx* ctx, const trhip_sensor* sensor, uint32_t spp, uint64_t seed, uint32_t sample_offset, const float* sample_L,
                          float* out_xyzw);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI -----------------------------------------------------
 * The scene is replicated; samples are independent.  A frame is sharded by global sample index (`sample_offset`: rank r of N
 * renders indices [r*spp_r, (r+1)*spp_r) of every pixel) into a private film; Film pixels are additive (xyz sums and
 * filter_weight_sum, film.jl:161-162, 190-191 — what merge_film_tile! relies on, film.jl:182-193), so ONE collective ends the frame.
 * The reference's only parallelism is Threads.@threads over tiles (integrators/sampler.jl:24) and photons (integrators/sppm.jl:334).
 *
 * trhip_comm_unique_id: rank 0 creates the 128-byte RCCL id (ncclGetUniqueId) and hands it to the other processes by any means
 *     (a file, MPI, torch.distributed's store ...).  trhip_comm_init: ncclCommInitRank on the context's GPU (collective: every
 *     rank calls it).  n_ranks == 1 is allowed (the collectives become copies).  RCCL is dlopen'ed on first use.
 * trhip_film_reduce: in-place ncclReduce(sum) of n_pixels x 4 floats (a device pointer, e.g. what trhip_render_path_device wrote)
 *     onto `root`; blocking.  trhip_film_allreduce: the same with every rank receiving the sum.
 * With a communicator, trhip_render_sppm shards the photon pass (rank r traces photon indices [r*P/N, (r+1)*P/N) of every
 *     iteration, sppm.jl:334) and all-reduces the per-pixel ϕ and M (sppm.jl:398-399) before _update_pixels! (sppm.jl:438-459):
 *     one exchange per iteration; every rank ends with the whole image.  The camera pass is replicated (1 path per pixel). */
#define TRHIP_UNIQUE_ID_BYTES 128
int trhip_comm_unique_id(uint8_t* out_id128);
int trhip_comm_init(trhip_ctx* ctx, const uint8_t* id128, int rank, int n_ranks);
int trhip_comm_destroy(trhip_ctx* ctx);
int trhip_comm_rank(const trhip_ctx* ctx, int* rank, int* n_ranks); /* 0 / 1 without a communicator */
int trhip_film_reduce(trhip_ctx* ctx, void* d_xyzw, uint64_t n_pixels, int root);
int trhip_film_allreduce(trhip_ctx* ctx, void* d_xyzw, uint64_t n_pixels);

/* ---- options ------------------------------------------------------------------------------------------------------- */
/* "count_visits" (0/1): instrumented traversal kernels fill nodes_visited / prims_tested.
 * "batch_paths": paths in flight per wavefront batch (0 = size from free HBM, the default).
 * "timing" (0/1): per-kernel HIP-event timing in trhip_stats (default 1).
 * "traversal" (1/2/3/4/6): 1 = literal accel/bvh.jl loop, 2 = children-in-parent nodes with per-lane ray replacement,
 *     3 (default) = 2 with the leaves of a wave postponed and tested together, 4 = 8-wide nodes with quantised child boxes walked
 *     in the binary tree's depth-first order (th_trace8.h; measured on par with 3); scenes 4 cannot take (foreign
 *     trees whose boxes do not nest, leaves of several primitives, more than 8 spheres, spheres not committed as a chain — see
 *     "compose_spheres") and rays it cannot take (a zero direction component) run 3; 6 = 3 with two rays per lane (th_trace4.h: better
 *     lane use, no fewer instructions: measured slower).  Same results bit for bit.
 * "compose_spheres" (-1/0/1): how trhip_scene_commit places up to 8 spheres of a scene that also has triangles: 1 = as a chain of
 *     single-sphere leaves above the triangles' subtree (what traversal 4 needs), 0 = inside one SAH tree, -1 (default) = 1 when
 *     "traversal" is 4 at commit time.  Either tree is a valid BVHAccel: results differ only in exact-t ties.
 * "overlap" (0/1): shadow rays of depth d on a second stream beside the closest-hit pass of depth d+1 (default 0: no gain any more).
 *     "stream2_priority" (-1/0/1): that stream's priority: lowest (default: the closest-hit rays are the critical path), the
 *     default level, highest; read when the streams are created (first render of a context).
 * "pipelines" (1..8): wavefront batches in flight at once (default 1).
 * "film_fused" (0/1, default 1): the path integrator's ray generation writes every sample's radiance record in the film pass's own layout, with its splat
 *     descriptor, so that a frame needs no memset of the records and no pack / re-lay pass before the gather (film.jl:134-164 replaced; same film bit for bit).
 * "bvh_builder" (-1/0/1/2/3): how trhip_scene_commit builds the BVH: 0 = binned SAH on the host, 1 = linear BVH on the device
 *     (Morton keys, radix sort, Karras hierarchy; 25-35 % more node visits per ray), 3 = the host builder's binned SAH run on the
 *     device (the same tree, 18 ms per million primitives instead of ~170 ms; scenes it cannot take go to the host builder),
 *     2 = the reference's own construction node for node (see trhip_scene_commit), -1 (default) = 3 from 64 Ki primitives on,
 *     0 below.  Every one of these trees is a valid BVHAccel: results differ only in exact-t ties.
 * "slab_margin_log2" (0..20, default 14): traversal 2 / 3 add to the reference's box test (bounds.jl:180-200) the two slab
 *     clauses it lost — it keeps the larger of the x and y exits — evaluated on boxes grown by 2^-N x the ray's reach; boxes on
 *     the path to a sphere keep the reference's test alone.  Fewer boxes visited, same results bit for bit;
 *     0 = the reference's test alone (its exact visit set, and the traversal tail that comes with it).
 * "tiny_scene_prims" (0..255): scenes of at most this many primitives are committed as ONE leaf (default 16; 0 = never).
 *     Read by trhip_scene_commit; results do not depend on it except through the order coincident hits are visited in.
 * "streaming" (-1/0/1): PathIntegrator on scenes with a real BVH as a streaming wavefront: rays that exceed a fetch budget
 *     are suspended and resumed in the next round instead of holding up their launch; same result bit for bit.  0 (default):
 *     never — since "slab_margin_log2" removed the 10^5-fetch rays it only costs; 1 always; -1 automatic (frames of
 *     at most 96 camera samples per primitive).  "stream_budget_min" (default 2048), "stream_budget_shift" (12) and
 *     "stream_list_cap" (0 = automatic) tune it.
 * "occluder_pretest" (0/1, default 1): any-hit rays test the scene's (at most 16) largest triangles first and go through the
 *     hierarchy only when none of them stops the ray; exact (th_trace2.h, k_any_occluders).
 * "sppm_batch": SPPM iterations whose camera / photon paths share the traversal launches (default 0 = as many as fit in
 *     free HBM, at most 128); the result does not depend on it.
 * "film_block" (0/1/2/3): film pixels per thread of the film gather: 1, 2 x 2, 1 x 4 (2, default), and (3) 1 x 4 reading one precomputed
 *     16-byte splat descriptor per sample (pixel range + filter-table indices, the same Float32 operations done once per sample instead of
 *     once per thread the sample reaches; filter radius <= 3, else 2; measured 20 % slower than 2); same film bit for bit.
 * "film_tiled" (0/1): LDS-staged film gather (default 0: measured slower).
 * "film_transpose" (0/1): film pass on pixel-group-major copies of the per-sample radiance / film positions (default 0: no gain).
 * "leaf_kernel" (0/1): one-leaf scenes (tiny_scene_prims) run the dedicated uniform-walk kernel instead of traversal 2 (default 1).
 * "band_tile_rows": PathIntegrator frames whose per-sample buffers (24 B per camera sample) do not fit in HBM are rendered in bands of
 *     whole 16-row tile rows into the same film — bit-identical to one band (tiles reach a film pixel in the reference's order either
 *     way); this option forces bands of N tile rows (tests; 0 = automatic).  trhip_last_sample_radiance needs a one-band frame.
 * "debug_trace_budget": DIAGNOSTIC ONLY, traversal abandons rays after this many node fetches (results wrong). */
int trhip_set_option(trhip_ctx* ctx, const char* name, int64_t value);
/* Whether this BUILD of the library carries the kernel families behind `name = value` (traversal 4 / 6 / 7, leaf_queue, leaf_sorted,
 * bvh_builder 1 exist only in the EXPERIMENTS build, -DTRHIP_EXPERIMENTS): 1 yes, 0 no (trhip_set_option would return
 * TRHIP_ERR_UNSUPPORTED).  A property of the binary: needs no context and no GPU — test suites decide at COLLECTION which variants exist. */
int trhip_option_in_build(const char* name, int64_t value);

/* The deterministic elementary functions of trace_detmath.h for hosts that cannot include a C header
 * (fn: 0 sin, 1 cos, 2 tan, 3 atan2(y, x), 4 acos, 5 log, 6 / 7 the sin / cos part of tm_sincosf); y may be NULL unless
 * fn == 3.  Needs no GPU. */
int trhip_detmath_f32(int fn, const float* x, const float* y, uint64_t n, float* out);
/* The same functions as the GPU kernels evaluate them (one thread per element): host and device must agree bit for bit. */
int trhip_detmath_f32_device(trhip_ctx* ctx, int fn, const float* x, const float* y, uint64_t n, float* out);

#ifdef __cplusplus
}
#endif
#endif /* TRACEHIP_H */
