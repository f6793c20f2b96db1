// oracle/orc_api.cpp — TEST INFRASTRUCTURE ONLY: C entry points over the restatement, loaded with ctypes by tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing under trace.jl_amd/ may link or load this.
// Matrices cross this API as 16 floats, row-major (m[row*4 + col]).
#include <cstdio>
#include <cstring>
#include <string>

#include "orc_build.h"
#include "orc_render.h"
#include "orc_sppm.h"

#ifdef _OPENMP
#include <omp.h>
#include <memory>
#endif

using namespace orc;

namespace {
M4 m4_from(const float* p) {
    M4 m;
    std::memcpy(m.m, p, sizeof m.m);
    return m;
}
void m4_to(const M4& m, float* p) { std::memcpy(p, m.m, sizeof m.m); }
Transformation tf_from(const float* m, const float* inv_m) { return Transformation(m4_from(m), m4_from(inv_m)); }

struct OrcScene {
    std::vector<Primitive> prims;  // caller order (before BVH build)
    Scene scene;
    bool committed = false;
    std::string err;
};
thread_local std::string g_err;
}  // namespace

extern "C" {

const char* orc_last_error() { return g_err.c_str(); }

// ---- scene construction ------------------------------------------------------------------------------------------------
void* orc_scene_new() { return new OrcScene(); }
void orc_scene_free(void* s) { delete (OrcScene*)s; }

// kind: 0 MATTE (Kd rgb, σ) | 1 MIRROR (Kr rgb) | 2 GLASS (Kr rgb, Kt rgb, u_rough, v_rough, index, remap) |
//       3 PLASTIC (Kd rgb, Ks rgb, roughness, remap)
int orc_scene_add_material(void* sp, int kind, const float* p, int n) {
    OrcScene* s = (OrcScene*)sp;
    Material m;
    m.kind = (Material::Kind)kind;
    switch (kind) {
    case 0:
        if (n != 4) return -1;
        m.Kd = RGB(p[0], p[1], p[2]);
        m.sigma = p[3];
        break;
    case 1:
        if (n != 3) return -1;
        m.Kr = RGB(p[0], p[1], p[2]);
        break;
    case 2:
        if (n != 10) return -1;
        m.Kr = RGB(p[0], p[1], p[2]);
        m.Kt = RGB(p[3], p[4], p[5]);
        m.u_roughness = p[6];
        m.v_roughness = p[7];
        m.index = p[8];
        m.remap_roughness = p[9] != 0;
        break;
    case 3:
        if (n != 8) return -1;
        m.Kd = RGB(p[0], p[1], p[2]);
        m.Ks = RGB(p[3], p[4], p[5]);
        m.roughness = p[6];
        m.remap_roughness = p[7] != 0;
        break;
    default: return -1;
    }
    s->scene.materials.push_back(m);
    return (int)s->scene.materials.size() - 1;
}

// create_triangle_mesh(core, n_tris, indices(1-based), n_verts, OBJECT-space vertices, normals|null) + one
// GeometricPrimitive per triangle with material_ids[k] (or -1).  Returns the index of the first primitive.
int orc_scene_add_triangle_mesh_ex(void* sp, const float* o2w, const float* o2w_inv, int reverse_orientation, const float* verts, uint32_t n_verts, const uint32_t* indices,
                                   uint32_t n_tris, const float* normals, const float* tangents, const float* uv_corners, const int32_t* material_ids);
int orc_scene_add_triangle_mesh(void* sp, const float* o2w, const float* o2w_inv, int reverse_orientation, const float* verts, uint32_t n_verts,
                                const uint32_t* indices, uint32_t n_tris, const float* normals, const int32_t* material_ids) {
    return orc_scene_add_triangle_mesh_ex(sp, o2w, o2w_inv, reverse_orientation, verts, n_verts, indices, n_tris, normals, nullptr, nullptr, material_ids);
}
// … with the mesh's optional tangents (n_verts x 3) and uvs (3 n_tris x 2, by corner position: triangle_mesh.jl:82)
int orc_scene_add_triangle_mesh_ex(void* sp, const float* o2w, const float* o2w_inv, int reverse_orientation, const float* verts, uint32_t n_verts, const uint32_t* indices,
                                   uint32_t n_tris, const float* normals, const float* tangents, const float* uv_corners, const int32_t* material_ids) {
    OrcScene* s = (OrcScene*)sp;
    ShapeCore core(tf_from(o2w, o2w_inv), reverse_orientation != 0);
    std::vector<V3> v(n_verts), nrm;
    for (uint32_t i = 0; i < n_verts; ++i) v[i] = V3(verts[3 * i], verts[3 * i + 1], verts[3 * i + 2]);
    if (normals) {
        nrm.resize(n_verts);
        for (uint32_t i = 0; i < n_verts; ++i) nrm[i] = V3(normals[3 * i], normals[3 * i + 1], normals[3 * i + 2]);
    }
    std::vector<V3> tan;
    if (tangents) {
        tan.resize(n_verts);
        for (uint32_t i = 0; i < n_verts; ++i) tan[i] = V3(tangents[3 * i], tangents[3 * i + 1], tangents[3 * i + 2]);
    }
    std::vector<V2> uvs;
    if (uv_corners) {
        uvs.resize(3 * (size_t)n_tris);
        for (size_t i = 0; i < uvs.size(); ++i) uvs[i] = V2{uv_corners[2 * i], uv_corners[2 * i + 1]};
    }
    std::vector<uint32_t> idx(indices, indices + 3 * (size_t)n_tris);
    auto mesh = std::make_shared<TriangleMesh>(core, idx, v, nrm, tan, uvs);
    const int first = (int)s->prims.size();
    for (uint32_t k = 0; k < n_tris; ++k) {
        Primitive p;
        p.kind = Primitive::TRIANGLE;
        p.triangle.mesh = mesh;
        p.triangle.i = k * 3 + 1;  // triangle_mesh.jl:40-42
        p.material = material_ids ? material_ids[k] : -1;
        p.user_id = (int)s->prims.size();
        s->prims.push_back(p);
    }
    return first;
}

int orc_scene_add_sphere(void* sp, const float* o2w, const float* o2w_inv, int reverse_orientation, float radius, float z_min, float z_max,
                         float phi_max_deg, int material_id) {
    OrcScene* s = (OrcScene*)sp;
    Primitive p;
    p.kind = Primitive::SPHERE;
    p.sphere = std::make_shared<Sphere>(ShapeCore(tf_from(o2w, o2w_inv), reverse_orientation != 0), radius, z_min, z_max, phi_max_deg);
    p.material = material_id;
    p.user_id = (int)s->prims.size();
    s->prims.push_back(p);
    return p.user_id;
}

int orc_scene_add_point_light(void* sp, const float* l2w, const float* l2w_inv, const float* I) {
    OrcScene* s = (OrcScene*)sp;
    s->scene.lights.push_back(PointLight(tf_from(l2w, l2w_inv), RGB(I[0], I[1], I[2])));
    return (int)s->scene.lights.size() - 1;
}
int orc_scene_add_spot_light(void* sp, const float* l2w, const float* l2w_inv, const float* I, float total_width_deg, float falloff_start_deg) {
    OrcScene* s = (OrcScene*)sp;
    s->scene.lights.push_back(SpotLight(tf_from(l2w, l2w_inv), RGB(I[0], I[1], I[2]), total_width_deg, falloff_start_deg));
    return (int)s->scene.lights.size() - 1;
}
// Read back what the restated light constructors computed (position, cosines) for host-mirror cross-checks.
int orc_scene_get_light(void* sp, int i, float* position3, float* cos2) {
    OrcScene* s = (OrcScene*)sp;
    if (i < 0 || i >= (int)s->scene.lights.size()) return -1;
    const Light& l = s->scene.lights[i];
    position3[0] = l.position.x;
    position3[1] = l.position.y;
    position3[2] = l.position.z;
    cos2[0] = l.cos_total_width;
    cos2[1] = l.cos_falloff_start;
    return 0;
}

// BVHAccel(prims, max_node_primitives) with the reference's own (quirky) builder.
int orc_scene_commit_reference_bvh(void* sp, int max_node_primitives) {
    OrcScene* s = (OrcScene*)sp;
    try {
        s->scene.aggregate = *build_reference_bvh(s->prims, max_node_primitives);
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
    s->committed = true;
    return 0;
}
// Adopt an externally built flattened BVH (the product's): nodes as 8 floats + 2 uint32 each, in the product's node
// layout: bounds min/max, then for a leaf (flags & 3) == 3: a = first ordered-primitive index (0-based), n = flags >> 2;
// for an interior node: a = second child index (0-based), axis = flags & 3 (0..2); first child is i + 1.
int orc_scene_commit_external_bvh(void* sp, const float* node_bounds /*n*6*/, const uint32_t* node_a, const uint32_t* node_flags, uint32_t n_nodes,
                                  const uint32_t* prim_order /* ordered slot -> caller primitive index */, uint32_t n_prims) {
    OrcScene* s = (OrcScene*)sp;
    BVHAccel& b = s->scene.aggregate;
    b.nodes.resize(n_nodes);
    for (uint32_t i = 0; i < n_nodes; ++i) {
        LinearNode& ln = b.nodes[i];
        ln.bounds = Bounds3(V3(node_bounds[6 * i], node_bounds[6 * i + 1], node_bounds[6 * i + 2]), V3(node_bounds[6 * i + 3], node_bounds[6 * i + 4], node_bounds[6 * i + 5]));
        if ((node_flags[i] & 3u) == 3u) {
            ln.leaf = true;
            ln.primitives_offset = node_a[i] + 1;
            ln.n_primitives = node_flags[i] >> 2;
        } else {
            ln.leaf = false;
            ln.second_child_offset = node_a[i] + 1;
            ln.split_axis = (uint8_t)((node_flags[i] & 3u) + 1);
        }
    }
    b.primitives.clear();
    b.primitives.reserve(n_prims);
    for (uint32_t i = 0; i < n_prims; ++i) {
        if (prim_order[i] >= s->prims.size()) {
            g_err = "prim_order out of range";
            return -1;
        }
        b.primitives.push_back(s->prims[prim_order[i]]);
    }
    s->committed = true;
    return 0;
}
uint32_t orc_scene_bvh_node_count(void* sp) { return (uint32_t)((OrcScene*)sp)->scene.aggregate.nodes.size(); }
uint32_t orc_scene_prim_count(void* sp) { return (uint32_t)((OrcScene*)sp)->scene.aggregate.primitives.size(); }
// Export the committed BVH in the external layout described above (+ ordered slot -> caller primitive index).
int orc_scene_get_bvh(void* sp, float* node_bounds, uint32_t* node_a, uint32_t* node_flags, uint32_t* prim_order) {
    OrcScene* s = (OrcScene*)sp;
    const BVHAccel& b = s->scene.aggregate;
    for (size_t i = 0; i < b.nodes.size(); ++i) {
        const LinearNode& ln = b.nodes[i];
        const float v[6] = {ln.bounds.p_min.x, ln.bounds.p_min.y, ln.bounds.p_min.z, ln.bounds.p_max.x, ln.bounds.p_max.y, ln.bounds.p_max.z};
        std::memcpy(node_bounds + 6 * i, v, sizeof v);
        if (ln.leaf) {
            node_a[i] = ln.primitives_offset - 1;
            node_flags[i] = (ln.n_primitives << 2) | 3u;
        } else {
            node_a[i] = ln.second_child_offset - 1;
            node_flags[i] = (uint32_t)(ln.split_axis - 1);
        }
    }
    for (size_t i = 0; i < b.primitives.size(); ++i) prim_order[i] = (uint32_t)b.primitives[i].user_id;
    return 0;
}
void orc_scene_world_bound(void* sp, float* out6) {
    const Bounds3 b = world_bound(((OrcScene*)sp)->scene.aggregate);
    out6[0] = b.p_min.x;
    out6[1] = b.p_min.y;
    out6[2] = b.p_min.z;
    out6[3] = b.p_max.x;
    out6[4] = b.p_max.y;
    out6[5] = b.p_max.z;
}

// ---- kernel-level queries ------------------------------------------------------------------------------------------------
// rays: n*8 floats (ox,oy,oz,tmax,dx,dy,dz,time).  out_t[n], out_prim[n] (ordered-primitive index, -1 = miss),
// out_geom (optional) n*15 floats: p(3) n(3) ns(3) wo(3) ss = normalize(shading.∂p∂u)(3).
int orc_trace_closest(void* sp, const float* rays, uint64_t n, float* out_t, int32_t* out_prim, float* out_geom, uint64_t* visit_counts2) {
    OrcScene* s = (OrcScene*)sp;
    uint64_t nodes = 0, prims = 0;
#pragma omp parallel reduction(+ : nodes, prims)
    {
        BVHAccel& local = s->scene.aggregate;
        counters() = Counters{};
#pragma omp for schedule(dynamic, 4096)
        for (int64_t i = 0; i < (int64_t)n; ++i) {
            const float* r = rays + 8 * i;
            Ray ray{V3(r[0], r[1], r[2]), V3(r[4], r[5], r[6]), r[3], r[7]};
            SurfaceInteraction si;
            const bool hit = bvh_intersect(local, ray, si);
            out_t[i] = hit ? ray.t_max : INF32;
            out_prim[i] = hit ? si.primitive : -1;
            if (out_geom) {
                float* g = out_geom + 15 * i;
                if (hit) {
                    const V3 ss = normalize(si.sh_dpdu);
                    const float v[15] = {si.p.x, si.p.y, si.p.z, si.n.x, si.n.y, si.n.z, si.sh_n.x, si.sh_n.y, si.sh_n.z, si.wo.x, si.wo.y, si.wo.z, ss.x, ss.y, ss.z};
                    std::memcpy(g, v, sizeof v);
                } else {
                    std::memset(g, 0, 15 * sizeof(float));
                }
            }
        }
        nodes += counters().nodes;
        prims += counters().prims;
    }
    if (visit_counts2) {
        visit_counts2[0] = nodes;
        visit_counts2[1] = prims;
    }
    return 0;
}
int orc_trace_any(void* sp, const float* rays, uint64_t n, uint8_t* out_occluded, uint64_t* visit_counts2) {
    OrcScene* s = (OrcScene*)sp;
    uint64_t nodes = 0, prims = 0;
#pragma omp parallel reduction(+ : nodes, prims)
    {
        BVHAccel& local = s->scene.aggregate;
        counters() = Counters{};
#pragma omp for schedule(dynamic, 4096)
        for (int64_t i = 0; i < (int64_t)n; ++i) {
            const float* r = rays + 8 * i;
            Ray ray{V3(r[0], r[1], r[2]), V3(r[4], r[5], r[6]), r[3], r[7]};
            out_occluded[i] = bvh_intersect_p(local, ray) ? 1 : 0;
        }
        nodes += counters().nodes;
        prims += counters().prims;
    }
    if (visit_counts2) {
        visit_counts2[0] = nodes;
        visit_counts2[1] = prims;
    }
    return 0;
}

// ---- sensor + render -------------------------------------------------------------------------------------------------------
struct orc_sensor {
    float camera_to_world[16], camera_to_world_inv[16];
    float screen_window[4];  // min.x, min.y, max.x, max.y
    float shutter_open, shutter_close, lens_radius, focal_distance, fov_deg;
    float resolution[2];
    float crop[4];  // fraction of the image: min.x, min.y, max.x, max.y
    float filter_radius[2], filter_tau;
    float film_scale;
};
static Film make_film(const orc_sensor* sn) {
    LanczosSincFilter flt;
    flt.radius = V2{sn->filter_radius[0], sn->filter_radius[1]};
    flt.tau = sn->filter_tau;
    return Film(V2{sn->resolution[0], sn->resolution[1]}, Bounds2{{sn->crop[0], sn->crop[1]}, {sn->crop[2], sn->crop[3]}}, flt, 1.0f, sn->film_scale);
}
static PerspectiveCamera make_camera(const orc_sensor* sn) {
    return PerspectiveCamera(tf_from(sn->camera_to_world, sn->camera_to_world_inv), Bounds2{{sn->screen_window[0], sn->screen_window[1]}, {sn->screen_window[2], sn->screen_window[3]}},
                             sn->shutter_open, sn->shutter_close, sn->lens_radius, sn->focal_distance, sn->fov_deg, V2{sn->resolution[0], sn->resolution[1]});
}
// What the restated Film / PerspectiveCamera constructors derive: for host-mirror cross-checks.
// out_i: width, height, sample-bounds min.x, min.y, max.x, max.y ; out_table: 256 floats (y,x); out_r2c: raster_to_camera.m
int orc_sensor_derived(const orc_sensor* sn, int32_t* out_i6, float* out_crop4, float* out_table256, float* out_r2c16) {
    Film f = make_film(sn);
    const Bounds2 sb = get_sample_bounds(f);
    out_i6[0] = f.width;
    out_i6[1] = f.height;
    out_i6[2] = (int)sb.p_min.x;
    out_i6[3] = (int)sb.p_min.y;
    out_i6[4] = (int)sb.p_max.x;
    out_i6[5] = (int)sb.p_max.y;
    out_crop4[0] = f.crop_bounds.p_min.x;
    out_crop4[1] = f.crop_bounds.p_min.y;
    out_crop4[2] = f.crop_bounds.p_max.x;
    out_crop4[3] = f.crop_bounds.p_max.y;
    std::memcpy(out_table256, f.filter_table, sizeof f.filter_table);
    const PerspectiveCamera cam = make_camera(sn);
    m4_to(cam.raster_to_camera.m, out_r2c16);
    return 0;
}
// generate_ray for a batch of camera samples (film.x, film.y, lens.x, lens.y, time) -> rays n*8
int orc_generate_rays(const orc_sensor* sn, const float* samples5, uint64_t n, float* out_rays8) {
    const PerspectiveCamera cam = make_camera(sn);
    for (uint64_t i = 0; i < n; ++i) {
        const float* c = samples5 + 5 * i;
        CameraSample cs{V2{c[0], c[1]}, V2{c[2], c[3]}, c[4]};
        const Ray r = generate_ray(cam, cs);
        const float v[8] = {r.o.x, r.o.y, r.o.z, r.t_max, r.d.x, r.d.y, r.d.z, r.time};
        std::memcpy(out_rays8 + 8 * i, v, sizeof v);
    }
    return 0;
}

struct orc_stats {
    uint64_t camera_samples, closest_rays, shadow_rays, nodes_visited, prims_tested;
};
// integrator: 0 Whitted, 1 Path.  out_xyzw: height*width*4 (xyz sums + filter_weight_sum, film.pixels (y,x) order).
// out_sample_L: optional, spp * n_sample_pixels * 3.  threads <= 1: the sequential tile loop of the restatement; threads > 1: the
// tiles of a tile row are rendered in parallel into private FilmTiles (like Threads.@threads, integrators/sampler.jl:24) and
// merged in tile order — the same film and samples bit for bit, sooner.
int orc_render(void* sp, const orc_sensor* sn, int integrator, int64_t spp, int max_depth, uint64_t seed, uint32_t sample_offset, int threads, float* out_xyzw,
               float* out_sample_L, orc_stats* stats) {
    OrcScene* s = (OrcScene*)sp;
    if (!s->committed) {
        g_err = "scene not committed";
        return -1;
    }
    Film film = make_film(sn);
    const PerspectiveCamera cam = make_camera(sn);
    RenderStats rs;
    if (threads <= 1) {
        render(s->scene, cam, film, (IntegratorKind)integrator, spp, max_depth, seed, sample_offset, out_sample_L, &rs);
    } else {
#ifdef _OPENMP
        const Bounds2 sb = get_sample_bounds(film);
        const int tile_size = 16;
        const long long width = (long long)std::floor((sb.p_max.x - sb.p_min.x + tile_size) / tile_size), height = (long long)std::floor((sb.p_max.y - sb.p_min.y + tile_size) / tile_size);
        uint64_t n_samples = 0, n_closest = 0, n_shadow = 0, n_nodes = 0, n_prims = 0;
        const int sbw = (int)(sb.p_max.x - sb.p_min.x) + 1, sbh = (int)(sb.p_max.y - sb.p_min.y) + 1;
        const size_t n_pix = (size_t)sbw * sbh;
        // bounded memory: bands of tile rows; the tiles of a band are rendered in parallel, then merged in k order — the film is the
        // sequential loop's bit for bit (the reference itself merges in completion order without a lock, film.jl:182-193)
        const long long band_rows = std::max<long long>(1, (8LL * threads + width - 1) / width);
        std::vector<std::unique_ptr<FilmTile>> tiles((size_t)(width * band_rows));
        for (long long row0 = 0; row0 < height; row0 += band_rows) {
            const long long n_band = std::min(band_rows, height - row0) * width;
#pragma omp parallel num_threads(threads) reduction(+ : n_samples, n_closest, n_shadow, n_nodes, n_prims)
        {
            Scene& local = s->scene;
            counters() = Counters{};
#pragma omp for schedule(dynamic, 1)
            for (long long kx = 0; kx < n_band; ++kx) {
                const float tx = (float)(kx % width), ty = (float)(row0 + kx / width);
                SeededSampler smp(spp, seed, sample_offset);
                const V2 tb_min{sb.p_min.x + tx * tile_size, sb.p_min.y + ty * tile_size};
                const V2 tb_max{jl_min(tb_min.x + (tile_size - 1), sb.p_max.x), jl_min(tb_min.y + (tile_size - 1), sb.p_max.y)};
                tiles[(size_t)kx].reset(new FilmTile(film, Bounds2{tb_min, tb_max}));
                FilmTile& tile = *tiles[(size_t)kx];
                for (float py = tb_min.y; py <= tb_max.y; py += 1.0f)
                    for (float px = tb_min.x; px <= tb_max.x; px += 1.0f) {
                        smp.start_pixel(V2{px, py});
                        while (smp.has_next_sample()) {
                            const CameraSample cs = smp.get_camera_sample(V2{px, py});
                            const Ray ray = generate_ray(cam, cs);
                            RGB l = integrator == 0 ? whitted_li(local, ray, max_depth, 1) : path_li(local, ray, smp, max_depth);
                            if (has_nan(l)) l = RGB(0.0f);
                            if (out_sample_L) {
                                const size_t pix = (size_t)(py - sb.p_min.y) * sbw + (size_t)(px - sb.p_min.x);
                                float* o = out_sample_L + ((size_t)(smp.current_sample - 1) * n_pix + pix) * 3;
                                o[0] = l.x;
                                o[1] = l.y;
                                o[2] = l.z;
                            }
                            add_sample(tile, cs.film, l, 1.0f);
                            n_samples++;
                            smp.start_next_sample();
                        }
                    }
            }
            n_closest += counters().closest;
            n_shadow += counters().shadow;
            n_nodes += counters().nodes;
            n_prims += counters().prims;
        }
            for (long long kx = 0; kx < n_band; ++kx) merge_film_tile(film, *tiles[(size_t)kx]);
        }
        rs.camera_samples = n_samples;
        rs.closest_rays = n_closest;
        rs.shadow_rays = n_shadow;
        rs.nodes_visited = n_nodes;
        rs.prims_tested = n_prims;
#else
        g_err = "built without OpenMP";
        return -1;
#endif
    }
    if (out_xyzw)
        for (size_t i = 0; i < film.pixels.size(); ++i) {
            out_xyzw[4 * i + 0] = film.pixels[i].xyz.x;
            out_xyzw[4 * i + 1] = film.pixels[i].xyz.y;
            out_xyzw[4 * i + 2] = film.pixels[i].xyz.z;
            out_xyzw[4 * i + 3] = film.pixels[i].filter_weight_sum;
        }
    if (stats) {
        stats->camera_samples = rs.camera_samples;
        stats->closest_rays = rs.closest_rays;
        stats->shadow_rays = rs.shadow_rays;
        stats->nodes_visited = rs.nodes_visited;
        stats->prims_tested = rs.prims_tested;
    }
    return 0;
}
// film.jl:204-222 minus the encoder: xyzw (H*W*4) -> linear RGB in [0,1] (H*W*3), rows not flipped.
int orc_film_to_rgb(const float* xyzw, int width, int height, float scale, float* out_rgb) {
    Film f;
    f.width = width;
    f.height = height;
    f.scale = scale;
    f.pixels.resize((size_t)width * height);
    for (size_t i = 0; i < f.pixels.size(); ++i) {
        f.pixels[i].xyz = V3(xyzw[4 * i], xyzw[4 * i + 1], xyzw[4 * i + 2]);
        f.pixels[i].filter_weight_sum = xyzw[4 * i + 3];
    }
    film_to_rgb(f, out_rgb);
    return 0;
}
int orc_num_threads() {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

// ---- known-answer helpers -----------------------------------------------------------------------------
// Transformations as the reference builds them: out = m (16) then inv_m (16).
void orc_translate(const float* d3, float* out32) {
    const Transformation t = translate(V3(d3[0], d3[1], d3[2]));
    m4_to(t.m, out32);
    m4_to(t.inv_m, out32 + 16);
}
void orc_scale(float x, float y, float z, float* out32) {
    const Transformation t = scale(x, y, z);
    m4_to(t.m, out32);
    m4_to(t.inv_m, out32 + 16);
}
void orc_look_at(const float* pos3, const float* target3, const float* up3, float* out32) {
    const Transformation t = look_at(V3(pos3[0], pos3[1], pos3[2]), V3(target3[0], target3[1], target3[2]), V3(up3[0], up3[1], up3[2]));
    m4_to(t.m, out32);
    m4_to(t.inv_m, out32 + 16);
}
void orc_perspective(float fov, float near, float far, float* out32) {
    const Transformation t = perspective(fov, near, far);
    m4_to(t.m, out32);
    m4_to(t.inv_m, out32 + 16);
}
void orc_transform_from_matrix(const float* m16, float* out32) {  // Transformation(::Mat4f) transformations.jl:7
    const Transformation t(m4_from(m16));
    m4_to(t.m, out32);
    m4_to(t.inv_m, out32 + 16);
}
void orc_transform_mul(const float* a32, const float* b32, float* out32) {  // transformations.jl:20-22
    const Transformation t = tf_from(a32, a32 + 16) * tf_from(b32, b32 + 16);
    m4_to(t.m, out32);
    m4_to(t.inv_m, out32 + 16);
}
void orc_transform_point(const float* t32, const float* p3, float* out3) {
    const V3 r = tf_from(t32, t32 + 16).point(V3(p3[0], p3[1], p3[2]));
    out3[0] = r.x;
    out3[1] = r.y;
    out3[2] = r.z;
}
void orc_coordinate_system(const float* v3, float* out6) {  // Trace.jl:139-146 -> v2, v3
    V3 a, b;
    coordinate_system(V3(v3[0], v3[1], v3[2]), a, b);
    out6[0] = a.x;
    out6[1] = a.y;
    out6[2] = a.z;
    out6[3] = b.x;
    out6[4] = b.y;
    out6[5] = b.z;
}
// intersect(b::Bounds3, ray) -> returns hit, t0, t1
int orc_bounds_intersect(const float* b6, const float* ray8, float* t01) {
    Ray r{V3(ray8[0], ray8[1], ray8[2]), V3(ray8[4], ray8[5], ray8[6]), ray8[3], ray8[7]};
    return bounds_intersect(Bounds3(V3(b6[0], b6[1], b6[2]), V3(b6[3], b6[4], b6[5])), r, t01[0], t01[1]) ? 1 : 0;
}
int orc_bounds_intersect_p(const float* b6, const float* ray8) {
    Ray r{V3(ray8[0], ray8[1], ray8[2]), V3(ray8[4], ray8[5], ray8[6]), ray8[3], ray8[7]};
    const V3 inv_dir(1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z);
    int neg[3];
    is_dir_negative(r.d, neg);
    return bounds_intersect_p(Bounds3(V3(b6[0], b6[1], b6[2]), V3(b6[3], b6[4], b6[5])), r, inv_dir, neg) ? 1 : 0;
}
// Direct shape queries on caller primitive `prim` (no BVH): intersect -> hit, t, geom(15: p n ns wo ss) + uv(2); intersect_p
int orc_prim_intersect(void* sp, int prim, const float* ray8, float* out_t, float* out_geom17) {
    OrcScene* s = (OrcScene*)sp;
    Ray r{V3(ray8[0], ray8[1], ray8[2]), V3(ray8[4], ray8[5], ray8[6]), ray8[3], ray8[7]};
    SurfaceInteraction si;
    float t;
    Primitive& p = s->prims[prim];
    const bool hit = p.kind == Primitive::SPHERE ? sphere_intersect(*p.sphere, r, t, si) : triangle_intersect(p.triangle, r, t, si);
    if (!hit) return 0;
    *out_t = t;
    const V3 ss = normalize(si.sh_dpdu);
    const float v[17] = {si.p.x, si.p.y, si.p.z, si.n.x, si.n.y, si.n.z, si.sh_n.x, si.sh_n.y, si.sh_n.z, si.wo.x, si.wo.y, si.wo.z, ss.x, ss.y, ss.z, si.uv.x, si.uv.y};
    std::memcpy(out_geom17, v, sizeof v);
    return 1;
}
int orc_prim_intersect_p(void* sp, int prim, const float* ray8) {
    OrcScene* s = (OrcScene*)sp;
    Ray r{V3(ray8[0], ray8[1], ray8[2]), V3(ray8[4], ray8[5], ray8[6]), ray8[3], ray8[7]};
    Primitive& p = s->prims[prim];
    return (p.kind == Primitive::SPHERE ? sphere_intersect_p(*p.sphere, r) : triangle_intersect_p(p.triangle, r)) ? 1 : 0;
}
void orc_prim_bounds(void* sp, int prim, float* world6, float* object6) {
    OrcScene* s = (OrcScene*)sp;
    Primitive& p = s->prims[prim];
    const Bounds3 w = world_bound(p);
    const Bounds3 o = p.kind == Primitive::SPHERE ? object_bound(*p.sphere) : object_bound(p.triangle);
    const float a[6] = {w.p_min.x, w.p_min.y, w.p_min.z, w.p_max.x, w.p_max.y, w.p_max.z};
    const float b[6] = {o.p_min.x, o.p_min.y, o.p_min.z, o.p_max.x, o.p_max.y, o.p_max.z};
    std::memcpy(world6, a, sizeof a);
    std::memcpy(object6, b, sizeof b);
}
float orc_triangle_area(void* sp, int prim) { return tri_area(((OrcScene*)sp)->prims[prim].triangle); }
// Nest the committed BVH of `inner` as one primitive of `outer` (test_intersection.jl:137-138).
int orc_scene_add_nested_bvh(void* outer, void* inner) {
    OrcScene* o = (OrcScene*)outer;
    OrcScene* in = (OrcScene*)inner;
    Primitive p;
    p.kind = Primitive::BVH;
    p.bvh = std::make_shared<BVHAccel>(in->scene.aggregate);
    p.user_id = (int)o->prims.size();
    o->prims.push_back(p);
    return p.user_id;
}

float orc_fresnel_dielectric(float c, float ei, float et) { return fresnel_dielectric(c, ei, et); }
void orc_fresnel_conductor(float c, const float* ei3, const float* et3, const float* k3, float* out3) {
    const RGB r = fresnel_conductor(c, RGB(ei3[0], ei3[1], ei3[2]), RGB(et3[0], et3[1], et3[2]), RGB(k3[0], k3[1], k3[2]));
    out3[0] = r.x;
    out3[1] = r.y;
    out3[2] = r.z;
}
float orc_roughness_to_alpha(float r) { return roughness_to_alpha(r); }
float orc_filter_eval(float rx, float ry, float tau, float px, float py) {
    LanczosSincFilter f;
    f.radius = V2{rx, ry};
    f.tau = tau;
    return filter_eval(f, V2{px, py});
}
// FilmTile(film, bounds): returns tile bounds (4) and size (2); then add samples and read filter_weight_sum / contrib back.
void* orc_filmtile_new(const orc_sensor* sn, const float* sample_bounds4, float* out_bounds4, int32_t* out_size2) {
    Film* f = new Film(make_film(sn));
    FilmTile* t = new FilmTile(*f, Bounds2{{sample_bounds4[0], sample_bounds4[1]}, {sample_bounds4[2], sample_bounds4[3]}});
    out_bounds4[0] = t->bounds.p_min.x;
    out_bounds4[1] = t->bounds.p_min.y;
    out_bounds4[2] = t->bounds.p_max.x;
    out_bounds4[3] = t->bounds.p_max.y;
    out_size2[0] = t->height;
    out_size2[1] = t->width;
    return t;
}
void orc_filmtile_add_sample(void* tp, float x, float y, const float* rgb3, float weight) { add_sample(*(FilmTile*)tp, V2{x, y}, RGB(rgb3[0], rgb3[1], rgb3[2]), weight); }
void orc_filmtile_read(void* tp, float* out_rgbw /* h*w*4 */) {
    FilmTile* t = (FilmTile*)tp;
    for (size_t i = 0; i < t->pixels.size(); ++i) {
        out_rgbw[4 * i] = t->pixels[i].contrib_sum.x;
        out_rgbw[4 * i + 1] = t->pixels[i].contrib_sum.y;
        out_rgbw[4 * i + 2] = t->pixels[i].contrib_sum.z;
        out_rgbw[4 * i + 3] = t->pixels[i].filter_weight_sum;
    }
}
// merge into a fresh Film and return its xyzw
void orc_filmtile_merge(void* tp, float* out_xyzw) {
    FilmTile* t = (FilmTile*)tp;
    Film f = *t->film;
    merge_film_tile(f, *t);
    for (size_t i = 0; i < f.pixels.size(); ++i) {
        out_xyzw[4 * i] = f.pixels[i].xyz.x;
        out_xyzw[4 * i + 1] = f.pixels[i].xyz.y;
        out_xyzw[4 * i + 2] = f.pixels[i].xyz.z;
        out_xyzw[4 * i + 3] = f.pixels[i].filter_weight_sum;
    }
}
void orc_filmtile_free(void* tp) {
    FilmTile* t = (FilmTile*)tp;
    delete t->film;
    delete t;
}

// BxDF-level queries.  bxdf kind: 0 LambertianR(r) 1 LambertianT(t) 2 OrenNayar(r, σ°) 3 SpecularR(r, fresnel) 4 SpecularT(t, ηa, ηb)
// 5 FresnelSpecular(r, t, ηa, ηb) 6 MicrofacetR(r, αx, αy, fresnel) 7 MicrofacetT(t, αx, αy, ηa, ηb).
// params: r(3) t(3) sigma alpha_x alpha_y eta_a eta_b fresnel_kind(0 noop,1 dielectric) fr_eta_i fr_eta_t  = 14 floats
static BxDF make_bxdf(int kind, const float* p) {
    const RGB r(p[0], p[1], p[2]), t(p[3], p[4], p[5]);
    Fresnel fr;
    if (p[11] == 1.0f) fr = FresnelDielectric(p[12], p[13]);
    switch (kind) {
    case 0: return LambertianReflection(r);
    case 1: return LambertianTransmission(t);
    case 2: return OrenNayar(r, p[6]);
    case 3: return SpecularReflection(r, fr);
    case 4: return SpecularTransmission(t, p[9], p[10]);
    case 5: return FresnelSpecular(r, t, p[9], p[10]);
    case 6: return MicrofacetReflection(r, TrowbridgeReitz(p[7], p[8]), fr);
    default: return MicrofacetTransmission(t, TrowbridgeReitz(p[7], p[8]), p[9], p[10]);
    }
}
int orc_bxdf_type(int kind, const float* params14) { return make_bxdf(kind, params14).type; }
// out: wi(3) pdf f(3) sampled_type(-1 = nothing)
void orc_bxdf_sample_f(int kind, const float* params14, const float* wo3, const float* u2, float* out8) {
    const BxDF b = make_bxdf(kind, params14);
    const BxDFSample s = bxdf_sample_f(b, V3(wo3[0], wo3[1], wo3[2]), V2{u2[0], u2[1]});
    const float v[8] = {s.wi.x, s.wi.y, s.wi.z, s.pdf, s.f.x, s.f.y, s.f.z, s.has_type ? (float)s.sampled_type : -1.0f};
    std::memcpy(out8, v, sizeof v);
}
void orc_bxdf_f_pdf(int kind, const float* params14, const float* wo3, const float* wi3, float* out4) {
    const BxDF b = make_bxdf(kind, params14);
    const RGB f = bxdf_f(b, V3(wo3[0], wo3[1], wo3[2]), V3(wi3[0], wi3[1], wi3[2]));
    out4[0] = f.x;
    out4[1] = f.y;
    out4[2] = f.z;
    out4[3] = bxdf_pdf(b, V3(wo3[0], wo3[1], wo3[2]), V3(wi3[0], wi3[1], wi3[2]));
}
// BSDF-level batch query through a material (the a10-a13 parity entry): for each i a shading frame is given by
// geom15 (p n ns wo ss as produced by orc_trace_closest; ts = ns × ss), the material by id.
//   mode 0: f(wo, wi, flags) and pdf -> out[0..3]       (dirs6 = wo(3) wi(3))
//   mode 1: sample_f(wo, u, flags) -> out = wi(3) f(3) pdf type      (dirs6 = wo(3) u(2) -)
int orc_bsdf_query(void* sp, int material, int allow_multiple_lobes, int mode, int flags, const float* frame9 /* ng ns ss */, const float* dirs6, uint64_t n,
                   float* out8) {
    OrcScene* s = (OrcScene*)sp;
    if (material < 0 || material >= (int)s->scene.materials.size()) return -1;
    for (uint64_t i = 0; i < n; ++i) {
        const float* fr = frame9 + 9 * i;
        const float* d = dirs6 + 6 * i;
        SurfaceInteraction si;
        si.n = V3(fr[0], fr[1], fr[2]);
        si.sh_n = V3(fr[3], fr[4], fr[5]);
        si.sh_dpdu = V3(fr[6], fr[7], fr[8]);
        const BSDF b = compute_scattering(s->scene.materials[material], si, allow_multiple_lobes != 0);
        float* o = out8 + 8 * i;
        const V3 wo(d[0], d[1], d[2]);
        if (mode == 0) {
            const V3 wi(d[3], d[4], d[5]);
            const RGB f = b.f(wo, wi, (uint8_t)flags);
            o[0] = f.x;
            o[1] = f.y;
            o[2] = f.z;
            o[3] = b.pdf(wo, wi, (uint8_t)flags);
            o[4] = o[5] = o[6] = o[7] = 0;
        } else {
            const BSDFSample r = bsdf_sample_f(b, wo, V2{d[3], d[4]}, (uint8_t)flags);
            const float v[8] = {r.wi.x, r.wi.y, r.wi.z, r.f.x, r.f.y, r.f.z, r.pdf, (float)r.sampled_type};
            std::memcpy(o, v, sizeof v);
        }
    }
    return 0;
}
// sample_li + unoccluded for a batch of reference points -> radiance(3) wi(3) pdf unoccluded
int orc_light_query(void* sp, int light, const float* points3, uint64_t n, float* out8) {
    OrcScene* s = (OrcScene*)sp;
    if (light < 0 || light >= (int)s->scene.lights.size()) return -1;
    for (uint64_t i = 0; i < n; ++i) {
        const LightSample ls = sample_li(s->scene.lights[light], V3(points3[3 * i], points3[3 * i + 1], points3[3 * i + 2]), 0.0f);
        const bool vis = s->committed ? unoccluded(s->scene, ls) : true;
        const float v[8] = {ls.radiance.x, ls.radiance.y, ls.radiance.z, ls.wi.x, ls.wi.y, ls.wi.z, ls.pdf, vis ? 1.0f : 0.0f};
        std::memcpy(out8 + 8 * i, v, sizeof v);
    }
    return 0;
}
// sampler values: u[i] = ts_uniform(ts_stream_key(seed, px, py, s), dim)
float orc_sampler_u(uint64_t seed, int32_t px, int32_t py, uint32_t s, uint32_t dim) { return ts_uniform(ts_stream_key(seed, px, py, s), dim); }
// deterministic math front ends, vectorised (tests/test_detmath.py)
void orc_detmath(int fn, const float* x, const float* y, uint64_t n, float* out) {
    for (uint64_t i = 0; i < n; ++i) {
        switch (fn) {
        case 0: out[i] = tm_sinf(x[i]); break;
        case 1: out[i] = tm_cosf(x[i]); break;
        case 2: out[i] = tm_tanf(x[i]); break;
        case 3: out[i] = tm_atan2f(y[i], x[i]); break;
        case 4: out[i] = tm_acosf(x[i]); break;
        case 5: out[i] = tm_logf(x[i]); break;
        default: out[i] = tm_atanf(x[i]); break;
        }
    }
}
void orc_detmath_f64(int fn, const double* x, uint64_t n, double* out) {
    for (uint64_t i = 0; i < n; ++i) out[i] = fn == 0 ? tm_sin(x[i]) : tm_cos(x[i]);
}

// ---- SPPM (orc_sppm.h) -------------------------------------------------------------------------------------------------------
// image: (height, width, 3) = _sppm_to_image after the last iteration.  Optional state dumps, all (height, width[, 3]):
// Ld, tau, radius, N (double), and from the last iteration before _update_pixels!: M (int64), phi, vp_p, vp_beta.
// info[0..5] = grid resolution xyz, grid entries, photon hits inside the grid (all iterations), photons per iteration.
// orc_sppm_ex: threads = OpenMP threads for the camera pass (tiles) and the photon pass (photons), as the reference's
// Threads.@threads (sppm.jl:184, 334) — ϕ then depends on the order of the atomic adds, like the reference's; [photon_begin,
// photon_end) + exchange(user, phi3, M, n_pixels): this process's photon slice and the per-iteration sum over processes.
typedef void (*orc_exchange_fn)(void* user, float* phi3, int64_t* M, uint64_t n_pixels);
int orc_sppm_ex(void* sp, const orc_sensor* sn, float initial_radius, int max_depth, int64_t n_iterations, int64_t photons_per_iteration, uint64_t seed, int threads,
                int64_t photon_begin, int64_t photon_end, orc_exchange_fn exchange, void* exchange_user, float* image, float* out_Ld, float* out_tau, float* out_radius, double* out_N,
                int64_t* out_M, float* out_phi, float* out_vp_p, float* out_vp_beta, int64_t* info, orc_stats* stats);
int orc_sppm(void* sp, const orc_sensor* sn, float initial_radius, int max_depth, int64_t n_iterations, int64_t photons_per_iteration, uint64_t seed, float* image,
             float* out_Ld, float* out_tau, float* out_radius, double* out_N, int64_t* out_M, float* out_phi, float* out_vp_p, float* out_vp_beta, int64_t* info,
             orc_stats* stats) {
    return orc_sppm_ex(sp, sn, initial_radius, max_depth, n_iterations, photons_per_iteration, seed, 1, 0, -1, nullptr, nullptr, image, out_Ld, out_tau, out_radius, out_N, out_M, out_phi,
                       out_vp_p, out_vp_beta, info, stats);
}
int orc_sppm_ex(void* sp, const orc_sensor* sn, float initial_radius, int max_depth, int64_t n_iterations, int64_t photons_per_iteration, uint64_t seed, int threads,
                int64_t photon_begin, int64_t photon_end, orc_exchange_fn exchange, void* exchange_user, float* image, float* out_Ld, float* out_tau, float* out_radius, double* out_N,
                int64_t* out_M, float* out_phi, float* out_vp_p, float* out_vp_beta, int64_t* info, orc_stats* stats) {
    OrcScene* s = (OrcScene*)sp;
    if (!s->committed) {
        g_err = "scene not committed";
        return -1;
    }
    const Film film = make_film(sn);
    const PerspectiveCamera cam = make_camera(sn);
    SPPMParams prm;
    prm.initial_search_radius = initial_radius;
    prm.max_depth = max_depth;
    prm.n_iterations = n_iterations;
    prm.photons_per_iteration = photons_per_iteration;
    prm.seed = seed;
    prm.threads = threads > 0 ? threads : 1;
    prm.photon_begin = photon_begin;
    prm.photon_end = photon_end;
    prm.exchange = exchange;
    prm.exchange_user = exchange_user;
    SPPMState st;
    if (!sppm_render(s->scene, cam, film, prm, st, image)) {
        g_err = "SPPM needs a film whose crop starts at pixel (1, 1) (sppm.jl:203 indexes pixels[y, x] with raster coordinates)";
        return -2;
    }
    const size_t n = st.pixels.size();
    for (size_t i = 0; i < n; ++i) {
        const SPPMPixel& p = st.pixels[i];
        if (out_Ld) out_Ld[3 * i] = p.Ld.x, out_Ld[3 * i + 1] = p.Ld.y, out_Ld[3 * i + 2] = p.Ld.z;
        if (out_tau) out_tau[3 * i] = p.tau.x, out_tau[3 * i + 1] = p.tau.y, out_tau[3 * i + 2] = p.tau.z;
        if (out_radius) out_radius[i] = p.radius;
        if (out_N) out_N[i] = p.N;
        if (out_M) out_M[i] = st.last_M[i];
    }
    if (out_phi) std::memcpy(out_phi, st.last_phi.data(), 3 * n * sizeof(float));
    if (out_vp_p) std::memcpy(out_vp_p, st.last_vp_p.data(), 3 * n * sizeof(float));
    if (out_vp_beta) std::memcpy(out_vp_beta, st.last_vp_beta.data(), 3 * n * sizeof(float));
    if (info) {
        info[0] = st.grid_res[0], info[1] = st.grid_res[1], info[2] = st.grid_res[2];
        info[3] = (int64_t)st.grid_entries;
        info[4] = (int64_t)st.photon_hits;
        info[5] = st.photons_per_iteration;
    }
    if (stats) {
        stats->camera_samples = (uint64_t)n * (uint64_t)n_iterations;
        stats->closest_rays = st.totals.closest;
        stats->shadow_rays = st.totals.shadow;
        stats->nodes_visited = st.totals.nodes;
        stats->prims_tested = st.totals.prims;
    }
    return 0;
}
// KAT helpers for the SPPM callees
float orc_radical_inverse(int64_t base_index, uint64_t a) { return radical_inverse(base_index, a); }
uint64_t orc_grid_hash(uint64_t x, uint64_t y, uint64_t z, uint64_t n) { return grid_hash(x, y, z, n) + 1; }  // 1-based like sppm.jl:497-501
// cdf must hold n + 1 floats; returns func_int
float orc_distribution1d(const float* func, int n, float* cdf) {
    const Distribution1D d(std::vector<float>(func, func + n));
    std::memcpy(cdf, d.cdf.data(), (size_t)(n + 1) * sizeof(float));
    return d.func_int;
}
// out3 = offset (1-based), pdf, u_remapped
void orc_sample_discrete(const float* func, int n, float u, float* out3) {
    const Distribution1D d(std::vector<float>(func, func + n));
    const DiscreteSample s = sample_discrete(d, u);
    out3[0] = (float)s.offset, out3[1] = s.pdf, out3[2] = s.u_remapped;
}
// bounds6 = p_min, p_max; res3; out4 = in_bounds, gx, gy, gz
void orc_to_grid(const float* p3, const float* bounds6, const int64_t* res3, int64_t* out4) {
    const GridPoint g = to_grid(V3(p3[0], p3[1], p3[2]), Bounds3(V3(bounds6[0], bounds6[1], bounds6[2]), V3(bounds6[3], bounds6[4], bounds6[5])), res3);
    out4[0] = g.in_bounds, out4[1] = (int64_t)g.g[0], out4[2] = (int64_t)g.g[1], out4[3] = (int64_t)g.g[2];
}
// light index into the scene's lights; out11 = le(3) o(3) d(3) pdf_pos pdf_dir
int orc_sample_le(void* sp, int light, const float* u2, float* out11) {
    OrcScene* s = (OrcScene*)sp;
    if (light < 0 || light >= (int)s->scene.lights.size()) return -1;
    const LeSample ls = sample_le(s->scene.lights[(size_t)light], V2{u2[0], u2[1]});
    const float v[11] = {ls.le.x, ls.le.y, ls.le.z, ls.ray.o.x, ls.ray.o.y, ls.ray.o.z, ls.ray.d.x, ls.ray.d.y, ls.ray.d.z, ls.pdf_pos, ls.pdf_dir};
    std::memcpy(out11, v, sizeof v);
    return 0;
}
}  // extern "C"
