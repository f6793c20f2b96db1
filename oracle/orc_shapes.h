// oracle/orc_shapes.h — TEST INFRASTRUCTURE ONLY.
// Restates surface_interaction.jl, shapes/Shape.jl, shapes/sphere.jl, shapes/triangle_mesh.jl, primitive.jl and the
// traversal half of accel/bvh.jl.  ∂n∂u/∂n∂v, ray differentials and compute_differentials! are dead data on the
// radiance path and are not carried.
#pragma once
#include <memory>

#include "orc_core.h"

namespace orc {

// Shape.jl:1-15
struct ShapeCore {
    Transformation object_to_world, world_to_object;
    bool reverse_orientation = false;
    bool transform_swaps_handedness = false;
    ShapeCore() = default;
    ShapeCore(const Transformation& o2w, bool reverse)
        : object_to_world(o2w), world_to_object(inv(o2w)), reverse_orientation(reverse), transform_swaps_handedness(swaps_handedness(o2w)) {}
    bool flips() const { return reverse_orientation != transform_swaps_handedness; }
};

// surface_interaction.jl:1-49 (fields that reach the radiance path)
struct SurfaceInteraction {
    // core
    V3 p;
    float time = 0;
    V3 wo;
    V3 n;
    // shading
    V3 sh_n, sh_dpdu, sh_dpdv;
    V2 uv;
    V3 dpdu, dpdv;
    int primitive = -1;                     // index into the ordered primitive list of the BVH that produced the hit
    const struct Primitive* prim = nullptr;  // interaction.primitive (primitive.jl:18)
};

// surface_interaction.jl:51-68
inline SurfaceInteraction make_interaction(V3 p, float time, V3 wo, V2 uv, V3 dpdu, V3 dpdv, const ShapeCore* shape) {
    V3 n = normalize(cross(dpdu, dpdv));
    if (shape && shape->flips()) n = n * -1.0f;
    SurfaceInteraction si;
    si.p = p;
    si.time = time;
    si.wo = wo;
    si.n = n;
    si.sh_n = n;
    si.sh_dpdu = dpdu;
    si.sh_dpdv = dpdv;
    si.uv = uv;
    si.dpdu = dpdu;
    si.dpdv = dpdv;
    return si;
}
// surface_interaction.jl:70-88
inline void set_shading_geometry(SurfaceInteraction& i, const ShapeCore* shape, V3 tangent, V3 bitangent, bool orientation_is_authoritative) {
    i.sh_n = normalize(cross(tangent, bitangent));
    if (shape && shape->flips()) i.sh_n = i.sh_n * -1.0f;
    if (orientation_is_authoritative)
        i.n = face_forward(i.n, i.sh_n);
    else
        i.sh_n = face_forward(i.sh_n, i.n);
    i.sh_dpdu = tangent;
    i.sh_dpdv = bitangent;
}
// surface_interaction.jl:154-181  (t::Transformation)(si)
inline SurfaceInteraction transform_interaction(const Transformation& t, const SurfaceInteraction& si) {
    SurfaceInteraction r = si;
    r.p = t.point(si.p);
    r.wo = normalize(t.vec(si.wo));
    r.n = normalize(t.normal(si.n));
    r.sh_n = normalize(t.normal(si.sh_n));
    r.sh_dpdu = t.vec(si.sh_dpdu);
    r.sh_dpdv = t.vec(si.sh_dpdv);
    r.dpdu = t.vec(si.dpdu);
    r.dpdv = t.vec(si.dpdv);
    return r;
}

// Trace.jl:196-211
inline Ray spawn_ray_to(V3 p0, float time, V3 p1, float delta = 1e-6f) {
    const V3 direction = p1 - p0;
    const V3 origin = p0 + delta * direction;
    return Ray{origin, direction, INF32, time};
}
inline Ray spawn_ray_dir(const SurfaceInteraction& si, V3 direction, float delta = 1e-6f) {
    const V3 origin = si.p + delta * direction;
    return Ray{origin, direction, INF32, si.time};
}

// ---- sphere.jl -----------------------------------------------------------------------------------------------------
struct Sphere {
    ShapeCore core;
    float radius = 1, z_min = -1, z_max = 1, theta_min = 0, theta_max = 0, phi_max = 0;
    Sphere() = default;
    // sphere.jl:13-26
    Sphere(const ShapeCore& c, float r, float zmin, float zmax, float phimax_deg) : core(c), radius(r) {
        z_min = jl_clamp(jl_min(zmin, zmax), -r, r);
        z_max = jl_clamp(jl_max(zmin, zmax), -r, r);
        theta_min = tm_acosf(jl_clamp(jl_min(zmin, zmax) / r, -1.0f, 1.0f));
        theta_max = tm_acosf(jl_clamp(jl_max(zmin, zmax) / r, -1.0f, 1.0f));
        phi_max = jl_deg2rad(jl_clamp(phimax_deg, 0.0f, 360.0f));
    }
    // sphere.jl:28-30
    Sphere(const ShapeCore& c, float r, float phimax_deg) : Sphere(c, r, -r, r, phimax_deg) {}
};
inline Bounds3 object_bound(const Sphere& s) { return {V3(-s.radius, -s.radius, s.z_min), V3(s.radius, s.radius, s.z_max)}; }  // :32-37
inline Bounds3 world_bound(const Sphere& s) { return s.core.object_to_world.bounds(object_bound(s)); }                          // Shape.jl:17-19

// sphere.jl:39-54
inline bool solve_quadratic(float a, float b, float c, float& t0, float& t1) {
    float d = b * b - 4 * a * c;
    if (d < 0) return false;
    d = std::sqrt(d);
    const float q = -0.5f * (b + (b < 0 ? -d : d));
    t0 = q / a;
    t1 = c / q;
    if (t0 > t1) std::swap(t0, t1);
    return true;
}
// sphere.jl:56-60
inline V3 refine_intersection(V3 p, const Sphere& s) {
    p = p * (s.radius / distance(V3(0.0f), p));
    if (p.x == 0 && p.y == 0) p = V3(1e-6f * s.radius, p.y, p.z);
    return p;
}
// sphere.jl:65-69
inline bool test_clipping(const Sphere& s, V3 p, float phi) {
    return (s.z_min > -s.radius && p.z < s.z_min) || (s.z_max < s.radius && p.z > s.z_max) || phi > s.phi_max;
}
// sphere.jl:71-75
inline float compute_phi(V3 p) {
    float phi = tm_atan2f(p.y, p.x);
    if (phi < 0.0f) phi += 2.0f * PI_F;
    return phi;
}
// sphere.jl:125-164.  Returns hit, t (shape_hit) and the world-space interaction.
inline bool sphere_intersect(const Sphere& s, const Ray& ray, float& t_hit, SurfaceInteraction& out) {
    const Ray r = s.core.world_to_object.ray(ray);
    const float nd = norm(r.d);
    const float a = nd * nd;
    const float b = dot(2.0f * r.o, r.d);  // `2 * or.o ⋅ or.d` parses as (2*o) ⋅ d (A.16e)
    const float no = norm(r.o);
    const float c = no * no - s.radius * s.radius;
    float t0, t1;
    if (!solve_quadratic(a, b, c, t0, t1)) return false;
    if (t0 > r.t_max || t1 < 0.0f) return false;
    if (t0 < 0) t0 = t1;  // no t_max re-check (A.8)

    float shape_hit = t0;
    V3 hit_point = refine_intersection(r.at(t0), s);
    float phi = compute_phi(hit_point);
    if (test_clipping(s, hit_point, phi)) {
        shape_hit = t1;
        hit_point = refine_intersection(r.at(t1), s);
        phi = compute_phi(hit_point);
        if (test_clipping(s, hit_point, phi)) return false;
    }
    const float u = phi / s.phi_max;
    const float theta = tm_acosf(jl_clamp(hit_point.z / s.radius, -1.0f, 1.0f));
    const float v = (theta - s.theta_min) / (s.theta_max - s.theta_min);
    // precompute_ϕ :77-83
    const float z_radius = std::sqrt(hit_point.x * hit_point.x + hit_point.y * hit_point.y);
    const float inv_z_radius = 1.0f / z_radius;
    const float cos_phi_ = hit_point.x * inv_z_radius;
    const float sin_phi_ = hit_point.y * inv_z_radius;
    // ∂p :88-94
    const V3 dpdu(-s.phi_max * hit_point.y, s.phi_max * hit_point.x, 0.0f);
    const V3 dpdv = (s.theta_max - s.theta_min) * V3(hit_point.z * cos_phi_, hit_point.z * sin_phi_, -s.radius * tm_sinf(theta));
    // interaction built in object space with the WORLD-space wo, then transformed (A.14) :159-162
    const SurfaceInteraction obj = make_interaction(hit_point, ray.time, -ray.d, V2{u, v}, dpdu, dpdv, &s.core);
    out = transform_interaction(s.core.object_to_world, obj);
    t_hit = shape_hit;
    return true;
}
// sphere.jl:166-191
inline bool sphere_intersect_p(const Sphere& s, const Ray& ray) {
    const Ray r = s.core.world_to_object.ray(ray);
    const float nd = norm(r.d);
    const float a = nd * nd;
    const float b = dot(2.0f * r.o, r.d);
    const float no = norm(r.o);
    const float c = no * no - s.radius * s.radius;
    float t0, t1;
    if (!solve_quadratic(a, b, c, t0, t1)) return false;
    if (t0 > r.t_max || t1 < 0.0f) return false;
    if (t0 < 0) t0 = t1;
    V3 hit_point = refine_intersection(r.at(t0), s);
    float phi = compute_phi(hit_point);
    if (test_clipping(s, hit_point, phi)) {
        hit_point = refine_intersection(r.at(t1), s);
        phi = compute_phi(hit_point);
        if (test_clipping(s, hit_point, phi)) return false;
    }
    return true;
}

// ---- triangle_mesh.jl ----------------------------------------------------------------------------------------------
struct TriangleMesh {  // :1-30 — vertices are moved to world space at construction (:23), normals are NOT (A.7)
    std::vector<V3> vertices;
    std::vector<uint32_t> indices;  // 1-based, as in the reference
    std::vector<V3> normals;        // empty = nothing
    std::vector<V3> tangents;       // empty = nothing; one per vertex, not transformed either (:27)
    std::vector<V2> uv;             // empty = nothing; read by CORNER position t.i + j, not through the indices (:82)
    ShapeCore core;                 // every Triangle of a mesh shares the ShapeCore it was created with (:45-58)
    TriangleMesh(const ShapeCore& c, const std::vector<uint32_t>& idx, const std::vector<V3>& verts, const std::vector<V3>& nrm, const std::vector<V3>& tan = {},
                 const std::vector<V2>& uvs = {})
        : indices(idx), normals(nrm), tangents(tan), uv(uvs), core(c) {
        vertices.reserve(verts.size());
        for (const V3& v : verts) vertices.push_back(c.object_to_world.point(v));
    }
};
struct Triangle {  // :32-43
    std::shared_ptr<TriangleMesh> mesh;
    uint32_t i = 1;  // 1-based position of the first index: i = 3k + 1
    const ShapeCore& core() const { return mesh->core; }
};
inline void tri_vertices(const Triangle& t, V3 vs[3]) {  // :70-72
    for (int j = 0; j < 3; ++j) vs[j] = t.mesh->vertices[t.mesh->indices[t.i - 1 + j] - 1];
}
inline void tri_normals(const Triangle& t, V3 ns[3]) {  // :73-75
    for (int j = 0; j < 3; ++j) ns[j] = t.mesh->normals[t.mesh->indices[t.i - 1 + j] - 1];
}
inline void tri_tangents(const Triangle& t, V3 ts[3]) {  // :76-78
    for (int j = 0; j < 3; ++j) ts[j] = t.mesh->tangents[t.mesh->indices[t.i - 1 + j] - 1];
}
inline void tri_uvs(const Triangle& t, V2 uv[3]) {  // :79-83
    if (t.mesh->uv.empty()) {
        uv[0] = V2{0, 0};
        uv[1] = V2{1, 0};
        uv[2] = V2{1, 1};
        return;
    }
    for (int j = 0; j < 3; ++j) uv[j] = t.mesh->uv[t.i - 1 + j];  // mesh.uv[t.i + j], 1-based
}
inline float tri_area(const Triangle& t) {  // :60-63
    V3 vs[3];
    tri_vertices(t, vs);
    return 0.5f * norm(cross(vs[1] - vs[0], vs[2] - vs[0]));
}
inline bool is_degenerate(const V3 vs[3]) {  // :65-68   (v⋅v) ≈ 0  <=>  == 0 (A.1)
    const V3 v = cross(vs[2] - vs[0], vs[1] - vs[0]);
    return dot(v, v) == 0.0f;
}
inline Bounds3 world_bound(const Triangle& t) {  // :97
    V3 vs[3];
    tri_vertices(t, vs);
    return bunion(bunion(Bounds3(vs[0]), Bounds3(vs[1])), Bounds3(vs[2]));
}
inline Bounds3 object_bound(const Triangle& t) {  // :93-96
    V3 vs[3];
    tri_vertices(t, vs);
    return bunion(bunion(Bounds3(t.core().world_to_object.point(vs[0])), Bounds3(t.core().world_to_object.point(vs[1]))),
                  Bounds3(t.core().world_to_object.point(vs[2])));
}
// :99-123 — permutation WITHOUT the winding-preserving swap (A.7)
inline void to_ray_coordinate_space(const V3 vs[3], const Ray& ray, V3 tvs[3], V3& shear) {
    const V3 ad = vabs(ray.d);
    int kz = 0;  // argmax returns the first maximum (A.16g)
    if (ad.y > ad[kz]) kz = 1;
    if (ad.z > ad[kz]) kz = 2;
    int kx = kz + 1;
    if (kx == 3) kx = 0;
    int ky = kx + 1;
    if (ky == 3) ky = 0;
    const V3 d(ray.d[kx], ray.d[ky], ray.d[kz]);
    const float denom = 1.0f / d.z;
    shear = V3(-d.x * denom, -d.y * denom, denom);
    for (int i = 0; i < 3; ++i) {
        const V3 vo = vs[i] - ray.o;
        const float dz = vs[i][kz] - ray.o[kz];
        tvs[i] = V3(vo[kx], vo[ky], vo[kz]) + V3(shear.x * dz, shear.y * dz, 0.0f);
    }
}
template <class T>
struct E3 {
    T a, b, c;
};
// :85-91
inline E3<float> edge_function(const V3 v[3]) {
    return {v[1].x * v[2].y - v[1].y * v[2].x, v[2].x * v[0].y - v[2].y * v[0].x, v[0].x * v[1].y - v[0].y * v[1].x};
}
inline E3<double> edge_function_f64(const V3 v[3]) {
    const double x0 = v[0].x, y0 = v[0].y, x1 = v[1].x, y1 = v[1].y, x2 = v[2].x, y2 = v[2].y;
    return {x1 * y2 - y1 * x2, x2 * y0 - y2 * x0, x0 * y1 - y0 * x1};
}
// Result of the edge/range tests shared by intersect and intersect_p (:189-214 == :247-270).
struct TriHit {
    bool hit = false;
    float t = 0;
    V3 bary;
};
template <class T>
inline bool tri_core_test(const E3<T>& e, const V3 tvs[3], float shear_z, float t_max, TriHit* out) {
    if ((e.a < 0 || e.b < 0 || e.c < 0) && (e.a > 0 || e.b > 0 || e.c > 0)) return false;
    const T det = e.a + e.b + e.c;
    if (det == 0) return false;
    const T t_scaled = e.a * tvs[0].z * shear_z + e.b * tvs[1].z * shear_z + e.c * tvs[2].z * shear_z;
    if (det < 0 && (t_scaled >= 0 || t_scaled < t_max * det)) return false;
    if (det > 0 && (t_scaled <= 0 || t_scaled > t_max * det)) return false;
    if (out) {
        const T inv_det = 1.0f / det;
        // (in the Float64 fall-back the reference would hand a Point3{Float64} to a Point3f-only constructor and
        //  throw; we round to Float32 instead — documented divergence)
        out->bary = V3((float)(e.a * inv_det), (float)(e.b * inv_det), (float)(e.c * inv_det));
        out->t = (float)(t_scaled * inv_det);
        out->hit = true;
    }
    return true;
}
inline bool tri_test(const V3 vs[3], const Ray& ray, TriHit* out) {
    if (is_degenerate(vs)) return false;  // (intersect_p returns a tuple from a ::Bool function here: treat as false, A.7)
    V3 tvs[3], shear;
    to_ray_coordinate_space(vs, ray, tvs, shear);
    const E3<float> e = edge_function(tvs);
    if (e.a == 0 && e.b == 0 && e.c == 0) {  // :195-197 fall back to double precision
        const E3<double> ed = edge_function_f64(tvs);
        return tri_core_test(ed, tvs, shear.z, ray.t_max, out);
    }
    return tri_core_test(e, tvs, shear.z, ray.t_max, out);
}
// :125-141 with the default uvs of :79-83
inline void tri_dp(const V3 vs[3], const V2 uv[3], V3& dpdu, V3& dpdv, V3& dp13, V3& dp23) {
    const V2 duv13 = uv[0] - uv[2], duv23 = uv[1] - uv[2];
    dp13 = vs[0] - vs[2];
    dp23 = vs[1] - vs[2];
    const float det = duv13.x * duv23.y - duv13.y * duv23.x;
    if (det == 0) {
        const V3 v = normalize(cross(vs[2] - vs[0], vs[1] - vs[0]));
        coordinate_system(v, dpdu, dpdv);
        return;
    }
    const float inv_det = 1.0f / det;
    dpdu = (duv23.y * dp13 - duv13.y * dp23) * inv_det;
    dpdv = (-duv23.x * dp13 + duv13.x * dp23) * inv_det;
}
// :187-243
inline bool triangle_intersect(const Triangle& t, const Ray& ray, float& t_hit, SurfaceInteraction& out) {
    V3 vs[3];
    tri_vertices(t, vs);
    TriHit h;
    if (!tri_test(vs, ray, &h)) return false;
    V2 uv[3];
    tri_uvs(t, uv);  // :215
    V3 dpdu, dpdv, dp13, dp23;
    tri_dp(vs, uv, dpdu, dpdv, dp13, dp23);
    const V3 hit_point = sum_mul(h.bary, vs);
    const V2 uv_hit = sum_mul(h.bary, uv);
    SurfaceInteraction si = make_interaction(hit_point, ray.time, -ray.d, uv_hit, dpdu, dpdv, &t.core());
    si.n = si.sh_n = normalize(cross(dp13, dp23));  // :230
    const bool has_normals = !t.mesh->normals.empty(), has_tangents = !t.mesh->tangents.empty();
    if (has_normals || has_tangents) {  // _init_triangle_shading_geometry! :160-185
        V3 ns = si.n;  // :168
        if (has_normals) {
            V3 nrm[3];
            tri_normals(t, nrm);
            ns = normalize(sum_mul(h.bary, nrm));
        }
        V3 ss;
        if (has_tangents) {  // :172-176
            V3 tg[3];
            tri_tangents(t, tg);
            ss = normalize(sum_mul(h.bary, tg));
        } else {
            ss = normalize(si.dpdu);
        }
        V3 ts = cross(ns, ss);
        if (dot(ts, ts) > 0) {
            ts = normalize(ts);
            ss = cross(ts, ns);
        } else {
            coordinate_system(ns, ss, ts);
        }
        set_shading_geometry(si, &t.core(), ss, ts, true);
    }
    if (has_normals) {
        si.n = face_forward(si.n, si.sh_n);  // :234-237
    } else if (t.core().flips()) {
        si.n = si.sh_n = -si.n;  // :238-240
    }
    out = si;
    t_hit = h.t;
    return true;
}
// :245-273
inline bool triangle_intersect_p(const Triangle& t, const Ray& ray) {
    V3 vs[3];
    tri_vertices(t, vs);
    return tri_test(vs, ray, nullptr);
}

// ---- primitive.jl + accel/bvh.jl:212-299 -----------------------------------------------------------------------------
struct BVHAccel;
struct Primitive {  // GeometricPrimitive{Sphere|Triangle} or a nested BVHAccel (test_intersection.jl:137-138)
    enum Kind { SPHERE, TRIANGLE, BVH } kind = SPHERE;
    std::shared_ptr<Sphere> sphere;
    Triangle triangle;
    std::shared_ptr<BVHAccel> bvh;
    int material = -1;  // index into the scene's material table; -1 = nothing
    int user_id = -1;   // position in the caller's primitive list
};
struct LinearNode {  // bvh.jl:38-48: LinearBVHLeaf / LinearBVHInterior, 1-based indices
    Bounds3 bounds;
    bool leaf = false;
    uint32_t primitives_offset = 0, n_primitives = 0;  // leaf
    uint32_t second_child_offset = 0;                  // interior
    uint8_t split_axis = 0;                            // interior, 1..3
};
struct BVHAccel {
    std::vector<Primitive> primitives;  // ordered_primitives
    std::vector<LinearNode> nodes;
};
// Visit / ray counters for the algorithmic-bytes and Mray/s figures (not in the reference); one set per thread.
struct Counters {
    uint64_t nodes = 0, prims = 0, closest = 0, shadow = 0;
};
inline Counters& counters() {
    static thread_local Counters c;
    return c;
}
inline Bounds3 world_bound(const BVHAccel& b) { return b.nodes.empty() ? Bounds3() : b.nodes[0].bounds; }  // bvh.jl:208-210
inline Bounds3 world_bound(const Primitive& p) {
    switch (p.kind) {
    case Primitive::SPHERE: return world_bound(*p.sphere);
    case Primitive::TRIANGLE: return world_bound(p.triangle);
    default: return world_bound(*p.bvh);
    }
}
bool bvh_intersect(BVHAccel& bvh, Ray& ray, SurfaceInteraction& out);
bool bvh_intersect_p(BVHAccel& bvh, Ray& ray);

// primitive.jl:12-20
inline bool primitive_intersect(Primitive& p, int index_in_parent, Ray& ray, SurfaceInteraction& out) {
    if (p.kind == Primitive::BVH) return bvh_intersect(*p.bvh, ray, out);
    float t_hit;
    SurfaceInteraction si;
    const bool hit = p.kind == Primitive::SPHERE ? sphere_intersect(*p.sphere, ray, t_hit, si) : triangle_intersect(p.triangle, ray, t_hit, si);
    if (!hit) return false;
    ray.t_max = t_hit;
    si.primitive = index_in_parent;
    si.prim = &p;
    out = si;
    return true;
}
// primitive.jl:22-26
inline bool primitive_intersect_p(Primitive& p, Ray& ray) {
    if (p.kind == Primitive::BVH) return bvh_intersect_p(*p.bvh, ray);
    return p.kind == Primitive::SPHERE ? sphere_intersect_p(*p.sphere, ray) : triangle_intersect_p(p.triangle, ray);
}

// bvh.jl:212-258
inline bool bvh_intersect(BVHAccel& bvh, Ray& ray, SurfaceInteraction& out) {
    bool hit = false;
    if (bvh.nodes.empty()) return false;
    check_direction(ray);
    const V3 inv_dir(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z);
    int neg[3];
    is_dir_negative(ray.d, neg);
    int to_visit_offset = 1, current = 1;
    int32_t nodes_to_visit[64] = {0};
    while (true) {
        const LinearNode& ln = bvh.nodes[current - 1];
        counters().nodes++;
        if (bounds_intersect_p(ln.bounds, ray, inv_dir, neg)) {
            if (ln.leaf && ln.n_primitives > 0) {
                for (uint32_t i = 0; i < ln.n_primitives; ++i) {
                    SurfaceInteraction tmp;
                    counters().prims++;
                    const int idx = (int)(ln.primitives_offset + i) - 1;
                    if (primitive_intersect(bvh.primitives[idx], idx, ray, tmp)) {
                        hit = true;
                        out = tmp;
                    }
                }
                if (to_visit_offset == 1) break;
                to_visit_offset -= 1;
                current = nodes_to_visit[to_visit_offset - 1];
            } else {
                if (neg[ln.split_axis - 1] == 2) {
                    nodes_to_visit[to_visit_offset - 1] = current + 1;
                    current = (int)ln.second_child_offset;
                } else {
                    nodes_to_visit[to_visit_offset - 1] = (int)ln.second_child_offset;
                    current += 1;
                }
                to_visit_offset += 1;
            }
        } else {
            if (to_visit_offset == 1) break;
            to_visit_offset -= 1;
            current = nodes_to_visit[to_visit_offset - 1];
        }
    }
    return hit;
}
// bvh.jl:260-299
inline bool bvh_intersect_p(BVHAccel& bvh, Ray& ray) {
    if (bvh.nodes.empty()) return false;
    check_direction(ray);
    const V3 inv_dir(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z);
    int neg[3];
    is_dir_negative(ray.d, neg);
    int to_visit_offset = 1, current = 1;
    int32_t nodes_to_visit[64] = {0};
    while (true) {
        const LinearNode& ln = bvh.nodes[current - 1];
        counters().nodes++;
        if (bounds_intersect_p(ln.bounds, ray, inv_dir, neg)) {
            if (ln.leaf && ln.n_primitives > 0) {
                for (uint32_t i = 0; i < ln.n_primitives; ++i) {
                    counters().prims++;
                    if (primitive_intersect_p(bvh.primitives[ln.primitives_offset + i - 1], ray)) return true;
                }
                if (to_visit_offset == 1) break;
                to_visit_offset -= 1;
                current = nodes_to_visit[to_visit_offset - 1];
            } else {
                if (neg[ln.split_axis - 1] == 2) {
                    nodes_to_visit[to_visit_offset - 1] = current + 1;
                    current = (int)ln.second_child_offset;
                } else {
                    nodes_to_visit[to_visit_offset - 1] = (int)ln.second_child_offset;
                    current += 1;
                }
                to_visit_offset += 1;
            }
        } else {
            if (to_visit_offset == 1) break;
            to_visit_offset -= 1;
            current = nodes_to_visit[to_visit_offset - 1];
        }
    }
    return false;
}

}  // namespace orc
