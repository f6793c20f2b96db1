// th_device.h — device-side geometry and scattering: Triangle / Sphere intersection (shapes/*.jl), SurfaceInteraction
// + BSDF frame (surface_interaction.jl, materials/bsdf.jl), BxDFs (reflection/*.jl), δ-lights (lights/*.jl).
// Written for gfx950 registers, not as a translation of the reference's object graph: a hit is {t, slot}; the
// interaction is rebuilt once per shaded vertex from the one hit primitive (the reference rebuilds it for every accepted
// candidate during traversal); a material is ≤2 precomputed lobes.  Operation order inside every formula follows the
// cited reference lines so that results equal the CPU oracle's bit-for-bit.
#pragma once
#include "th_scene.h"

namespace th {

// Read-only scene data addressed wave-uniformly (same index in every lane): viewed through the constant address space so that the
// loads are scalar (s_load) and their results live in SGPRs.  The scene is never written while a kernel that reads it runs.
template <class T>
TH_D T uniform_load(const T* p, uint32_t i) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(4))) T* ConstPtr;
    return ((ConstPtr)(uintptr_t)p)[i];
#else
    return p[i];
#endif
}


// ---- triangle (shapes/triangle_mesh.jl) ---------------------------------------------------------------------------------
struct TriTest {
    float t;
    f3 bary;
};
// :65-68  is_degenerate
TH_D bool tri_degenerate(f3 v0, f3 v1, f3 v2) {
    const f3 v = cross(v2 - v0, v1 - v0);
    return dot(v, v) == 0.0f;
}
// :99-123 + :189-218 (== :247-270).  t_max test is inclusive of equality (A.6); no kx/ky swap (A.7).
// What the triangle test derives from the ray alone (:99-114): the dominant axis and the shear.  Computed once per ray by the
// traversal kernels instead of once per triangle (one correctly rounded division among it).
struct RayShear {
    int kz;
    float sx, sy, sz;
};
TH_HD RayShear ray_shear(f3 d) {
    const float ax = fabs_(d.x), ay = fabs_(d.y), az = fabs_(d.z);
    RayShear r;
    r.kz = 0;
    float am = ax;
    if (ay > am) {
        r.kz = 1;
        am = ay;
    }
    if (az > am) r.kz = 2;
    const float dpx = r.kz == 0 ? d.y : (r.kz == 1 ? d.z : d.x), dpy = r.kz == 0 ? d.z : (r.kz == 1 ? d.x : d.y), dpz = r.kz == 0 ? d.x : (r.kz == 1 ? d.y : d.z);
    const float denom = 1.0f / dpz;
    r.sx = -dpx * denom;
    r.sy = -dpy * denom;
    r.sz = denom;
    return r;
}
// the same with 1 / d at hand (a walk's refill computes it for the box tests): 1 / d[kz] IS the reciprocal of that component, bit for bit — one division less per ray
TH_HD RayShear ray_shear(f3 d, f3 inv_d) {
    const float ax = fabs_(d.x), ay = fabs_(d.y), az = fabs_(d.z);
    RayShear r;
    r.kz = 0;
    float am = ax;
    if (ay > am) {
        r.kz = 1;
        am = ay;
    }
    if (az > am) r.kz = 2;
    const float dpx = r.kz == 0 ? d.y : (r.kz == 1 ? d.z : d.x), dpy = r.kz == 0 ? d.z : (r.kz == 1 ? d.x : d.y);
    const float denom = r.kz == 0 ? inv_d.x : (r.kz == 1 ? inv_d.y : inv_d.z);
    r.sx = -dpx * denom;
    r.sy = -dpy * denom;
    r.sz = denom;
    return r;
}
template <bool WANT_HIT, bool SELECTS = true>
TH_D bool tri_intersect_sheared(f3 v0, f3 v1, f3 v2, f3 o, const RayShear& rs, float t_max, TriTest* out);
template <bool WANT_HIT>
TH_D bool tri_intersect(f3 v0, f3 v1, f3 v2, f3 o, f3 d, float t_max, TriTest* out) {  // the shading kernels' entry (hit geometry rebuilt from one primitive)
    if (tri_degenerate(v0, v1, v2)) return false;
    return tri_intersect_sheared<WANT_HIT, false>(v0, v1, v2, o, ray_shear(d), t_max, out);
}
// The test proper, for a triangle already known not to be degenerate (PRIM_DEGENERATE is set at scene commit).
// SELECTS: the permutation by 18 selects instead of a three-way branch — a wave of a traversal kernel mixes rays of all three dominant axes
// and ran the branch three times with a third of its lanes each (S-cornell closest-hit 50.8 -> 49.6 ms, S-blob 188.6 -> 186.4); in the shading
// kernels, where the test is a cold path, the selects cost registers (S-mesh shading 60 -> 66 ms): they keep the branch.
template <bool WANT_HIT, bool SELECTS>
TH_D bool tri_intersect_sheared(f3 v0, f3 v1, f3 v2, f3 o, const RayShear& rs, float t_max, TriTest* out) {
    const int kz = rs.kz;
    // permute so that kz is last: (kx, ky, kz) = (kz+1, kz+2, kz) mod 3; vertices[i][kz] - ray.o[kz] (:116-117) is the same subtraction as the
    // translated vertex's kz component
    const f3 p0 = v0 - o, p1 = v1 - o, p2 = v2 - o;
    f3 a0, a1, a2;
    if (SELECTS) {
        const bool k0 = kz == 0, k1 = kz == 1;
        a0 = mk3(k0 ? p0.y : (k1 ? p0.z : p0.x), k0 ? p0.z : (k1 ? p0.x : p0.y), k0 ? p0.x : (k1 ? p0.y : p0.z));
        a1 = mk3(k0 ? p1.y : (k1 ? p1.z : p1.x), k0 ? p1.z : (k1 ? p1.x : p1.y), k0 ? p1.x : (k1 ? p1.y : p1.z));
        a2 = mk3(k0 ? p2.y : (k1 ? p2.z : p2.x), k0 ? p2.z : (k1 ? p2.x : p2.y), k0 ? p2.x : (k1 ? p2.y : p2.z));
    } else if (kz == 0) {
        a0 = mk3(p0.y, p0.z, p0.x);
        a1 = mk3(p1.y, p1.z, p1.x);
        a2 = mk3(p2.y, p2.z, p2.x);
    } else if (kz == 1) {
        a0 = mk3(p0.z, p0.x, p0.y);
        a1 = mk3(p1.z, p1.x, p1.y);
        a2 = mk3(p2.z, p2.x, p2.y);
    } else {
        a0 = p0;
        a1 = p1;
        a2 = p2;
    }
    const float dz0 = a0.z, dz1 = a1.z, dz2 = a2.z;
    const float sx = rs.sx, sy = rs.sy, sz = rs.sz;
    const float x0 = a0.x + sx * dz0, y0 = a0.y + sy * dz0, z0 = a0.z + 0.0f;
    const float x1 = a1.x + sx * dz1, y1 = a1.y + sy * dz1, z1 = a1.z + 0.0f;
    const float x2 = a2.x + sx * dz2, y2 = a2.y + sy * dz2, z2 = a2.z + 0.0f;
    // _edge_function :85-91
    const float e0 = x1 * y2 - y1 * x2;
    const float e1 = x2 * y0 - y2 * x0;
    const float e2 = x0 * y1 - y0 * x1;
    if (e0 == 0.0f && e1 == 0.0f && e2 == 0.0f) {
        // :195-197 Float64 fall-back; everything downstream is then Float64 in the reference (promotion)
        const double E0 = (double)x1 * (double)y2 - (double)y1 * (double)x2;
        const double E1 = (double)x2 * (double)y0 - (double)y2 * (double)x0;
        const double E2 = (double)x0 * (double)y1 - (double)y0 * (double)x1;
        if ((E0 < 0 || E1 < 0 || E2 < 0) && (E0 > 0 || E1 > 0 || E2 > 0)) return false;
        const double det = E0 + E1 + E2;
        if (det == 0) return false;
        const double ts = E0 * z0 * sz + E1 * z1 * sz + E2 * z2 * sz;
        if (det < 0 && (ts >= 0 || ts < t_max * det)) return false;
        if (det > 0 && (ts <= 0 || ts > t_max * det)) return false;
        if (WANT_HIT) {
            const double inv_det = 1.0f / det;
            out->bary = mk3((float)(E0 * inv_det), (float)(E1 * inv_det), (float)(E2 * inv_det));
            out->t = (float)(ts * inv_det);
        }
        return true;
    }
    if ((e0 < 0 || e1 < 0 || e2 < 0) && (e0 > 0 || e1 > 0 || e2 > 0)) return false;
    const float det = e0 + e1 + e2;
    if (det == 0.0f) return false;
    const float ts = e0 * z0 * sz + e1 * z1 * sz + e2 * z2 * sz;
    if (det < 0 && (ts >= 0 || ts < t_max * det)) return false;
    if (det > 0 && (ts <= 0 || ts > t_max * det)) return false;
    if (WANT_HIT) {
        const float inv_det = 1.0f / det;
        out->bary = mk3(e0 * inv_det, e1 * inv_det, e2 * inv_det);
        out->t = ts * inv_det;
    }
    return true;
}

// ---- sphere (shapes/sphere.jl) --------------------------------------------------------------------------------------------
TH_D bool solve_quadratic(float a, float b, float c, float& t0, float& t1) {  // :39-54
    float d = b * b - 4 * a * c;
    if (d < 0) return false;
    d = sqrt_(d);
    const float q = -0.5f * (b + (b < 0 ? -d : d));
    t0 = q / a;
    t1 = c / q;
    if (t0 > t1) {
        const float tmp = t0;
        t0 = t1;
        t1 = tmp;
    }
    return true;
}
TH_D f3 sphere_refine(f3 p, float radius) {  // :56-60
    p = p * (radius / norm(mk3(0.0f, 0.0f, 0.0f) - p));
    if (p.x == 0.0f && p.y == 0.0f) p = mk3(1e-6f * radius, p.y, p.z);
    return p;
}
TH_D float sphere_phi(f3 p) {  // :71-75
    float phi = tm_atan2f(p.y, p.x);
    if (phi < 0.0f) phi += 2.0f * kPi;
    return phi;
}
TH_D bool sphere_clipped(const SphereRec& s, f3 p, float phi) {  // :65-69
    return (s.z_min > -s.radius && p.z < s.z_min) || (s.z_max < s.radius && p.z > s.z_max) || phi > s.phi_max;
}
struct SphereHit {
    float t;
    f3 p_obj;  // refined object-space hit point
    float phi;
};
// :125-158 / :166-191 up to the interaction.  `t0 < 0 && (t0 = t1)` without re-checking t_max (A.8).
// FULL_ONLY: the caller guarantees `never_clipped` for every sphere of the scene (trhip_scene: partial_spheres == false), and the
// clipped-sphere code — a Float64 atan2 that sets the register count of every kernel it is inlined into — is compiled out.
template <bool WANT_POINT, bool FULL_ONLY = false>
TH_D bool sphere_intersect(const SphereRec& s, f3 o, f3 d, float t_max, SphereHit& h) {
    const f3 oo = xf_point(s.o2w_inv, o);
    const f3 od = xf_vec(s.o2w_inv, d);
    const float nd = norm(od);
    const float a = nd * nd;
    const float b = dot(2.0f * oo, od);
    const float no = norm(oo);
    const float c = no * no - s.radius * s.radius;
    float t0, t1;
    if (!solve_quadratic(a, b, c, t0, t1)) return false;
    if (t0 > t_max || t1 < 0.0f) return false;
    if (t0 < 0) t0 = t1;
    float shape_hit = t0;
    if (FULL_ONLY || s.never_clipped) {
        // Full sphere: test_clipping (sphere.jl:65-69) is false for every hit point — the z clauses are off and ϕ, which is
        // atan(y, x) (+ 2π if negative) and therefore never exceeds Float32(2π), cannot exceed ϕ_max >= Float32(2π).
        // The Float64 atan2 and, for traversal, the refined hit point are not needed to decide the hit.
        if (WANT_POINT) h.p_obj = sphere_refine(oo + od * t0, s.radius);
        h.t = shape_hit;
        h.phi = 0.0f;
        return true;
    }
    f3 hp = sphere_refine(oo + od * t0, s.radius);
    float phi = sphere_phi(hp);
    if (sphere_clipped(s, hp, phi)) {
        shape_hit = t1;
        hp = sphere_refine(oo + od * t1, s.radius);
        phi = sphere_phi(hp);
        if (sphere_clipped(s, hp, phi)) return false;
    }
    h.t = shape_hit;
    h.p_obj = hp;
    h.phi = phi;
    return true;
}

// ---- interaction + BSDF frame -----------------------------------------------------------------------------------------------
struct Shading {  // what BSDF(si) (bsdf.jl:41-50) and the integrators read from a SurfaceInteraction
    f3 p;         // core.p
    f3 wo;        // core.wo
    f3 ng;        // core.n
    f3 ns;        // shading.n
    f3 ss;        // normalize(shading.∂p∂u)
    f3 ts;        // ns × ss (not re-normalised, A.11)
};
// triangle_mesh.jl:219-243 + 125-141 + 160-185 + surface_interaction.jl:51-88
// The two quantities of a triangle's interaction that depend on nothing but its vertices: the geometric normal (:230) and the
// normalised ∂p∂u.  k_shade_constants evaluates this once per slot at commit (same code, same device: same bits) and leaves the
// results in the slot's shading record; shade_triangle then takes them from there (`pre`) instead of redoing two cross / normalise
// chains per path vertex.
struct TriConstants {
    f3 n, ss;
};
// uv: the triangle's three (u, v) pairs {u0, v0, u1, v1, u2, v2} (uvs(t), :79-83), or null for the default (0,0) (1,0) (1,1)
TH_D TriConstants triangle_constants(f3 v0, f3 v1, f3 v2, const float* uv = nullptr) {
    // ∂p (:125-141)
    const float u0 = uv ? uv[0] : 0.0f, w0 = uv ? uv[1] : 0.0f, u1 = uv ? uv[2] : 1.0f, w1 = uv ? uv[3] : 0.0f, u2 = uv ? uv[4] : 1.0f, w2 = uv ? uv[5] : 1.0f;
    const float du13x = u0 - u2, du13y = w0 - w2, du23x = u1 - u2, du23y = w1 - w2;
    const f3 dp13 = v0 - v2, dp23 = v1 - v2;
    const float det = du13x * du23y - du13y * du23x;
    f3 dpdu;
    if (det == 0.0f) {  // `det ≈ 0` (:132) is true for 0 only; unreachable with the default uvs
        f3 t2;
        coordinate_system(normalize(cross(v2 - v0, v1 - v0)), dpdu, t2);
    } else {
        const float inv_det = 1.0f / det;
        dpdu = (du23y * dp13 - du13y * dp23) * inv_det;
    }
    TriConstants c;
    c.n = normalize(cross(dp13, dp23));  // :230 overrides the constructor's normal
    c.ss = normalize(dpdu);
    return c;
}
// tan: the slot's three vertex tangents (float4 records) when the mesh has them, else null — fetched HERE, after the normals have been used up, so that a scene
// without tangents (every scene of the reference) keeps the register budget it had
TH_D Shading shade_triangle(f3 v0, f3 v1, f3 v2, bool has_normals, f3 n0, f3 n1, f3 n2, bool flip, f3 bary, f3 ray_d, const TriConstants* pre = nullptr, const float4* tan = nullptr) {
    const bool has_tangents = tan != nullptr;
    Shading s;
    const TriConstants tc = pre ? *pre : triangle_constants(v0, v1, v2);
    s.p = bary.x * v0 + bary.y * v1 + bary.z * v2;  // sum_mul(barycentric, vs) :222
    s.wo = -ray_d;
    f3 n = tc.n;
    f3 shn = n;
    bool ss_is_unit = true;  // s.ss below is normalize(∂p∂u): tc.ss itself unless the shading normals replace ∂p∂u
    f3 sh_dpdu = tc.ss;
    if (has_normals || has_tangents) {  // _init_triangle_shading_geometry! (:160-185)
        const f3 nsn = has_normals ? normalize(bary.x * n0 + bary.y * n1 + bary.z * n2) : n;
        f3 ss = tc.ss;
        if (has_tangents) {
            const float4 ta = tan[0], tb = tan[1], tcn = tan[2];
            ss = normalize(bary.x * mk3(ta.x, ta.y, ta.z) + bary.y * mk3(tb.x, tb.y, tb.z) + bary.z * mk3(tcn.x, tcn.y, tcn.z));
        }
        f3 ts = cross(nsn, ss);
        if (dot(ts, ts) > 0.0f) {
            ts = normalize(ts);
            ss = cross(ts, nsn);
        } else {
            coordinate_system(nsn, ss, ts);
        }
        // set_shading_geometry!(…, orientation_is_authoritative = true)
        shn = normalize(cross(ss, ts));
        if (flip) shn = shn * -1.0f;
        n = face_forward(n, shn);
        sh_dpdu = ss;
        ss_is_unit = false;
        if (has_normals) {
            n = face_forward(n, shn);  // :234-237
        } else if (flip) {             // :238-239 (a mesh with tangents only)
            n = -n;
            shn = n;
        }
    } else if (flip) {
        n = -n;
        shn = n;
    }
    s.ng = n;
    s.ns = shn;
    s.ss = ss_is_unit ? sh_dpdu : normalize(sh_dpdu);
    s.ts = cross(s.ns, s.ss);
    return s;
}
// sphere.jl:144-162 + surface_interaction.jl:51-68 + 154-181
TH_D Shading shade_sphere(const SphereRec& s, const SphereHit& h, f3 ray_d) {
    const f3 hp = h.p_obj;
    const float theta = tm_acosf(jclamp(hp.z / s.radius, -1.0f, 1.0f));
    const float z_radius = sqrt_(hp.x * hp.x + hp.y * hp.y);
    const float inv_z_radius = 1.0f / z_radius;
    const float cphi = hp.x * inv_z_radius, sphi = hp.y * inv_z_radius;
    const f3 dpdu = mk3(-s.phi_max * hp.y, s.phi_max * hp.x, 0.0f);
    const f3 dpdv = (s.theta_max - s.theta_min) * mk3(hp.z * cphi, hp.z * sphi, -s.radius * tm_sinf(theta));
    f3 n = normalize(cross(dpdu, dpdv));
    if (s.flip) n = n * -1.0f;
    Shading r;
    r.p = xf_point(s.o2w, hp);
    r.wo = normalize(xf_vec(s.o2w, -ray_d));  // world-space wo pushed through object_to_world (A.14)
    r.ng = normalize(xf_normal(s.o2w_inv, n));
    r.ns = normalize(xf_normal(s.o2w_inv, n));
    r.ss = normalize(xf_vec(s.o2w, dpdu));
    r.ts = cross(r.ns, r.ss);
    return r;
}

// ---- BxDFs (reflection/*.jl) ---------------------------------------------------------------------------------------------
TH_D f3 lobe_r(const Lobe& l) { return mk3(l.r[0], l.r[1], l.r[2]); }
TH_D f3 lobe_t(const Lobe& l) { return mk3(l.t[0], l.t[1], l.t[2]); }
TH_D bool lobe_matches(const Lobe& l, int flags) { return (l.type & flags) == l.type; }  // bxdf.jl:9-11
TH_D bool same_hemisphere(f3 w, f3 wp) { return w.z * wp.z > 0.0f; }                    // bxdf.jl:13-15
// bxdf.jl:52-62
TH_D bool refract(f3 wi, f3 n, float eta, f3& wt) {
    const float cos_i = dot(n, wi);
    const float sin2_i = jmax(0.0f, 1.0f - cos_i * cos_i);
    const float sin2_t = (eta * eta) * sin2_i;
    if (sin2_t >= 1.0f) {
        wt = splat3(0.0f);
        return false;
    }
    const float cos_t = sqrt_(1.0f - sin2_t);
    wt = (-eta) * wi + (eta * cos_i - cos_t) * n;
    return true;
}
// bxdf.jl:74-95
TH_D float fresnel_dielectric(float cos_i, float eta_i, float eta_t) {
    cos_i = jclamp(cos_i, -1.0f, 1.0f);
    if (cos_i <= 0.0f) {
        const float tmp = eta_i;
        eta_i = eta_t;
        eta_t = tmp;
        cos_i = fabs_(cos_i);
    }
    const float sin_i = sqrt_(jmax(0.0f, 1.0f - cos_i * cos_i));
    const float sin_t = sin_i * eta_i / eta_t;
    if (sin_t >= 1.0f) return 1.0f;
    const float cos_t = sqrt_(jmax(0.0f, 1.0f - sin_t * sin_t));
    const float r_par = (eta_t * cos_i - eta_i * cos_t) / (eta_t * cos_i + eta_i * cos_t);
    const float r_perp = (eta_i * cos_i - eta_t * cos_t) / (eta_i * cos_i + eta_t * cos_t);
    return 0.5f * (r_par * r_par + r_perp * r_perp);
}
TH_D f3 lobe_fresnel(const Lobe& l, float cos_i) {  // bxdf.jl:127-140
    if (l.fresnel == FRESNEL_NOOP) return splat3(1.0f);
    return splat3(fresnel_dielectric(cos_i, l.fr_eta_i, l.fr_eta_t));
}
// TrowbridgeReitzDistribution (microfacet.jl:53-201), α_x = l.a, α_y = l.b, sample_visible_area = true
TH_D float tr_lambda(float ax, float ay, f3 w) {  // :68-75
    const float th = fabs_(tan_theta(w));
    if (isinf_(th)) return 0.0f;
    const float cp = cos_phi(w), sp = sin_phi(w);
    const float alpha = sqrt_(cp * cp * (ax * ax) + sp * sp * (ay * ay));
    const float at = alpha * th;
    return (-1.0f + sqrt_(1.0f + at * at)) / 2.0f;
}
TH_D float tr_G1(float ax, float ay, f3 w) { return 1.0f / (1.0f + tr_lambda(ax, ay, w)); }                               // :89-91
TH_D float tr_G(float ax, float ay, f3 wo, f3 wi) { return 1.0f / (1.0f + tr_lambda(ax, ay, wo) + tr_lambda(ax, ay, wi)); }  // :93-95
TH_D float tr_D(float ax, float ay, f3 w) {                                                                               // :101-108
    const float tt = tan_theta(w);
    const float tan2 = tt * tt;
    if (isinf_(tan2)) return 0.0f;
    const float cos4 = pow4(cos_theta(w));
    const float cp = cos_phi(w), sp = sin_phi(w);
    const float e = (cp * cp / (ax * ax) + sp * sp / (ay * ay)) * tan2;
    const float ope = 1.0f + e;
    return 1.0f / (kPi * ax * ay * cos4 * (ope * ope));
}
TH_D float tr_pdf(float ax, float ay, f3 wo, f3 wh) {  // :110-113
    return tr_D(ax, ay, wh) * tr_G1(ax, ay, wo) * fabs_(dot(wo, wh)) / fabs_(cos_theta(wo));
}
TH_D void tr_sample_11(float cos_t, float u1, float u2, float& slope_x, float& slope_y) {  // :115-155
    if (cos_t > 0.9999f) {
        const float r = sqrt_(u1 / (1.0f - u1));
        const double phi = 6.28318530718 * (double)u2;
        slope_x = (float)((double)r * tm_cos(phi));
        slope_y = (float)((double)r * tm_sin(phi));
        return;
    }
    const float sin_t = sqrt_(jmax(0.0f, 1.0f - cos_t * cos_t));
    const float tan_t = sin_t / cos_t;
    float a = 1.0f / tan_t;
    const float g1 = 2.0f / (1.0f + sqrt_(1.0f + 1.0f / (a * a)));
    a = 2.0f * u1 / g1 - 1.0f;
    float tmp = 1.0f / (a * a - 1.0f);
    if (tmp > 1e10f) tmp = 1e10f;
    const float b = tan_t;
    const float b2 = b * b;
    const float d = sqrt_(jmax(0.0f, b2 * (tmp * tmp) - (a * a - b2) * tmp));
    const float sx1 = b * tmp - d, sx2 = b * tmp + d;
    slope_x = (a < 0 || sx2 > 1.0f / tan_t) ? sx1 : sx2;
    float s;
    if (u2 > 0.5f) {
        s = 1.0f;
        u2 = 2.0f * (u2 - 0.5f);
    } else {
        s = -1.0f;
        u2 = 2.0f * (0.5f - u2);
    }
    const float z = (u2 * (u2 * (u2 * 0.27385f - 0.73369f) + 0.46341f)) / (u2 * (u2 * (u2 * 0.093073f + 0.309420f) - 1.0f) + 0.597999f);
    slope_y = s * z * sqrt_(1.0f + slope_x * slope_x);
}
TH_D f3 tr_sample_wh(float ax, float ay, f3 wo, f2 u) {  // :157-184
    const bool flip = wo.z < 0.0f;
    const f3 wi = flip ? -wo : wo;
    const f3 ws = normalize(mk3(wi.x * ax, wi.y * ay, wi.z));
    float sx, sy;
    tr_sample_11(cos_theta(ws), u.x, u.y, sx, sy);
    const float c = cos_phi(ws), s = sin_phi(ws);
    const float tmp = c * sx - s * sy;
    sy = s * sx + c * sy;
    sx = tmp;
    sx *= ax;
    sy *= ay;
    const f3 wh = normalize(mk3(-sx, -sy, 1.0f));
    return flip ? -wh : wh;
}
TH_D bool vec_isapprox_zero(f3 w) {  // isapprox(wh, Vec3f(0)) microfacet.jl:229
    const float d = norm(w);
    if (!isnan_(d) && !isinf_(d)) return d <= 0.00034526698f * jmax(d, 0.0f);
    return w.x == 0.0f && w.y == 0.0f && w.z == 0.0f;
}

// f(wo, wi) in the local frame
TH_D f3 lobe_f(const Lobe& l, f3 wo, f3 wi) {
    switch (l.kind) {
    case LOBE_LAMBERT_R:
    case LOBE_LAMBERT_T: return lobe_r(l) * kInvPi;  // lambertian.jl:22-24, 58-60
    case LOBE_OREN_NAYAR: {                           // microfacet.jl:22-42
        const float sin_i = sin_theta(wi), sin_o = sin_theta(wo);
        float max_cos = 0.0f;
        if (sin_i > 1e-4f && sin_o > 1e-4f) {
            const float spi = sin_phi(wi), cpi = cos_phi(wi), spo = sin_phi(wo), cpo = cos_phi(wo);
            max_cos = jmax(0.0f, cpi * cpo + spi * spo);
        }
        float sin_a, tan_b;
        if (cos_theta(wi) > fabs_(cos_theta(wo))) {
            sin_a = sin_o;
            tan_b = sin_i / fabs_(cos_theta(wi));
        } else {
            sin_a = sin_i;
            tan_b = sin_o / fabs_(cos_theta(wo));
        }
        return lobe_r(l) * kInvPi * (l.a + l.b * max_cos * sin_a * tan_b);
    }
    case LOBE_MICROFACET_R: {  // microfacet.jl:221-234
        const float cos_o = fabs_(cos_theta(wo)), cos_i = fabs_(cos_theta(wi));
        f3 wh = wi + wo;
        if (cos_i == 0.0f || cos_o == 0.0f) return splat3(0.0f);
        if (vec_isapprox_zero(wh)) return splat3(0.0f);
        wh = normalize(wh);
        const f3 f = lobe_fresnel(l, dot(wi, face_forward(wh, mk3(0.0f, 0.0f, 1.0f))));
        return lobe_r(l) * tr_D(l.a, l.b, wh) * tr_G(l.a, l.b, wo, wi) * f / (4.0f * cos_i * cos_o);
    }
    case LOBE_MICROFACET_T: {  // microfacet.jl:281-304
        if (same_hemisphere(wo, wi)) return splat3(0.0f);
        const float cos_o = cos_theta(wo), cos_i = cos_theta(wi);
        if (cos_o == 0.0f || cos_i == 0.0f) return splat3(0.0f);
        const float eta = cos_theta(wo) > 0.0f ? (l.eta_b / l.eta_a) : (l.eta_a / l.eta_b);
        f3 wh = normalize(wo + wi * eta);
        if (wh.z < 0.0f) wh = -wh;
        const float d_o = dot(wo, wh), d_i = dot(wi, wh);
        if (d_o * d_i > 0.0f) return splat3(0.0f);
        const f3 f = lobe_fresnel(l, d_o);
        const float denom = d_o + eta * d_i;
        const float factor = 1.0f;  // `T isa Radiance` is always false (A.11)
        const float dd = tr_D(l.a, l.b, wh), dg = tr_G(l.a, l.b, wo, wi);
        return (splat3(1.0f) - f) * lobe_r(l) * fabs_(dd * dg * d_o * d_i * (eta * eta) * (factor * factor) / (cos_i * cos_o * (denom * denom)));
    }
    default: return splat3(0.0f);  // specular lobes: specular.jl:23-27, 73-77, 132-136
    }
}
// compute_pdf(bxdf, wo, wi)
TH_D float lobe_pdf(const Lobe& l, f3 wo, f3 wi) {
    switch (l.kind) {
    case LOBE_LAMBERT_T: return !same_hemisphere(wo, wi) ? fabs_(cos_theta(wi)) * kInvPi : 0.0f;  // lambertian.jl:83-87
    case LOBE_FRESNEL_SPECULAR: return 0.0f;                                                      // specular.jl:138
    case LOBE_MICROFACET_R: {                                                                     // microfacet.jl:252-258
        if (!same_hemisphere(wo, wi)) return 0.0f;
        const f3 wh = normalize(wo + wi);
        return tr_pdf(l.a, l.b, wo, wh) / dot(4.0f * wo, wh);
    }
    case LOBE_MICROFACET_T: {  // microfacet.jl:322-337
        if (same_hemisphere(wo, wi)) return 0.0f;
        const float eta = cos_theta(wo) > 0.0f ? (l.eta_b / l.eta_a) : (l.eta_a / l.eta_b);
        const f3 wh = normalize(wo + wi * eta);
        const float d_o = dot(wo, wh), d_i = dot(wi, wh);
        if (d_o * d_i > 0.0f) return 0.0f;
        const float denom = d_o + eta * d_i;
        const float dwh_dwi = fabs_(d_i * (eta * eta) / (denom * denom));
        return tr_pdf(l.a, l.b, wo, wh) * dwh_dwi;
    }
    default: return same_hemisphere(wo, wi) ? fabs_(cos_theta(wi)) * kInvPi : 0.0f;  // bxdf.jl:23-25
    }
}
struct LobeSample {
    f3 wi;
    float pdf;
    f3 f;
    int sampled_type;  // -1 = nothing
};
// sample_f(bxdf, wo, u)
TH_D LobeSample lobe_sample_f(const Lobe& l, f3 wo, f2 u) {
    LobeSample s;
    s.wi = splat3(0.0f);
    s.pdf = 0.0f;
    s.f = splat3(0.0f);
    s.sampled_type = -1;
    switch (l.kind) {
    case LOBE_SPECULAR_R: {  // specular.jl:34-39
        s.wi = mk3(-wo.x, -wo.y, wo.z);
        s.pdf = 1.0f;
        s.f = lobe_fresnel(l, cos_theta(s.wi)) * lobe_r(l) / fabs_(cos_theta(s.wi));
        return s;
    }
    case LOBE_SPECULAR_T: {  // specular.jl:84-104
        const bool entering = cos_theta(wo) > 0.0f;
        const float eta_i = entering ? l.eta_a : l.eta_b;
        const float eta_t = entering ? l.eta_b : l.eta_a;
        f3 wi;
        if (!refract(wo, face_forward(mk3(0.0f, 0.0f, 1.0f), wo), eta_i / eta_t, wi)) return s;
        s.wi = wi;
        s.pdf = 1.0f;
        const float cos_wi = cos_theta(wi);
        const f3 ft = lobe_r(l) * (splat3(1.0f) - lobe_fresnel(l, cos_wi));
        s.f = ft / fabs_(cos_wi);
        return s;
    }
    case LOBE_FRESNEL_SPECULAR: {  // specular.jl:143-173
        const float fd = fresnel_dielectric(cos_theta(wo), l.eta_a, l.eta_b);
        if (u.x < fd) {
            s.wi = mk3(-wo.x, -wo.y, wo.z);
            s.sampled_type = BSDF_SPECULAR | BSDF_REFLECTION;
            s.pdf = fd;
            s.f = fd * lobe_r(l) / fabs_(cos_theta(s.wi));
            return s;
        }
        float eta_i, eta_t;
        if (cos_theta(wo) > 0.0f) {
            eta_i = l.eta_a;
            eta_t = l.eta_b;
        } else {
            eta_i = l.eta_b;
            eta_t = l.eta_a;
        }
        f3 wi;
        if (!refract(wo, face_forward(mk3(0.0f, 0.0f, 1.0f), wo), eta_i / eta_t, wi)) {
            s.wi = wi;
            s.pdf = fd;  // pdf = fd, f = 0 on total internal reflection (A.11)
            return s;
        }
        s.wi = wi;
        s.pdf = 1.0f - fd;
        const f3 ft = lobe_t(l) * s.pdf;
        s.sampled_type = BSDF_SPECULAR | BSDF_TRANSMISSION;
        s.f = ft / fabs_(cos_theta(wi));
        return s;
    }
    case LOBE_MICROFACET_R: {  // microfacet.jl:236-250
        if (wo.z == 0.0f) return s;
        const f3 wh = tr_sample_wh(l.a, l.b, wo, u);
        if (dot(wo, wh) < 0.0f) return s;
        const f3 wi = reflect(wo, wh);
        if (!same_hemisphere(wo, wi)) return s;
        s.wi = wi;
        s.pdf = lobe_pdf(l, wo, wh);  // wh passed where wi is expected (A.11)
        s.f = lobe_f(l, wo, wi);
        return s;
    }
    case LOBE_MICROFACET_T: {  // microfacet.jl:306-320
        if (wo.z == 0.0f) return s;
        const f3 wh = tr_sample_wh(l.a, l.b, wo, u);
        if (dot(wo, wh) < 0.0f) return s;
        const float eta = cos_theta(wo) > 0.0f ? (l.eta_b / l.eta_a) : (l.eta_a / l.eta_b);
        f3 wi;
        if (!refract(wo, wh, eta, wi)) return s;
        s.wi = wi;
        s.pdf = lobe_pdf(l, wo, wi);
        s.f = lobe_f(l, wo, wi);
        return s;
    }
    case LOBE_LAMBERT_T: {  // lambertian.jl:72-81
        f3 wi = cosine_sample_hemisphere(u);
        if (wo.z > 0.0f) wi = mk3(wi.x, wi.y, -wi.z);
        s.wi = wi;
        s.pdf = lobe_pdf(l, wo, wi);
        s.f = lobe_f(l, wo, wi);
        return s;
    }
    default: {  // bxdf.jl:34-42
        f3 wi = cosine_sample_hemisphere(u);
        if (wo.z < 0.0f) wi = mk3(wi.x, wi.y, -wi.z);
        s.wi = wi;
        s.pdf = lobe_pdf(l, wo, wi);
        s.f = lobe_f(l, wo, wi);
        return s;
    }
    }
}

// ---- BSDF (materials/bsdf.jl) -------------------------------------------------------------------------------------------
TH_D f3 bsdf_to_local(const Shading& s, f3 v) { return mk3(dot(v, s.ss), dot(v, s.ts), dot(v, s.ns)); }  // :68-70
TH_D f3 bsdf_to_world(const Shading& s, f3 v) {                                                           // :72-74
    return mk3(s.ss.x * v.x + s.ts.x * v.y + s.ns.x * v.z, s.ss.y * v.x + s.ts.y * v.y + s.ns.y * v.z, s.ss.z * v.x + s.ts.z * v.y + s.ns.z * v.z);
}
TH_D int bsdf_num_components(const LobeSet& b, int flags) {  // :195-201
    int n = 0;
    for (int i = 0; i < b.n; ++i) n += lobe_matches(b.lobe[i], flags) ? 1 : 0;
    return n;
}
// :79-100
TH_D f3 bsdf_f(const LobeSet& b, const Shading& s, f3 wo_w, f3 wi_w, int flags) {
    const f3 wo = bsdf_to_local(s, wo_w);
    if (wo.z == 0.0f) return splat3(0.0f);
    const f3 wi = bsdf_to_local(s, wi_w);
    const bool refl = (dot(wi_w, s.ng) * dot(wo_w, s.ng)) > 0.0f;
    f3 out = splat3(0.0f);
    for (int i = 0; i < b.n; ++i) {
        const Lobe& l = b.lobe[i];
        if (lobe_matches(l, flags) && ((refl && (l.type & BSDF_REFLECTION) != 0) || (!refl && (l.type & BSDF_TRANSMISSION) != 0))) out = out + lobe_f(l, wo, wi);
    }
    return out;
}
// bsdf_f for MANY wi against ONE (frame, wo) — the SPPM gather evaluates a visible point's BSDF for every photon inside its radius (sppm.jl:374-391).  What
// bsdf_f derives from (frame, wo) alone is evaluated once: wo in the local frame, wo · ng, and Λ(wo) of the microfacet lobes (three square roots and three
// divisions of tr_lambda: a quarter of the plastic floor's BSDF).  Every operation of bsdf_f / lobe_f is kept, in its order — G = 1 / ((1 + Λ(wo)) + Λ(wi)) —
// so the value is bsdf_f's bit for bit (tests/test_gpu_sppm.py compares the gather with the oracle's bsdf evaluation per photon).
struct BsdfWo {
    f3 wo;         // bsdf_to_local(s, wo_w)
    float wo_ng;   // dot(wo_w, s.ng)
    float lam[2];  // tr_lambda(α_x, α_y, wo) of lobe i where it is a microfacet lobe (a LobeSet holds at most two lobes)
};
TH_D BsdfWo bsdf_wo_terms(const LobeSet& b, const Shading& s, f3 wo_w) {
    BsdfWo p;
    p.wo = bsdf_to_local(s, wo_w);
    p.wo_ng = dot(wo_w, s.ng);
    for (int i = 0; i < 2; ++i) {
        const bool mf = i < b.n && (b.lobe[i].kind == LOBE_MICROFACET_R || b.lobe[i].kind == LOBE_MICROFACET_T);
        p.lam[i] = mf ? tr_lambda(b.lobe[i].a, b.lobe[i].b, p.wo) : 0.0f;
    }
    return p;
}
TH_D f3 lobe_f_wo(const Lobe& l, f3 wo, f3 wi, float lam_o) {
    switch (l.kind) {
    case LOBE_MICROFACET_R: {  // microfacet.jl:221-234, as lobe_f
        const float cos_o = fabs_(cos_theta(wo)), cos_i = fabs_(cos_theta(wi));
        f3 wh = wi + wo;
        if (cos_i == 0.0f || cos_o == 0.0f) return splat3(0.0f);
        if (vec_isapprox_zero(wh)) return splat3(0.0f);
        wh = normalize(wh);
        const f3 f = lobe_fresnel(l, dot(wi, face_forward(wh, mk3(0.0f, 0.0f, 1.0f))));
        const float g = 1.0f / (1.0f + lam_o + tr_lambda(l.a, l.b, wi));  // tr_G
        return lobe_r(l) * tr_D(l.a, l.b, wh) * g * f / (4.0f * cos_i * cos_o);
    }
    case LOBE_MICROFACET_T: {  // microfacet.jl:281-304, as lobe_f
        if (same_hemisphere(wo, wi)) return splat3(0.0f);
        const float cos_o = cos_theta(wo), cos_i = cos_theta(wi);
        if (cos_o == 0.0f || cos_i == 0.0f) return splat3(0.0f);
        const float eta = cos_theta(wo) > 0.0f ? (l.eta_b / l.eta_a) : (l.eta_a / l.eta_b);
        f3 wh = normalize(wo + wi * eta);
        if (wh.z < 0.0f) wh = -wh;
        const float d_o = dot(wo, wh), d_i = dot(wi, wh);
        if (d_o * d_i > 0.0f) return splat3(0.0f);
        const f3 f = lobe_fresnel(l, d_o);
        const float denom = d_o + eta * d_i;
        const float factor = 1.0f;
        const float dd = tr_D(l.a, l.b, wh), dg = 1.0f / (1.0f + lam_o + tr_lambda(l.a, l.b, wi));
        return (splat3(1.0f) - f) * lobe_r(l) * fabs_(dd * dg * d_o * d_i * (eta * eta) * (factor * factor) / (cos_i * cos_o * (denom * denom)));
    }
    default: return lobe_f(l, wo, wi);
    }
}
TH_D f3 bsdf_f_wo(const LobeSet& b, const Shading& s, const BsdfWo& pre, f3 wi_w, int flags) {
    if (pre.wo.z == 0.0f) return splat3(0.0f);
    const f3 wi = bsdf_to_local(s, wi_w);
    const bool refl = (dot(wi_w, s.ng) * pre.wo_ng) > 0.0f;
    f3 out = splat3(0.0f);
    for (int i = 0; i < b.n; ++i) {
        const Lobe& l = b.lobe[i];
        if (lobe_matches(l, flags) && ((refl && (l.type & BSDF_REFLECTION) != 0) || (!refl && (l.type & BSDF_TRANSMISSION) != 0))) out = out + lobe_f_wo(l, pre.wo, wi, pre.lam[i < 2 ? i : 1]);
    }
    return out;
}
// :177-193
TH_D float bsdf_pdf(const LobeSet& b, const Shading& s, f3 wo_w, f3 wi_w, int flags) {
    if (b.n == 0) return 0.0f;
    const f3 wo = bsdf_to_local(s, wo_w);
    if (wo.z == 0.0f) return 0.0f;
    const f3 wi = bsdf_to_local(s, wi_w);
    float p = 0.0f;
    int matching = 0;
    for (int i = 0; i < b.n; ++i)
        if (lobe_matches(b.lobe[i], flags)) {
            matching++;
            p += lobe_pdf(b.lobe[i], wo, wi);
        }
    return matching > 0 ? p / (float)matching : 0.0f;
}
struct BsdfSample {
    f3 wi;
    f3 f;
    float pdf;
    int sampled_type;
};
// :107-175
TH_D BsdfSample bsdf_sample_f(const LobeSet& b, const Shading& s, f3 wo_w, f2 u, int type) {
    BsdfSample none;
    none.wi = splat3(0.0f);
    none.f = splat3(0.0f);
    none.pdf = 0.0f;
    none.sampled_type = BSDF_NONE;
    const int matching = bsdf_num_components(b, type);
    if (matching == 0) return none;
    int component = (int)__builtin_ceilf(u.x * (float)matching);
    if (component < 1) component = 1;
    if (component > matching) component = matching;
    int count = component;
    component -= 1;
    int chosen = 0;
    for (int i = 0; i < b.n; ++i)
        if (lobe_matches(b.lobe[i], type)) {
            if (count == 1) {
                chosen = i;
                break;
            }
            count -= 1;
        }
    const Lobe& l = b.lobe[chosen];
    const f2 ur{jmin(u.x * (float)matching - (float)component, 1.0f), u.y};
    const f3 wo = bsdf_to_local(s, wo_w);
    if (wo.z == 0.0f) return none;
    int sampled_type = l.type;
    const LobeSample ls = lobe_sample_f(l, wo, ur);
    float pdf = ls.pdf;
    f3 f = ls.f;
    if (ls.sampled_type >= 0) sampled_type = ls.sampled_type;
    if (pdf == 0.0f) return none;
    const f3 wi_w = bsdf_to_world(s, ls.wi);
    const bool specular = (l.type & BSDF_SPECULAR) != 0;
    if (!specular && matching > 1)
        for (int i = 0; i < b.n; ++i)
            if (i != chosen && lobe_matches(b.lobe[i], type)) pdf += lobe_pdf(b.lobe[i], wo, ls.wi);
    if (matching > 1) pdf /= (float)matching;
    if (!specular) {
        const bool refl = (dot(wi_w, s.ng) * dot(wo_w, s.ng)) > 0.0f;
        f = splat3(0.0f);
        for (int i = 0; i < b.n; ++i) {
            const Lobe& x = b.lobe[i];
            if (lobe_matches(x, type) && ((refl && (x.type & BSDF_REFLECTION) != 0) || (!refl && (x.type & BSDF_TRANSMISSION) != 0))) f = f + lobe_f(x, wo, ls.wi);
        }
    }
    BsdfSample r;
    r.wi = wi_w;
    r.f = f;
    r.pdf = pdf;
    r.sampled_type = sampled_type;
    return r;
}

// ---- the commonest BSDF, specialised ---------------------------------------------------------------------------------------
// A BSDF that is exactly one LambertianReflection lobe (MatteMaterial with σ = 0, material.jl:16-31; type DIFFUSE|REFLECTION,
// which every flag set the integrators pass matches).  bsdf_f / bsdf_sample_f evaluated for that case: the same operations
// in the same order, without the lobe loops and kind switches the general code needs (they cost registers and ~40 % of the
// shading kernel's instructions).
TH_D bool bsdf_is_single_lambert(const LobeSet& b) { return b.n == 1 && b.lobe[0].kind == LOBE_LAMBERT_R; }
TH_D f3 lambert_bsdf_f(const Lobe& l, const Shading& s, f3 wo_w, f3 wi_w) {
    const f3 wo = bsdf_to_local(s, wo_w);
    if (wo.z == 0.0f) return splat3(0.0f);
    const bool refl = (dot(wi_w, s.ng) * dot(wo_w, s.ng)) > 0.0f;
    f3 out = splat3(0.0f);
    if (refl) out = out + lobe_r(l) * kInvPi;
    return out;
}
TH_D BsdfSample lambert_bsdf_sample_f(const Lobe& l, const Shading& s, f3 wo_w, f2 u) {
    BsdfSample r;
    r.wi = splat3(0.0f);
    r.f = splat3(0.0f);
    r.pdf = 0.0f;
    r.sampled_type = BSDF_NONE;
    const f2 ur{jmin(u.x * 1.0f - 0.0f, 1.0f), u.y};  // one matching lobe: component 1, remapped u (bsdf.jl:118-131)
    const f3 wo = bsdf_to_local(s, wo_w);
    if (wo.z == 0.0f) return r;
    f3 wi = cosine_sample_hemisphere(ur);  // bxdf.jl:34-42
    if (wo.z < 0.0f) wi = mk3(wi.x, wi.y, -wi.z);
    const float pdf = same_hemisphere(wo, wi) ? fabs_(cos_theta(wi)) * kInvPi : 0.0f;  // bxdf.jl:23-25
    if (pdf == 0.0f) return r;
    const f3 wi_w = bsdf_to_world(s, wi);
    const bool refl = (dot(wi_w, s.ng) * dot(wo_w, s.ng)) > 0.0f;
    f3 f = splat3(0.0f);
    if (refl) f = f + lobe_r(l) * kInvPi;
    r.wi = wi_w;
    r.f = f;
    r.pdf = pdf;
    r.sampled_type = l.type;
    return r;
}

// ---- lights (lights/point.jl:50-58, lights/spot.jl:22-40) -----------------------------------------------------------------
struct LightSample {
    f3 radiance;
    f3 wi;
    float pdf;
};
TH_D LightSample sample_li(const LightRec& l, f3 p) {
    LightSample s;
    const f3 lp = mk3(l.position[0], l.position[1], l.position[2]);
    s.wi = normalize(lp - p);
    s.pdf = 1.0f;
    const f3 I = mk3(l.I[0], l.I[1], l.I[2]);
    const f3 dv = lp - p;  // distance_squared(position, ref.p)  bounds.jl:132-135
    const float d2 = dot(dv, dv);
    if (l.kind == 0) {
        s.radiance = I / d2;
    } else {
        const f3 w = -s.wi;
        const f3 wl = normalize(mk3(l.w2l[0] * w.x + l.w2l[1] * w.y + l.w2l[2] * w.z, l.w2l[3] * w.x + l.w2l[4] * w.y + l.w2l[5] * w.z,
                                    l.w2l[6] * w.x + l.w2l[7] * w.y + l.w2l[8] * w.z));
        const float c = wl.z;
        float fall;
        if (c < l.cos_total_width)
            fall = 0.0f;
        else if (c >= l.cos_falloff_start)
            fall = 1.0f;
        else
            fall = pow4((c - l.cos_total_width) / (l.cos_falloff_start - l.cos_total_width));
        s.radiance = I * fall / d2;
    }
    return s;
}

}  // namespace th
