// oracle/orc_scatter.h — TEST INFRASTRUCTURE ONLY.
// Restates reflection/{bxdf,lambertian,specular,microfacet}.jl, materials/{bsdf,material}.jl, textures/basic.jl:1-10
// (ConstantTexture folds to a constant), lights/{light,point,spot}.jl.  Quirk ledger: SURVEY.md A.11.
#pragma once
#include "orc_shapes.h"

namespace orc {

// bxdf.jl:1-7
constexpr uint8_t BSDF_NONE = 0, BSDF_REFLECTION = 1, BSDF_TRANSMISSION = 2, BSDF_DIFFUSE = 4, BSDF_GLOSSY = 8, BSDF_SPECULAR = 16, BSDF_ALL = 31;

inline bool same_hemisphere(V3 w, V3 wp) { return w.z * wp.z > 0; }  // bxdf.jl:13-15
const float INV_PI = 1.0f / PI_F;                                   // `1f0 / π`

// bxdf.jl:52-62
inline bool refract(V3 wi, V3 n, float eta, V3& wt) {
    const float cos_i = dot(n, wi);
    const float sin2_i = jl_max(0.0f, 1.0f - cos_i * cos_i);
    const float sin2_t = (eta * eta) * sin2_i;
    if (sin2_t >= 1) {
        wt = V3(0.0f);
        return false;
    }
    const float cos_t = std::sqrt(1.0f - sin2_t);
    wt = (-eta) * wi + (eta * cos_i - cos_t) * n;
    return true;
}
// bxdf.jl:74-95
inline float fresnel_dielectric(float cos_i, float eta_i, float eta_t) {
    cos_i = jl_clamp(cos_i, -1.0f, 1.0f);
    if (cos_i <= 0.0f) {
        std::swap(eta_i, eta_t);
        cos_i = std::fabs(cos_i);
    }
    const float sin_i = std::sqrt(jl_max(0.0f, 1.0f - cos_i * cos_i));
    const float sin_t = sin_i * eta_i / eta_t;
    if (sin_t >= 1.0f) return 1.0f;
    const float cos_t = std::sqrt(jl_max(0.0f, 1.0f - sin_t * sin_t));
    const float r_par = (eta_t * cos_i - eta_i * cos_t) / (eta_t * cos_i + eta_i * cos_t);
    const float r_perp = (eta_i * cos_i - eta_t * cos_t) / (eta_i * cos_i + eta_t * cos_t);
    return 0.5f * (r_par * r_par + r_perp * r_perp);
}
// bxdf.jl:102-125, one spectrum channel
inline float fresnel_conductor_1(float cos_i, float eta_i, float eta_t, float k) {
    cos_i = jl_clamp(cos_i, -1.0f, 1.0f);
    const float eta = eta_t / eta_i;
    const float etak = k / eta_i;
    const float cos2 = cos_i * cos_i;
    const float sin2 = 1.0f - cos2;
    const float eta2 = eta * eta;
    const float etak2 = etak * etak;
    const float t0 = eta2 - etak2 - sin2;
    const float a2pb2 = std::sqrt(t0 * t0 + 4.0f * eta2 * etak2);
    const float t1 = a2pb2 + cos2;
    const float a = std::sqrt(0.5f * (a2pb2 + t0));
    const float t2 = 2.0f * cos_i * a;
    const float r_perp = (t1 - t2) / (t1 + t2);
    const float t3 = cos2 * a2pb2 + sin2 * sin2;
    const float t4 = t2 * sin2;
    const float r_par = r_perp * (t3 - t4) / (t3 + t4);
    return 0.5f * (r_par + r_perp);
}
inline RGB fresnel_conductor(float cos_i, RGB eta_i, RGB eta_t, RGB k) {
    return {fresnel_conductor_1(cos_i, eta_i.x, eta_t.x, k.x), fresnel_conductor_1(cos_i, eta_i.y, eta_t.y, k.y), fresnel_conductor_1(cos_i, eta_i.z, eta_t.z, k.z)};
}
struct Fresnel {  // bxdf.jl:127-140
    enum Kind { NOOP, DIELECTRIC, CONDUCTOR } kind = NOOP;
    float eta_i = 1, eta_t = 1;
    RGB c_eta_i, c_eta_t, c_k;
    RGB eval(float cos_i) const {
        if (kind == NOOP) return RGB(1.0f);
        if (kind == DIELECTRIC) return RGB(fresnel_dielectric(cos_i, eta_i, eta_t));
        return fresnel_conductor(cos_i, c_eta_i, c_eta_t, c_k);
    }
};

// microfacet.jl:53-66
struct TrowbridgeReitz {
    float alpha_x = 1, alpha_y = 1;
    bool sample_visible_area = true;
    TrowbridgeReitz() = default;
    TrowbridgeReitz(float ax, float ay, bool vis = true) : alpha_x(jl_max(1e-3f, ax)), alpha_y(jl_max(1e-3f, ay)), sample_visible_area(vis) {}
};
// microfacet.jl:68-75
inline float tr_lambda(const TrowbridgeReitz& d, V3 w) {
    const float th = std::fabs(tan_theta(w));
    if (std::isinf(th)) return 0.0f;
    const float cp = cos_phi(w), sp = sin_phi(w);
    const float alpha = std::sqrt(cp * cp * (d.alpha_x * d.alpha_x) + sp * sp * (d.alpha_y * d.alpha_y));
    const float at = alpha * th;
    const float a2t2 = at * at;
    return (-1.0f + std::sqrt(1.0f + a2t2)) / 2.0f;
}
// microfacet.jl:82-87
inline float roughness_to_alpha(float roughness) {
    roughness = jl_max(1e-3f, roughness);
    const float x = tm_logf(roughness);
    return 1.62142f + 0.819955f * x + 0.1734f * (x * x) + 0.0171201f * (x * x * x) + 0.000640711f * jl_pow4(x);
}
inline float tr_G1(const TrowbridgeReitz& d, V3 w) { return 1.0f / (1.0f + tr_lambda(d, w)); }                          // :89-91
inline float tr_G(const TrowbridgeReitz& d, V3 wo, V3 wi) { return 1.0f / (1.0f + tr_lambda(d, wo) + tr_lambda(d, wi)); }  // :93-95
// microfacet.jl:101-108
inline float tr_D(const TrowbridgeReitz& d, V3 w) {
    const float tt = tan_theta(w);
    const float tan2 = tt * tt;
    if (std::isinf(tan2)) return 0.0f;
    const float cos4 = jl_pow4(cos_theta(w));
    const float cp = cos_phi(w), sp = sin_phi(w);
    const float e = (cp * cp / (d.alpha_x * d.alpha_x) + sp * sp / (d.alpha_y * d.alpha_y)) * tan2;
    const float ope = 1.0f + e;
    return 1.0f / (PI_F * d.alpha_x * d.alpha_y * cos4 * (ope * ope));
}
// microfacet.jl:110-113
inline float tr_pdf(const TrowbridgeReitz& d, V3 wo, V3 wh) {
    if (!d.sample_visible_area) return tr_D(d, wh) * std::fabs(cos_theta(wh));
    return tr_D(d, wh) * tr_G1(d, wo) * std::fabs(dot(wo, wh)) / std::fabs(cos_theta(wo));
}
// microfacet.jl:115-155
inline void tr_sample_11(float cos_t, float u1, float u2, float& slope_x, float& slope_y) {
    if (cos_t > 0.9999f) {
        const float r = std::sqrt(u1 / (1.0f - u1));
        const double phi = 6.28318530718 * (double)u2;  // the Float64 literal of :117
        slope_x = (float)((double)r * tm_cos(phi));
        slope_y = (float)((double)r * tm_sin(phi));
        return;
    }
    const float sin_t = std::sqrt(jl_max(0.0f, 1.0f - cos_t * cos_t));
    const float tan_t = sin_t / cos_t;
    float a = 1.0f / tan_t;
    const float g1 = 2.0f / (1.0f + std::sqrt(1.0f + 1.0f / (a * a)));
    a = 2.0f * u1 / g1 - 1.0f;
    float tmp = 1.0f / (a * a - 1.0f);
    if (tmp > 1e10f) tmp = 1e10f;
    const float b = tan_t;
    const float b2 = b * b;
    const float d = std::sqrt(jl_max(0.0f, b2 * (tmp * tmp) - (a * a - b2) * tmp));
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
    slope_y = s * z * std::sqrt(1.0f + slope_x * slope_x);
}
// microfacet.jl:157-173
inline V3 tr_sample(V3 wi, float ax, float ay, float u1, float u2) {
    const V3 ws = normalize(V3(wi.x * ax, wi.y * ay, wi.z));
    float sx, sy;
    tr_sample_11(cos_theta(ws), u1, u2, sx, sy);
    const float c = cos_phi(ws), s = sin_phi(ws);
    const float tmp = c * sx - s * sy;
    sy = s * sx + c * sy;
    sx = tmp;
    sx *= ax;
    sy *= ay;
    return normalize(V3(-sx, -sy, 1.0f));
}
// microfacet.jl:175-201
inline V3 tr_sample_wh(const TrowbridgeReitz& d, V3 wo, V2 u) {
    if (d.sample_visible_area) {
        const bool flip = wo.z < 0.0f;
        const V3 wh = tr_sample(flip ? -wo : wo, d.alpha_x, d.alpha_y, u.x, u.y);
        return flip ? -wh : wh;
    }
    float cos_t = 0.0f;
    float phi = 2.0f * PI_F * u.y;
    if (jl_isapprox(d.alpha_x, d.alpha_y)) {
        const float tan2 = d.alpha_x * d.alpha_x * u.x / (1.0f - u.x);
        cos_t = 1.0f / std::sqrt(1.0f + tan2);
    } else {
        phi = tm_atanf(d.alpha_y / d.alpha_x * tm_tanf(2.0f * PI_F * u.y + 0.5f * PI_F));
        if (u.y > 0.5f) phi += PI_F;
        const float sp = tm_sinf(phi), cp = tm_cosf(phi);
        const float ax2 = d.alpha_x * d.alpha_x, ay2 = d.alpha_y * d.alpha_y;
        const float a2 = 1.0f / (cp * cp / ax2 + sp * sp / ay2);
        const float tan2 = a2 * u.x / (1.0f - u.x);
        cos_t = 1.0f / std::sqrt(1.0f + tan2);
    }
    const float sin_t = std::sqrt(jl_max(0.0f, 1.0f - cos_t * cos_t));
    const V3 wh = spherical_direction(sin_t, cos_t, phi);
    return same_hemisphere(wo, wh) ? wh : -wh;
}

// ---- BxDF: one tagged struct instead of Julia's dispatch -----------------------------------------------------------
struct BxDF {
    enum Kind { LAMBERTIAN_R, LAMBERTIAN_T, OREN_NAYAR, SPECULAR_R, SPECULAR_T, FRESNEL_SPECULAR, MICROFACET_R, MICROFACET_T } kind = LAMBERTIAN_R;
    uint8_t type = 0;
    RGB r, t;                  // reflectance / transmittance
    float a = 0, b = 0;        // OrenNayar
    Fresnel fresnel;           // SpecularReflection, Microfacet*, SpecularTransmission
    float eta_a = 1, eta_b = 1;
    TrowbridgeReitz dist;
    bool matches(uint8_t flags) const { return (type & flags) == type; }  // bxdf.jl:9-11
};
inline BxDF LambertianReflection(RGB r) {  // lambertian.jl:5-16
    BxDF b;
    b.kind = BxDF::LAMBERTIAN_R;
    b.r = r;
    b.type = BSDF_DIFFUSE | BSDF_REFLECTION;
    return b;
}
inline BxDF LambertianTransmission(RGB t) {  // lambertian.jl:48-56
    BxDF b;
    b.kind = BxDF::LAMBERTIAN_T;
    b.t = t;
    b.type = BSDF_DIFFUSE | BSDF_TRANSMISSION;
    return b;
}
inline BxDF OrenNayar(RGB r, float sigma_deg) {  // microfacet.jl:6-20
    BxDF b;
    b.kind = BxDF::OREN_NAYAR;
    b.r = r;
    const float s = jl_deg2rad(sigma_deg);
    const float s2 = s * s;
    b.a = 1.0f - (s2 / (2.0f * (s2 + 0.33f)));
    b.b = 0.45f * s2 / (s2 + 0.09f);
    b.type = BSDF_DIFFUSE | BSDF_REFLECTION;
    return b;
}
inline BxDF SpecularReflection(RGB r, Fresnel f) {  // specular.jl:1-16
    BxDF b;
    b.kind = BxDF::SPECULAR_R;
    b.r = r;
    b.fresnel = f;
    b.type = BSDF_SPECULAR | BSDF_REFLECTION;
    return b;
}
inline Fresnel FresnelDielectric(float ei, float et) {
    Fresnel f;
    f.kind = Fresnel::DIELECTRIC;
    f.eta_i = ei;
    f.eta_t = et;
    return f;
}
inline BxDF SpecularTransmission(RGB t, float ea, float eb) {  // specular.jl:41-66
    BxDF b;
    b.kind = BxDF::SPECULAR_T;
    b.t = t;
    b.eta_a = ea;
    b.eta_b = eb;
    b.fresnel = FresnelDielectric(ea, eb);
    b.type = BSDF_SPECULAR | BSDF_TRANSMISSION;
    return b;
}
inline BxDF FresnelSpecular(RGB r, RGB t, float ea, float eb) {  // specular.jl:107-130
    BxDF b;
    b.kind = BxDF::FRESNEL_SPECULAR;
    b.r = r;
    b.t = t;
    b.eta_a = ea;
    b.eta_b = eb;
    b.type = BSDF_SPECULAR | BSDF_TRANSMISSION | BSDF_REFLECTION;
    return b;
}
inline BxDF MicrofacetReflection(RGB r, TrowbridgeReitz d, Fresnel f) {  // microfacet.jl:204-219
    BxDF b;
    b.kind = BxDF::MICROFACET_R;
    b.r = r;
    b.dist = d;
    b.fresnel = f;
    b.type = BSDF_REFLECTION | BSDF_GLOSSY;
    return b;
}
inline BxDF MicrofacetTransmission(RGB t, TrowbridgeReitz d, float ea, float eb) {  // microfacet.jl:261-279
    BxDF b;
    b.kind = BxDF::MICROFACET_T;
    b.t = t;
    b.dist = d;
    b.eta_a = ea;
    b.eta_b = eb;
    b.fresnel = FresnelDielectric(ea, eb);
    b.type = BSDF_TRANSMISSION | BSDF_GLOSSY;
    return b;
}

// isapprox(wh, Vec3f(0)) for arrays (microfacet.jl:229): norm-based when finite, component-wise otherwise.
inline bool vec_isapprox_zero(V3 w) {
    const float d = norm(w);
    if (std::isfinite(d)) return d <= 0.00034526698f * jl_max(d, 0.0f);
    return w.x == 0 && w.y == 0 && w.z == 0;
}

// f(wo, wi) of each BxDF
inline RGB bxdf_f(const BxDF& b, V3 wo, V3 wi) {
    switch (b.kind) {
    case BxDF::LAMBERTIAN_R: return b.r * INV_PI;  // lambertian.jl:22-24
    case BxDF::LAMBERTIAN_T: return b.t * INV_PI;  // lambertian.jl:58-60
    case BxDF::OREN_NAYAR: {                       // microfacet.jl:22-42
        const float sin_i = sin_theta(wi), sin_o = sin_theta(wo);
        float max_cos = 0.0f;
        if (sin_i > 1e-4f && sin_o > 1e-4f) {
            const float spi = sin_phi(wi), cpi = cos_phi(wi), spo = sin_phi(wo), cpo = cos_phi(wo);
            max_cos = jl_max(0.0f, cpi * cpo + spi * spo);
        }
        float sin_a, tan_b;
        if (cos_theta(wi) > std::fabs(cos_theta(wo))) {  // `abs(cos_θ(wi) > abs(cos_θ(wo)))`: abs of a Bool (A.11)
            sin_a = sin_o;
            tan_b = sin_i / std::fabs(cos_theta(wi));
        } else {
            sin_a = sin_i;
            tan_b = sin_o / std::fabs(cos_theta(wo));
        }
        return b.r * INV_PI * (b.a + b.b * max_cos * sin_a * tan_b);
    }
    case BxDF::SPECULAR_R:
    case BxDF::SPECULAR_T:
    case BxDF::FRESNEL_SPECULAR: return RGB(0.0f);  // specular.jl:23-27, 73-77, 132-136
    case BxDF::MICROFACET_R: {                      // microfacet.jl:221-234
        const float cos_o = std::fabs(cos_theta(wo)), cos_i = std::fabs(cos_theta(wi));
        V3 wh = wi + wo;
        if (cos_i == 0 || cos_o == 0) return RGB(0.0f);
        if (vec_isapprox_zero(wh)) return RGB(0.0f);
        wh = normalize(wh);
        const RGB f = b.fresnel.eval(dot(wi, face_forward(wh, V3(0, 0, 1))));
        return b.r * tr_D(b.dist, wh) * tr_G(b.dist, wo, wi) * f / (4.0f * cos_i * cos_o);
    }
    case BxDF::MICROFACET_T: {  // microfacet.jl:281-304
        if (same_hemisphere(wo, wi)) return RGB(0.0f);
        const float cos_o = cos_theta(wo), cos_i = cos_theta(wi);
        if (cos_o == 0 || cos_i == 0) return RGB(0.0f);
        const float eta = cos_theta(wo) > 0.0f ? (b.eta_b / b.eta_a) : (b.eta_a / b.eta_b);
        V3 wh = normalize(wo + wi * eta);
        if (wh.z < 0) wh = -wh;
        const float d_o = dot(wo, wh), d_i = dot(wi, wh);
        if (d_o * d_i > 0) return RGB(0.0f);
        const RGB f = b.fresnel.eval(d_o);
        const float denom = d_o + eta * d_i;
        const float factor = 1.0f;  // `T isa Radiance` is always false (A.11)
        const float dd = tr_D(b.dist, wh), dg = tr_G(b.dist, wo, wi);
        return (RGB(1.0f) - f) * b.t * std::fabs(dd * dg * d_o * d_i * (eta * eta) * (factor * factor) / (cos_i * cos_o * (denom * denom)));
    }
    }
    return RGB(0.0f);
}
// compute_pdf(bxdf, wo, wi)
inline float bxdf_pdf(const BxDF& b, V3 wo, V3 wi) {
    switch (b.kind) {
    case BxDF::LAMBERTIAN_T: return !same_hemisphere(wo, wi) ? std::fabs(cos_theta(wi)) * INV_PI : 0.0f;  // lambertian.jl:83-87
    case BxDF::FRESNEL_SPECULAR: return 0.0f;                                                            // specular.jl:138
    case BxDF::MICROFACET_R: {                                                                           // microfacet.jl:252-258
        if (!same_hemisphere(wo, wi)) return 0.0f;
        const V3 wh = normalize(wo + wi);
        return tr_pdf(b.dist, wo, wh) / dot(4.0f * wo, wh);
    }
    case BxDF::MICROFACET_T: {  // microfacet.jl:322-337
        if (same_hemisphere(wo, wi)) return 0.0f;
        const float eta = cos_theta(wo) > 0.0f ? (b.eta_b / b.eta_a) : (b.eta_a / b.eta_b);
        const V3 wh = normalize(wo + wi * eta);
        const float d_o = dot(wo, wh), d_i = dot(wi, wh);
        if (d_o * d_i > 0) return 0.0f;
        const float denom = d_o + eta * d_i;
        const float dwh_dwi = std::fabs(d_i * (eta * eta) / (denom * denom));
        return tr_pdf(b.dist, wo, wh) * dwh_dwi;
    }
    default: return same_hemisphere(wo, wi) ? std::fabs(cos_theta(wi)) * INV_PI : 0.0f;  // bxdf.jl:23-25
    }
}
struct BxDFSample {
    V3 wi;
    float pdf = 0;
    RGB f;
    bool has_type = false;
    uint8_t sampled_type = 0;
};
// sample_f(bxdf, wo, u)
inline BxDFSample bxdf_sample_f(const BxDF& b, V3 wo, V2 u) {
    BxDFSample s;
    switch (b.kind) {
    case BxDF::SPECULAR_R: {  // specular.jl:34-39
        s.wi = V3(-wo.x, -wo.y, wo.z);
        s.pdf = 1.0f;
        s.f = b.fresnel.eval(cos_theta(s.wi)) * b.r / std::fabs(cos_theta(s.wi));
        return s;
    }
    case BxDF::SPECULAR_T: {  // specular.jl:84-104
        const bool entering = cos_theta(wo) > 0;
        const float eta_i = entering ? b.eta_a : b.eta_b;
        const float eta_t = entering ? b.eta_b : b.eta_a;
        V3 wi;
        if (!refract(wo, face_forward(V3(0, 0, 1), wo), eta_i / eta_t, wi)) return s;  // (0, 0, black)
        s.wi = wi;
        s.pdf = 1.0f;
        const float cos_wi = cos_theta(wi);
        const RGB ft = b.t * (RGB(1.0f) - b.fresnel.eval(cos_wi));
        s.f = ft / std::fabs(cos_wi);  // no (η_i/η_t)² factor: `T isa Radiance` is false (A.11)
        return s;
    }
    case BxDF::FRESNEL_SPECULAR: {  // specular.jl:143-173
        const float fd = fresnel_dielectric(cos_theta(wo), b.eta_a, b.eta_b);
        if (u.x < fd) {
            s.wi = V3(-wo.x, -wo.y, wo.z);
            s.has_type = true;
            s.sampled_type = BSDF_SPECULAR | BSDF_REFLECTION;
            s.pdf = fd;
            s.f = fd * b.r / std::fabs(cos_theta(s.wi));
            return s;
        }
        float eta_i, eta_t;
        if (cos_theta(wo) > 0) {
            eta_i = b.eta_a;
            eta_t = b.eta_b;
        } else {
            eta_i = b.eta_b;
            eta_t = b.eta_a;
        }
        V3 wi;
        if (!refract(wo, face_forward(V3(0, 0, 1), wo), eta_i / eta_t, wi)) {
            s.wi = wi;
            s.pdf = fd;  // pdf = fd with f = 0 on total internal reflection (A.11)
            s.f = RGB(0.0f);
            return s;
        }
        s.wi = wi;
        s.pdf = 1.0f - fd;
        const RGB ft = b.t * s.pdf;
        s.has_type = true;
        s.sampled_type = BSDF_SPECULAR | BSDF_TRANSMISSION;
        s.f = ft / std::fabs(cos_theta(wi));
        return s;
    }
    case BxDF::MICROFACET_R: {  // microfacet.jl:236-250
        if (wo.z == 0) return s;
        const V3 wh = tr_sample_wh(b.dist, wo, u);
        if (dot(wo, wh) < 0) return s;
        const V3 wi = reflect(wo, wh);
        if (!same_hemisphere(wo, wi)) return s;
        s.wi = wi;
        s.pdf = bxdf_pdf(b, wo, wh);  // passes wh where wi is expected (A.11)
        s.f = bxdf_f(b, wo, wi);
        return s;
    }
    case BxDF::MICROFACET_T: {  // microfacet.jl:306-320
        if (wo.z == 0) return s;
        const V3 wh = tr_sample_wh(b.dist, wo, u);
        if (dot(wo, wh) < 0) return s;
        const float eta = cos_theta(wo) > 0.0f ? (b.eta_b / b.eta_a) : (b.eta_a / b.eta_b);  // inverse of refract's convention (A.11)
        V3 wi;
        if (!refract(wo, wh, eta, wi)) return s;
        s.wi = wi;
        s.pdf = bxdf_pdf(b, wo, wi);
        s.f = bxdf_f(b, wo, wi);
        return s;
    }
    case BxDF::LAMBERTIAN_T: {  // lambertian.jl:72-81
        V3 wi = cosine_sample_hemisphere(u);
        if (wo.z > 0) wi = V3(wi.x, wi.y, -wi.z);
        s.wi = wi;
        s.pdf = bxdf_pdf(b, wo, wi);
        s.f = bxdf_f(b, wo, wi);
        return s;
    }
    default: {  // bxdf.jl:34-42
        V3 wi = cosine_sample_hemisphere(u);
        if (wo.z < 0) wi = V3(wi.x, wi.y, -wi.z);
        s.wi = wi;
        s.pdf = bxdf_pdf(b, wo, wi);
        s.f = bxdf_f(b, wo, wi);
        return s;
    }
    }
}

// ---- materials/bsdf.jl ----------------------------------------------------------------------------------------------
struct BSDF {
    float eta = 1;
    V3 ng, ns, ss, ts;
    int n_bxdfs = 0;
    BxDF bxdfs[8];  // MAX_BxDF = 8 (bsdf.jl:4)
    bool valid = false;
    BSDF() = default;
    BSDF(const SurfaceInteraction& si, float eta_ = 1.0f) : eta(eta_), valid(true) {  // bsdf.jl:41-50
        ng = si.n;
        ns = si.sh_n;
        ss = normalize(si.sh_dpdu);
        ts = cross(ns, ss);  // not re-normalised (A.11)
    }
    void add(const BxDF& b) { bxdfs[n_bxdfs++] = b; }                                    // bsdf.jl:53-57
    V3 world_to_local(V3 v) const { return {dot(v, ss), dot(v, ts), dot(v, ns)}; }        // bsdf.jl:68-70
    V3 local_to_world(V3 v) const {                                                       // bsdf.jl:72-74: Mat3f0(ss..., ts..., ns...) * v
        return {ss.x * v.x + ts.x * v.y + ns.x * v.z, ss.y * v.x + ts.y * v.y + ns.y * v.z, ss.z * v.x + ts.z * v.y + ns.z * v.z};
    }
    int num_components(uint8_t flags) const {  // bsdf.jl:195-201
        int n = 0;
        for (int i = 0; i < n_bxdfs; ++i)
            if (bxdfs[i].matches(flags)) n++;
        return n;
    }
    // bsdf.jl:79-100
    RGB f(V3 wo_world, V3 wi_world, uint8_t flags = BSDF_ALL) const {
        const V3 wo = world_to_local(wo_world);
        if (wo.z == 0.0f) return RGB(0.0f);
        const V3 wi = world_to_local(wi_world);
        const bool reflect_ = (dot(wi_world, ng) * dot(wo_world, ng)) > 0;
        RGB out(0.0f);
        for (int i = 0; i < n_bxdfs; ++i) {
            const BxDF& b = bxdfs[i];
            if (b.matches(flags) && ((reflect_ && (b.type & BSDF_REFLECTION) != 0) || (!reflect_ && (b.type & BSDF_TRANSMISSION) != 0)))
                out = out + bxdf_f(b, wo, wi);
        }
        return out;
    }
    // bsdf.jl:177-193
    float pdf(V3 wo_world, V3 wi_world, uint8_t flags) const {
        if (n_bxdfs == 0) return 0.0f;
        const V3 wo = world_to_local(wo_world);
        if (wo.z == 0.0f) return 0.0f;
        const V3 wi = world_to_local(wi_world);
        float p = 0.0f;
        int matching = 0;
        for (int i = 0; i < n_bxdfs; ++i)
            if (bxdfs[i].matches(flags)) {
                matching++;
                p += bxdf_pdf(bxdfs[i], wo, wi);
            }
        return matching > 0 ? p / (float)matching : 0.0f;
    }
};
struct BSDFSample {
    V3 wi;
    RGB f;
    float pdf = 0;
    uint8_t sampled_type = BSDF_NONE;
};
// bsdf.jl:107-175
inline BSDFSample bsdf_sample_f(const BSDF& b, V3 wo_world, V2 u, uint8_t type) {
    BSDFSample none;
    const int matching = b.num_components(type);
    if (matching == 0) return none;
    // Int64(ceil(u[1] * matching)) — throws on NaN in the reference (A.16i); u is never NaN here.
    long long component = (long long)std::ceil(u.x * (float)matching);
    if (component < 1) component = 1;
    if (component > matching) component = matching;
    long long count = component;
    component -= 1;
    int chosen = -1;
    for (int i = 0; i < b.n_bxdfs; ++i)
        if (b.bxdfs[i].matches(type)) {
            if (count == 1) {
                chosen = i;
                break;
            }
            count -= 1;
        }
    const BxDF& bxdf = b.bxdfs[chosen];
    const V2 u_remapped{jl_min(u.x * (float)matching - (float)component, 1.0f), u.y};
    const V3 wo = b.world_to_local(wo_world);
    if (wo.z == 0.0f) return none;
    uint8_t sampled_type = bxdf.type;
    const BxDFSample s = bxdf_sample_f(bxdf, wo, u_remapped);
    V3 wi = s.wi;
    float pdf = s.pdf;
    RGB f = s.f;
    if (s.has_type) sampled_type = s.sampled_type;
    if (pdf == 0.0f) return none;
    const V3 wi_world = b.local_to_world(wi);
    if (!((bxdf.type & BSDF_SPECULAR) != 0) && matching > 1) {
        for (int i = 0; i < b.n_bxdfs; ++i)
            if (i != chosen && b.bxdfs[i].matches(type)) pdf += bxdf_pdf(b.bxdfs[i], wo, wi);  // `!=` is egal on lobes; lobes of one BSDF are never identical
    }
    if (matching > 1) pdf /= (float)matching;
    if (!((bxdf.type & BSDF_SPECULAR) != 0)) {
        const bool reflect_ = (dot(wi_world, b.ng) * dot(wo_world, b.ng)) > 0;
        f = RGB(0.0f);
        for (int i = 0; i < b.n_bxdfs; ++i) {
            const BxDF& x = b.bxdfs[i];
            if (x.matches(type) && ((reflect_ && (x.type & BSDF_REFLECTION) != 0) || (!reflect_ && (x.type & BSDF_TRANSMISSION) != 0)))
                f = f + bxdf_f(x, wo, wi);
        }
    }
    BSDFSample r;
    r.wi = wi_world;
    r.f = f;
    r.pdf = pdf;
    r.sampled_type = sampled_type;
    return r;
}

// ---- materials/material.jl (textures are ConstantTexture: textures/basic.jl:4-10) -----------------------------------
struct Material {
    enum Kind { MATTE = 0, MIRROR = 1, GLASS = 2, PLASTIC = 3 } kind = MATTE;
    RGB Kd, Ks, Kr, Kt;
    float sigma = 0, u_roughness = 0, v_roughness = 0, index = 1, roughness = 0;
    bool remap_roughness = true;
};
// material.jl:16-31 / 39-46 / 75-116 / 135-151.  Transport mode T never changes a value (A.11) and is dropped.
inline BSDF compute_scattering(const Material& m, const SurfaceInteraction& si, bool allow_multiple_lobes) {
    switch (m.kind) {
    case Material::MATTE: {
        BSDF bsdf(si);
        const RGB r = clamp_spectrum(m.Kd);
        if (is_black(r)) return bsdf;
        const float sigma = jl_clamp(m.sigma, 0.0f, 90.0f);
        if (sigma == 0.0f)
            bsdf.add(LambertianReflection(r));
        else
            bsdf.add(OrenNayar(r, sigma));
        return bsdf;
    }
    case Material::MIRROR: {
        BSDF bsdf(si);
        const RGB r = clamp_spectrum(m.Kr);
        if (is_black(r)) return bsdf;
        bsdf.add(SpecularReflection(r, Fresnel{}));
        return bsdf;
    }
    case Material::GLASS: {
        const float eta = m.index;
        float ur = m.u_roughness, vr = m.v_roughness;
        BSDF bsdf(si, eta);
        const RGB r = clamp_spectrum(m.Kr), t = clamp_spectrum(m.Kt);
        if (is_black(r) && is_black(t)) return bsdf;
        const bool is_specular = ur == 0 && vr == 0;
        if (is_specular && allow_multiple_lobes) {
            bsdf.add(FresnelSpecular(r, t, 1.0f, eta));
            return bsdf;
        }
        if (m.remap_roughness) {
            ur = roughness_to_alpha(ur);
            vr = roughness_to_alpha(vr);
        }
        TrowbridgeReitz dist;
        if (!is_specular) dist = TrowbridgeReitz(ur, vr);
        if (!is_black(r)) {
            const Fresnel fr = FresnelDielectric(1.0f, eta);
            if (is_specular)
                bsdf.add(SpecularReflection(r, fr));
            else
                bsdf.add(MicrofacetReflection(r, dist, fr));
        }
        if (!is_black(t)) {
            if (is_specular)
                bsdf.add(SpecularTransmission(t, 1.0f, eta));
            else
                bsdf.add(MicrofacetTransmission(t, dist, 1.0f, eta));
        }
        return bsdf;
    }
    case Material::PLASTIC: {
        BSDF bsdf(si);
        const RGB kd = clamp_spectrum(m.Kd);
        if (!is_black(kd)) bsdf.add(LambertianReflection(kd));
        const RGB ks = clamp_spectrum(m.Ks);
        if (is_black(ks)) return bsdf;
        const Fresnel fr = FresnelDielectric(1.5f, 1.0f);
        float rough = m.roughness;
        if (m.remap_roughness) rough = roughness_to_alpha(rough);
        bsdf.add(MicrofacetReflection(ks, TrowbridgeReitz(rough, rough), fr));
        return bsdf;
    }
    }
    return BSDF();
}

// ---- lights/{light,point,spot}.jl ------------------------------------------------------------------------------------
struct Light {
    enum Kind { POINT = 0, SPOT = 1 } kind = POINT;
    Transformation light_to_world, world_to_light;
    RGB i;
    V3 position;
    float cos_total_width = 0, cos_falloff_start = 0;
};
inline Light PointLight(const Transformation& l2w, RGB i) {  // point.jl:19-24
    Light l;
    l.kind = Light::POINT;
    l.light_to_world = l2w;
    l.world_to_light = inv(l2w);
    l.i = i;
    l.position = l2w.point(V3(0.0f));
    return l;
}
inline Light SpotLight(const Transformation& l2w, RGB i, float total_width_deg, float falloff_start_deg) {  // spot.jl:10-19
    Light l;
    l.kind = Light::SPOT;
    l.light_to_world = l2w;
    l.world_to_light = inv(l2w);
    l.position = l2w.point(V3(0.0f));
    l.i = i;
    l.cos_total_width = tm_cosf(jl_deg2rad(total_width_deg));
    l.cos_falloff_start = tm_cosf(jl_deg2rad(falloff_start_deg));
    return l;
}
// spot.jl:32-40
inline float spot_falloff(const Light& s, V3 w) {
    const V3 wl = normalize(s.world_to_light.vec(w));
    const float c = wl.z;
    if (c < s.cos_total_width) return 0.0f;
    if (c >= s.cos_falloff_start) return 1.0f;
    const float d = (c - s.cos_total_width) / (s.cos_falloff_start - s.cos_total_width);
    return jl_pow4(d);
}
struct LightSample {
    RGB radiance;
    V3 wi;
    float pdf = 1;
    V3 p0, p1;  // VisibilityTester end points (light.jl:12-15)
    float time = 0;
};
// point.jl:50-58, spot.jl:22-30
inline LightSample sample_li(const Light& l, V3 ref_p, float ref_time) {
    LightSample s;
    s.wi = normalize(l.position - ref_p);
    s.pdf = 1.0f;
    s.p0 = ref_p;
    s.p1 = l.position;
    s.time = ref_time;
    if (l.kind == Light::POINT)
        s.radiance = l.i / distance_squared(l.position, ref_p);
    else
        s.radiance = l.i * spot_falloff(l, -s.wi) / distance_squared(l.position, ref_p);
    return s;
}
// point.jl:74-76, spot.jl:42-44
inline RGB light_power(const Light& l) {
    if (l.kind == Light::POINT) return 4.0f * PI_F * l.i;
    return l.i * 2.0f * PI_F * (1.0f - 0.5f * (l.cos_falloff_start + l.cos_total_width));
}

// Trace.jl:176-194
struct Scene {
    std::vector<Light> lights;
    std::vector<Material> materials;
    BVHAccel aggregate;
};
inline bool scene_intersect(Scene& sc, Ray& ray, SurfaceInteraction& si) {
    counters().closest++;
    return bvh_intersect(sc.aggregate, ray, si);
}
inline bool scene_intersect_p(Scene& sc, Ray& ray) {
    counters().shadow++;
    return bvh_intersect_p(sc.aggregate, ray);
}
// light.jl:17-19
inline bool unoccluded(Scene& sc, const LightSample& v) {
    Ray r = spawn_ray_to(v.p0, v.time, v.p1);
    return !scene_intersect_p(sc, r);
}
// primitive.jl:29-35 + surface_interaction.jl:141-147.  Returns an invalid BSDF when the primitive has no material.
inline BSDF compute_scattering(const Scene& sc, const SurfaceInteraction& si, bool allow_multiple_lobes) {
    if (!si.prim || si.prim->material < 0) return BSDF();
    return compute_scattering(sc.materials[si.prim->material], si, allow_multiple_lobes);
}

}  // namespace orc
