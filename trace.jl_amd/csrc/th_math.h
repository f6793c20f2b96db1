// th_math.h — Float32 vector math for the gfx950 kernels (and the host-side scene flattener) with the reference's
// Julia semantics: left-to-right sums of products, no FMA contraction (build flag -ffp-contract=off), NaN-propagating
// max/min, normalize(v) = (1/‖v‖)·v.  This is product code; it shares nothing with oracle/.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/trace_detmath.h"
#include "../../include/trace_sampler.h"

#define TH_HD __host__ __device__ __forceinline__
#define TH_D __device__ __forceinline__

namespace th {

constexpr float kInf = __builtin_huge_valf();
constexpr float kPi = TM_PI_F;            // Float32(π)
constexpr float kInvPi = 1.0f / TM_PI_F;  // `1f0 / π`

TH_HD float fabs_(float x) { return __builtin_fabsf(x); }
TH_HD float sqrt_(float x) { return __builtin_sqrtf(x); }  // IEEE correctly rounded (-fhip-fp32-correctly-rounded-divide-sqrt)
TH_HD bool isnan_(float x) { return x != x; }
TH_HD bool isinf_(float x) { return fabs_(x) == kInf; }
TH_HD bool signbit_(float x) { return (__builtin_bit_cast(uint32_t, x) >> 31) != 0; }
// Base.max / Base.min / Base.clamp for Float32 (A.16f)
TH_HD float jmax(float a, float b) {
    if (a != a || b != b) return a + b;
    if (a > b) return a;
    if (b > a) return b;
    return signbit_(a) ? b : a;
}
TH_HD float jmin(float a, float b) {
    if (a != a || b != b) return a + b;
    if (a < b) return a;
    if (b < a) return b;
    return signbit_(a) ? a : b;
}
TH_HD float jclamp(float x, float lo, float hi) { return x > hi ? hi : (x < lo ? lo : x); }
TH_HD float deg2rad(float x) { return x * (kPi / 180.0f); }  // A.16c
TH_HD float pow4(float x) {                                   // A.16d: Float64 power-by-squaring, one rounding
    const double d = (double)x, d2 = d * d;
    return (float)(d2 * d2);
}

struct f2 {
    float x, y;
};
struct f3 {
    float x, y, z;
};
TH_HD f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
TH_HD f3 splat3(float a) { return f3{a, a, a}; }
TH_HD f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
TH_HD f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
TH_HD f3 operator-(f3 a) { return f3{-a.x, -a.y, -a.z}; }
TH_HD f3 operator*(float s, f3 a) { return f3{s * a.x, s * a.y, s * a.z}; }
TH_HD f3 operator*(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
TH_HD f3 operator*(f3 a, f3 b) { return f3{a.x * b.x, a.y * b.y, a.z * b.z}; }
TH_HD f3 operator/(f3 a, float s) { return f3{a.x / s, a.y / s, a.z / s}; }
TH_HD f3 operator/(f3 a, f3 b) { return f3{a.x / b.x, a.y / b.y, a.z / b.z}; }
TH_HD float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
TH_HD f3 cross(f3 a, f3 b) { return f3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
TH_HD float norm(f3 a) { return sqrt_(a.x * a.x + a.y * a.y + a.z * a.z); }
TH_HD f3 normalize(f3 a) { return (1.0f / norm(a)) * a; }
TH_HD bool is_black(f3 a) { return a.x == 0.0f && a.y == 0.0f && a.z == 0.0f; }
TH_HD bool has_nan(f3 a) { return a.x != a.x || a.y != a.y || a.z != a.z; }
TH_HD float comp(f3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
TH_HD f3 face_forward(f3 n, f3 v) { return dot(n, v) < 0.0f ? -n : n; }  // Trace.jl:170
TH_HD float to_Y(f3 s) { return 0.212671f * s.x + 0.715160f * s.y + 0.072169f * s.z; }  // spectrum.jl:64-66
TH_HD f3 rgb_to_xyz(f3 c) {                                                            // spectrum.jl:8-14
    return f3{0.412453f * c.x + 0.357580f * c.y + 0.180423f * c.z, 0.212671f * c.x + 0.715160f * c.y + 0.072169f * c.z,
              0.019334f * c.x + 0.119193f * c.y + 0.950227f * c.z};
}
TH_HD f3 xyz_to_rgb(f3 c) {  // spectrum.jl:1-7
    return f3{3.240479f * c.x - 1.537150f * c.y - 0.498535f * c.z, -0.969256f * c.x + 1.875991f * c.y + 0.041556f * c.z,
              0.055648f * c.x - 0.204043f * c.y + 1.057311f * c.z};
}

// 4x4 matrix, row-major, used for (t::Transformation)(p | v | n)  transformations.jl:132-140
struct m44 {
    float m[16];
};
TH_HD f3 xf_point(const float* m, f3 p) {
    const float x = m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3] * 1.0f;
    const float y = m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7] * 1.0f;
    const float z = m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11] * 1.0f;
    const float w = m[12] * p.x + m[13] * p.y + m[14] * p.z + m[15] * 1.0f;
    if (w == 1.0f) return f3{x, y, z};
    return f3{x / w, y / w, z / w};
}
TH_HD f3 xf_vec(const float* m, f3 v) {
    return f3{m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z, m[8] * v.x + m[9] * v.y + m[10] * v.z};
}
// transpose(inv_m[1:3,1:3]) * n
TH_HD f3 xf_normal(const float* inv_m, f3 n) {
    return f3{inv_m[0] * n.x + inv_m[4] * n.y + inv_m[8] * n.z, inv_m[1] * n.x + inv_m[5] * n.y + inv_m[9] * n.z,
              inv_m[2] * n.x + inv_m[6] * n.y + inv_m[10] * n.z};
}

// Trace.jl:48-67
TH_HD f2 concentric_sample_disk(f2 u) {
    const float ox = 2.0f * u.x - 1.0f, oy = 2.0f * u.y - 1.0f;
    if (ox == 0.0f && oy == 0.0f) return f2{0.0f, 0.0f};
    float r, th;
    if (fabs_(ox) > fabs_(oy)) {
        r = ox;
        th = (oy / ox) * kPi / 4.0f;
    } else {
        r = oy;
        th = kPi / 2.0f - (ox / oy) * kPi / 4.0f;
    }
    float sn, cs;
    tm_sincosf(th, &sn, &cs);  // == tm_sinf(th), tm_cosf(th): one reduction, no quadrant branches
    return f2{r * cs, r * sn};
}
TH_HD f3 cosine_sample_hemisphere(f2 u) {
    const f2 d = concentric_sample_disk(u);
    const float z = sqrt_(jmax(0.0f, 1.0f - d.x * d.x - d.y * d.y));
    return f3{d.x, d.y, z};
}
// Trace.jl:109-121
TH_HD float cos_theta(f3 w) { return w.z; }
TH_HD float sin_theta2(f3 w) { return jmax(0.0f, 1.0f - w.z * w.z); }
TH_HD float sin_theta(f3 w) { return sqrt_(sin_theta2(w)); }
TH_HD float tan_theta(f3 w) { return sin_theta(w) / cos_theta(w); }
TH_HD float cos_phi(f3 w) {
    const float s = sin_theta(w);
    return s == 0.0f ? 1.0f : jclamp(w.x / s, -1.0f, 1.0f);
}
TH_HD float sin_phi(f3 w) {
    const float s = sin_theta(w);
    return s == 0.0f ? 1.0f : jclamp(w.y / s, -1.0f, 1.0f);
}
TH_HD f3 reflect(f3 wo, f3 n) { return -wo + (2.0f * dot(wo, n)) * n; }  // Trace.jl:126
// Trace.jl:139-146
TH_HD void coordinate_system(f3 v1, f3& v2, f3& v3) {
    if (fabs_(v1.x) > fabs_(v1.y))
        v2 = mk3(-v1.z, 0.0f, v1.x) / sqrt_(v1.x * v1.x + v1.z * v1.z);
    else
        v2 = mk3(0.0f, v1.z, -v1.y) / sqrt_(v1.y * v1.y + v1.z * v1.z);
    v3 = cross(v1, v2);
}
// ray.jl:25-29: -0.0 -> +0.0
TH_HD f3 check_direction(f3 d) { return f3{d.x == 0.0f ? 0.0f : d.x, d.y == 0.0f ? 0.0f : d.y, d.z == 0.0f ? 0.0f : d.z}; }

}  // namespace th
