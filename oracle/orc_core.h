// oracle/orc_core.h — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement of pxl-th/Trace.jl @ 2024_10_08: value types and helpers of layers L0/L1
// (SURVEY.md §1).  Every function cites the reference file:line (relative to /root/reference/src) it follows
// and keeps the reference's operation order in Float32; the behavioural ledger is SURVEY.md Appendix A.
// Build flags are part of the contract: -O2 -ffp-contract=off -fno-fast-math (oracle/Makefile).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
// Parity status: pinned against the reference's own unit-test vectors by
// tests/test_oracle_kat.py; the reference itself cannot run here (no Julia), see DESIGN.md.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#include "../include/trace_detmath.h"

namespace orc {

constexpr float INF32 = std::numeric_limits<float>::infinity();
constexpr float PI_F = TM_PI_F;  // Float32(π): how Julia promotes π in Float32 expressions (A.16b)

// ---- Julia scalar semantics -------------------------------------------------------------------
// Base.max / Base.min propagate NaN and order signed zeros; Base.clamp passes NaN through.
inline float jl_max(float a, float b) {
    if (a != a || b != b) return a + b;
    if (a > b) return a;
    if (b > a) return b;
    return std::signbit(a) ? b : a;
}
inline float jl_min(float a, float b) {
    if (a != a || b != b) return a + b;
    if (a < b) return a;
    if (b < a) return b;
    return std::signbit(a) ? a : b;
}
inline float jl_clamp(float x, float lo, float hi) { return x > hi ? hi : (x < lo ? lo : x); }
inline double jl_maxd(double a, double b) {
    if (a != a || b != b) return a + b;
    if (a > b) return a;
    if (b > a) return b;
    return std::signbit(a) ? b : a;
}
inline float jl_abs(float x) { return std::fabs(x); }
// isapprox(x, y) with default rtol = sqrt(eps(Float32)), atol = 0 (A.1).
inline bool jl_isapprox(float x, float y) {
    if (x == y) return true;
    if (!std::isfinite(x) || !std::isfinite(y)) return false;
    const float rtol = 0.00034526698f;
    return std::fabs(x - y) <= rtol * jl_max(std::fabs(x), std::fabs(y));
}
// deg2rad(x::Float32) = x * (Float32(π) / 180f0)   (A.16c)
inline float jl_deg2rad(float x) { return x * (PI_F / 180.0f); }
// x^4 on Float32 = power_by_squaring in Float64, one rounding (A.16d)
inline float jl_pow4(float x) {
    const double d = (double)x;
    const double d2 = d * d;
    return (float)(d2 * d2);
}

// ---- vectors -----------------------------------------------------------------------------------------------------
struct V2 {
    float x = 0, y = 0;
    float operator[](int i) const { return i == 0 ? x : y; }
};
inline V2 operator+(V2 a, V2 b) { return {a.x + b.x, a.y + b.y}; }
inline V2 operator-(V2 a, V2 b) { return {a.x - b.x, a.y - b.y}; }
inline V2 operator*(float s, V2 a) { return {s * a.x, s * a.y}; }
inline V2 operator*(V2 a, float s) { return {a.x * s, a.y * s}; }

struct V3 {
    float x = 0, y = 0, z = 0;
    V3() = default;
    V3(float a, float b, float c) : x(a), y(b), z(c) {}
    explicit V3(float a) : x(a), y(a), z(a) {}
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    float& at(int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }  // broadcast .*
inline V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline V3 operator/(V3 a, V3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
inline bool operator==(V3 a, V3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
// StaticArrays: dot = left-to-right sum of products; cross = 3 differences of products; norm = sqrt(Σx²);
// normalize(v) = inv(norm(v)) * v   (SURVEY.md §8c "assumed third-party semantics")
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float norm(V3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
inline V3 normalize(V3 a) { return (1.0f / norm(a)) * a; }
inline V3 vabs(V3 a) { return {std::fabs(a.x), std::fabs(a.y), std::fabs(a.z)}; }
inline bool is_zero(V3 a) { return a.x == 0 && a.y == 0 && a.z == 0; }
inline bool has_nan(V3 a) { return a.x != a.x || a.y != a.y || a.z != a.z; }
inline V3 vmin(V3 a, V3 b) { return {jl_min(a.x, b.x), jl_min(a.y, b.y), jl_min(a.z, b.z)}; }
inline V3 vmax(V3 a, V3 b) { return {jl_max(a.x, b.x), jl_max(a.y, b.y), jl_max(a.z, b.z)}; }

// Trace.jl:98  sum_mul(a, b) = a[1]*b[1] + a[2]*b[2] + a[3]*b[3]  (b is a triple of vectors)
inline V3 sum_mul(V3 a, const V3 b[3]) { return a.x * b[0] + a.y * b[1] + a.z * b[2]; }
inline V2 sum_mul(V3 a, const V2 b[3]) { return a.x * b[0] + a.y * b[1] + a.z * b[2]; }

// ---- Trace.jl:48-168 helpers ---------------------------------------------------------------------------------------
// Trace.jl:48-61
inline V2 concentric_sample_disk(V2 u) {
    const V2 offset = 2.0f * u - V2{1.0f, 1.0f};
    if (offset.x == 0 && offset.y == 0) return {0, 0};
    float r, th;
    if (std::fabs(offset.x) > std::fabs(offset.y)) {
        r = offset.x;
        th = (offset.y / offset.x) * PI_F / 4.0f;
    } else {
        r = offset.y;
        th = PI_F / 2.0f - (offset.x / offset.y) * PI_F / 4.0f;
    }
    return r * V2{tm_cosf(th), tm_sinf(th)};
}
// Trace.jl:63-67
inline V3 cosine_sample_hemisphere(V2 u) {
    const V2 d = concentric_sample_disk(u);
    const float z = std::sqrt(jl_max(0.0f, 1.0f - d.x * d.x - d.y * d.y));
    return {d.x, d.y, z};
}
// Trace.jl:69-74
inline V3 uniform_sample_sphere(V2 u) {
    const float z = 1.0f - 2.0f * u.x;
    const float r = std::sqrt(jl_max(0.0f, 1.0f - z * z));
    const float phi = 2.0f * PI_F * u.y;
    return {r * tm_cosf(phi), r * tm_sinf(phi), z};
}
// Trace.jl:76-81
inline V3 uniform_sample_cone(V2 u, float cos_max) {
    const float c = 1.0f - u.x + u.x * cos_max;
    const float s = std::sqrt(1.0f - c * c);
    const float phi = u.y * 2.0f * PI_F;
    return {tm_cosf(phi) * s, tm_sinf(phi) * s, c};
}
inline float uniform_sphere_pdf() { return 1.0f / (4.0f * PI_F); }                         // Trace.jl:92
inline float uniform_cone_pdf(float cos_max) { return 1.0f / (2.0f * PI_F * (1.0f - cos_max)); }  // Trace.jl:94-96

// Trace.jl:109-121 (sin_ϕ returns 1 at the pole — A.11)
inline float cos_theta(V3 w) { return w.z; }
inline float sin_theta2(V3 w) { return jl_max(0.0f, 1.0f - cos_theta(w) * cos_theta(w)); }
inline float sin_theta(V3 w) { return std::sqrt(sin_theta2(w)); }
inline float tan_theta(V3 w) { return sin_theta(w) / cos_theta(w); }
inline float cos_phi(V3 w) {
    const float s = sin_theta(w);
    return s == 0.0f ? 1.0f : jl_clamp(w.x / s, -1.0f, 1.0f);
}
inline float sin_phi(V3 w) {
    const float s = sin_theta(w);
    return s == 0.0f ? 1.0f : jl_clamp(w.y / s, -1.0f, 1.0f);
}
// Trace.jl:126   reflect(wo, n) = -wo + 2f0 * (wo ⋅ n) * n
inline V3 reflect(V3 wo, V3 n) { return -wo + (2.0f * dot(wo, n)) * n; }
// Trace.jl:139-146
inline void coordinate_system(V3 v1, V3& v2, V3& v3) {
    if (std::fabs(v1.x) > std::fabs(v1.y))
        v2 = V3(-v1.z, 0, v1.x) / std::sqrt(v1.x * v1.x + v1.z * v1.z);
    else
        v2 = V3(0, v1.z, -v1.y) / std::sqrt(v1.y * v1.y + v1.z * v1.z);
    v3 = cross(v1, v2);
}
// Trace.jl:148-157
inline V3 spherical_direction(float sin_t, float cos_t, float phi) {
    return {sin_t * tm_cosf(phi), sin_t * tm_sinf(phi), cos_t};
}
// Trace.jl:170   face_forward(n, v) = (n ⋅ v) < 0 ? -n : n
inline V3 face_forward(V3 n, V3 v) { return dot(n, v) < 0 ? -n : n; }

// ---- ray.jl --------------------------------------------------------------------------------------------------------
struct Ray {  // ray.jl:1-6 (differentials are dead data, A.10, and are not carried)
    V3 o, d;
    float t_max = INF32;
    float time = 0;
    V3 at(float t) const { return o + d * t; }  // ray.jl:31-33
};
// ray.jl:25-29: only -0.0 -> +0.0 (A.2)
inline void check_direction(Ray& r) {
    r.d = V3(r.d.x == 0.0f ? 0.0f : r.d.x, r.d.y == 0.0f ? 0.0f : r.d.y, r.d.z == 0.0f ? 0.0f : r.d.z);
}

// ---- bounds.jl -----------------------------------------------------------------------------------------------------
struct Bounds2 {
    V2 p_min{INF32, INF32}, p_max{-INF32, -INF32};
};
struct Bounds3 {
    V3 p_min{INF32, INF32, INF32}, p_max{-INF32, -INF32, -INF32};  // bounds.jl:13 invalid by default
    Bounds3() = default;
    Bounds3(V3 a, V3 b) : p_min(a), p_max(b) {}
    explicit Bounds3(V3 p) : p_min(p), p_max(p) {}
    const V3& operator[](int i) const { return i == 1 ? p_min : p_max; }  // 1-based like bounds.jl:25-29
};
inline Bounds3 bunion(const Bounds3& a, const Bounds3& b) { return {vmin(a.p_min, b.p_min), vmax(a.p_max, b.p_max)}; }  // :62-64
inline Bounds2 bintersect(const Bounds2& a, const Bounds2& b) {  // bounds.jl:66-68
    return {{jl_max(a.p_min.x, b.p_min.x), jl_max(a.p_min.y, b.p_min.y)}, {jl_min(a.p_max.x, b.p_max.x), jl_min(a.p_max.y, b.p_max.y)}};
}
inline bool is_valid(const Bounds3& b) {  // bounds.jl:30-32
    return b.p_min.x != INF32 && b.p_min.y != INF32 && b.p_min.z != INF32 && b.p_max.x != -INF32 && b.p_max.y != -INF32 &&
           b.p_max.z != -INF32;
}
inline V3 corner(const Bounds3& b, int c) {  // bounds.jl:50-58 (c is 1-based)
    c -= 1;
    return {b[(c & 1) + 1].x, b[(c & 2) != 0 ? 2 : 1].y, b[(c & 4) != 0 ? 2 : 1].z};
}
inline V3 diagonal(const Bounds3& b) { return b.p_max - b.p_min; }  // bounds.jl:85
inline float surface_area(const Bounds3& b) {                        // bounds.jl:87-90 (note: this is the Bounds3 one)
    const V3 d = diagonal(b);
    return 2 * (d.x * d.y + d.x * d.z + d.y * d.z);
}
inline int maximum_extent(const Bounds3& b) {  // bounds.jl:116-124, 1-based axis
    const V3 d = diagonal(b);
    if (d.x > d.y && d.x > d.z) return 1;
    if (d.y > d.z) return 2;
    return 3;
}
inline float lerp(float v1, float v2, float t) { return (1 - t) * v1 + t * v2; }  // bounds.jl:126
inline float distance_squared(V3 p1, V3 p2) {                                     // bounds.jl:132-135
    const V3 p = p1 - p2;
    return dot(p, p);
}
inline float distance(V3 p1, V3 p2) { return norm(p1 - p2); }  // bounds.jl:131
inline V3 offset(const Bounds3& b, V3 p) {                     // bounds.jl:138-147
    const V3 o = p - b.p_min;
    const bool g0 = b.p_max.x > b.p_min.x, g1 = b.p_max.y > b.p_min.y, g2 = b.p_max.z > b.p_min.z;
    if (!(g0 || g1 || g2)) return o;
    return {o.x / (g0 ? b.p_max.x - b.p_min.x : 1.0f), o.y / (g1 ? b.p_max.y - b.p_min.y : 1.0f), o.z / (g2 ? b.p_max.z - b.p_min.z : 1.0f)};
}
inline Bounds3 expand(const Bounds3& b, float d) { return {b.p_min - V3(d), b.p_max + V3(d)}; }  // bounds.jl:84

// bounds.jl:155-173  intersect(b, ray) -> (hit, t0, t1)
inline bool bounds_intersect(const Bounds3& b, const Ray& ray, float& t0o, float& t1o) {
    float t0 = 0.0f, t1 = ray.t_max;
    for (int i = 0; i < 3; ++i) {
        const float inv = 1.0f / ray.d[i];
        float tn = (b.p_min[i] - ray.o[i]) * inv;
        float tf = (b.p_max[i] - ray.o[i]) * inv;
        if (tn > tf) std::swap(tn, tf);
        t0 = tn > t0 ? tn : t0;
        t1 = tf < t1 ? tf : t1;
        if (t0 > t1) {
            t0o = t1o = 0;
            return false;
        }
    }
    t0o = t0;
    t1o = t1;
    return true;
}
// bounds.jl:175-181: 1 = positive, 2 = negative
inline void is_dir_negative(V3 d, int neg[3]) {
    neg[0] = d.x < 0 ? 2 : 1;
    neg[1] = d.y < 0 ? 2 : 1;
    neg[2] = d.z < 0 ? 2 : 1;
}
// bounds.jl:186-206 — note line 197 `ty_max > tx_max && (tx_max = ty_max)` (takes the LARGER far value) and no
// (1+2γ) robustness factor: restated as written.
inline bool bounds_intersect_p(const Bounds3& b, const Ray& ray, V3 inv_dir, const int neg[3]) {
    float tx_min = (b[neg[0]].x - ray.o.x) * inv_dir.x;
    float tx_max = (b[3 - neg[0]].x - ray.o.x) * inv_dir.x;
    const float ty_min = (b[neg[1]].y - ray.o.y) * inv_dir.y;
    const float ty_max = (b[3 - neg[1]].y - ray.o.y) * inv_dir.y;
    if (tx_min > ty_max || ty_min > tx_max) return false;
    if (ty_min > tx_min) tx_min = ty_min;
    if (ty_max > tx_max) tx_max = ty_max;
    const float tz_min = (b[neg[2]].z - ray.o.z) * inv_dir.z;
    const float tz_max = (b[3 - neg[2]].z - ray.o.z) * inv_dir.z;
    if (tx_min > tz_max || tz_min > tx_max) return false;
    if (tz_min > tx_min) tx_min = tz_min;
    if (tz_max < tx_max) tx_max = tz_max;
    return tx_min < ray.t_max && tx_max > 0;
}

// ---- transformations.jl:1-165 ---------------------------------------------------------------------------------------
struct M4 {
    float m[4][4];  // m[row][col]
    static M4 identity() {
        M4 r;
        std::memset(r.m, 0, sizeof r.m);
        for (int i = 0; i < 4; ++i) r.m[i][i] = 1.0f;
        return r;
    }
    static M4 rows(float a00, float a01, float a02, float a03, float a10, float a11, float a12, float a13, float a20, float a21,
                   float a22, float a23, float a30, float a31, float a32, float a33) {
        M4 r;
        const float v[16] = {a00, a01, a02, a03, a10, a11, a12, a13, a20, a21, a22, a23, a30, a31, a32, a33};
        for (int i = 0; i < 16; ++i) r.m[i / 4][i % 4] = v[i];
        return r;
    }
};
inline M4 transpose(const M4& a) {
    M4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) r.m[i][j] = a.m[j][i];
    return r;
}
// StaticArrays 4x4 * 4x4: c[i,j] = a[i,1]*b[1,j] + a[i,2]*b[2,j] + a[i,3]*b[3,j] + a[i,4]*b[4,j], left to right.
inline M4 mul(const M4& a, const M4& b) {
    M4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j] + a.m[i][3] * b.m[3][j];
    return r;
}
// inv(::Mat4f): StaticArrays' closed form = cofactor(i,j) * (1 / det).  The term order inside StaticArrays'
// cofactor polynomials could not be checked offline; it only matters for general matrices (tolerance source,
// DESIGN.md), not for the perspective matrix whose cofactors are single products.
inline float det3(float a, float b, float c, float d, float e, float f, float g, float h, float i) {
    return a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
}
inline M4 inv(const M4& A) {
    float cof[4][4];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            float s[9];
            int k = 0;
            for (int i = 0; i < 4; ++i) {
                if (i == r) continue;
                for (int j = 0; j < 4; ++j) {
                    if (j == c) continue;
                    s[k++] = A.m[i][j];
                }
            }
            const float d = det3(s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7], s[8]);
            cof[r][c] = ((r + c) & 1) ? -d : d;
        }
    const float det = A.m[0][0] * cof[0][0] + A.m[0][1] * cof[0][1] + A.m[0][2] * cof[0][2] + A.m[0][3] * cof[0][3];
    const float idet = 1.0f / det;
    M4 R;
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) R.m[r][c] = cof[c][r] * idet;
    return R;
}

struct Transformation {  // transformations.jl:1-4
    M4 m = M4::identity(), inv_m = M4::identity();
    Transformation() = default;
    Transformation(const M4& a, const M4& b) : m(a), inv_m(b) {}
    explicit Transformation(const M4& a) : m(a), inv_m(inv(a)) {}  // :7

    // transformations.jl:132-138
    V3 point(V3 p) const {
        const float x = m.m[0][0] * p.x + m.m[0][1] * p.y + m.m[0][2] * p.z + m.m[0][3] * 1.0f;
        const float y = m.m[1][0] * p.x + m.m[1][1] * p.y + m.m[1][2] * p.z + m.m[1][3] * 1.0f;
        const float z = m.m[2][0] * p.x + m.m[2][1] * p.y + m.m[2][2] * p.z + m.m[2][3] * 1.0f;
        const float w = m.m[3][0] * p.x + m.m[3][1] * p.y + m.m[3][2] * p.z + m.m[3][3] * 1.0f;
        if (w == 1) return {x, y, z};
        return {x / w, y / w, z / w};
    }
    // :139
    V3 vec(V3 v) const {
        return {m.m[0][0] * v.x + m.m[0][1] * v.y + m.m[0][2] * v.z, m.m[1][0] * v.x + m.m[1][1] * v.y + m.m[1][2] * v.z,
                m.m[2][0] * v.x + m.m[2][1] * v.y + m.m[2][2] * v.z};
    }
    // :140   transpose(inv_m[1:3,1:3]) * n
    V3 normal(V3 n) const {
        return {inv_m.m[0][0] * n.x + inv_m.m[1][0] * n.y + inv_m.m[2][0] * n.z, inv_m.m[0][1] * n.x + inv_m.m[1][1] * n.y + inv_m.m[2][1] * n.z,
                inv_m.m[0][2] * n.x + inv_m.m[1][2] * n.y + inv_m.m[2][2] * n.z};
    }
    // :141-143  mapreduce(i -> Bounds3(t(corner(b, i))), ∪, 1:8)
    Bounds3 bounds(const Bounds3& b) const {
        Bounds3 r(point(corner(b, 1)));
        for (int i = 2; i <= 8; ++i) r = bunion(r, Bounds3(point(corner(b, i))));
        return r;
    }
    // :144
    Ray ray(const Ray& r) const { return Ray{point(r.o), vec(r.d), r.t_max, r.time}; }
};
inline Transformation inv(const Transformation& t) { return {t.inv_m, t.m}; }  // :12
// :20-22 — inverse multiplied in the SAME order (A.3, load-bearing)
inline Transformation operator*(const Transformation& a, const Transformation& b) { return {mul(a.m, b.m), mul(a.inv_m, b.inv_m)}; }
// :24-38
inline Transformation translate(V3 d) {
    return {M4::rows(1, 0, 0, d.x, 0, 1, 0, d.y, 0, 0, 1, d.z, 0, 0, 0, 1), M4::rows(1, 0, 0, -d.x, 0, 1, 0, -d.y, 0, 0, 1, -d.z, 0, 0, 0, 1)};
}
// :40-54
inline Transformation scale(float x, float y, float z) {
    return {M4::rows(x, 0, 0, 0, 0, y, 0, 0, 0, 0, z, 0, 0, 0, 0, 1), M4::rows(1 / x, 0, 0, 0, 0, 1 / y, 0, 0, 0, 0, 1 / z, 0, 0, 0, 0, 1)};
}
// :105-117
inline Transformation look_at(V3 position, V3 target, V3 up) {
    const V3 z_axis = normalize(position - target);
    const V3 x_axis = normalize(cross(up, z_axis));
    const V3 y_axis = cross(z_axis, x_axis);
    const M4 m = M4::rows(x_axis.x, y_axis.x, z_axis.x, 0, x_axis.y, y_axis.y, z_axis.y, 0, x_axis.z, y_axis.z, z_axis.z, 0, 0, 0, 0, 1);
    return translate(position) * Transformation(m, transpose(m));
}
// :119-130 — the Mat4f literal is NOT wrapped in transpose(): column-major fill (A.4, load-bearing)
inline Transformation perspective(float fov, float near, float far) {
    M4 p;
    std::memset(p.m, 0, sizeof p.m);
    // columns: (1,0,0,0) (0,1,0,0) (0,0,far/(far-near),-far*near/(far-near)) (0,0,1,0)
    p.m[0][0] = 1;
    p.m[1][1] = 1;
    p.m[2][2] = far / (far - near);
    p.m[3][2] = -far * near / (far - near);
    p.m[2][3] = 1;
    p.m[3][3] = 0;
    const float inv_tan = 1.0f / tm_tanf(jl_deg2rad(fov) / 2.0f);
    return scale(inv_tan, inv_tan, 1.0f) * Transformation(p);
}
// :161-163
inline bool swaps_handedness(const Transformation& t) {
    const float d = det3(t.m.m[0][0], t.m.m[0][1], t.m.m[0][2], t.m.m[1][0], t.m.m[1][1], t.m.m[1][2], t.m.m[2][0], t.m.m[2][1], t.m.m[2][2]);
    return d < 0;
}

// ---- spectrum.jl ---------------------------------------------------------------------------------------------------
using RGB = V3;  // RGBSpectrum.c :: Point3f; all operators are componentwise (spectrum.jl:16-31)
inline V3 XYZ_to_RGB(V3 xyz) {  // spectrum.jl:1-7
    return {3.240479f * xyz.x - 1.537150f * xyz.y - 0.498535f * xyz.z, -0.969256f * xyz.x + 1.875991f * xyz.y + 0.041556f * xyz.z,
            0.055648f * xyz.x - 0.204043f * xyz.y + 1.057311f * xyz.z};
}
inline V3 RGB_to_XYZ(V3 rgb) {  // spectrum.jl:8-14
    return {0.412453f * rgb.x + 0.357580f * rgb.y + 0.180423f * rgb.z, 0.212671f * rgb.x + 0.715160f * rgb.y + 0.072169f * rgb.z,
            0.019334f * rgb.x + 0.119193f * rgb.y + 0.950227f * rgb.z};
}
inline float to_Y(RGB s) { return 0.212671f * s.x + 0.715160f * s.y + 0.072169f * s.z; }  // spectrum.jl:64-66
inline bool is_black(RGB c) { return is_zero(c); }                                         // spectrum.jl:54
inline RGB clamp_spectrum(RGB c, float lo = 0.0f, float hi = INF32) {                      // spectrum.jl:34-38
    return {jl_clamp(c.x, lo, hi), jl_clamp(c.y, lo, hi), jl_clamp(c.z, lo, hi)};
}

}  // namespace orc
