/* trace_detmath.h — deterministic elementary functions shared by every build of this repo.
 *
 * Why this exists (SURVEY.md §7 "Hard parts", Appendix A.16): the reference (Trace.jl) calls Julia's own
 * sin/cos/tan/atan/acos/log.  Julia evaluates the Float32 versions of sin/cos/tan/log through Float64
 * kernels and rounds once, so they are (almost always) the correctly rounded value; atan/acos are <= 1 ulp.
 * glibc's libm (CPU) and OCML (gfx950) are each a different <= 1-2 ulp function, and a 1-ulp difference flips
 * hit/miss decisions at silhouettes.  To make "GPU result == CPU-oracle result" a bit-for-bit statement we
 * evaluate every transcendental through the SAME Float64 code on both sides and round once to Float32:
 *   - only IEEE-754 correctly rounded operations are used (+ - * / sqrt, int<->fp conversions),
 *   - no FMA contraction (-ffp-contract=off is mandatory for every translation unit that includes this),
 *   - no table lookups that depend on the platform.
 * Accuracy of the Float64 kernels is ~1e-16 relative on the domains used by the path (|x| <= ~1e3 for
 * sin/cos/tan), so the Float32 results are the correctly rounded values except in ~1e-8 of cases.
 * tests/test_detmath.py checks that against mpmath/numpy.
 *
 * This header is part of the *specification* of the boundary (like the sampler in trace_sampler.h), not of the
 * oracle: both oracle/ and trace.jl_amd/csrc include it.
 */
#ifndef TRACE_DETMATH_H
#define TRACE_DETMATH_H

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define TM_HD __host__ __device__ inline
#else
#define TM_HD inline
#endif

#define TM_PI_D 3.14159265358979311600e+00      /* 0x1.921fb54442d18p+1 */
#define TM_PIO2_D 1.57079632679489655800e+00    /* 0x1.921fb54442d18p+0 */
#define TM_PIO2_LO_D 6.12323399573676603587e-17 /* 0x1.1a62633145c07p-54 */
#define TM_LN2_D 6.93147180559945286227e-01     /* 0x1.62e42fefa39efp-1 */
#define TM_INVPIO2_D 6.36619772367581382433e-01 /* 0x1.45f306dc9c883p-1 */

/* Float32(pi) as Julia promotes it in Float32 expressions (A.16b). */
#define TM_PI_F 3.14159274101257324219f

TM_HD double tm_bits_to_double(unsigned long long u) {
    double d;
    __builtin_memcpy(&d, &u, sizeof d);
    return d;
}
TM_HD unsigned long long tm_double_to_bits(double d) {
    unsigned long long u;
    __builtin_memcpy(&u, &d, sizeof u);
    return u;
}
TM_HD bool tm_isnan(double x) { return x != x; }
TM_HD double tm_abs(double x) { return x < 0.0 ? -x : (x == 0.0 ? 0.0 : x); }

/* sin on |r| <= pi/4 (+ a little): Taylor to r^19, exact 1/n! coefficients, Horner in r^2. */
TM_HD double tm_sin_kernel(double r) {
    const double z = r * r;
    double p = -1.0 / 121645100408832000.0;       /* -1/19! */
    p = p * z + 1.0 / 355687428096000.0;          /*  1/17! */
    p = p * z - 1.0 / 1307674368000.0;            /* -1/15! */
    p = p * z + 1.0 / 6227020800.0;               /*  1/13! */
    p = p * z - 1.0 / 39916800.0;                 /* -1/11! */
    p = p * z + 1.0 / 362880.0;                   /*  1/9!  */
    p = p * z - 1.0 / 5040.0;                     /* -1/7!  */
    p = p * z + 1.0 / 120.0;                      /*  1/5!  */
    p = p * z - 1.0 / 6.0;                        /* -1/3!  */
    return r + r * (z * p);
}
/* cos on |r| <= pi/4 (+ a little): Taylor to r^20. */
TM_HD double tm_cos_kernel(double r) {
    const double z = r * r;
    double p = 1.0 / 2432902008176640000.0;       /*  1/20! */
    p = p * z - 1.0 / 6402373705728000.0;         /* -1/18! */
    p = p * z + 1.0 / 20922789888000.0;           /*  1/16! */
    p = p * z - 1.0 / 87178291200.0;              /* -1/14! */
    p = p * z + 1.0 / 479001600.0;                /*  1/12! */
    p = p * z - 1.0 / 3628800.0;                  /* -1/10! */
    p = p * z + 1.0 / 40320.0;                    /*  1/8!  */
    p = p * z - 1.0 / 720.0;                      /* -1/6!  */
    p = p * z + 1.0 / 24.0;                       /*  1/4!  */
    return 1.0 - (0.5 * z - (z * z) * p);
}
/* Cody–Waite reduction x = k*pi/2 + r; good to ~1e-16*|k| absolute, which is all the path needs. */
TM_HD double tm_rem_pio2(double x, long long* k_out) {
    const double t = x * TM_INVPIO2_D;
    const long long k = (long long)(t >= 0.0 ? t + 0.5 : t - 0.5);
    const double kd = (double)k;
    const double r = (x - kd * TM_PIO2_D) - kd * TM_PIO2_LO_D;
    *k_out = k;
    return r;
}
TM_HD double tm_sin(double x) {
    if (!(x == x) || x - x != 0.0) return x - x; /* NaN, Inf -> NaN */
    long long k;
    const double r = tm_rem_pio2(x, &k);
    switch ((int)(k & 3)) {
    case 0: return tm_sin_kernel(r);
    case 1: return tm_cos_kernel(r);
    case 2: return -tm_sin_kernel(r);
    default: return -tm_cos_kernel(r);
    }
}
TM_HD double tm_cos(double x) {
    if (!(x == x) || x - x != 0.0) return x - x;
    long long k;
    const double r = tm_rem_pio2(x, &k);
    switch ((int)(k & 3)) {
    case 0: return tm_cos_kernel(r);
    case 1: return -tm_sin_kernel(r);
    case 2: return -tm_cos_kernel(r);
    default: return tm_sin_kernel(r);
    }
}
TM_HD double tm_tan(double x) { return tm_sin(x) / tm_cos(x); }
/* sin and cos of the same argument: one reduction, each kernel once, the quadrant applied by selection.  The values are
 * those of tm_sin(x) and tm_cos(x) bit for bit (negation is exact); on a GPU the lanes of a wave no longer serialise over
 * the four quadrant cases of two switches. */
TM_HD void tm_sincos(double x, double* s_out, double* c_out) {
    if (!(x == x) || x - x != 0.0) {
        *s_out = *c_out = x - x;
        return;
    }
    long long k;
    const double r = tm_rem_pio2(x, &k);
    const double sk = tm_sin_kernel(r), ck = tm_cos_kernel(r);
    const int q = (int)(k & 3);
    const double s = (q & 1) ? ck : sk, c = (q & 1) ? sk : ck;
    *s_out = (q & 2) ? -s : s;
    *c_out = (q == 1 || q == 2) ? -c : c;
}

/* atan for t >= 0. */
TM_HD double tm_atan_pos(double t) {
    /* atan(k/8), k = 0..8, correctly rounded (generated with mpmath at 200 bits). */
    const double atan_c[9] = {
        0.0,
        1.24354994546761438e-01, /* 0x1.fd5ba9aac2f6ep-4 */
        2.44978663126864143e-01, /* 0x1.f5b75f92c80ddp-3 */
        3.58770670270572245e-01, /* 0x1.6f61941e4def1p-2 */
        4.63647609000806094e-01, /* 0x1.dac670561bb4fp-2 */
        5.58599315343562441e-01, /* 0x1.1e00babdefeb4p-1 */
        6.43501108793284371e-01, /* 0x1.4978fa3269ee1p-1 */
        7.18829999621624491e-01, /* 0x1.700a7c5784634p-1 */
        7.85398163397448279e-01, /* 0x1.921fb54442d18p-1 */
    };
    bool inv = false;
    if (t > 1.0) {
        t = 1.0 / t;
        inv = true;
    }
    const int k = (int)(t * 8.0 + 0.5);
    const double c = (double)k * 0.125;
    const double u = (t - c) / (1.0 + t * c);
    const double z = u * u;
    double p = -1.0 / 15.0;
    p = p * z + 1.0 / 13.0;
    p = p * z - 1.0 / 11.0;
    p = p * z + 1.0 / 9.0;
    p = p * z - 1.0 / 7.0;
    p = p * z + 1.0 / 5.0;
    p = p * z - 1.0 / 3.0;
    const double a = atan_c[k] + (u + u * (z * p));
    return inv ? (TM_PIO2_D - a) + TM_PIO2_LO_D : a;
}
TM_HD double tm_atan(double x) {
    if (x != x) return x;
    return x < 0.0 ? -tm_atan_pos(-x) : tm_atan_pos(x);
}
/* atan2 with C / Julia `atan(y, x)` conventions for the finite cases the path can produce. */
TM_HD double tm_atan2(double y, double x) {
    if (x != x || y != y) return x + y;
    const bool yneg = (tm_double_to_bits(y) >> 63) != 0;
    const bool xneg = (tm_double_to_bits(x) >> 63) != 0;
    if (y == 0.0) {
        const double r = xneg ? TM_PI_D : 0.0;
        return yneg ? -r : r;
    }
    if (x == 0.0) return yneg ? -TM_PIO2_D : TM_PIO2_D;
    const double ay = yneg ? -y : y;
    const double ax = xneg ? -x : x;
    double a;
    if (ay - ay != 0.0) {                           /* |y| = Inf */
        a = (ax - ax != 0.0) ? 0.5 * TM_PIO2_D : TM_PIO2_D;
    } else if (ax - ax != 0.0) {                    /* |x| = Inf */
        a = 0.0;
    } else {
        a = tm_atan_pos(ay / ax);
    }
    if (xneg) a = (TM_PI_D - a) + 2.0 * TM_PIO2_LO_D;
    return yneg ? -a : a;
}
/* acos on [-1, 1] (NaN outside), via atan2(sqrt((1-x)(1+x)), x). */
TM_HD double tm_acos(double x) {
    if (!(x >= -1.0 && x <= 1.0)) return __builtin_nan("");
    const double s = __builtin_sqrt((1.0 - x) * (1.0 + x));
    return tm_atan2(s, x);
}
/* natural log, x > 0 finite normal double (every Float32 > 0 converts to one). */
TM_HD double tm_log(double x) {
    if (x != x) return x;
    if (x < 0.0) return __builtin_nan("");
    if (x == 0.0) return -__builtin_huge_val();
    if (x - x != 0.0) return x;
    unsigned long long u = tm_double_to_bits(x);
    long long e = (long long)((u >> 52) & 0x7ff) - 1023;
    u = (u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m = tm_bits_to_double(u);                /* [1, 2) */
    if (m > 1.41421356237309514547) {
        m = m * 0.5;
        e += 1;
    }
    const double s = (m - 1.0) / (m + 1.0);
    const double z = s * s;
    double p = 1.0 / 25.0;
    p = p * z + 1.0 / 23.0;
    p = p * z + 1.0 / 21.0;
    p = p * z + 1.0 / 19.0;
    p = p * z + 1.0 / 17.0;
    p = p * z + 1.0 / 15.0;
    p = p * z + 1.0 / 13.0;
    p = p * z + 1.0 / 11.0;
    p = p * z + 1.0 / 9.0;
    p = p * z + 1.0 / 7.0;
    p = p * z + 1.0 / 5.0;
    p = p * z + 1.0 / 3.0;
    const double lm = 2.0 * (s + s * (z * p));
    return (double)e * TM_LN2_D + lm;
}

/* Float32 front ends: one rounding from the Float64 kernel. */
TM_HD float tm_sinf(float x) { return (float)tm_sin((double)x); }
TM_HD float tm_cosf(float x) { return (float)tm_cos((double)x); }
TM_HD void tm_sincosf(float x, float* s_out, float* c_out) { /* == tm_sinf(x), tm_cosf(x) */
    double s, c;
    tm_sincos((double)x, &s, &c);
    *s_out = (float)s;
    *c_out = (float)c;
}
TM_HD float tm_tanf(float x) { return (float)tm_tan((double)x); }
TM_HD float tm_atanf(float x) { return (float)tm_atan((double)x); }
TM_HD float tm_atan2f(float y, float x) { return (float)tm_atan2((double)y, (double)x); }
TM_HD float tm_acosf(float x) { return (float)tm_acos((double)x); }
TM_HD float tm_logf(float x) { return (float)tm_log((double)x); }

#endif /* TRACE_DETMATH_H */
