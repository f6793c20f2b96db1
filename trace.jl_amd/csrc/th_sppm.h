// th_sppm.h — SPPMIntegrator (integrators/sppm.jl, whole file) as wavefront passes on the device.
//
// The reference's iteration is camera pass -> grid of visible points -> photon pass (atomic ϕ, M) -> pixel update.  Here:
//   * camera paths and photon paths do not depend on the pixel statistics, so a BATCH of iterations shares every traversal
//     and shading launch (k_sppm_raygen / k_shade_sppm, k_photon_gen / k_shade_photon): B x the rays per launch instead of
//     B x the launches.  Camera: one path per (iteration, pixel), stopping at the first diffuse (or, at max depth, glossy)
//     vertex (sppm.jl:208-266); direct light goes through the shadow queue WITHOUT β (A.12) into per-depth term slots that
//     k_sppm_fold_ld adds to Ld in the reference's order.  The sampler stream of iteration k is (seed, pixel, sample k-1).
//     Photons: Halton dimensions by radical inverse (sampler/sampling.jl:43-60); the emission weight β is never updated
//     along the path (A.13); every hit at depth >= 2 is recorded.
//   * then per iteration, in order: grid bounds / resolution from the visible points and the current radii (:278-302), the
//     iteration's photon hits counting-sorted by the hash of their cell, a per-pixel gather of ϕ and M over the buckets of
//     the cells the visible point registers in (the same pairs the reference's bucket walk finds, k_sppm_gather), and the
//     Float64 pixel update (:438-459).
// M, radius, N, Ld and the visible points are bit-exact against the oracle; ϕ/τ are sums of the same terms in another order.
#pragma once
#include "th_kernels.h"

namespace th {

// primes.jl: the first primes, 2 omitted.  Dimension k >= 1 of radical_inverse uses kOddPrimes[k - 1].
__constant__ int kOddPrimes[256] = {
    3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59,
    61, 67, 71, 73, 79, 83, 89, 97, 101, 103, 107, 109, 113, 127, 131, 137,
    139, 149, 151, 157, 163, 167, 173, 179, 181, 191, 193, 197, 199, 211, 223, 227,
    229, 233, 239, 241, 251, 257, 263, 269, 271, 277, 281, 283, 293, 307, 311, 313,
    317, 331, 337, 347, 349, 353, 359, 367, 373, 379, 383, 389, 397, 401, 409, 419,
    421, 431, 433, 439, 443, 449, 457, 461, 463, 467, 479, 487, 491, 499, 503, 509,
    521, 523, 541, 547, 557, 563, 569, 571, 577, 587, 593, 599, 601, 607, 613, 617,
    619, 631, 641, 643, 647, 653, 659, 661, 673, 677, 683, 691, 701, 709, 719, 727,
    733, 739, 743, 751, 757, 761, 769, 773, 787, 797, 809, 811, 821, 823, 827, 829,
    839, 853, 857, 859, 863, 877, 881, 883, 887, 907, 911, 919, 929, 937, 941, 947,
    953, 967, 971, 977, 983, 991, 997, 1009, 1013, 1019, 1021, 1031, 1033, 1039, 1049, 1051,
    1061, 1063, 1069, 1087, 1091, 1093, 1097, 1103, 1109, 1117, 1123, 1129, 1151, 1153, 1163, 1171,
    1181, 1187, 1193, 1201, 1213, 1217, 1223, 1229, 1231, 1237, 1249, 1259, 1277, 1279, 1283, 1289,
    1291, 1297, 1301, 1303, 1307, 1319, 1321, 1327, 1361, 1367, 1373, 1381, 1399, 1409, 1423, 1427,
    1429, 1433, 1439, 1447, 1451, 1453, 1459, 1471, 1481, 1483, 1487, 1489, 1493, 1499, 1511, 1523,
    1531, 1543, 1549, 1553, 1559, 1567, 1571, 1579, 1583, 1597, 1601, 1607, 1609, 1613, 1619, 1621,
};

// sampler/sampling.jl:43-60.  The reference takes digits with floor(a / base) in Float64, which equals the integer quotient
// for every index below 2^53 / base.
TH_D float radical_inverse(int base_index, uint64_t a) {
    if (base_index == 0) return (float)((double)__brevll(a) * 5.4210108624275222e-20);
    const uint32_t base = (uint32_t)kOddPrimes[base_index - 1];
    const float inv_base = 1.0f / (float)base;
    uint64_t reversed = 0;
    float inv_base_n = 1.0f;
    if (a < (1ull << 32)) {  // 32-bit division is several times cheaper
        uint32_t a32 = (uint32_t)a;
        while (a32 > 0) {
            const uint32_t next = a32 / base;
            reversed = reversed * base + (a32 - next * base);
            inv_base_n *= inv_base;
            a32 = next;
        }
    } else {
        while (a > 0) {
            const uint64_t next = a / base;
            reversed = reversed * base + (a - next * base);
            inv_base_n *= inv_base;
            a = next;
        }
    }
    return jmin((float)reversed * inv_base_n, 1.0f);
}

struct LightDistribution {  // Distribution1D over to_Y(power(light)) (sampling.jl:3-31, sppm.jl:564-569), built on the host
    const float* func;      // n
    const float* cdf;       // n + 1
    float func_int;
    int32_t n;
};

struct VisiblePoints {  // per film pixel (y, x) row-major; β == 0 <=> no visible point this iteration
    float4* p_mat;      // p, as_float(material)
    float4* wo;
    float4* beta;
    float4* ng;
    float4* ns;
    float4* ss;
    float4* ts;
};
struct PixelStats {
    float4* Ld;
    float4* tau;
    float* radius;
    double* N;
    float* phi;    // 3 per pixel, atomics
    uint32_t* M;   // atomics
};
struct GridInfo {  // device-resident
    uint32_t enc_min[3], enc_max[3], enc_max_radius;  // order-preserving encodings for atomicMin / atomicMax
    float bmin[3], bmax[3];
    int32_t res[3];
    uint32_t valid;
    uint32_t overflow;  // unused
    uint32_t total;     // in-bounds photon hits of the last iteration
    unsigned long long photon_hits;    // in-bounds photon hits, all iterations of the call
    unsigned long long registrations;  // (visible point, cell) pairs = list nodes of the reference's grid, last iteration
    uint32_t n_hot;                    // pixels deferred to k_sppm_gather_hot
    // option "count_visits" (trhip_stats::count_sub), summed over the iterations of the call: what the gather's byte model is made of
    unsigned long long stat_candidates;     // (pixel, photon) pairs distance-tested
    unsigned long long stat_accepted;       // pairs inside the radius: BSDF evaluated
    unsigned long long stat_visible_points; // pixels with a visible point
};
TH_D uint32_t enc_f32(float f) {
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
TH_D float dec_f32(uint32_t e) { return __uint_as_float((e & 0x80000000u) ? (e & 0x7fffffffu) : ~e); }

// sppm.jl:479-495 with bounds.jl:134-143
TH_D bool to_grid(const GridInfo& g, f3 p, uint32_t out[3]) {
    const f3 o = mk3(p.x - g.bmin[0], p.y - g.bmin[1], p.z - g.bmin[2]);
    const bool g0 = g.bmax[0] > g.bmin[0], g1 = g.bmax[1] > g.bmin[1], g2 = g.bmax[2] > g.bmin[2];
    f3 po = o;
    if (g0 || g1 || g2) po = mk3(o.x / (g0 ? g.bmax[0] - g.bmin[0] : 1.0f), o.y / (g1 ? g.bmax[1] - g.bmin[1] : 1.0f), o.z / (g2 ? g.bmax[2] - g.bmin[2] : 1.0f));
    const float pf[3] = {po.x, po.y, po.z};
    bool in_bounds = true;
    for (int a = 0; a < 3; ++a) {
        const float fl = __builtin_floorf((float)g.res[a] * pf[a]);
        // Int64(floor(...)): clamp in float first (|fl| can exceed the int range far outside the grid)
        const bool inside = fl >= 0.0f && fl < (float)g.res[a];
        if (!inside) in_bounds = false;
        int32_t gi = !(fl >= 0.0f) ? 0 : (fl > (float)(g.res[a] - 1) ? g.res[a] - 1 : (int32_t)fl);
        out[a] = (uint32_t)gi;
    }
    return in_bounds;
}
TH_D uint32_t grid_hash(uint32_t x, uint32_t y, uint32_t z, uint32_t hash_size) {  // sppm.jl:497-501 (0-based)
    const uint64_t h = ((uint64_t)x * 73856093ull) ^ ((uint64_t)y * 19349663ull) ^ ((uint64_t)z * 83492791ull);
    return (uint32_t)(h % (uint64_t)hash_size);
}

// ---- camera pass ----------------------------------------------------------------------------------------------------------------
// Camera rays of a BATCH of iterations: entry e = (iteration it0 + e / n_pix, pixel e % n_pix).  The camera paths of different
// iterations do not depend on each other (only Ld accumulates, in order, see k_sppm_fold_ld), so they share the launches.
template <int TH_ONE_COPY = 0> __global__ __launch_bounds__(kBlock) void k_sppm_raygen(const DeviceSensor* __restrict__ sep, uint32_t n, uint32_t n_pix, uint32_t width, uint64_t seed, uint32_t it0, PathQueue q,
                                                        uint32_t cap, Counters* ctr) {
    const DeviceSensor& se = *sep;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const uint32_t it_local = i / n_pix, pix = i - it_local * n_pix;
        const int py = 1 + (int)(pix / width), px = 1 + (int)(pix - (pix / width) * width);  // crop_bounds start at (1, 1)
        const uint64_t key = ts_stream_key(seed, px, py, it0 + it_local - 1u);
        const f2 film{(float)px + ts_uniform(key, TS_DIM_FILM_X), (float)py + ts_uniform(key, TS_DIM_FILM_Y)};
        const f2 lens{ts_uniform(key, TS_DIM_LENS_X), ts_uniform(key, TS_DIM_LENS_Y)};
        f3 o, d;
        float time;
        generate_ray(se, film, lens, ts_uniform(key, TS_DIM_TIME), o, d, time);
        d = check_direction(d);
        const uint32_t w = i >> 6;
        const uint32_t phys = (w % kSeg) * cap + (w / kSeg) * 64u + (i & 63u);
        q.o[phys] = make_float4(o.x, o.y, o.z, __uint_as_float(i));
        q.d[phys] = make_float4(d.x, d.y, d.z, __uint_as_float(0u));  // .w: specular_bounce
        q.beta[phys] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
    }
    if (blockIdx.x == 0 && threadIdx.x < kSeg) {
        const uint32_t sgm = threadIdx.x, W = (n + 63u) >> 6;
        uint32_t cnt = ((W + kSeg - 1 - sgm) / kSeg) * 64u;
        if (W > 0 && (W - 1) % kSeg == sgm && (n & 63u)) cnt -= 64u - (n & 63u);
        ctr->n_queue[0][sgm * kCtrStride] = cnt;
    }
}

TH_D void add_nan_where(float4* L, uint32_t slot, uint32_t poison) {  // L += β · 0 with a non-finite β
    float4 l = L[slot];
    const float nanv = __builtin_nanf("");
    if (poison & 1u) l.x += nanv;
    if (poison & 2u) l.y += nanv;
    if (poison & 4u) l.z += nanv;
    L[slot] = l;
}

// One level of the camera pass (sppm.jl:208-266) for a batch of iterations.  slot = it_local * n_pix + pixel indexes the
// visible points of the batch; what the reference adds to pixel.Ld at this depth goes to the term slot
// ((it_local * max_depth + depth - 1) * n_pix + pixel) of a zeroed buffer — written by at most one path, so the `+=` of the
// shadow kernel is race-free — and k_sppm_fold_ld adds the terms to Ld in the reference's order afterwards.
#ifndef TH_SHADE_SPPM_WAVES
#define TH_SHADE_SPPM_WAVES 3  // 180 VGPRs unconstrained (2 waves per SIMD); capped at 3: C4 shading section 162.9 -> 159.3 ms, at 4 (spills) 162.2
#endif
template <bool TAN = true>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(TH_SHADE_SPPM_WAVES))) void k_shade_sppm(DeviceScene sc, PathQueue qin, PathQueue qout, ShadowQueue sq, uint32_t cap, const float4* __restrict__ hits, VisiblePoints vp,
                                                       float4* __restrict__ Ld, Counters* ctr, int depth, int max_depth, uint64_t seed, uint32_t it0, uint32_t n_pix, uint32_t width) {
    __shared__ SegView sv;
    const SegQueue qv{ctr->n_queue[depth - 1], cap, 0u};
    seg_load(qv, sv);
    const uint32_t total = sv.prefix[kSeg];
    uint32_t seg_in = 0;  // (carried over the iterations: seg_locate_from)
    for (uint32_t flat = blockIdx.x * kBlock + threadIdx.x; flat < total; flat += gridDim.x * kBlock) {
        uint32_t lb;
        seg_locate_from(sv, flat & ~63u, seg_in, lb);
        const uint32_t local = lb + (flat & 63u);
        const bool valid = local < sv.count[seg_in];
        const uint32_t i = seg_in * cap + local;
        const uint32_t seg_out = (flat >> 6) % kSeg;
        bool want_shadow = false, want_next = false;
        float4 so4, sd4, sc4, no4, nd4, nb4;
        if (valid) {
            const float4 h4 = hits[i];
            const int prim = __float_as_int(h4.y);
            const float4 o4 = qin.o[i], d4 = qin.d[i], b4 = qin.beta[i];
            const uint32_t slot = __float_as_uint(o4.w);
            const uint32_t it_local = slot / n_pix, pix = slot - it_local * n_pix;
            const uint32_t term = (it_local * (uint32_t)max_depth + (uint32_t)(depth - 1)) * n_pix + pix;
            f3 beta = mk3(b4.x, b4.y, b4.z);
            const uint32_t poison = ((isnan_(beta.x) || isinf_(beta.x)) ? 1u : 0u) | ((isnan_(beta.y) || isinf_(beta.y)) ? 2u : 0u) | ((isnan_(beta.z) || isinf_(beta.z)) ? 4u : 0u);
            if (prim < 0) {
                if (poison && sc.n_lights > 0) add_nan_where(Ld, term, poison);  // Ld += β * le(light, ray) = β * 0 (:211-216)
            } else {
                const f3 o = mk3(o4.x, o4.y, o4.z), d = mk3(d4.x, d4.y, d4.z);
                const bool specular_bounce = __float_as_uint(d4.w) != 0u;
                Shading sh;
                uint32_t material;
                if (rebuild_shading<false, TAN>(sc, prim, o, d, sh, material) && material != PRIM_NO_MATERIAL) {
                    const LobeSet& bsdf = sc.materials[material].set[1];
                    const int py = 1 + (int)(pix / width), px = 1 + (int)(pix - (pix / width) * width);
                    const uint64_t key = ts_stream_key(seed, px, py, it0 + it_local - 1u);
                    const uint32_t v = (uint32_t)(depth - 1);
                    const f3 wo = -d;
                    if ((depth == 1 || specular_bounce) && poison) add_nan_where(Ld, term, poison);  // Ld += β * le(si, wo) = β * 0 (:227-229)
                    // uniform_sample_one_light, not weighted by β (:230-232, A.12)
                    if (sc.n_lights > 0) {
                        const int nl = (int)sc.n_lights;
                        int ln = (int)__builtin_ceilf(ts_uniform(key, ts_vertex_dim(v, TS_V_LIGHT_PICK)) * (float)nl);
                        if (ln > nl) ln = nl;
                        if (ln < 1) ln = 1;
                        const float light_pdf = 1.0f / (float)nl;
                        const LightRec& light = sc.lights[ln - 1];
                        const LightSample ls = sample_li(light, sh.p);
                        if (ls.pdf > 0.0f && !is_black(ls.radiance)) {
                            const f3 f = bsdf_f(bsdf, sh, sh.wo, ls.wi, BSDF_ALL & ~BSDF_SPECULAR) * fabs_(dot(ls.wi, sh.ns));
                            if (!is_black(f)) {
                                const f3 c = (splat3(0.0f) + f * ls.radiance / ls.pdf) / light_pdf;
                                const f3 lp = mk3(light.position[0], light.position[1], light.position[2]);
                                const f3 dir = lp - sh.p;
                                const f3 org = sh.p + 1e-6f * dir;
                                const f3 cd = check_direction(dir);
                                so4 = make_float4(org.x, org.y, org.z, __uint_as_float(term));
                                sd4 = make_float4(cd.x, cd.y, cd.z, __uint_as_float(0u));
                                sc4 = make_float4(c.x, c.y, c.z, 0.0f);
                                want_shadow = true;
                            }
                        }
                    }
                    const bool is_diffuse = bsdf_num_components(bsdf, BSDF_DIFFUSE | BSDF_REFLECTION | BSDF_TRANSMISSION) > 0;
                    const bool is_glossy = bsdf_num_components(bsdf, BSDF_GLOSSY | BSDF_REFLECTION | BSDF_TRANSMISSION) > 0;
                    if (is_diffuse || (is_glossy && depth == max_depth)) {  // :239-245
                        vp.p_mat[slot] = make_float4(sh.p.x, sh.p.y, sh.p.z, __uint_as_float(material));
                        vp.wo[slot] = make_float4(wo.x, wo.y, wo.z, 0.0f);
                        vp.beta[slot] = make_float4(beta.x, beta.y, beta.z, 0.0f);
                        vp.ng[slot] = make_float4(sh.ng.x, sh.ng.y, sh.ng.z, 0.0f);
                        vp.ns[slot] = make_float4(sh.ns.x, sh.ns.y, sh.ns.z, 0.0f);
                        vp.ss[slot] = make_float4(sh.ss.x, sh.ss.y, sh.ss.z, 0.0f);
                        vp.ts[slot] = make_float4(sh.ts.x, sh.ts.y, sh.ts.z, 0.0f);
                    } else if (depth < max_depth) {
                        const f2 u{ts_uniform(key, ts_vertex_dim(v, TS_V_BSDF_U0)), ts_uniform(key, ts_vertex_dim(v, TS_V_BSDF_U1))};
                        const BsdfSample bs = bsdf_sample_f(bsdf, sh, wo, u, BSDF_ALL);
                        if (!(bs.pdf == 0.0f || is_black(bs.f))) {
                            beta = beta * (bs.f * fabs_(dot(bs.wi, sh.ns)) / bs.pdf);
                            const float by = to_Y(beta);
                            bool alive = true;
                            if (by < 0.25f) {
                                const float cont = jmin(1.0f, by);
                                if (ts_uniform(key, ts_vertex_dim(v, TS_V_RR)) > cont)
                                    alive = false;
                                else
                                    beta = beta / cont;
                            }
                            if (alive) {
                                const f3 org = sh.p + 1e-6f * bs.wi;
                                const f3 nd = check_direction(bs.wi);
                                no4 = make_float4(org.x, org.y, org.z, __uint_as_float(slot));
                                nd4 = make_float4(nd.x, nd.y, nd.z, __uint_as_float((bs.sampled_type & BSDF_SPECULAR) != 0 ? 1u : 0u));
                                nb4 = make_float4(beta.x, beta.y, beta.z, 0.0f);
                                want_next = true;
                            }
                        }
                    }
                }
            }
        }
        const uint32_t si = seg_out * cap + wave_compact(want_shadow, &ctr->n_shadow[depth - 1][seg_out * kCtrStride]);
        if (want_shadow) {
            sq.o[si] = so4;
            sq.d[si] = sd4;
            sq.c[si] = sc4;
        }
        const uint32_t ni = seg_out * cap + wave_compact(want_next, &ctr->n_queue[depth][seg_out * kCtrStride]);
        if (want_next) {
            qout.o[ni] = no4;
            qout.d[ni] = nd4;
            qout.beta[ni] = nb4;
        }
    }
}

// ---- grid (sppm.jl:278-318) --------------------------------------------------------------------------------------------------------
TH_D float wave_min(float v) {
    for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off));
    return v;
}
TH_D float wave_max(float v) {
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}
template <int TH_ONE_COPY = 0> __global__ __launch_bounds__(kBlock) void k_sppm_grid_reset(GridInfo* g) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        for (int a = 0; a < 3; ++a) {
            g->enc_min[a] = enc_f32(kInf);
            g->enc_max[a] = enc_f32(-kInf);
        }
        g->enc_max_radius = enc_f32(0.0f);
        g->valid = 0;
        g->total = 0;  // photon_hits is per render call
        g->registrations = 0;
        g->n_hot = 0;
    }
}
// grid_bounds = ∪ expand(Bounds3(vp.p), radius), max_radius (:285-292)
template <int TH_ONE_COPY = 0> __global__ __launch_bounds__(kBlock) void k_sppm_grid_bounds(VisiblePoints vp, const float* __restrict__ radius, uint32_t n, GridInfo* g) {
    float mn[3] = {kInf, kInf, kInf}, mx[3] = {-kInf, -kInf, -kInf}, mr = 0.0f;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const float4 b = vp.beta[i];
        if (b.x == 0.0f && b.y == 0.0f && b.z == 0.0f) continue;
        const float4 p = vp.p_mat[i];
        const float r = radius[i];
        const float pp[3] = {p.x, p.y, p.z};
        for (int a = 0; a < 3; ++a) {
            mn[a] = fminf(mn[a], pp[a] - r);
            mx[a] = fmaxf(mx[a], pp[a] + r);
        }
        mr = fmaxf(mr, r);
    }
    for (int a = 0; a < 3; ++a) {
        mn[a] = wave_min(mn[a]);
        mx[a] = wave_max(mx[a]);
    }
    mr = wave_max(mr);
    // one set of atomics per block (seven words shared by the whole grid: one set per wave was 80 µs of the launch's 97)
    __shared__ float s_red[kBlock / 64][7];
    const uint32_t wv = threadIdx.x >> 6;
    if (lane_id() == 0) {
        for (int a = 0; a < 3; ++a) {
            s_red[wv][a] = mn[a];
            s_red[wv][3 + a] = mx[a];
        }
        s_red[wv][6] = mr;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (uint32_t w = 1; w < kBlock / 64; ++w) {
            for (int a = 0; a < 3; ++a) {
                s_red[0][a] = fminf(s_red[0][a], s_red[w][a]);
                s_red[0][3 + a] = fmaxf(s_red[0][3 + a], s_red[w][3 + a]);
            }
            s_red[0][6] = fmaxf(s_red[0][6], s_red[w][6]);
        }
        if (s_red[0][6] > 0.0f) {
            for (int a = 0; a < 3; ++a) {
                atomicMin(&g->enc_min[a], enc_f32(s_red[0][a]));
                atomicMax(&g->enc_max[a], enc_f32(s_red[0][3 + a]));
            }
            atomicMax(&g->enc_max_radius, enc_f32(s_red[0][6]));
        }
    }
}
// grid resolution (:293-302)
template <int TH_ONE_COPY = 0> __global__ void k_sppm_grid_setup(GridInfo* g) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const float max_radius = dec_f32(g->enc_max_radius);
    if (!(max_radius > 0.0f)) {
        g->valid = 0;
        g->res[0] = g->res[1] = g->res[2] = 1;
        return;
    }
    float diag[3];
    for (int a = 0; a < 3; ++a) {
        g->bmin[a] = dec_f32(g->enc_min[a]);
        g->bmax[a] = dec_f32(g->enc_max[a]);
        diag[a] = g->bmax[a] - g->bmin[a];
    }
    const float max_diag = jmax(jmax(diag[0], diag[1]), diag[2]);
    const float base_res = __builtin_floorf(max_diag / max_radius);  // Int64(floor(...)): exact in Float32 below 2^24
    for (int a = 0; a < 3; ++a) {
        const float r = __builtin_floorf(base_res * diag[a] / max_diag);
        g->res[a] = r > 1.0f ? (int32_t)r : 1;
    }
    g->valid = 1;
}
// The reference hangs every visible point into the hash buckets of the cells its sphere overlaps and lets each photon walk
// the bucket of its own cell, adding to the pixels with atomics (:303-318, :366-391).  The same set of (registration, photon
// hit) pairs — equal bucket, distance <= radius; a visible point registered in two cells that hash alike is credited twice,
// as in the reference — is enumerated here from the other side: the photon hits of the iteration are counting-sorted by the
// hash of their cell, and every pixel walks the buckets of the cells it would have registered in.  One thread owns a pixel:
// ϕ and M are plain sums in registers, no atomics, no contention under a caustic.
struct PhotonRecords;
// pass = 0: count the in-bounds photon hits per bucket; pass = 1: fill (counts[h] counts down from the bucket size).
// Record slots of one iteration: (d, first_photon + i), d < n_depths, i < n_photons.
template <int TH_ONE_COPY = 0> __global__ __launch_bounds__(kBlock) void k_sppm_hit_bin(const float4* __restrict__ rec_p, const uint8_t* __restrict__ rec_valid, uint32_t n_batch_photons, uint32_t first_photon,
                                                         uint32_t n_photons, uint32_t n_depths, uint32_t hash_size, GridInfo* gp, uint32_t* __restrict__ counts,
                                                         const uint32_t* __restrict__ starts, float4* __restrict__ hit_sorted, int pass) {
    const GridInfo& g = *gp;
    if (!g.valid) return;
    const uint32_t total = n_photons * n_depths;
    for (uint32_t k = blockIdx.x * kBlock + threadIdx.x; k < total; k += gridDim.x * kBlock) {
        const uint32_t d = k / n_photons, i = k - d * n_photons;
        const uint32_t r = d * n_batch_photons + first_photon + i;
        if (!rec_valid[r]) continue;
        const float4 hp = rec_p[r];
        uint32_t gi[3];
        if (!to_grid(g, mk3(hp.x, hp.y, hp.z), gi)) continue;  // `in_bounds` (:370-373)
        const uint32_t h = grid_hash(gi[0], gi[1], gi[2], hash_size);
        if (pass == 0) {
            atomicAdd(&counts[h], 1u);  // the call's hit total is kept by k_sppm_scan (a counter per wave here was 8 192 atomics on one word: 90 µs of the launch's 130)
        } else {
            // the position travels with the record index: the gather rejects most candidates on the distance alone, without a second,
            // dependent fetch through the index
            hit_sorted[starts[h] + atomicSub(&counts[h], 1u) - 1u] = make_float4(hp.x, hp.y, hp.z, __uint_as_float(r));
        }
    }
}
// Exclusive prefix sums of the bucket sizes in three launches: tiles of kScanTile counts (local prefix + tile total), the
// tile totals (k_sppm_scan, one block), then the tile offsets added back.
constexpr uint32_t kScanTile = 1024;
template <int TH_ONE_COPY = 0> __global__ __launch_bounds__(kBlock) void k_sppm_scan_tiles(const uint32_t* __restrict__ counts, uint32_t* __restrict__ starts, uint32_t n, uint32_t* __restrict__ tile_sums) {
    __shared__ uint32_t wsum[kBlock / 64];
    const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * 4u;
    uint32_t v[4];
    for (int k = 0; k < 4; ++k) v[k] = base + k < n ? counts[base + k] : 0u;
    const uint32_t mine = v[0] + v[1] + v[2] + v[3];
    uint32_t incl = mine;  // inclusive scan across the wave
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(incl, off);
        if ((int)lane_id() >= off) incl += up;
    }
    const uint32_t w = threadIdx.x >> 6;
    if (lane_id() == 63) wsum[w] = incl;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t k = 0; k < w; ++k) before += wsum[k];
    uint32_t acc = before + incl - mine;
    for (int k = 0; k < 4; ++k) {
        if (base + k < n) starts[base + k] = acc;
        acc += v[k];
    }
    if (threadIdx.x == kBlock - 1) tile_sums[blockIdx.x] = acc;
}
template <int TH_ONE_COPY = 0> __global__ __launch_bounds__(kBlock) void k_sppm_scan_add(uint32_t* __restrict__ starts, uint32_t n, const uint32_t* __restrict__ tile_offsets, uint32_t n_tiles) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i <= n; i += gridDim.x * kBlock) {
        if (i == n)
            starts[n] = tile_offsets[n_tiles];
        else
            starts[i] += tile_offsets[i / kScanTile];
    }
}
// starts[0..n] = exclusive prefix sums of counts[0..n-1]; one block of 1024 threads.
template <int TH_ONE_COPY = 0> __global__ __launch_bounds__(1024) void k_sppm_scan(const uint32_t* __restrict__ counts, uint32_t* __restrict__ starts, uint32_t n, GridInfo* g) {
    __shared__ uint32_t part[1024];
    const uint32_t t = threadIdx.x;
    const uint32_t chunk = (n + 1023u) / 1024u;
    const uint32_t b = min(n, t * chunk), e = min(n, b + chunk);
    uint32_t s = 0;
    for (uint32_t i = b; i < e; ++i) s += counts[i];
    part[t] = s;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan
        const uint32_t v = t >= off ? part[t - off] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t acc = t ? part[t - 1] : 0u;
    for (uint32_t i = b; i < e; ++i) {
        starts[i] = acc;
        acc += counts[i];
    }
    if (t == 1023) {
        starts[n] = part[1023];
        g->total = part[1023];
        g->photon_hits += part[1023];  // in-bounds photon hits, all iterations of the call
    }
}

// ---- photon pass (sppm.jl:320-436) ----------------------------------------------------------------------------------------------
// Photon ray leaving the light: sample_discrete over light power, sample_le (point.jl:60-69, spot.jl:46-55), β.
// per_iter / p_lo / p_hi: a multi-GPU job (trhip_comm_init) traces photon indices [p_lo, p_hi) of every iteration on this rank
// (Threads.@threads over photon_index, sppm.jl:334, spread over processes); the index space and the records keep the full layout.
template <int TH_ONE_COPY = 0> __global__ __launch_bounds__(kBlock) void k_photon_gen(DeviceScene sc, LightDistribution ld, uint32_t n_photons, uint64_t halton_base, PathQueue q, uint32_t cap, Counters* ctr,
                                                       uint32_t per_iter, uint32_t p_lo, uint32_t p_hi) {
    const uint32_t total = (n_photons + 63u) & ~63u;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < total; i += gridDim.x * kBlock) {
        bool want = false;
        float4 o4, d4, b4;
        const uint32_t p_in_iter = per_iter ? i % per_iter : 0u;
        if (i < n_photons && p_in_iter >= p_lo && p_in_iter < p_hi) {
            const uint64_t hi = halton_base + i;
            const float light_sample = radical_inverse(0, hi);
            int offset = 0;  // findlast(cdf[i] ≤ u), 1-based
            for (int k = ld.n + 1; k >= 1; --k)
                if (ld.cdf[k - 1] <= light_sample) {
                    offset = k;
                    break;
                }
            offset = offset < 1 ? 1 : (offset > ld.n ? ld.n : offset);
            const float light_pdf = ld.func_int > 0.0f ? ld.func[offset - 1] / (ld.func_int * (float)ld.n) : 0.0f;
            const LightRec& l = sc.lights[offset - 1];
            const f2 u{radical_inverse(1, hi), radical_inverse(2, hi)};
            const f3 I = mk3(l.I[0], l.I[1], l.I[2]);
            f3 dir, le;
            float pdf_dir;
            if (l.kind == 0) {  // uniform_sample_sphere Trace.jl:69-74
                const float z = 1.0f - 2.0f * u.x;
                const float r = sqrt_(jmax(0.0f, 1.0f - z * z));
                const float phi = 2.0f * kPi * u.y;
                dir = mk3(r * tm_cosf(phi), r * tm_sinf(phi), z);
                pdf_dir = 1.0f / (4.0f * kPi);
                le = I;
            } else {  // uniform_sample_cone Trace.jl:76-81, light_to_world on the vector
                const float c = 1.0f - u.x + u.x * l.cos_total_width;
                const float s = sqrt_(1.0f - c * c);
                const float phi = u.y * 2.0f * kPi;
                const f3 w = mk3(tm_cosf(phi) * s, tm_sinf(phi) * s, c);
                dir = mk3(l.l2w[0] * w.x + l.l2w[1] * w.y + l.l2w[2] * w.z, l.l2w[3] * w.x + l.l2w[4] * w.y + l.l2w[5] * w.z, l.l2w[6] * w.x + l.l2w[7] * w.y + l.l2w[8] * w.z);
                pdf_dir = 1.0f / (2.0f * kPi * (1.0f - l.cos_total_width));
                const f3 wl = normalize(mk3(l.w2l[0] * dir.x + l.w2l[1] * dir.y + l.w2l[2] * dir.z, l.w2l[3] * dir.x + l.w2l[4] * dir.y + l.w2l[5] * dir.z,
                                            l.w2l[6] * dir.x + l.w2l[7] * dir.y + l.w2l[8] * dir.z));
                const float ct = wl.z;
                float fall;
                if (ct < l.cos_total_width)
                    fall = 0.0f;
                else if (ct >= l.cos_falloff_start)
                    fall = 1.0f;
                else
                    fall = pow4((ct - l.cos_total_width) / (l.cos_falloff_start - l.cos_total_width));
                le = I * fall;
            }
            const float pdf_pos = 1.0f;
            if (!(pdf_dir == 0.0f || is_black(le))) {
                const f3 beta = fabs_(dot(dir, dir)) * le / (light_pdf * pdf_pos * pdf_dir);  // light_normal = ray.d
                if (!is_black(beta)) {
                    const f3 cd = check_direction(dir);
                    o4 = make_float4(l.position[0], l.position[1], l.position[2], __uint_as_float(i));
                    d4 = make_float4(cd.x, cd.y, cd.z, 0.0f);
                    b4 = make_float4(beta.x, beta.y, beta.z, 0.0f);
                    want = true;
                }
            }
        }
        const uint32_t seg_out = (i >> 6) % kSeg;
        const uint32_t k = seg_out * cap + wave_compact(want, &ctr->n_queue[0][seg_out * kCtrStride]);
        if (want) {
            q.o[k] = o4;
            q.d[k] = d4;
            q.beta[k] = b4;
        }
    }
}

// Photon hits of a batch of iterations.  Photon paths do not depend on the pixels (β is never updated, the grid is only
// read when depositing), so the photons of several iterations are traced together and every hit at depth >= 2 is recorded
// in the slot ((depth - 2) * n_batch_photons + photon): no counter, no atomics.  k_sppm_deposit replays the records of one
// iteration against that iteration's grid.
struct PhotonRecords {
    float4* p;       // hit point, unused
    float4* wi;      // -photon_ray.d, unused
    float4* beta;    // emission weight β
    uint8_t* valid;  // zeroed per batch
};

// One photon bounce: record the hit at depth > 1 (:366-391 happens in k_sppm_deposit), then sample the next direction and play
// Russian roulette (:393-418).
// Consecutive photons (Halton indices) land anywhere: taken 64 at a time in queue order, a wave mixes glass, plastic and matte hits and
// runs each material's lobe code for a third of its lanes (PMC: 21 of 64 lanes per VALU instruction).  So a wave only CLASSIFIES its
// entries as they come (material index mod kPhotonRings, from the hit slot's record) and parks their indices in per-class rings in
// LDS; a class is shaded whenever 64 of its entries wait, all lanes in the same lobe code (the trick of k_shade_path).  Every photon
// is processed by exactly the code it was processed by before — a photon's record slot and Halton dimensions depend on nothing
// but its index and depth — so the results do not change.
constexpr int kPhotonRings = 4;
#ifndef TH_SHADE_PHOTON_WAVES
#define TH_SHADE_PHOTON_WAVES 4
#endif
template <bool TAN = true>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(TH_SHADE_PHOTON_WAVES))) void k_shade_photon(DeviceScene sc, PathQueue qin, PathQueue qout, uint32_t cap, const float4* __restrict__ hits, PhotonRecords rec,
                                                         uint32_t n_batch_photons, Counters* ctr, int depth, int max_depth, uint64_t halton_base) {
    __shared__ SegView sv;
    __shared__ uint32_t s_ring[kBlock / 64][kPhotonRings][128];
    const SegQueue qv{ctr->n_queue[depth - 1], cap, 0u};
    seg_load(qv, sv);
    const uint32_t total = sv.prefix[kSeg];
    const uint32_t lane = lane_id(), wv = threadIdx.x >> 6;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const uint32_t seg_out = ((blockIdx.x * kBlock + threadIdx.x) >> 6) % kSeg;  // every wave-iteration of a wave feeds the same output segment (k_shade_path)
    uint32_t ring_head[kPhotonRings], ring_cnt[kPhotonRings];  // wave-uniform
    for (int c = 0; c < kPhotonRings; ++c) ring_head[c] = ring_cnt[c] = 0u;
    // one photon-path vertex for queue entry i (a real hit), if `on`; called by the whole wave
    auto vertex = [&](bool on, uint32_t i) {
        bool want_next = false;
        float4 no4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), nd4 = no4, nb4 = no4;
        if (on) {
            const float4 h4 = hits[i];
            const int prim = __float_as_int(h4.y);
            const float4 o4 = qin.o[i], d4 = qin.d[i], b4 = qin.beta[i];
            const uint32_t photon = __float_as_uint(o4.w);
            const f3 o = mk3(o4.x, o4.y, o4.z), d = mk3(d4.x, d4.y, d4.z);
            const f3 beta = mk3(b4.x, b4.y, b4.z);
            Shading sh;
            uint32_t material;
            if (rebuild_shading<false, TAN>(sc, prim, o, d, sh, material) && material != PRIM_NO_MATERIAL) {
                const f3 wi_photon = -d;
                if (depth > 1) {
                    const size_t r = (size_t)(depth - 2) * n_batch_photons + photon;
                    rec.p[r] = make_float4(sh.p.x, sh.p.y, sh.p.z, 0.0f);
                    rec.wi[r] = make_float4(wi_photon.x, wi_photon.y, wi_photon.z, 0.0f);
                    rec.beta[r] = b4;
                    rec.valid[r] = 1;
                }
                const LobeSet& bsdf = sc.materials[material].set[1];  // compute_scattering!(…, true, Importance): the mode changes nothing (A.11)
                const uint64_t hidx = halton_base + photon;
                const int dim = 6 + 3 * (depth - 1);
                const f2 u{radical_inverse(dim, hidx), radical_inverse(dim + 1, hidx)};
                const BsdfSample bs = bsdf_sample_f(bsdf, sh, wi_photon, u, BSDF_ALL);
                if (!(is_black(bs.f) || bs.pdf == 0.0f) && depth < max_depth) {
                    const f3 beta_new = beta * bs.f * fabs_(dot(bs.wi, sh.ns)) / bs.pdf;
                    const float q = jmax(0.0f, 1.0f - to_Y(beta_new) / to_Y(beta));
                    if (!(radical_inverse(dim + 2, hidx) < q)) {
                        const f3 org = sh.p + 1e-6f * bs.wi;
                        const f3 nd = check_direction(bs.wi);
                        no4 = make_float4(org.x, org.y, org.z, __uint_as_float(photon));
                        nd4 = make_float4(nd.x, nd.y, nd.z, 0.0f);
                        nb4 = b4;
                        want_next = true;
                    }
                }
            }
        }
        const uint32_t ni = seg_out * cap + wave_compact(want_next, &ctr->n_queue[depth][seg_out * kCtrStride]);
        if (want_next) {
            qout.o[ni] = no4;
            qout.d[ni] = nd4;
            qout.beta[ni] = nb4;
        }
    };
    uint32_t seg_in = 0;  // (carried over the iterations: seg_locate_from)
    for (uint32_t flat = blockIdx.x * kBlock + threadIdx.x; flat < total; flat += gridDim.x * kBlock) {
        uint32_t lb;
        seg_locate_from(sv, flat & ~63u, seg_in, lb);
        const uint32_t local = lb + (flat & 63u);
        const uint32_t i = seg_in * cap + local;
        int cls = -1;  // -1: nothing to do (padding, miss)
        if (local < sv.count[seg_in]) {
            const int prim = __float_as_int(hits[i].y);
            if (prim >= 0) cls = (int)((__float_as_uint(sc.shade[8 * (size_t)prim].w) & PRIM_MATERIAL_MASK) % (uint32_t)kPhotonRings);
        }
        auto park = [&](auto cc) {  // class cc (a compile-time constant: the ring counters stay in registers)
            constexpr int c = decltype(cc)::value;
            const unsigned long long m = __ballot(cls == c);
            if (m == 0ull) return;
            if (cls == c) s_ring[wv][c][(ring_head[c] + ring_cnt[c] + (uint32_t)__popcll(m & lt_mask)) & 127u] = i;
            ring_cnt[c] += (uint32_t)__popcll(m);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (ring_cnt[c] >= 64u) {
                vertex(true, s_ring[wv][c][(ring_head[c] + lane) & 127u]);
                ring_head[c] = (ring_head[c] + 64u) & 127u;
                ring_cnt[c] -= 64u;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
        };
        static_assert(kPhotonRings == 4, "one call per class");
        park(std::integral_constant<int, 0>{});
        park(std::integral_constant<int, 1>{});
        park(std::integral_constant<int, 2>{});
        park(std::integral_constant<int, 3>{});
    }
    auto flush = [&](auto cc) {  // the last, partial batches
        constexpr int c = decltype(cc)::value;
        if (ring_cnt[c]) vertex(lane < ring_cnt[c], s_ring[wv][c][(ring_head[c] + lane) & 127u]);
    };
    flush(std::integral_constant<int, 0>{});
    flush(std::integral_constant<int, 1>{});
    flush(std::integral_constant<int, 2>{});
    flush(std::integral_constant<int, 3>{});
}

// `pixel.ϕ += β · f_vp(wo_vp, wi)`, `pixel.M += 1` (sppm.jl:374-391) for one iteration, gathered per pixel: for every cell the
// visible point registers in (:306-317) walk the bucket of photon hits with that hash and take those within the radius.
// A caustic puts thousands of photons into the radius of a few thousand pixels (and a handful into the rest): pixels with
// more than kHotCandidates candidates are deferred to k_sppm_gather_hot, where a whole wave shares one pixel's buckets.
#ifndef TH_SPPM_HOT
#define TH_SPPM_HOT 192
#endif
constexpr uint32_t kHotCandidates = TH_SPPM_HOT;
struct GatherSum {
    f3 phi;
    uint32_t M;
};
// Candidates e0 + first, e0 + first + stride, … of every bucket the visible point registers in.
TH_D GatherSum gather_pixel(const DeviceScene& sc, const PhotonRecords& rec, const VisiblePoints& vp, uint32_t i, float4 p4, float rad, const uint32_t lo[3], const uint32_t hi[3],
                            const uint32_t* __restrict__ starts, const float4* __restrict__ hit_sorted, uint32_t hash_size, uint32_t first, uint32_t stride) {
    const f3 vpp = mk3(p4.x, p4.y, p4.z);
    Shading vs;
    bool have_frame = false;
    GatherSum s{splat3(0.0f), 0u};
    for (uint32_t z = lo[2]; z <= hi[2]; ++z)
        for (uint32_t y = lo[1]; y <= hi[1]; ++y)
            for (uint32_t x = lo[0]; x <= hi[0]; ++x) {
                const uint32_t h = grid_hash(x, y, z, hash_size);
                const uint32_t e0 = starts[h], e1 = starts[h + 1];
                for (uint32_t e = e0 + first; e < e1; e += stride) {
                    const float4 hp = hit_sorted[e];
                    const uint32_t r = __float_as_uint(hp.w);
                    const f3 dv = vpp - mk3(hp.x, hp.y, hp.z);  // distance_squared(vp.p, p)
                    if (dot(dv, dv) > rad * rad) continue;
                    if (!have_frame) {
                        const float4 wo4 = vp.wo[i], ng4 = vp.ng[i], ns4 = vp.ns[i], ss4 = vp.ss[i], ts4 = vp.ts[i];
                        vs.p = vpp;
                        vs.wo = mk3(wo4.x, wo4.y, wo4.z);
                        vs.ng = mk3(ng4.x, ng4.y, ng4.z);
                        vs.ns = mk3(ns4.x, ns4.y, ns4.z);
                        vs.ss = mk3(ss4.x, ss4.y, ss4.z);
                        vs.ts = mk3(ts4.x, ts4.y, ts4.z);
                        have_frame = true;
                    }
                    const float4 w4 = rec.wi[r], b4 = rec.beta[r];
                    const LobeSet& vb = sc.materials[__float_as_uint(p4.w)].set[1];
                    s.phi = s.phi + mk3(b4.x, b4.y, b4.z) * bsdf_f(vb, vs, vs.wo, mk3(w4.x, w4.y, w4.z), BSDF_ALL);
                    s.M++;
                }
            }
    return s;
}
// One thread per pixel; also counts the registrations (the reference's list nodes) for trhip_sppm_state.
#ifndef TH_SPPM_GATHER_WAVES
#define TH_SPPM_GATHER_WAVES 4
#endif
#ifndef TH_SPPM_HOT_WAVES
#define TH_SPPM_HOT_WAVES 4  // 144 VGPRs unconstrained (3 waves); capped at 4: C4 shading section 160.4 -> 156.4 ms
#endif
template <int TH_ONE_COPY = 0> __global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(TH_SPPM_GATHER_WAVES))) void k_sppm_gather(DeviceScene sc, PhotonRecords rec, VisiblePoints vp, PixelStats px, uint32_t n, GridInfo* gp, const uint32_t* __restrict__ starts,
                                                        const float4* __restrict__ hit_sorted, uint32_t hash_size, uint32_t* __restrict__ hot_list, uint32_t count_registrations, uint32_t count_stats) {
    // A wave takes 64 pixels.  Every lane walks the buckets of ITS pixel one candidate per round (a cursor over cells and entries);
    // the (pixel, photon) pairs that pass the distance test are parked in a per-wave ring and evaluated 64 at a time — frame and
    // material of the pair's pixel are fetched by whichever lane gets the pair — and summed per pixel in LDS.  (One thread per pixel
    // evaluating its own pairs in place ran the BSDF with 13 of 64 lanes.)  Pixels with more than kHotCandidates candidates go to
    // k_sppm_gather_hot as before.  Also counts the registrations (the reference's list nodes) for trhip_sppm_state.
    __shared__ uint32_t s_ring[kBlock / 64][128];   // photon record of a parked pair
    __shared__ uint32_t s_src[kBlock / 64][128];    // … and the lane whose pixel it belongs to
    __shared__ float s_phi[kBlock / 64][64][3];
    __shared__ uint32_t s_m[kBlock / 64][64];
    // the wave's 64 visible points as the BSDF needs them, staged once per batch of pixels: frame (ss, ts, ns, ng: 12 floats), what bsdf_f derives from wo
    // alone (BsdfWo: 6 floats, th_device.h) and the material — [word][lane], so that a lane shading a parked pair reads its pixel's column from LDS instead of
    // re-fetching six 16-byte records per pair
    __shared__ float s_vp[kBlock / 64][19][64];
    const GridInfo& g = *gp;
    if (!g.valid) return;
    const uint32_t lane = lane_id(), wv = threadIdx.x >> 6;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    unsigned long long n_reg = 0, n_cand = 0, n_acc = 0, n_vp = 0;
    const uint32_t total = (n + 63u) & ~63u;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < total; i += gridDim.x * kBlock) {
        bool hot = false, walk = false;
        float4 p4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        float rad = 0.0f;
        uint32_t lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
        if (i < n) {
            const float4 b = vp.beta[i];
            if (!(b.x == 0.0f && b.y == 0.0f && b.z == 0.0f)) {
                n_vp++;
                p4 = vp.p_mat[i];
                rad = px.radius[i];
                to_grid(g, mk3(p4.x - rad, p4.y - rad, p4.z - rad), lo);
                to_grid(g, mk3(p4.x + rad, p4.y + rad, p4.z + rad), hi);
                uint32_t candidates = 0;
                for (uint32_t z = lo[2]; z <= hi[2]; ++z)
                    for (uint32_t y = lo[1]; y <= hi[1]; ++y)
                        for (uint32_t x = lo[0]; x <= hi[0]; ++x) {
                            n_reg++;
                            const uint32_t h = grid_hash(x, y, z, hash_size);
                            candidates += starts[h + 1] - starts[h];
                        }
                hot = candidates > kHotCandidates;
                walk = !hot && candidates > 0;
                n_cand += candidates;  // hot pixels too: k_sppm_gather_hot tests every one of them
            }
        }
        const uint32_t k = wave_compact(hot, &gp->n_hot);
        if (hot) hot_list[k] = i;
        if (__ballot(walk) == 0ull) continue;
        s_phi[wv][lane][0] = s_phi[wv][lane][1] = s_phi[wv][lane][2] = 0.0f;
        s_m[wv][lane] = 0u;
        if (walk) {  // this lane's visible point, for whichever lane shades one of its pairs
            const float4 wo4 = vp.wo[i], ng4 = vp.ng[i], ns4 = vp.ns[i], ss4 = vp.ss[i], ts4 = vp.ts[i];
            Shading vs;
            vs.p = mk3(p4.x, p4.y, p4.z);
            vs.wo = mk3(wo4.x, wo4.y, wo4.z);
            vs.ng = mk3(ng4.x, ng4.y, ng4.z);
            vs.ns = mk3(ns4.x, ns4.y, ns4.z);
            vs.ss = mk3(ss4.x, ss4.y, ss4.z);
            vs.ts = mk3(ts4.x, ts4.y, ts4.z);
            const BsdfWo pre = bsdf_wo_terms(sc.materials[__float_as_uint(p4.w)].set[1], vs, vs.wo);
            float* col = &s_vp[wv][0][lane];
            col[0 * 64] = vs.ss.x, col[1 * 64] = vs.ss.y, col[2 * 64] = vs.ss.z;
            col[3 * 64] = vs.ts.x, col[4 * 64] = vs.ts.y, col[5 * 64] = vs.ts.z;
            col[6 * 64] = vs.ns.x, col[7 * 64] = vs.ns.y, col[8 * 64] = vs.ns.z;
            col[9 * 64] = vs.ng.x, col[10 * 64] = vs.ng.y, col[11 * 64] = vs.ng.z;
            col[12 * 64] = pre.wo.x, col[13 * 64] = pre.wo.y, col[14 * 64] = pre.wo.z;
            col[15 * 64] = pre.wo_ng, col[16 * 64] = pre.lam[0], col[17 * 64] = pre.lam[1];
            col[18 * 64] = p4.w;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const f3 vpp = mk3(p4.x, p4.y, p4.z);
        // cursor: cell (cx, cy, cz), entries [e, e1) of its bucket
        uint32_t cx = lo[0], cy = lo[1], cz = lo[2], e = 0, e1 = 0;
        bool more = walk;
        if (more) {
            const uint32_t h = grid_hash(cx, cy, cz, hash_size);
            e = starts[h];
            e1 = starts[h + 1];
        }
        uint32_t ring_head = 0, ring_cnt = 0;  // wave-uniform
        auto shade = [&](uint32_t cnt) {        // the first cnt parked pairs, one per lane
            if (lane < cnt) {
                const uint32_t r = s_ring[wv][(ring_head + lane) & 127u], src = s_src[wv][(ring_head + lane) & 127u];
                const float4 w4 = rec.wi[r], b4 = rec.beta[r];
                const float* col = &s_vp[wv][0][src];
                Shading vs;  // bsdf_f reads the frame (ss, ts, ns), ng and — through BsdfWo — wo; p is not used
                vs.ss = mk3(col[0 * 64], col[1 * 64], col[2 * 64]);
                vs.ts = mk3(col[3 * 64], col[4 * 64], col[5 * 64]);
                vs.ns = mk3(col[6 * 64], col[7 * 64], col[8 * 64]);
                vs.ng = mk3(col[9 * 64], col[10 * 64], col[11 * 64]);
                BsdfWo pre;
                pre.wo = mk3(col[12 * 64], col[13 * 64], col[14 * 64]);
                pre.wo_ng = col[15 * 64];
                pre.lam[0] = col[16 * 64];
                pre.lam[1] = col[17 * 64];
                const LobeSet& vb = sc.materials[__float_as_uint(col[18 * 64])].set[1];
                const f3 c = mk3(b4.x, b4.y, b4.z) * bsdf_f_wo(vb, vs, pre, mk3(w4.x, w4.y, w4.z), BSDF_ALL);
                n_acc++;
                atomicAdd(&s_phi[wv][src][0], c.x);
                atomicAdd(&s_phi[wv][src][1], c.y);
                atomicAdd(&s_phi[wv][src][2], c.z);
                atomicAdd(&s_m[wv][src], 1u);
            }
        };
        while (__ballot(more) != 0ull) {
            if (more) {
                // skip empty buckets / advance to the next cell
                while (e >= e1) {
                    if (++cx > hi[0]) {
                        cx = lo[0];
                        if (++cy > hi[1]) {
                            cy = lo[1];
                            if (++cz > hi[2]) {
                                more = false;
                                break;
                            }
                        }
                    }
                    const uint32_t h = grid_hash(cx, cy, cz, hash_size);
                    e = starts[h];
                    e1 = starts[h + 1];
                }
            }
            // kGatherAhead candidates of the lane's bucket per round, fetched together: the walk is a chain of dependent 16-byte loads (a round per candidate
            // kept the kernel waiting on memory: 54 M candidates per iteration took 0.6 ms for ~3 ms worth of arithmetic per 100 iterations)
#ifndef TH_SPPM_GATHER_AHEAD
#define TH_SPPM_GATHER_AHEAD 4
#endif
            constexpr uint32_t kGatherAhead = TH_SPPM_GATHER_AHEAD;
            float4 hp[kGatherAhead];
            const uint32_t n_here = more ? min(kGatherAhead, e1 - e) : 0u;
#pragma unroll
            for (uint32_t u = 0; u < kGatherAhead; ++u) hp[u] = u < n_here ? hit_sorted[e + u] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            e += n_here;
#pragma unroll
            for (uint32_t u = 0; u < kGatherAhead; ++u) {
                bool acc_u = false;
                if (u < n_here) {
                    const f3 dv = vpp - mk3(hp[u].x, hp[u].y, hp[u].z);  // distance_squared(vp.p, p)
                    acc_u = !(dot(dv, dv) > rad * rad);
                }
                const unsigned long long m = __ballot(acc_u);
                if (m) {
                    if (acc_u) {
                        const uint32_t at = (ring_head + ring_cnt + (uint32_t)__popcll(m & lt_mask)) & 127u;
                        s_ring[wv][at] = __float_as_uint(hp[u].w);
                        s_src[wv][at] = lane;
                    }
                    ring_cnt += (uint32_t)__popcll(m);
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    if (ring_cnt >= 64u) {
                        shade(64u);
                        ring_head = (ring_head + 64u) & 127u;
                        ring_cnt -= 64u;
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    }
                }
            }
        }
        if (ring_cnt) shade(ring_cnt);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (walk && s_m[wv][lane]) {  // ϕ and M are zero between iterations (_update_pixels! clears them)
            px.phi[3 * i + 0] = s_phi[wv][lane][0];
            px.phi[3 * i + 1] = s_phi[wv][lane][1];
            px.phi[3 * i + 2] = s_phi[wv][lane][2];
            px.M[i] = s_m[wv][lane];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    if (count_registrations) {  // trhip_sppm_state reports the last iteration's: only that launch pays for 8 192 atomics on one word (90 µs)
        n_reg = wave_sum(n_reg);
        if (lane_id() == 0 && n_reg) atomicAdd(&gp->registrations, n_reg);
    }
    if (count_stats) {  // option "count_visits": the instrumented pass of bench.py
        n_cand = wave_sum(n_cand);
        n_acc = wave_sum(n_acc);
        n_vp = wave_sum(n_vp);
        if (lane_id() == 0) {
            if (n_cand) atomicAdd(&gp->stat_candidates, n_cand);
            if (n_acc) atomicAdd(&gp->stat_accepted, n_acc);
            if (n_vp) atomicAdd(&gp->stat_visible_points, n_vp);
        }
    }
}
template <int TH_ONE_COPY = 0> __global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(TH_SPPM_HOT_WAVES))) void k_sppm_gather_hot(DeviceScene sc, PhotonRecords rec, VisiblePoints vp, PixelStats px, GridInfo* gp, const uint32_t* __restrict__ starts,
                                                            const float4* __restrict__ hit_sorted, uint32_t hash_size, const uint32_t* __restrict__ hot_list, uint32_t count_stats) {
    // One wave per hot pixel.  The distance test runs over a bucket with all 64 lanes; the photons that pass (about one in eight) are
    // not shaded where they are found — a handful of lanes would run the BSDF while the rest wait — but parked in a per-wave ring
    // and shaded 64 at a time (the trick of k_shade_path).
    __shared__ uint32_t s_ring[kBlock / 64][128];
    const GridInfo& g = *gp;
    const uint32_t n_hot = g.n_hot;
    const uint32_t lane = lane_id(), wv = threadIdx.x >> 6;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const uint32_t wave = (blockIdx.x * kBlock + threadIdx.x) >> 6, n_waves = (gridDim.x * kBlock) >> 6;
    for (uint32_t w = wave; w < n_hot; w += n_waves) {
        const uint32_t i = hot_list[w];
        const float4 p4 = vp.p_mat[i];
        const float rad = px.radius[i];
        const f3 vpp = mk3(p4.x, p4.y, p4.z);
        uint32_t lo[3], hi[3];
        to_grid(g, mk3(p4.x - rad, p4.y - rad, p4.z - rad), lo);
        to_grid(g, mk3(p4.x + rad, p4.y + rad, p4.z + rad), hi);
        const float4 wo4 = vp.wo[i], ng4 = vp.ng[i], ns4 = vp.ns[i], ss4 = vp.ss[i], ts4 = vp.ts[i];
        Shading vs;
        vs.p = vpp;
        vs.wo = mk3(wo4.x, wo4.y, wo4.z);
        vs.ng = mk3(ng4.x, ng4.y, ng4.z);
        vs.ns = mk3(ns4.x, ns4.y, ns4.z);
        vs.ss = mk3(ss4.x, ss4.y, ss4.z);
        vs.ts = mk3(ts4.x, ts4.y, ts4.z);
        const LobeSet& vb = sc.materials[__float_as_uint(p4.w)].set[1];
        const BsdfWo pre = bsdf_wo_terms(vb, vs, vs.wo);  // once per pixel instead of once per accepted photon (th_device.h)
        GatherSum s{splat3(0.0f), 0u};
        uint32_t ring_head = 0, ring_cnt = 0;  // wave-uniform
        auto shade = [&](uint32_t n) {         // the first n parked photons, one per lane
            if (lane < n) {
                const uint32_t r = s_ring[wv][(ring_head + lane) & 127u];
                const float4 w4 = rec.wi[r], b4 = rec.beta[r];
                s.phi = s.phi + mk3(b4.x, b4.y, b4.z) * bsdf_f_wo(vb, vs, pre, mk3(w4.x, w4.y, w4.z), BSDF_ALL);
                s.M++;
            }
        };
        // the buckets of all the cells the visible point registers in, fetched at once: lane c hashes cell c (x fastest, then y, then z: the order of the
        // reference's loops, sppm.jl:306-317) and loads its bucket's bounds — one trip to memory instead of one per cell in front of every bucket walk
        const uint32_t nxc = hi[0] - lo[0] + 1u, nyc = hi[1] - lo[1] + 1u, nzc = hi[2] - lo[2] + 1u;
        const uint32_t ncell = nxc * nyc * nzc;
        uint32_t e0_l = 0u, e1_l = 0u;
        for (uint32_t cb = 0; cb < ncell; cb += 64u) {  // (more than 64 cells: a radius many cells wide — rare; 64 at a time)
            const uint32_t c_l = cb + lane;
            if (c_l < ncell) {
                const uint32_t cx = c_l % nxc, cyz = c_l / nxc;
                const uint32_t h = grid_hash(lo[0] + cx, lo[1] + cyz % nyc, lo[2] + cyz / nyc, hash_size);
                e0_l = starts[h];
                e1_l = starts[h + 1];
            }
            const uint32_t c_end = min(64u, ncell - cb);
            for (uint32_t c = 0; c < c_end; ++c) {
                    const uint32_t e0 = __shfl(e0_l, (int)c), e1 = __shfl(e1_l, (int)c);
                    for (uint32_t eb = e0; eb < e1; eb += 64u) {  // wave-uniform trip count
                        const uint32_t e = eb + lane;
                        bool acc = false;
                        uint32_t r = 0;
                        if (e < e1) {
                            const float4 hp = hit_sorted[e];
                            r = __float_as_uint(hp.w);
                            const f3 dv = vpp - mk3(hp.x, hp.y, hp.z);  // distance_squared(vp.p, p)
                            acc = !(dot(dv, dv) > rad * rad);
                        }
                        const unsigned long long m = __ballot(acc);
                        if (m) {
                            if (acc) s_ring[wv][(ring_head + ring_cnt + (uint32_t)__popcll(m & lt_mask)) & 127u] = r;
                            ring_cnt += (uint32_t)__popcll(m);
                            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                            if (ring_cnt >= 64u) {
                                shade(64u);
                                ring_head = (ring_head + 64u) & 127u;
                                ring_cnt -= 64u;
                                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                            }
                        }
                    }
            }
        }
        if (ring_cnt) shade(ring_cnt);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (int off = 32; off > 0; off >>= 1) {
            s.phi.x += __shfl_down(s.phi.x, off);
            s.phi.y += __shfl_down(s.phi.y, off);
            s.phi.z += __shfl_down(s.phi.z, off);
            s.M += __shfl_down(s.M, off);
        }
        if (lane == 0 && s.M) {
            px.phi[3 * i + 0] = s.phi.x;
            px.phi[3 * i + 1] = s.phi.y;
            px.phi[3 * i + 2] = s.phi.z;
            px.M[i] = s.M;
            if (count_stats) atomicAdd(&gp->stat_accepted, (unsigned long long)s.M);
        }
    }
}

// pixel.Ld += every term of the batch in the reference's order: iterations ascending, inside an iteration by depth
// (sppm.jl:211-232).  A depth nothing was added at holds +0, and x + 0 == x for every x this sum can take.
template <int TH_ONE_COPY = 0> __global__ __launch_bounds__(kBlock) void k_sppm_fold_ld(uint32_t n_pix, uint32_t n_iter, uint32_t max_depth, const float4* __restrict__ terms, float4* __restrict__ Ld) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n_pix; i += gridDim.x * kBlock) {
        float4 l = Ld[i];
        for (uint32_t k = 0; k < n_iter * max_depth; ++k) {
            const float4 t = terms[(size_t)k * n_pix + i];
            l.x += t.x;
            l.y += t.y;
            l.z += t.z;
        }
        Ld[i] = l;
    }
}

// ---- _update_pixels! (sppm.jl:438-459) and _sppm_to_image (:461-472) ------------------------------------------------------------------
template <int TH_ONE_COPY = 0> __global__ __launch_bounds__(kBlock) void k_sppm_update(uint32_t n, float gamma, PixelStats px, VisiblePoints vp) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const uint32_t M = px.M[i];
        if (M > 0) {
            const double N = px.N[i];
            const float radius = px.radius[i];
            const double n_new = N + (double)(gamma * (float)M);
            const double radius_new = (double)radius * __builtin_sqrt(n_new / (N + (double)M));
            const double ratio = radius_new / (double)radius, r2 = ratio * ratio;
            const float4 t = px.tau[i];
            const float sx = t.x + px.phi[3 * i], sy = t.y + px.phi[3 * i + 1], sz = t.z + px.phi[3 * i + 2];
            px.tau[i] = make_float4((float)((double)sx * r2), (float)((double)sy * r2), (float)((double)sz * r2), 0.0f);
            px.radius[i] = (float)radius_new;
            px.N[i] = n_new;
            px.phi[3 * i] = px.phi[3 * i + 1] = px.phi[3 * i + 2] = 0.0f;
            px.M[i] = 0u;
        }
        vp.beta[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
}
// film pixel = set_image!(film, image): xyz = to_XYZ(image[i]), filter_weight_sum = 1 (film.jl:195-202)
template <int TH_ONE_COPY = 0> __global__ __launch_bounds__(kBlock) void k_sppm_image(uint32_t n, uint32_t iteration, uint64_t photons_per_iteration, PixelStats px, float4* __restrict__ film) {
    const double Np = (double)((uint64_t)iteration * photons_per_iteration) * 3.141592653589793;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const float4 ld = px.Ld[i], t = px.tau[i];
        const float r = px.radius[i];
        const f3 a = mk3(ld.x, ld.y, ld.z) / (float)iteration;
        const double den = Np * (double)(r * r);
        const f3 b = mk3((float)((double)t.x / den), (float)((double)t.y / den), (float)((double)t.z / den));
        const f3 xyz = rgb_to_xyz(a + b);
        film[i] = make_float4(xyz.x, xyz.y, xyz.z, 1.0f);
    }
}

}  // namespace th
