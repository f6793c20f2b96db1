// th_whitted.h — WhittedIntegrator (integrators/sampler.jl:58-199) as a wavefront: the reference's recursion
//
//     li = Σ_lights f·Li·|wi·n|/pdf  +  f_r · li(reflected) · |wi·ns| / pdf_r  +  f_t · li(transmitted) · |wi·ns| / pdf_t
//
// multiplies each child's TOTAL radiance, so carrying a throughput down the tree would change the rounding.  Instead the
// ray tree is built level by level (one node per ray: parent id, branch, f, |wi·ns|, pdf), direct light is added to each
// node in light order as its shadow rays resolve, and the tree is folded bottom-up — reflected child first, then the
// transmitted one, exactly the order of `l += specular_reflect(…); l += specular_transmit(…)` (:96-99) — so every
// sample's radiance equals the recursive evaluation bit for bit.  The get_2d() values Whitted draws are ignored by δ-lights
// and by single specular lobes: none are consumed.
#pragma once
#include "th_kernels.h"

namespace th {

struct WhittedPool {      // one entry per ray-tree node; node id = level_base + physical queue index
    float4* L;            // radiance of the subtree rooted here (rgb)
    uint32_t* parent;     // node id of the parent
    float4* coef;         // f.rgb, |wi·ns|
    float2* pdf_branch;   // pdf, as_float(branch: 0 reflected, 1 transmitted)
};
struct WhittedFlags {
    uint32_t overflow;  // a level outgrew its queue capacity
};

// One tree level: interaction, BSDF (allow_multiple_lobes = false), per-light shadow rays, ≤ 2 specular children.
template <int TH_ONE_COPY = 0> __global__ __launch_bounds__(kBlock) void k_shade_whitted(DeviceScene sc, PathQueue qin, PathQueue qout, ShadowQueue sq, uint32_t cap, uint32_t cap_shadow, const float4* __restrict__ hits,
                                                          WhittedPool pool, uint32_t base_in, uint32_t base_out, Counters* ctr, WhittedFlags* flags, int depth, int max_depth) {
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
        bool have = false;
        Shading sh;
        const LobeSet* bsdf = nullptr;
        if (valid) {
            const int prim = __float_as_int(hits[i].y);
            if (prim >= 0) {
                const float4 o4 = qin.o[i], d4 = qin.d[i];
                uint32_t material;
                if (rebuild_shading(sc, prim, mk3(o4.x, o4.y, o4.z), mk3(d4.x, d4.y, d4.z), sh, material) && material != PRIM_NO_MATERIAL) {
                    bsdf = &sc.materials[material].set[0];  // compute_scattering!(si, ray): allow_multiple_lobes = false (:75)
                    have = true;
                }
            }
        }
        const uint32_t node = base_in + i;
        if (depth == 1 && valid) pool.parent[node] = __float_as_uint(qin.o[i].w);  // roots remember their sample slot
        // ---- direct light, one shadow ray per (hit, light) (:84-94) ----
        for (uint32_t l = 0; l < sc.n_lights; ++l) {
            bool want = false;
            float4 so4, sd4, sc4;
            if (have) {
                const LightRec& light = sc.lights[l];
                const LightSample ls = sample_li(light, sh.p);
                if (!(is_black(ls.radiance) || ls.pdf == 0.0f)) {
                    const f3 f = bsdf_f(*bsdf, sh, sh.wo, ls.wi, BSDF_ALL);
                    if (!is_black(f)) {
                        const f3 c = f * ls.radiance * fabs_(dot(ls.wi, sh.ns)) / ls.pdf;
                        const f3 lp = mk3(light.position[0], light.position[1], light.position[2]);
                        const f3 dir = lp - sh.p;
                        const f3 org = sh.p + 1e-6f * dir;
                        const f3 cd = check_direction(dir);
                        so4 = make_float4(org.x, org.y, org.z, __uint_as_float(node));
                        sd4 = make_float4(cd.x, cd.y, cd.z, __uint_as_float(l));
                        sc4 = make_float4(c.x, c.y, c.z, 0.0f);
                        want = true;
                    }
                }
            }
            const uint32_t k = wave_compact(want, &ctr->n_shadow[depth - 1][seg_out * kCtrStride]);
            if (want) {
                if (k < cap_shadow) {
                    const uint32_t si = seg_out * cap_shadow + k;
                    sq.o[si] = so4;
                    sq.d[si] = sd4;
                    sq.c[si] = sc4;
                } else {
                    flags->overflow = 1u;
                }
            }
        }
        // ---- specular_reflect (:103-143) then specular_transmit (:145-199) ----
        if (depth + 1 <= max_depth) {
            for (int pass = 0; pass < 2; ++pass) {
                bool want = false;
                float4 no4, nd4, cf4;
                float pdf = 0.0f;
                if (have) {
                    const int type = (pass == 0 ? BSDF_REFLECTION : BSDF_TRANSMISSION) | BSDF_SPECULAR;
                    const BsdfSample s = bsdf_sample_f(*bsdf, sh, sh.wo, f2{0.0f, 0.0f}, type);
                    const float a = fabs_(dot(s.wi, sh.ns));
                    if (s.pdf > 0.0f && !is_black(s.f) && a != 0.0f) {
                        const f3 org = sh.p + 1e-6f * s.wi;
                        const f3 nd = check_direction(s.wi);
                        no4 = make_float4(org.x, org.y, org.z, 0.0f);
                        nd4 = make_float4(nd.x, nd.y, nd.z, 0.0f);
                        cf4 = make_float4(s.f.x, s.f.y, s.f.z, a);
                        pdf = s.pdf;
                        want = true;
                    }
                }
                const uint32_t k = wave_compact(want, &ctr->n_queue[depth][seg_out * kCtrStride]);
                if (want) {
                    if (k < cap) {
                        const uint32_t ni = seg_out * cap + k;
                        qout.o[ni] = no4;
                        qout.d[ni] = nd4;
                        const uint32_t child = base_out + ni;
                        pool.parent[child] = node;
                        pool.coef[child] = cf4;
                        pool.pdf_branch[child] = make_float2(pdf, __uint_as_float((uint32_t)pass));
                    } else {
                        flags->overflow = 1u;
                    }
                }
            }
        }
    }
}

// `l += f * sampled_li * abs(wi ⋅ n) / pdf` for light `light`, for every unoccluded shadow ray of this level (:91-93).
template <int TH_ONE_COPY = 0> __global__ __launch_bounds__(kBlock) void k_whitted_direct(SegQueue q, ShadowQueue sq, const uint8_t* __restrict__ occluded, uint32_t light, float4* __restrict__ node_L) {
    __shared__ SegView sv;
    seg_load(q, sv);
    const uint32_t total = sv.prefix[kSeg];
    uint32_t seg = 0;  // (carried over the iterations: seg_locate_from)
    for (uint32_t flat = blockIdx.x * kBlock + threadIdx.x; flat < total; flat += gridDim.x * kBlock) {
        uint32_t lb;
        seg_locate_from(sv, flat & ~63u, seg, lb);
        const uint32_t local = lb + (flat & 63u);
        if (local >= sv.count[seg]) continue;
        const uint32_t i = seg * q.cap + local;
        if (__float_as_uint(sq.d[i].w) != light || occluded[i]) continue;
        const uint32_t node = __float_as_uint(sq.o[i].w);
        const float4 c = sq.c[i];
        float4 l = node_L[node];
        l.x += c.x;
        l.y += c.y;
        l.z += c.z;
        node_L[node] = l;
    }
}
// Fold one level into its parents: `l += f * li(child) * abs(wi ⋅ ns) / pdf` for the children of branch `branch`.
template <int TH_ONE_COPY = 0> __global__ __launch_bounds__(kBlock) void k_whitted_resolve(SegQueue q, WhittedPool pool, uint32_t base, uint32_t branch) {
    __shared__ SegView sv;
    seg_load(q, sv);
    const uint32_t total = sv.prefix[kSeg];
    uint32_t seg = 0;  // (carried over the iterations: seg_locate_from)
    for (uint32_t flat = blockIdx.x * kBlock + threadIdx.x; flat < total; flat += gridDim.x * kBlock) {
        uint32_t lb;
        seg_locate_from(sv, flat & ~63u, seg, lb);
        const uint32_t local = lb + (flat & 63u);
        if (local >= sv.count[seg]) continue;
        const uint32_t node = base + seg * q.cap + local;
        const float2 pb = pool.pdf_branch[node];
        if (__float_as_uint(pb.y) != branch) continue;
        const float4 cf = pool.coef[node], lc = pool.L[node];
        const f3 term = mk3(cf.x, cf.y, cf.z) * mk3(lc.x, lc.y, lc.z) * cf.w / pb.x;
        const uint32_t parent = pool.parent[node];
        float4 l = pool.L[parent];
        l.x += term.x;
        l.y += term.y;
        l.z += term.z;
        pool.L[parent] = l;
    }
}
// Level 1 (camera rays): node radiance -> per-sample radiance buffer indexed by slot.
template <int TH_ONE_COPY = 0> __global__ __launch_bounds__(kBlock) void k_whitted_finish(SegQueue q, const uint32_t* __restrict__ root_slot, const float4* __restrict__ node_L, float4* __restrict__ L) {
    __shared__ SegView sv;
    seg_load(q, sv);
    const uint32_t total = sv.prefix[kSeg];
    uint32_t seg = 0;  // (carried over the iterations: seg_locate_from)
    for (uint32_t flat = blockIdx.x * kBlock + threadIdx.x; flat < total; flat += gridDim.x * kBlock) {
        uint32_t lb;
        seg_locate_from(sv, flat & ~63u, seg, lb);
        const uint32_t local = lb + (flat & 63u);
        if (local >= sv.count[seg]) continue;
        const uint32_t i = seg * q.cap + local;
        L[root_slot[i]] = node_L[i];
    }
}

}  // namespace th
