// tu_scene.hip — scene flattening into HBM: materials, primitives, BVHAccel commit (host SAH / device LBVH / a caller's tree), the
// derived node arrays of the traversal kernels.
#include "th_host.h"
#include "th_bvh_ref.h"

// Where the children-in-parent node of interior node i of a depth-first FlatBVH lives (th_trace2.h wnodes): widx[i]; returns the number of 64-byte slots (>= the number of
// interior nodes: layout 1 pads).  layout 0: interior nodes in depth-first order (a first child directly behind its parent).  layout 1 (option "node_layout"): the two
// interior children of a node SIDE BY SIDE in one aligned 128-byte line — a node that misses L2 arrives as a 128-byte line either way (round 3's request-size counters on the
// 10 M-triangle scene: 9.7e8 requests of 128 B, none of 32 / 64): with the siblings in it, the far child the walk comes back for is already on the chip.  Subtrees stay
// contiguous (pairs are allocated when their parent is visited, depth first), the root stays slot 0.  Slots left empty by the alignment are zero-filled and never referenced.
static uint32_t wide_node_order(const FlatBVH& t, int layout, std::vector<uint32_t>& widx) {
    const uint32_t n = (uint32_t)t.a.size();
    widx.assign(n, 0u);
    auto interior = [&](uint32_t i) { return (t.flags[i] & 3u) != 3u; };
    uint32_t next = 0;
    if (layout != 1) {
        for (uint32_t i = 0; i < n; ++i)
            if (interior(i)) widx[i] = next++;
        return next;
    }
    if (n == 0 || !interior(0)) return 0;
    widx[0] = next++;
    std::vector<uint32_t> stack{0u};
    while (!stack.empty()) {
        const uint32_t i = stack.back();
        stack.pop_back();
        const uint32_t c0 = i + 1, c1 = t.a[i];
        const bool i0 = c0 < n && interior(c0), i1 = c1 < n && interior(c1);
        if (i0 && i1) {
            next = (next + 1u) & ~1u;  // the pair starts a 128-byte line
            widx[c0] = next++;
            widx[c1] = next++;
        } else if (i0) {
            widx[c0] = next++;
        } else if (i1) {
            widx[c1] = next++;
        }
        if (i1) stack.push_back(c1);  // the first child's subtree is laid out first
        if (i0) stack.push_back(c0);
    }
    return next;
}

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <mutex>

namespace {
// TRHIP_COMMIT_TIMING=1: where trhip_scene_commit spends its time, one line per stage on stderr
struct CommitClock {
    bool on = std::getenv("TRHIP_COMMIT_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void tick(const char* what) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[commit] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    }
};

// ---- materials: the lobes each Material adds (materials/material.jl), precomputed per material ---------------------------------
float roughness_to_alpha(float roughness) {  // microfacet.jl:82-87
    roughness = jmax(1e-3f, roughness);
    const float x = tm_logf(roughness);
    return 1.62142f + 0.819955f * x + 0.1734f * (x * x) + 0.0171201f * (x * x * x) + 0.000640711f * pow4(x);
}
void clamp_rgb(const float* in, float* out) {  // clamp(spectrum) spectrum.jl:34-38
    for (int i = 0; i < 3; ++i) out[i] = jclamp(in[i], 0.0f, kInf);
}
bool black(const float* c) { return c[0] == 0.0f && c[1] == 0.0f && c[2] == 0.0f; }
Lobe base_lobe(int kind, int type) {
    Lobe l;
    std::memset(&l, 0, sizeof l);
    l.kind = kind;
    l.type = type;
    l.fresnel = FRESNEL_NOOP;
    l.eta_a = l.eta_b = l.fr_eta_i = l.fr_eta_t = 1.0f;
    return l;
}
void set_rgb(float* dst, const float* src) {
    dst[0] = src[0];
    dst[1] = src[1];
    dst[2] = src[2];
}
Lobe microfacet_lobe(int kind, int type, const float* rgb, float ax, float ay) {
    Lobe l = base_lobe(kind, type);
    set_rgb(l.r, rgb);
    l.a = jmax(1e-3f, ax);  // TrowbridgeReitzDistribution ctor microfacet.jl:61-65
    l.b = jmax(1e-3f, ay);
    return l;
}
int build_material(int kind, const float* p, int n, MaterialRec& m) {
    std::memset(&m, 0, sizeof m);
    for (int multi = 0; multi < 2; ++multi) {
        LobeSet& s = m.set[multi];
        s.n = 0;
        s.eta = 1.0f;
        switch (kind) {
        case TRHIP_MATTE: {  // material.jl:16-31
            if (n != 4) return -1;
            float r[3];
            clamp_rgb(p, r);
            if (black(r)) break;
            const float sigma = jclamp(p[3], 0.0f, 90.0f);
            if (sigma == 0.0f) {
                Lobe l = base_lobe(LOBE_LAMBERT_R, BSDF_DIFFUSE | BSDF_REFLECTION);
                set_rgb(l.r, r);
                s.lobe[s.n++] = l;
            } else {  // OrenNayar ctor microfacet.jl:12-19
                Lobe l = base_lobe(LOBE_OREN_NAYAR, BSDF_DIFFUSE | BSDF_REFLECTION);
                set_rgb(l.r, r);
                const float sg = deg2rad(sigma);
                const float s2 = sg * sg;
                l.a = 1.0f - (s2 / (2.0f * (s2 + 0.33f)));
                l.b = 0.45f * s2 / (s2 + 0.09f);
                s.lobe[s.n++] = l;
            }
            break;
        }
        case TRHIP_MIRROR: {  // material.jl:39-46
            if (n != 3) return -1;
            float r[3];
            clamp_rgb(p, r);
            if (black(r)) break;
            Lobe l = base_lobe(LOBE_SPECULAR_R, BSDF_SPECULAR | BSDF_REFLECTION);
            set_rgb(l.r, r);
            s.lobe[s.n++] = l;
            break;
        }
        case TRHIP_GLASS: {  // material.jl:75-116
            if (n != 10) return -1;
            const float eta = p[8];
            float ur = p[6], vr = p[7];
            const bool remap = p[9] != 0.0f;
            s.eta = eta;
            float r[3], t[3];
            clamp_rgb(p, r);
            clamp_rgb(p + 3, t);
            if (black(r) && black(t)) break;
            const bool is_specular = ur == 0.0f && vr == 0.0f;
            if (is_specular && multi) {
                Lobe l = base_lobe(LOBE_FRESNEL_SPECULAR, BSDF_SPECULAR | BSDF_TRANSMISSION | BSDF_REFLECTION);
                set_rgb(l.r, r);
                set_rgb(l.t, t);
                l.eta_a = 1.0f;
                l.eta_b = eta;
                s.lobe[s.n++] = l;
                break;
            }
            if (remap) {
                ur = roughness_to_alpha(ur);
                vr = roughness_to_alpha(vr);
            }
            if (!black(r)) {
                Lobe l = is_specular ? base_lobe(LOBE_SPECULAR_R, BSDF_SPECULAR | BSDF_REFLECTION) : microfacet_lobe(LOBE_MICROFACET_R, BSDF_REFLECTION | BSDF_GLOSSY, r, ur, vr);
                set_rgb(l.r, r);
                l.fresnel = FRESNEL_DIELECTRIC;
                l.fr_eta_i = 1.0f;
                l.fr_eta_t = eta;
                s.lobe[s.n++] = l;
            }
            if (!black(t)) {
                Lobe l = is_specular ? base_lobe(LOBE_SPECULAR_T, BSDF_SPECULAR | BSDF_TRANSMISSION) : microfacet_lobe(LOBE_MICROFACET_T, BSDF_TRANSMISSION | BSDF_GLOSSY, t, ur, vr);
                set_rgb(l.r, t);
                l.eta_a = 1.0f;
                l.eta_b = eta;
                l.fresnel = FRESNEL_DIELECTRIC;  // FresnelDielectric(η_a, η_b) specular.jl:60, microfacet.jl:275
                l.fr_eta_i = 1.0f;
                l.fr_eta_t = eta;
                s.lobe[s.n++] = l;
            }
            break;
        }
        case TRHIP_PLASTIC: {  // material.jl:135-151
            if (n != 8) return -1;
            float kd[3], ks[3];
            clamp_rgb(p, kd);
            if (!black(kd)) {
                Lobe l = base_lobe(LOBE_LAMBERT_R, BSDF_DIFFUSE | BSDF_REFLECTION);
                set_rgb(l.r, kd);
                s.lobe[s.n++] = l;
            }
            clamp_rgb(p + 3, ks);
            if (black(ks)) break;
            float rough = p[6];
            if (p[7] != 0.0f) rough = roughness_to_alpha(rough);
            Lobe l = microfacet_lobe(LOBE_MICROFACET_R, BSDF_REFLECTION | BSDF_GLOSSY, ks, rough, rough);
            l.fresnel = FRESNEL_DIELECTRIC;
            l.fr_eta_i = 1.5f;
            l.fr_eta_t = 1.0f;
            s.lobe[s.n++] = l;
            break;
        }
        default: return -1;
        }
    }
    return 0;
}

// world_bound(sphere) = object_to_world(object_bound) (Shape.jl:17-19, transformations.jl:141-143)
HostAABB sphere_world_bound(const SphereRec& s) {
    HostAABB b;
    b.reset();
    const float lo[3] = {-s.radius, -s.radius, s.z_min}, hi[3] = {s.radius, s.radius, s.z_max};
    for (int c = 0; c < 8; ++c) {
        const f3 p = xf_point(s.o2w, mk3((c & 1) ? hi[0] : lo[0], (c & 2) ? hi[1] : lo[1], (c & 4) ? hi[2] : lo[2]));
        const float q[3] = {p.x, p.y, p.z};
        b.grow_point(q);
    }
    return b;
}
float det3(const float* m) {  // rows of the upper-left 3x3 of a row-major 4x4
    return m[0] * (m[5] * m[10] - m[6] * m[9]) - m[1] * (m[4] * m[10] - m[6] * m[8]) + m[2] * (m[4] * m[9] - m[5] * m[8]);
}

// does the subtree rooted at flat node `root` hold a sphere?  (depth-first layout: the subtree is a contiguous index range)
bool has_sphere_subtree(const trhip_scene* s, uint32_t root) {
    const uint32_t n_nodes = (uint32_t)s->bvh.a.size(), n_prims = (uint32_t)s->bvh.order.size();
    if (root >= n_nodes) return true;
    // end of the subtree: follow second children until a leaf
    uint32_t end = root;
    while ((s->bvh.flags[end] & 3u) != 3u) end = s->bvh.a[end];
    for (uint32_t i = root; i <= end && i < n_nodes; ++i)
        if ((s->bvh.flags[i] & 3u) == 3u)
            for (uint32_t k = s->bvh.a[i]; k < s->bvh.a[i] + (s->bvh.flags[i] >> 2) && k < n_prims; ++k)
                if (s->prims[s->bvh.order[k]].kind == 1) return true;
    return false;
}
}  // namespace

int upload_scene(trhip_scene* s) {
    trhip_ctx* ctx = s->ctx;
    CommitClock clk;
    const uint32_t n_nodes = (uint32_t)s->bvh.a.size(), n_prims = (uint32_t)s->bvh.order.size();
    RawArray<float4> nodes((size_t)n_nodes * 2), prims((size_t)n_prims * 3), nrm((size_t)n_prims * 3);  // every element is written below
    parallel_for(n_nodes, [&](size_t i0, size_t i1) {
        for (size_t i = i0; i < i1; ++i) {
            const float* b = &s->bvh.bounds[6 * i];
            nodes[2 * i] = make_float4(b[0], b[1], b[2], __builtin_bit_cast(float, s->bvh.a[i]));
            nodes[2 * i + 1] = make_float4(b[3], b[4], b[5], __builtin_bit_cast(float, s->bvh.flags[i]));
        }
    });
    parallel_for(n_prims, [&](size_t k0, size_t k1) {
        for (size_t k = k0; k < k1; ++k) {
            const HostPrim& p = s->prims[s->bvh.order[k]];
            if (p.kind == 1) {
                prims[3 * k] = make_float4(__builtin_bit_cast(float, p.sphere_id), 0, 0, __builtin_bit_cast(float, p.meta));
                prims[3 * k + 1] = prims[3 * k + 2] = make_float4(0, 0, 0, 0);
                nrm[3 * k] = nrm[3 * k + 1] = nrm[3 * k + 2] = make_float4(0, 0, 0, 0);
            } else {
                prims[3 * k] = make_float4(p.v[0], p.v[1], p.v[2], __builtin_bit_cast(float, p.meta));
                prims[3 * k + 1] = make_float4(p.v[3], p.v[4], p.v[5], 0);
                prims[3 * k + 2] = make_float4(p.v[6], p.v[7], p.v[8], 0);
                const uint32_t mat = p.meta & PRIM_MATERIAL_MASK;
                const bool fast = mat != PRIM_NO_MATERIAL && mat < s->materials.size() && s->materials[mat].set[1].n == 1 && s->materials[mat].set[1].lobe[0].kind == LOBE_LAMBERT_R;
                if (fast) prims[3 * k].w = __builtin_bit_cast(float, p.meta | PRIM_FAST);
                for (int j = 0; j < 3; ++j) nrm[3 * k + j] = make_float4(p.n[3 * j], p.n[3 * j + 1], p.n[3 * j + 2], fast ? s->materials[mat].set[1].lobe[0].r[j] : 0.0f);
            }
        }
    });
    clk.tick("upload: node / prim records");
    if (int rc = upload(ctx, s->d_nodes, nodes.data(), nodes.size() * sizeof(float4))) return rc;
    if (int rc = upload(ctx, s->d_prims, prims.data(), prims.size() * sizeof(float4))) return rc;
    if (int rc = upload(ctx, s->d_nrm, nrm.data(), nrm.size() * sizeof(float4))) return rc;
    clk.tick("upload: 3 copies");
    // the two optional mesh arrays (no scene of the reference sets them): tangents stay resident for the shading kernels, (u, v)s only feed k_shade_constants
    const bool any_tan = !s->prim_tan.empty(), any_uv = !s->prim_uv.empty();
    if (any_tan) s->prim_tan.resize(9 * s->prims.size(), 0.0f);  // primitives added after the last mesh with tangents / (u, v)s: zeros
    if (any_uv) s->prim_uv.resize(7 * s->prims.size(), 0.0f);
    s->has_materialless_prim = false;
    for (const HostPrim& p : s->prims) s->has_materialless_prim = s->has_materialless_prim || (p.meta & PRIM_MATERIAL_MASK) == PRIM_NO_MATERIAL;
    s->dev.tri_tan = nullptr;
    if (any_tan) {
        std::vector<float4> tan((size_t)n_prims * 3, make_float4(0, 0, 0, 0));
        parallel_for(n_prims, [&](size_t k0, size_t k1) {
            for (size_t k = k0; k < k1; ++k) {
                const uint32_t id = s->bvh.order[k];
                const HostPrim& p = s->prims[id];
                const float* tg = &s->prim_tan[9 * (size_t)id];
                if (p.kind == 0 && (p.meta & PRIM_HAS_TANGENTS))
                    for (int j = 0; j < 3; ++j) tan[3 * k + j] = make_float4(tg[3 * j], tg[3 * j + 1], tg[3 * j + 2], 0.0f);
            }
        });
        if (int rc = upload(ctx, s->d_tan, tan.data(), tan.size() * sizeof(float4))) return rc;
        s->dev.tri_tan = (const float4*)s->d_tan.p;
    } else {
        release(s->d_tan);
    }
    DevBuf d_uv;
    if (any_uv) {
        std::vector<float4> uvs((size_t)n_prims * 2, make_float4(0, 0, 0, 0));
        parallel_for(n_prims, [&](size_t k0, size_t k1) {
            for (size_t k = k0; k < k1; ++k) {
                const uint32_t id = s->bvh.order[k];
                const float* uv = &s->prim_uv[7 * (size_t)id];
                if (s->prims[id].kind == 0 && uv[6] != 0.0f) {
                    uvs[2 * k] = make_float4(uv[0], uv[1], uv[2], uv[3]);
                    uvs[2 * k + 1] = make_float4(uv[4], uv[5], 1.0f, 0.0f);
                }
            }
        });
        if (int rc = upload(ctx, d_uv, uvs.data(), uvs.size() * sizeof(float4))) {
            release(d_uv);
            return rc;
        }
    }
    {  // the shading kernels' interleaved view (th_scene.h): one 128-byte line per slot, put together on the device from the two arrays just uploaded
        if (int rc = ensure(ctx, s->d_shade, (size_t)n_prims * 8 * sizeof(float4))) {
            release(d_uv);
            return rc;
        }
        hipLaunchKernelGGL(k_shade_constants, dim3(std::max(1u, std::min((n_prims + kBlock - 1) / kBlock, 4096u))), dim3(kBlock), 0, ctx->stream, (float4*)s->d_shade.p, (const float4*)s->d_prims.p, (const float4*)s->d_nrm.p, (const float4*)d_uv.p, n_prims);
        const hipError_t e1 = hipGetLastError(), e2 = hipStreamSynchronize(ctx->stream);
        release(d_uv);
        HIP_TRY(ctx, e1);
        HIP_TRY(ctx, e2);
    }
    clk.tick("upload: shade records");
    if (int rc = upload(ctx, s->d_spheres, s->spheres.data(), s->spheres.size() * sizeof(SphereRec))) return rc;
    if (int rc = upload(ctx, s->d_materials, s->materials.data(), s->materials.size() * sizeof(MaterialRec))) return rc;
    if (int rc = upload(ctx, s->d_lights, s->lights.data(), s->lights.size() * sizeof(LightRec))) return rc;
    s->dev.nodes = (const float4*)s->d_nodes.p;
    s->dev.prims = (const float4*)s->d_prims.p;
    s->dev.tri_nrm = (const float4*)s->d_nrm.p;
    s->dev.shade = (const float4*)s->d_shade.p;
    s->dev.spheres = (const SphereRec*)s->d_spheres.p;
    s->dev.materials = (const MaterialRec*)s->d_materials.p;
    s->dev.lights = (const LightRec*)s->d_lights.p;
    s->dev.n_nodes = n_nodes;
    s->dev.n_prims = n_prims;
    s->dev.n_spheres = (uint32_t)s->spheres.size();
    s->dev.n_materials = (uint32_t)s->materials.size();
    s->dev.n_lights = (uint32_t)s->lights.size();
    // ---- children-in-parent nodes for k_trace2 (th_trace2.h) ----
    bool has_empty_leaf = false;
    s->wide_ok = false;
    std::memset(&s->wide, 0, sizeof s->wide);
    s->wide.root_ref = kRefNone;
    if (n_nodes > 0 && n_prims < (1u << 24) && !s->literal_only) {
        std::vector<uint32_t> widx;
        const uint32_t n_int = wide_node_order(s->bvh, ctx->node_layout, widx);
        bool ok = n_int < (1u << 24);
        RawArray<float4> wn((size_t)n_int * 4);  // every interior node writes its four records below (layout 1: the alignment gaps are zeroed first)
        if (ctx->node_layout == 1 && ok) std::memset((void*)wn.data(), 0, (size_t)n_int * 4 * sizeof(float4));
        has_empty_leaf = false;
        // subtrees that hold a sphere keep the reference's loose slab test (th_trace2.h, slab_test2): the fp32 sphere quadratic
        // (sphere.jl:120-150) accepts rays that pass the sphere at a distance far beyond the tight test's margin
        std::vector<uint8_t> has_sphere(n_nodes, 0);
        if (!s->spheres.empty()) {
            // the ordered slots that hold spheres (the .w lane of a slot's first record carries PRIM_SPHERE)
            std::vector<uint32_t> sphere_slots;
            std::mutex lock;
            parallel_for(n_prims, [&](size_t k0, size_t k1) {
                std::vector<uint32_t> mine;
                for (size_t k = k0; k < k1; ++k)
                    if (__builtin_bit_cast(uint32_t, prims[3 * k].w) & PRIM_SPHERE) mine.push_back((uint32_t)k);
                if (!mine.empty()) {
                    std::lock_guard<std::mutex> g(lock);
                    sphere_slots.insert(sphere_slots.end(), mine.begin(), mine.end());
                }
            });
            bool marked = false;
            if (sphere_slots.size() <= 4096) {
                // few spheres: mark the root path of each one's leaf.  In the depth-first layout a subtree's slots are contiguous and start at its leftmost leaf's first
                // slot, so "which child holds slot k" is one comparison with the second child's leftmost slot.  (A tree that is not laid out that way — a caller's
                // — fails the checks below and takes the bottom-up pass.)
                auto leftmost = [&](uint32_t i, uint32_t& slot) {
                    for (uint32_t guard = 0; guard < 4096u && i < n_nodes; ++guard) {
                        if ((s->bvh.flags[i] & 3u) == 3u) {
                            slot = s->bvh.a[i];
                            return true;
                        }
                        ++i;
                    }
                    return false;
                };
                marked = true;
                for (uint32_t k : sphere_slots) {
                    uint32_t i = 0;
                    bool found = false;
                    for (uint32_t guard = 0; guard < 4096u && i < n_nodes; ++guard) {
                        has_sphere[i] = 1;
                        if ((s->bvh.flags[i] & 3u) == 3u) {
                            found = k - s->bvh.a[i] < (s->bvh.flags[i] >> 2);
                            break;
                        }
                        const uint32_t second = s->bvh.a[i];
                        uint32_t lo2 = 0;
                        if (second <= i + 1 || second >= n_nodes || !leftmost(second, lo2)) break;
                        i = k >= lo2 ? second : i + 1;
                    }
                    if (!found) {
                        marked = false;
                        break;
                    }
                }
                if (!marked) std::fill(has_sphere.begin(), has_sphere.end(), (uint8_t)0);
            }
            if (!marked)  // many spheres, or a layout the shortcut does not understand: bottom-up over all nodes
                for (uint32_t i = n_nodes; i-- > 0;) {
                    if ((s->bvh.flags[i] & 3u) == 3u) {
                        const uint32_t first = s->bvh.a[i], cnt = s->bvh.flags[i] >> 2;
                        for (uint32_t k = first; k < first + cnt && k < n_prims; ++k) has_sphere[i] |= (__builtin_bit_cast(uint32_t, prims[3 * (size_t)k].w) & PRIM_SPHERE) != 0;
                    } else {
                        has_sphere[i] = has_sphere[i + 1] | (s->bvh.a[i] < n_nodes ? has_sphere[s->bvh.a[i]] : 1);
                    }
                }
        }
        std::atomic<bool> bad{!ok}, empty_leaf{false};
        parallel_for(n_nodes, [&](size_t i0, size_t i1) {
            static const float kNanBox[6] = {NAN, NAN, NAN, NAN, NAN, NAN};
            for (size_t i = i0; i < i1; ++i) {
                if ((s->bvh.flags[i] & 3u) == 3u) continue;
                const uint32_t c[2] = {(uint32_t)i + 1, s->bvh.a[i]};
                uint32_t ref[2], cnt[2];
                for (int k = 0; k < 2; ++k) {
                    if ((s->bvh.flags[c[k]] & 3u) == 3u) {
                        ref[k] = s->bvh.a[c[k]];
                        cnt[k] = s->bvh.flags[c[k]] >> 2;
                        if (cnt[k] == 0) {
                            // the reference's builder can emit a leaf of 0 primitives with the invalid bounds (+Inf, -Inf) (A.6, th_bvh_ref.h): no ray's box
                            // test passes on it (tx_min = +Inf, bounds.jl:186-188), so the child word is never read — any leaf-shaped word will do.
                            // An empty leaf with a REAL box (a foreign tree) could be "entered": the literal kernel walks those.
                            const float* eb = &s->bvh.bounds[6 * (size_t)c[k]];
                            if (eb[0] == INFINITY && eb[1] == INFINITY && eb[2] == INFINITY && eb[3] == -INFINITY && eb[4] == -INFINITY && eb[5] == -INFINITY) {
                                ref[k] = 0;
                                cnt[k] = 1;
                                empty_leaf = true;
                            } else {
                                bad = true;
                            }
                        } else if (cnt[k] > 255) {
                            bad = true;  // oversized leaves only come from foreign BVHs: use the literal kernel
                        }
                    } else {
                        ref[k] = widx[c[k]];
                        cnt[k] = 0;
                    }
                }
                const float* l = &s->bvh.bounds[6 * (size_t)c[0]];
                const float* r = &s->bvh.bounds[6 * (size_t)c[1]];
                // an empty leaf's box (+Inf, -Inf) goes in as NaNs: every comparison with a NaN product is false, so the child is missed by the select form of the box
                // test (slab_test2) AND by the min / max forms (th_trace7.h), where min(+Inf x, -Inf x) would turn the inverted box into "everything"
                if ((s->bvh.flags[c[0]] & 3u) == 3u && (s->bvh.flags[c[0]] >> 2) == 0u) l = kNanBox;
                if ((s->bvh.flags[c[1]] & 3u) == 3u && (s->bvh.flags[c[1]] >> 2) == 0u) r = kNanBox;
                float4* w = &wn[4 * (size_t)widx[i]];
                w[0] = make_float4(l[0], l[1], l[2], l[3]);
                w[1] = make_float4(l[4], l[5], r[0], r[1]);
                w[2] = make_float4(r[2], r[3], r[4], r[5]);
                // child word = ref | count << 24 (the stack entry format); meta = split axis | "subtree holds a sphere" bits 2 (first) / 3 (second)
                w[3] = make_float4(__builtin_bit_cast(float, ref[0] | (cnt[0] << 24)), __builtin_bit_cast(float, ref[1] | (cnt[1] << 24)),
                                   __builtin_bit_cast(float, (s->bvh.flags[i] & 3u) | ((uint32_t)has_sphere[c[0]] << 2) | ((uint32_t)has_sphere[c[1]] << 3)), 0.0f);
            }
        });
        ok = !bad;
        has_empty_leaf = empty_leaf;
        // k_trace7 re-derives a leaf's box from its triangles' vertices (min / max are exact): true for every tree built here; a caller's tree (trhip_scene_set_bvh) may
        // carry larger leaf boxes, and then the kernel keeps the reference's test on every box instead
        std::atomic<bool> loose{false};
        parallel_for(n_nodes, [&](size_t i0, size_t i1) {
            for (size_t i = i0; i < i1 && !loose; ++i) {
                if ((s->bvh.flags[i] & 3u) != 3u) continue;
                const uint32_t first = s->bvh.a[i], cnt = s->bvh.flags[i] >> 2;
                if (cnt == 0) continue;
                HostAABB u;
                u.reset();
                bool sphere = false;
                for (uint32_t k = first; k < first + cnt && k < n_prims; ++k) {
                    const float4* pr = &prims[3 * (size_t)k];  // the slot's records, in slot order (sequential here; the host primitives are in caller order)
                    if (__builtin_bit_cast(uint32_t, pr[0].w) & PRIM_SPHERE) {
                        sphere = true;
                        break;
                    }
                    for (int j = 0; j < 3; ++j) {
                        const float v[3] = {pr[j].x, pr[j].y, pr[j].z};
                        u.grow_point(v);
                    }
                }
                if (sphere) continue;  // sphere leaves are reached through exact tests anyway
                const float* b = &s->bvh.bounds[6 * i];
                if (!(u.mn[0] == b[0] && u.mn[1] == b[1] && u.mn[2] == b[2] && u.mx[0] == b[3] && u.mx[1] == b[4] && u.mx[2] == b[5])) loose = true;
            }
        });
        const bool leaf_tight = !loose;
        if (ok) {
            if (int rc = upload(ctx, s->d_wnodes, wn.data(), wn.size() * sizeof(float4))) return rc;
            s->wide.leaf_tight = leaf_tight ? 1u : 0u;
            // the smallest sphere, in world units (half the smallest extent of its world bound): how far outside itself its Float32 quadratic can report a hit
            float r_min = INFINITY;
            for (const HostAABB& sb : s->sphere_bounds)
                for (int ax = 0; ax < 3; ++ax) r_min = std::fmin(r_min, 0.5f * (sb.mx[ax] - sb.mn[ax]));
            s->wide.sphere_lag = (s->sphere_bounds.empty() || !(r_min > 0.0f)) ? 0.0f : 3.8e-6f / r_min;
            s->wide.wnodes = (const float4*)s->d_wnodes.p;
            s->wide.n_wnodes = n_int;
            std::memcpy(s->wide.root_box, &s->bvh.bounds[0], 6 * sizeof(float));
            if ((s->bvh.flags[0] & 3u) == 3u) {
                s->wide.root_ref = s->bvh.a[0];
                s->wide.root_cnt = s->bvh.flags[0] >> 2;
                ok = s->wide.root_cnt > 0 && s->wide.root_cnt <= 255;
            } else {
                s->wide.root_ref = 0;
                s->wide.root_cnt = 0;
            }
            s->wide_ok = ok;
        }
    }
    clk.tick("upload: wnodes");
    // ---- 8-wide nodes over the triangles' subtree for k_trace8 (th_wide8.h) ----
    s->w8_ok = false;
    std::memset(&s->w8, 0, sizeof s->w8);
#ifdef TRHIP_EXPERIMENTS  // (traversal 4 is a kernel family of the EXPERIMENTS build: the default commit does not build its view)
    if (s->wide_ok && s->wide.root_cnt == 0 && n_nodes >= 3 && !has_empty_leaf) {
        // root of the triangles' subtree: the whole tree when the scene has no sphere; with spheres the commit composed
        // root -> {leaf of all spheres (flat node 1), triangles (flat node 2)} (compose_bvh)
        uint32_t sub_root = 0, n_sph = 0;
        bool shape_ok = true;
        if (!s->spheres.empty()) {
            n_sph = (uint32_t)s->spheres.size();
            sub_root = 2 * n_sph;
            shape_ok = n_sph <= (uint32_t)kW8MaxSpheres && sub_root < n_nodes;
            for (uint32_t i = 0; i < n_sph && shape_ok; ++i)
                shape_ok = (s->bvh.flags[2 * i] & 3u) != 3u && (s->bvh.flags[2 * i] & 3u) == (s->bvh.flags[0] & 3u) && s->bvh.a[2 * i] == 2 * i + 2 &&
                           s->bvh.flags[2 * i + 1] == ((1u << 2) | 3u) && s->bvh.a[2 * i + 1] == i && s->prims[s->bvh.order[i]].kind == 1;
            shape_ok = shape_ok && !has_sphere_subtree(s, sub_root);
        }
        if (shape_ok) {
            Wide8Host wh = build_wide8(s->bvh, sub_root, [&](uint32_t slot, float* v, uint32_t& meta) {
                if (slot >= n_prims) return false;
                const HostPrim& p = s->prims[s->bvh.order[slot]];
                if (p.kind != 0) return false;
                std::memcpy(v, p.v, 9 * sizeof(float));
                meta = p.meta;
                return true;
            });
            if (wh.ok) {
                if (int rc = upload(ctx, s->d_w8nodes, wh.nodes.data(), wh.nodes.size() * sizeof(uint32_t))) return rc;
                if (int rc = upload(ctx, s->d_w8tris, wh.tris.data(), wh.tris.size() * sizeof(float))) return rc;
                s->w8.nodes = (const uint4*)s->d_w8nodes.p;
                s->w8.tris = (const float4*)s->d_w8tris.p;
                std::memcpy(s->w8.root_box, &s->bvh.bounds[0], 6 * sizeof(float));
                std::memcpy(s->w8.tri_box, &s->bvh.bounds[6 * (size_t)sub_root], 6 * sizeof(float));
                for (uint32_t i = 0; i < n_sph; ++i) std::memcpy(s->w8.sph_box[i], &s->bvh.bounds[6 * (size_t)(2 * i + 1)], 6 * sizeof(float));
                s->w8.n_sph = n_sph;
                s->w8.chain_axis = s->bvh.flags[0] & 3u;
                s->w8_nodes = (uint32_t)(wh.nodes.size() / kW8NodeDwords);
                s->w8_depth = wh.depth;
                s->w8_ok = true;
            }
        }
    }
#endif
    clk.tick("upload: 8-wide view");
    // ---- one-leaf scenes: the boxes of the leaf's triangles, for the candidate masks of k_leaf_sorted (th_leaf2.h) ----
    release(s->d_leaf_boxes);
    if (s->wide_ok && s->wide.root_cnt > 0 && s->wide.root_cnt <= 30) {
        const uint32_t first = s->wide.root_ref, cnt = s->wide.root_cnt;
        std::vector<float> boxes(6 * (size_t)cnt, 0.0f);
        for (uint32_t k = 0; k < cnt && first + k < n_prims; ++k) {
            const float4* pr = &prims[3 * (size_t)(first + k)];
            if (__builtin_bit_cast(uint32_t, pr[0].w) & PRIM_SPHERE) continue;
            HostAABB u;
            u.reset();
            for (int j = 0; j < 3; ++j) {
                const float v[3] = {pr[j].x, pr[j].y, pr[j].z};
                u.grow_point(v);
            }
            std::memcpy(&boxes[6 * (size_t)k], u.mn, 3 * sizeof(float));
            std::memcpy(&boxes[6 * (size_t)k + 3], u.mx, 3 * sizeof(float));
        }
        if (int rc = upload(ctx, s->d_leaf_boxes, boxes.data(), boxes.size() * sizeof(float))) return rc;
    }
    // ---- one-leaf scenes: the order in which any-hit rays try the leaf's primitives (th_trace2.h, k_any_leaf) ----
    // A shadow ray runs from the surface THROUGH the light (t_max = Inf): what stops it at the latest is what the light sees, so the
    // primitives subtending the largest solid angle at the lights come first (triangles: Van Oosterom & Strackee; spheres: the cap of
    // their bounding sphere).  Any order gives the same boolean.
    s->wide.leaf_order = nullptr;
    if (s->wide_ok && s->wide.root_cnt > 1 && !s->lights.empty()) {
        const uint32_t first = s->wide.root_ref, cnt = s->wide.root_cnt;
        std::vector<std::pair<double, uint32_t>> ord;
        for (uint32_t k = 0; k < cnt; ++k) {
            const HostPrim& p = s->prims[s->bvh.order[first + k]];
            double w = 0.0;
            for (const LightRec& l : s->lights) {
                const float* lp = l.position;
                if (p.kind == 1) {
                    const HostAABB& b = s->sphere_bounds[p.sphere_id];
                    double c[3], r = 0.0, d2 = 0.0;
                    for (int a = 0; a < 3; ++a) {
                        c[a] = 0.5 * ((double)b.mn[a] + b.mx[a]);
                        r = std::max(r, 0.5 * ((double)b.mx[a] - b.mn[a]));
                        d2 += (c[a] - lp[a]) * (c[a] - lp[a]);
                    }
                    w += d2 <= r * r ? 4.0 * 3.14159265358979 : 2.0 * 3.14159265358979 * (1.0 - std::sqrt(std::max(0.0, 1.0 - r * r / d2)));
                } else {
                    double r[3][3], len[3];
                    for (int v = 0; v < 3; ++v) {
                        for (int c = 0; c < 3; ++c) r[v][c] = (double)p.v[3 * v + c] - lp[c];
                        len[v] = std::sqrt(r[v][0] * r[v][0] + r[v][1] * r[v][1] + r[v][2] * r[v][2]);
                    }
                    const double det = r[0][0] * (r[1][1] * r[2][2] - r[1][2] * r[2][1]) - r[0][1] * (r[1][0] * r[2][2] - r[1][2] * r[2][0]) + r[0][2] * (r[1][0] * r[2][1] - r[1][1] * r[2][0]);
                    auto dot3 = [&](int a, int b) { return r[a][0] * r[b][0] + r[a][1] * r[b][1] + r[a][2] * r[b][2]; };
                    const double den = len[0] * len[1] * len[2] + dot3(0, 1) * len[2] + dot3(0, 2) * len[1] + dot3(1, 2) * len[0];
                    w += 2.0 * std::fabs(std::atan2(det, den));
                }
            }
            ord.push_back({w, k});
        }
        std::stable_sort(ord.begin(), ord.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
        std::vector<uint32_t> order(cnt);
        for (uint32_t k = 0; k < cnt; ++k) order[k] = ord[k].second;
        if (int rc = upload(ctx, s->d_leaf_order, order.data(), order.size() * sizeof(uint32_t))) return rc;
        s->wide.leaf_order = (const uint32_t*)s->d_leaf_order.p;
    }
    // ---- largest triangles: the any-hit pre-pass (th_trace2.h, k_any_occluders) ----
    s->n_occluders = 0;
    if (s->wide_ok && s->wide.root_cnt == 0) {
        const float* rb = &s->bvh.bounds[0];
        const double ex = (double)rb[3] - rb[0], ey = (double)rb[4] - rb[1], ez = (double)rb[5] - rb[2];
        const double face = std::max(ex * ey, std::max(ex * ez, ey * ez));
        std::vector<std::pair<double, uint32_t>> big;  // (area, ordered slot)
        std::mutex big_lock;
        parallel_for(n_prims, [&](size_t k0, size_t k1) {
            std::vector<std::pair<double, uint32_t>> mine;
            for (size_t k = k0; k < k1; ++k) {
                const HostPrim& p = s->prims[s->bvh.order[k]];
                if (p.kind != 0 || (p.meta & PRIM_DEGENERATE)) continue;
                const double ax = (double)p.v[3] - p.v[0], ay = (double)p.v[4] - p.v[1], az = (double)p.v[5] - p.v[2];
                const double bx = (double)p.v[6] - p.v[0], by = (double)p.v[7] - p.v[1], bz = (double)p.v[8] - p.v[2];
                const double cx = ay * bz - az * by, cy = az * bx - ax * bz, cz = ax * by - ay * bx;
                const double area = 0.5 * std::sqrt(cx * cx + cy * cy + cz * cz);
                if (area >= 0.02 * face) mine.push_back({area, (uint32_t)k});
            }
            if (!mine.empty()) {
                std::lock_guard<std::mutex> g(big_lock);
                big.insert(big.end(), mine.begin(), mine.end());
            }
        });
        if (!big.empty() && big.size() * 8 <= (size_t)n_prims) {  // a few walls around much else; not a scene that consists of large triangles
            std::sort(big.begin(), big.end(), [](const auto& a, const auto& b) { return a.first > b.first || (a.first == b.first && a.second < b.second); });
            if (big.size() > 16) big.resize(16);
            // test order: a shadow ray runs from the surface THROUGH the light (t_max = Inf) — what stops it at the latest is what the light
            // sees, so the triangles subtending the largest solid angle at the lights come first (Van Oosterom & Strackee)
            auto solid_angle = [&](uint32_t k, const float* lp) {
                const HostPrim& p = s->prims[s->bvh.order[k]];
                double r[3][3], len[3];
                for (int v = 0; v < 3; ++v) {
                    for (int c = 0; c < 3; ++c) r[v][c] = (double)p.v[3 * v + c] - lp[c];
                    len[v] = std::sqrt(r[v][0] * r[v][0] + r[v][1] * r[v][1] + r[v][2] * r[v][2]);
                }
                const double det = r[0][0] * (r[1][1] * r[2][2] - r[1][2] * r[2][1]) - r[0][1] * (r[1][0] * r[2][2] - r[1][2] * r[2][0]) + r[0][2] * (r[1][0] * r[2][1] - r[1][1] * r[2][0]);
                auto dot3 = [&](int a, int b) { return r[a][0] * r[b][0] + r[a][1] * r[b][1] + r[a][2] * r[b][2]; };
                const double den = len[0] * len[1] * len[2] + dot3(0, 1) * len[2] + dot3(0, 2) * len[1] + dot3(1, 2) * len[0];
                return 2.0 * std::fabs(std::atan2(det, den));
            };
            if (!s->lights.empty()) {
                for (auto& b : big) {
                    double w = 0.0;
                    for (const LightRec& l : s->lights) w += solid_angle(b.second, l.position);
                    b.first = w;
                }
                std::stable_sort(big.begin(), big.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
            }
            // the leaf that holds each of the (at most 16) chosen slots
            std::vector<std::atomic<uint32_t>> leaf_of(big.size());
            for (auto& l : leaf_of) l = 0xffffffffu;
            parallel_for(n_nodes, [&](size_t i0, size_t i1) {
                for (size_t i = i0; i < i1; ++i) {
                    if ((s->bvh.flags[i] & 3u) != 3u) continue;
                    const uint32_t first = s->bvh.a[i], cnt = s->bvh.flags[i] >> 2;
                    for (size_t q = 0; q < big.size(); ++q)
                        if (big[q].second - first < cnt) leaf_of[q] = (uint32_t)i;
                }
            });
            std::vector<uint32_t> slots;
            std::vector<float> boxes;
            for (size_t q = 0; q < big.size(); ++q) {
                const uint32_t leaf = leaf_of[q];
                if (leaf == 0xffffffffu) continue;
                slots.push_back(big[q].second);
                for (int a = 0; a < 6; ++a) boxes.push_back(s->bvh.bounds[6 * (size_t)leaf + a]);
            }
            if (slots.size() >= 6) {  // an enclosure (three quads or more); a lone floor stops few shadow rays and the pre-pass only costs (S-caustic)
                if (int rc = upload(ctx, s->d_occ_slots, slots.data(), slots.size() * sizeof(uint32_t))) return rc;
                if (int rc = upload(ctx, s->d_occ_boxes, boxes.data(), boxes.size() * sizeof(float))) return rc;
                s->n_occluders = (uint32_t)slots.size();
            }
        }
    }
    clk.tick("upload: occluders");
    s->committed = true;
    return 0;
}

// ---- the accelerator of the hybrid mode (th_trace3c.h) ------------------------------------------------------------------------------------------------
// `s->bvh` (canonical: the reference's construction or the host's own tree) has been uploaded by upload_scene; `s->acc` is the library's tree over the same
// primitives (acc.order[k] = caller primitive).  Derives: the accelerator's primitive records in ITS leaf order with the canonical slot in the second
// record's .w lane, its children-in-parent nodes, the per-sphere / per-slot canonical leaf boxes the certificate reads — and verifies what the certificate
// assumes: every accelerator leaf has, bit for bit, the box of the canonical leaf of each of its primitives.  Anything that does not fit leaves
// hybrid_ok = false: every ray then walks the canonical tree (same answers).
static bool conform_accelerator(const FlatBVH& acc, const std::vector<uint32_t>& cslot, const std::vector<uint32_t>& leaf_of_slot, const std::vector<float>& slot_box, uint32_t n_prims,
                                FlatBVH& out) {
    const uint32_t n = (uint32_t)acc.a.size();
    out.bounds.reserve((size_t)n * 6 + 64);
    out.a.reserve(n + 16);
    out.flags.reserve(n + 16);
    out.order.reserve(n_prims);
    out.max_depth = 0;
    bool ok = true;
    auto new_node = [&]() {
        out.bounds.insert(out.bounds.end(), 6, 0.0f);
        out.a.push_back(0u);
        out.flags.push_back(0u);
        return (uint32_t)out.a.size() - 1u;
    };
    auto unite = [&](uint32_t dst, uint32_t l, uint32_t r) {
        for (int k = 0; k < 3; ++k) {
            out.bounds[6 * (size_t)dst + k] = std::fmin(out.bounds[6 * (size_t)l + k], out.bounds[6 * (size_t)r + k]);
            out.bounds[6 * (size_t)dst + 3 + k] = std::fmax(out.bounds[6 * (size_t)l + 3 + k], out.bounds[6 * (size_t)r + 3 + k]);
        }
    };
    std::vector<std::pair<uint32_t, uint32_t>> items;  // (canonical leaf, caller primitive) of one accelerator leaf
    // groups [g0, g1) of `items` (runs of one canonical leaf, bounded by `cuts`): one leaf each, a balanced subtree above them
    std::vector<uint32_t> cuts;
    std::function<uint32_t(uint32_t, uint32_t, uint32_t)> emit_groups = [&](uint32_t g0, uint32_t g1, uint32_t depth) -> uint32_t {
        out.max_depth = std::max(out.max_depth, depth);
        const uint32_t idx = new_node();
        if (g1 - g0 == 1) {
            const uint32_t b = cuts[g0], e = cuts[g0 + 1];
            if (e - b > 255u) ok = false;
            out.a[idx] = (uint32_t)out.order.size();
            out.flags[idx] = ((e - b) << 2) | 3u;
            for (uint32_t k = b; k < e; ++k) out.order.push_back(items[k].second);
            std::memcpy(&out.bounds[6 * (size_t)idx], &slot_box[6 * (size_t)cslot[items[b].second]], 6 * sizeof(float));
            return idx;
        }
        const uint32_t mid = g0 + (g1 - g0) / 2;
        const uint32_t l = emit_groups(g0, mid, depth + 1);
        const uint32_t r = emit_groups(mid, g1, depth + 1);
        out.a[idx] = r;
        out.flags[idx] = 0u;  // (any axis: the accelerator's visiting order does not matter)
        unite(idx, l, r);
        return idx;
    };
    std::function<uint32_t(uint32_t, uint32_t)> emit = [&](uint32_t i, uint32_t depth) -> uint32_t {
        if (!ok || depth > 200u) {
            ok = false;
            return 0u;
        }
        if ((acc.flags[i] & 3u) == 3u) {
            const uint32_t first = acc.a[i], cnt = acc.flags[i] >> 2;
            if (cnt == 0 || (uint64_t)first + cnt > n_prims) {
                ok = false;
                return 0u;
            }
            items.clear();
            for (uint32_t k = first; k < first + cnt; ++k) items.emplace_back(leaf_of_slot[cslot[acc.order[k]]], acc.order[k]);
            std::stable_sort(items.begin(), items.end(), [](const std::pair<uint32_t, uint32_t>& x, const std::pair<uint32_t, uint32_t>& y) { return x.first < y.first; });
            cuts.clear();
            for (uint32_t k = 0; k < cnt; ++k)
                if (k == 0 || items[k].first != items[k - 1].first) cuts.push_back(k);
            const uint32_t n_groups = (uint32_t)cuts.size();
            cuts.push_back(cnt);
            return emit_groups(0, n_groups, depth);
        }
        const uint32_t idx = new_node();
        if (acc.a[i] <= i + 1 || acc.a[i] >= n) {
            ok = false;
            return idx;
        }
        const uint32_t l = emit(i + 1, depth + 1);
        const uint32_t r = emit(acc.a[i], depth + 1);
        out.a[idx] = r;
        out.flags[idx] = acc.flags[i] & 3u;
        if (ok) unite(idx, l, r);
        return idx;
    };
    emit(0, 1);
    return ok && out.order.size() == n_prims && (out.flags[0] & 3u) != 3u;
}

static int upload_accelerator_impl(trhip_scene* s, bool conformed);
int upload_accelerator(trhip_scene* s) { return upload_accelerator_impl(s, false); }
static int upload_accelerator_conformed(trhip_scene* s) { return upload_accelerator_impl(s, true); }
static int upload_accelerator_impl(trhip_scene* s, bool conformed) {
    trhip_ctx* ctx = s->ctx;
    CommitClock clk;
    s->hybrid_ok = false;
    std::memset(&s->wide_acc, 0, sizeof s->wide_acc);
    s->wide_acc.root_ref = kRefNone;
    s->cert = CertScene{};
    s->dev_acc = s->dev;
    const uint32_t n_prims = (uint32_t)s->bvh.order.size(), n_cnodes = (uint32_t)s->bvh.a.size(), n_anodes = (uint32_t)s->acc.a.size();
    if (!s->wide_ok || s->literal_only || n_prims == 0 || n_anodes == 0 || s->acc.order.size() != n_prims || n_prims >= (1u << 24)) return 0;
    if (s->acc.max_depth > (uint32_t)(kStackLds + kStackSpill)) return 0;  // the certified walk's stack holds 64 entries like every other kernel's (a regrouped leaf adds levels)
    if (s->wide.root_cnt > 0) return 0;  // the canonical tree is one leaf: nothing to accelerate
    // canonical slot of every caller primitive, and the box of the canonical leaf that holds each canonical slot
    std::vector<uint32_t> cslot(s->prims.size(), 0xffffffffu);
    for (uint32_t k = 0; k < n_prims; ++k) cslot[s->bvh.order[k]] = k;
    std::vector<float> slot_box((size_t)n_prims * 6, 0.0f);
    std::vector<uint32_t> leaf_of_slot(n_prims, 0u);  // the canonical leaf (node index) that holds each canonical slot
    std::vector<uint8_t> covered(n_prims, 0);
    parallel_for(n_cnodes, [&](size_t i0, size_t i1) {
        for (size_t i = i0; i < i1; ++i) {
            if ((s->bvh.flags[i] & 3u) != 3u) continue;
            const uint32_t first = s->bvh.a[i], cnt = s->bvh.flags[i] >> 2;
            for (uint32_t k = first; k < first + cnt && k < n_prims; ++k) {
                std::memcpy(&slot_box[6 * (size_t)k], &s->bvh.bounds[6 * i], 6 * sizeof(float));
                leaf_of_slot[k] = (uint32_t)i;
                covered[k] = 1;
            }
        }
    });
    for (uint32_t k = 0; k < n_prims; ++k)
        if (!covered[k]) return 0;  // (a foreign tree that leaves a primitive out)
    for (uint32_t k = 0; k < n_prims; ++k)
        if (s->acc.order[k] >= s->prims.size() || cslot[s->acc.order[k]] == 0xffffffffu) return 0;
    // per sphere: its canonical leaf's box
    std::vector<float> sph_box(std::max<size_t>(1, s->spheres.size()) * 6, 0.0f);
    for (uint32_t k = 0; k < n_prims; ++k) {
        const HostPrim& p = s->prims[s->bvh.order[k]];
        if (p.kind == 1 && p.sphere_id < s->spheres.size()) std::memcpy(&sph_box[6 * (size_t)p.sphere_id], &slot_box[6 * (size_t)k], 6 * sizeof(float));
    }
    std::vector<uint32_t> sph_slot(std::max<size_t>(1, s->spheres.size()), 0u);
    for (uint32_t k = 0; k < n_prims; ++k) {
        const HostPrim& p = s->prims[s->bvh.order[k]];
        if (p.kind == 1 && p.sphere_id < s->spheres.size()) sph_slot[p.sphere_id] = k;
    }
    if (s->spheres.size() > kCertMaxSpheres) return 0;  // the certified walk tests EVERY sphere when it fetches a ray: a scene of many spheres keeps the canonical tree alone
    if (int rc = upload(ctx, s->d_sphere_boxes, sph_box.data(), sph_box.size() * sizeof(float))) return rc;
    if (int rc = upload(ctx, s->d_sphere_slots, sph_slot.data(), sph_slot.size() * sizeof(uint32_t))) return rc;
    s->cert.sphere_boxes = (const float*)s->d_sphere_boxes.p;
    s->cert.sphere_slots = (const uint32_t*)s->d_sphere_slots.p;
    s->cert.n_spheres = (uint32_t)s->spheres.size();
    {
        std::vector<SphereCert> sc(std::max<size_t>(1, s->spheres.size()));
        for (size_t k = 0; k < s->spheres.size(); ++k) {
            std::memcpy(sc[k].box, &sph_box[6 * k], 6 * sizeof(float));
            sc[k].radius = s->spheres[k].radius;
            sc[k].slot = sph_slot[k];
            std::memcpy(sc[k].o2w_inv, s->spheres[k].o2w_inv, 16 * sizeof(float));
            sc[k].never_clipped = s->spheres[k].never_clipped;
            sc[k].pad[0] = sc[k].pad[1] = sc[k].pad[2] = 0u;
        }
        if (int rc = upload(ctx, s->d_sphere_cert, sc.data(), sc.size() * sizeof(SphereCert))) return rc;
        s->cert.sphere_cert = s->d_sphere_cert.p;
    }
    std::memcpy(s->wide_acc.root_box, &s->bvh.bounds[0], 6 * sizeof(float));  // the union of all primitives: the same box in every tree (bvh.jl:226 tests it first)
    const bool one_leaf = n_anodes == 1 && (s->acc.flags[0] & 3u) == 3u;
    if (one_leaf) {
        // every primitive in one leaf: walked over the CANONICAL records in canonical slot order (k_trace_leaf_c / k_any_leaf_c)
        if (n_prims > 255) return 0;
        if (int rc = upload(ctx, s->d_slot_boxes, slot_box.data(), slot_box.size() * sizeof(float))) return rc;
        s->cert.slot_boxes = (const float*)s->d_slot_boxes.p;
        s->wide_acc.root_ref = 0;
        s->wide_acc.root_cnt = n_prims;
        s->wide_acc.leaf_order = nullptr;
        if (n_prims > 1 && !s->lights.empty()) {  // any-hit rays try what subtends the largest solid angle at the lights first (as k_any_leaf: any order gives the same boolean)
            std::vector<std::pair<double, uint32_t>> ord;
            for (uint32_t k = 0; k < n_prims; ++k) {
                const HostPrim& p = s->prims[s->bvh.order[k]];
                double w = 0.0;
                for (const LightRec& l : s->lights) {
                    const float* lp = l.position;
                    if (p.kind == 1) {
                        const HostAABB& b = s->sphere_bounds[p.sphere_id];
                        double c[3], r = 0.0, d2 = 0.0;
                        for (int a = 0; a < 3; ++a) {
                            c[a] = 0.5 * ((double)b.mn[a] + b.mx[a]);
                            r = std::max(r, 0.5 * ((double)b.mx[a] - b.mn[a]));
                            d2 += (c[a] - lp[a]) * (c[a] - lp[a]);
                        }
                        w += d2 <= r * r ? 4.0 * 3.14159265358979 : 2.0 * 3.14159265358979 * (1.0 - std::sqrt(std::max(0.0, 1.0 - r * r / d2)));
                    } else {
                        double r[3][3], len[3];
                        for (int v = 0; v < 3; ++v) {
                            for (int c = 0; c < 3; ++c) r[v][c] = (double)p.v[3 * v + c] - lp[c];
                            len[v] = std::sqrt(r[v][0] * r[v][0] + r[v][1] * r[v][1] + r[v][2] * r[v][2]);
                        }
                        const double det = r[0][0] * (r[1][1] * r[2][2] - r[1][2] * r[2][1]) - r[0][1] * (r[1][0] * r[2][2] - r[1][2] * r[2][0]) + r[0][2] * (r[1][0] * r[2][1] - r[1][1] * r[2][0]);
                        auto dot3 = [&](int a, int b) { return r[a][0] * r[b][0] + r[a][1] * r[b][1] + r[a][2] * r[b][2]; };
                        const double den = len[0] * len[1] * len[2] + dot3(0, 1) * len[2] + dot3(0, 2) * len[1] + dot3(1, 2) * len[0];
                        w += 2.0 * std::fabs(std::atan2(det, den));
                    }
                }
                ord.push_back({w, k});
            }
            std::stable_sort(ord.begin(), ord.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
            std::vector<uint32_t> order(n_prims);
            for (uint32_t k = 0; k < n_prims; ++k) order[k] = ord[k].second;
            if (int rc = upload(ctx, s->d_acc_leaf_order, order.data(), order.size() * sizeof(uint32_t))) return rc;
            s->wide_acc.leaf_order = (const uint32_t*)s->d_acc_leaf_order.p;
        }
        s->hybrid_ok = true;
        clk.tick("accelerator: one leaf");
        return 0;
    }
    // ---- a hierarchy: leaf boxes must be the canonical leaves' ----
    std::atomic<bool> bad{false};
    parallel_for(n_anodes, [&](size_t i0, size_t i1) {
        for (size_t i = i0; i < i1 && !bad; ++i) {
            if ((s->acc.flags[i] & 3u) != 3u) {
                if (s->acc.a[i] <= i + 1 || s->acc.a[i] >= n_anodes) bad = true;
                continue;
            }
            const uint32_t first = s->acc.a[i], cnt = s->acc.flags[i] >> 2;
            if (cnt == 0 || cnt > 255 || (uint64_t)first + cnt > n_prims) {
                bad = true;
                continue;
            }
            for (uint32_t k = first; k < first + cnt; ++k)
                if (std::memcmp(&s->acc.bounds[6 * i], &slot_box[6 * (size_t)cslot[s->acc.order[k]]], 6 * sizeof(float)) != 0) bad = true;
        }
    });
    if ((s->acc.flags[0] & 3u) == 3u) return 0;
    if (bad && conformed) return 0;  // (cannot happen: the conformed tree has the canonical leaf boxes by construction)
    if (bad) {
        // The library's builder draws its leaves where ITS cost function says (primitives with coincident centroids share one; a node that is not worth splitting stays
        // whole), the reference's where its own does.  The certificate needs every accelerator leaf to be (part of) ONE canonical leaf, under that leaf's box: the
        // accelerator is ours to change — its leaves are regrouped by canonical leaf (a leaf that straddles several becomes a small subtree), every leaf takes its canonical
        // leaf's box, the interior boxes are re-derived bottom-up (so they still nest).  A looser leaf box costs the accelerator a few visits, never an answer.
        FlatBVH out;
        if (!conform_accelerator(s->acc, cslot, leaf_of_slot, slot_box, n_prims, out)) return 0;
        s->acc = std::move(out);
        return upload_accelerator_conformed(s);
    }
    if (std::memcmp(&s->acc.bounds[0], &s->bvh.bounds[0], 6 * sizeof(float)) != 0) return 0;
    clk.tick("accelerator: leaf boxes");
    // ---- per canonical slot, per sphere: where the reference's walk meets the primitive relative to the sphere (th_trace3c.h: a ray that starts inside a sphere only counts what
    //      the reference tests AFTER that sphere).  3 bits per sphere: the split axis of the canonical node where the two root paths part (3: they share a leaf) and whether the
    //      primitive is in that node's SECOND child (in a shared leaf: behind the sphere).  The depth-first layout makes a subtree's slots one contiguous range. ----
    std::vector<uint32_t> order_word(n_prims, 0u);
    {
        auto leftmost = [&](uint32_t i, uint32_t& slot) {
            for (uint32_t guard = 0; guard < 4096u && i < n_cnodes; ++guard) {
                if ((s->bvh.flags[i] & 3u) == 3u) {
                    slot = s->bvh.a[i];
                    return true;
                }
                ++i;
            }
            return false;
        };
        for (uint32_t sid = 0; sid < std::min<uint32_t>((uint32_t)s->spheres.size(), kCertOrderSpheres); ++sid) {  // (30 of the word's 32 bits; later spheres have no order bits: th_trace3c.h)
            const uint32_t ks = sph_slot[sid], sh = 3u * sid;
            uint32_t i = 0, lo = 0, hi = n_prims;
            bool ok = false;
            for (uint32_t guard = 0; guard < 4096u && i < n_cnodes; ++guard) {
                if ((s->bvh.flags[i] & 3u) == 3u) {
                    const uint32_t first = s->bvh.a[i], cnt = s->bvh.flags[i] >> 2;
                    ok = ks - first < cnt && first == lo && first + cnt == hi;
                    for (uint32_t k = first; ok && k < first + cnt; ++k)
                        if (k != ks) order_word[k] |= (3u | (k > ks ? 4u : 0u)) << sh;
                    break;
                }
                const uint32_t second = s->bvh.a[i], axis = s->bvh.flags[i] & 3u;
                uint32_t lo2 = 0;
                if (second <= i + 1 || second >= n_cnodes || !leftmost(second, lo2) || lo2 < lo || lo2 > hi) break;
                if (ks >= lo2) {
                    for (uint32_t k = lo; k < lo2; ++k) order_word[k] |= axis << sh;
                    lo = lo2;
                    i = second;
                } else {
                    for (uint32_t k = lo2; k < hi; ++k) order_word[k] |= (axis | 4u) << sh;
                    hi = lo2;
                    i = i + 1;
                }
            }
            if (!ok) return 0;  // (a layout this walk does not understand: the canonical tree alone)
        }
    }
    // primitive records in the accelerator's order; the canonical slot rides in the second record's .w lane, the order word in the third's
    RawArray<float4> aprims((size_t)n_prims * 3);
    std::vector<uint8_t> is_sphere(n_prims, 0);
    parallel_for(n_prims, [&](size_t k0, size_t k1) {
        for (size_t k = k0; k < k1; ++k) {
            const HostPrim& p = s->prims[s->acc.order[k]];
            const float cs = __builtin_bit_cast(float, cslot[s->acc.order[k]]);
            if (p.kind == 1) {
                aprims[3 * k] = make_float4(__builtin_bit_cast(float, p.sphere_id), 0, 0, __builtin_bit_cast(float, p.meta));
                aprims[3 * k + 1] = make_float4(0, 0, 0, cs);
                aprims[3 * k + 2] = make_float4(0, 0, 0, 0);
                is_sphere[k] = 1;
            } else {
                aprims[3 * k] = make_float4(p.v[0], p.v[1], p.v[2], __builtin_bit_cast(float, p.meta));
                aprims[3 * k + 1] = make_float4(p.v[3], p.v[4], p.v[5], cs);
                aprims[3 * k + 2] = make_float4(p.v[6], p.v[7], p.v[8], __builtin_bit_cast(float, order_word[cslot[s->acc.order[k]]]));
            }
        }
    });
    if (int rc = upload(ctx, s->d_acc_prims, aprims.data(), aprims.size() * sizeof(float4))) return rc;
    clk.tick("accelerator: primitive records");
    // ---- what the certificate's lower bound needs of the TRIANGLES (th_trace3c.h; spheres are never hidden from the walk): per axis the largest extent of a NON-FLAT triangle
    //      (the computed t of a primitive differs from the depth of the ray's point on it by at most the primitive's extent along the ray's dominant axis), and for FLAT ones
    //      (zero extent in some axis: walls, floors — of any size) the largest L^3 / (2 A) (how far the edge functions' rounding can move that point) ----
    {
        std::mutex lock;
        float mle[3] = {0.0f, 0.0f, 0.0f}, sq = 0.0f;
        parallel_for(s->prims.size(), [&](size_t i0, size_t i1) {
            float m[3] = {0.0f, 0.0f, 0.0f}, q = 0.0f;
            for (size_t i = i0; i < i1; ++i) {
                const HostPrim& p = s->prims[i];
                if (p.kind != 0 || (p.meta & PRIM_DEGENERATE)) continue;
                float e[3];
                for (int c = 0; c < 3; ++c) e[c] = std::fmax(std::fmax(p.v[c], p.v[3 + c]), p.v[6 + c]) - std::fmin(std::fmin(p.v[c], p.v[3 + c]), p.v[6 + c]);
                if (e[0] == 0.0f || e[1] == 0.0f || e[2] == 0.0f) {
                    double ab[3], ac[3];
                    for (int c = 0; c < 3; ++c) {
                        ab[c] = (double)p.v[3 + c] - p.v[c];
                        ac[c] = (double)p.v[6 + c] - p.v[c];
                    }
                    const double bc[3] = {ac[0] - ab[0], ac[1] - ab[1], ac[2] - ab[2]};
                    const double l2 = std::max({ab[0] * ab[0] + ab[1] * ab[1] + ab[2] * ab[2], ac[0] * ac[0] + ac[1] * ac[1] + ac[2] * ac[2], bc[0] * bc[0] + bc[1] * bc[1] + bc[2] * bc[2]});
                    const double cx = ab[1] * ac[2] - ab[2] * ac[1], cy = ab[2] * ac[0] - ab[0] * ac[2], cz = ab[0] * ac[1] - ab[1] * ac[0];
                    const double area2 = std::sqrt(cx * cx + cy * cy + cz * cz);  // 2 A
                    if (area2 > 0.0) q = std::fmax(q, (float)(l2 * std::sqrt(l2) / area2) * 1.0001f);
                } else {
                    for (int a = 0; a < 3; ++a)
                        if (e[a] < INFINITY) m[a] = std::fmax(m[a], e[a]);
                }
            }
            std::lock_guard<std::mutex> g(lock);
            for (int a = 0; a < 3; ++a) mle[a] = std::fmax(mle[a], m[a]);
            sq = std::fmax(sq, q);
        });
        for (int a = 0; a < 3; ++a) s->cert.mle_small[a] = mle[a];
        s->cert.sq_flat = sq;
        if (std::getenv("TRHIP_COMMIT_TIMING")) std::fprintf(stderr, "[commit] certificate: mle %g %g %g, sq_flat %g\n", mle[0], mle[1], mle[2], sq);
    }
    // children-in-parent nodes (as upload_scene's, th_trace2.h)
    std::vector<uint32_t> widx;
    const uint32_t n_int = wide_node_order(s->acc, ctx->node_layout, widx);
    if (n_int >= (1u << 24)) return 0;
    std::vector<uint8_t> has_sphere(n_anodes, 0);
    for (uint32_t i = n_anodes; i-- > 0;) {  // bottom-up (second children and first children both come later in the depth-first layout)
        if ((s->acc.flags[i] & 3u) == 3u) {
            const uint32_t first = s->acc.a[i], cnt = s->acc.flags[i] >> 2;
            if (!s->spheres.empty())
                for (uint32_t k = first; k < first + cnt; ++k) has_sphere[i] |= is_sphere[k];
        } else {
            has_sphere[i] = has_sphere[i + 1] | has_sphere[s->acc.a[i]];
        }
    }
    RawArray<float4> wn((size_t)n_int * 4);
    if (ctx->node_layout == 1) std::memset((void*)wn.data(), 0, (size_t)n_int * 4 * sizeof(float4));
    parallel_for(n_anodes, [&](size_t i0, size_t i1) {
        for (size_t i = i0; i < i1; ++i) {
            if ((s->acc.flags[i] & 3u) == 3u) continue;
            const uint32_t c[2] = {(uint32_t)i + 1, s->acc.a[i]};
            uint32_t ref[2], cnt[2];
            for (int k = 0; k < 2; ++k) {
                if ((s->acc.flags[c[k]] & 3u) == 3u) {
                    ref[k] = s->acc.a[c[k]];
                    cnt[k] = s->acc.flags[c[k]] >> 2;
                } else {
                    ref[k] = widx[c[k]];
                    cnt[k] = 0;
                }
            }
            const float* l = &s->acc.bounds[6 * (size_t)c[0]];
            const float* r = &s->acc.bounds[6 * (size_t)c[1]];
            float4* w = &wn[4 * (size_t)widx[i]];
            // the ACCELERATOR's layout: each axis' two planes side by side — {min x, max x, min y, max y}, {min z, max z | min x, max x}, {min y, max y, min z, max z} — so that
            // k_trace3c's packed instructions take a pair as it was loaded (th_trace3c.h "The step"); the canonical tree's nodes keep {min xyz, max xyz} (upload_scene)
            w[0] = make_float4(l[0], l[3], l[1], l[4]);
            w[1] = make_float4(l[2], l[5], r[0], r[3]);
            w[2] = make_float4(r[1], r[4], r[2], r[5]);
            // child word = ref | count << 24 (the stack entry format); meta = split axis | "subtree holds a sphere" bits 2 (first) / 3 (second)
            w[3] = make_float4(__builtin_bit_cast(float, ref[0] | (cnt[0] << 24)), __builtin_bit_cast(float, ref[1] | (cnt[1] << 24)),
                               __builtin_bit_cast(float, (s->acc.flags[i] & 3u) | ((uint32_t)has_sphere[c[0]] << 2) | ((uint32_t)has_sphere[c[1]] << 3)), 0.0f);
        }
    });
    if (int rc = upload(ctx, s->d_acc_wnodes, wn.data(), wn.size() * sizeof(float4))) return rc;
    s->wide_acc.wnodes = (const float4*)s->d_acc_wnodes.p;
    s->wide_acc.n_wnodes = n_int;
    // ---- the same tree FOUR children wide (th_trace3c4.h): every other level of the binary tree folded into its parent, the widest child first (by surface area) while fewer
    //      than four slots are taken.  128 bytes per node = one L2 line: {min, max} pairs per axis of child 0 … 3 (six float4), the four child words (leaf: first primitive |
    //      count << 24 — the accelerator's leaves, i.e. parts of canonical leaves with exactly their boxes, untouched; interior: index of the 4-wide node), one spare float4.
    //      An empty slot holds NaN planes (every comparison false: never entered).  The certified walk is free in its order and in its topology (th_trace3c.h header): half
    //      the dependent node fetches per ray.
    s->wide_acc.w4nodes = nullptr;
    s->wide_acc.n_w4nodes = 0;
    if (ctx->wide4) {
        struct Frame { uint32_t node, out; };
        std::vector<float4> w4;
        w4.reserve((size_t)n_int / 2 * 8 + 64);
        std::vector<Frame> todo;
        todo.push_back({0u, 0u});
        w4.resize(8);
        uint32_t max_depth4 = 0;
        std::vector<uint32_t> depth4{1u};  // per 4-wide node
        bool ok4 = (s->acc.flags[0] & 3u) != 3u;
        auto area = [&](uint32_t n) {
            const float* b = &s->acc.bounds[6 * (size_t)n];
            const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
            return dx * dy + dx * dz + dy * dz;
        };
        // the first nodes breadth-first — the root, its children, theirs: the first 85 indices, of which k_trace3c4 keeps as many as fit in LDS (TH_TRACE3C4_TOP) —, the rest depth-first
        // (a subtree's nodes near each other)
        size_t bfs_head = 0;
        while (ok4 && bfs_head < todo.size()) {
            Frame f;
            if (w4.size() / 8 < 85u) {
                f = todo[bfs_head++];  // (first in, first out while the top is being numbered)
            } else {
                f = todo.back();
                todo.pop_back();
            }
            uint32_t kids[4];
            int nk = 2;
            kids[0] = f.node + 1;
            kids[1] = s->acc.a[f.node];
            while (nk < 4) {
                int best = -1;
                float best_a = -1.0f;
                for (int k = 0; k < nk; ++k)
                    if ((s->acc.flags[kids[k]] & 3u) != 3u && area(kids[k]) > best_a) {
                        best_a = area(kids[k]);
                        best = k;
                    }
                if (best < 0) break;
                const uint32_t c = kids[best];
                kids[best] = c + 1;
                kids[nk++] = s->acc.a[c];
            }
            float4 rec[8];
            const float qn = std::nanf("");
            float planes[24];
            uint32_t enc[4] = {1u << 24, 1u << 24, 1u << 24, 1u << 24};
            for (int k = 0; k < 24; ++k) planes[k] = qn;
            for (int k = 0; k < nk; ++k) {
                const uint32_t c = kids[k];
                const float* b = &s->acc.bounds[6 * (size_t)c];
                for (int a = 0; a < 3; ++a) {
                    planes[6 * k + 2 * a] = b[a];
                    planes[6 * k + 2 * a + 1] = b[3 + a];
                }
                if ((s->acc.flags[c] & 3u) == 3u) {
                    const uint32_t cnt = s->acc.flags[c] >> 2;
                    if (cnt == 0 || cnt > 255) ok4 = false;
                    enc[k] = s->acc.a[c] | (cnt << 24);
                } else {
                    const uint32_t idx = (uint32_t)(w4.size() / 8);
                    if (idx >= (1u << 24)) ok4 = false;
                    enc[k] = idx;
                    w4.resize(w4.size() + 8);
                    depth4.push_back(depth4[f.out] + 1);
                    max_depth4 = std::max(max_depth4, depth4.back());
                    todo.push_back({c, idx});
                }
            }
            for (int j = 0; j < 6; ++j) rec[j] = make_float4(planes[4 * j], planes[4 * j + 1], planes[4 * j + 2], planes[4 * j + 3]);
            rec[6] = make_float4(__builtin_bit_cast(float, enc[0]), __builtin_bit_cast(float, enc[1]), __builtin_bit_cast(float, enc[2]), __builtin_bit_cast(float, enc[3]));
            rec[7] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            std::memcpy(&w4[8 * (size_t)f.out], rec, sizeof rec);
        }
        // the walk pushes up to three entries per level: its 64-entry stack holds a tree of at most 21 four-wide levels (a deeper one keeps the binary walk)
#if defined(TH_TRACE3C4_CHEAP) && TH_TRACE3C4_CHEAP == 3
        // EXPERIMENT (th_trace3c4.h CHEAP = 3): the same nodes in 64 bytes — {lo.xyz, scale.x}, 24 plane bytes (child k, axis a: byte 6 k + 2 a the low plane, + 1 the high one) +
        // {scale.y, scale.z}, four child words (an empty slot: kRefNone) —, plane = lo + q x scale with a power-of-two scale per axis, low planes rounded down, high planes up
        if (ok4 && 3 * (max_depth4 + 1) <= (uint32_t)kStack2Total) {
            const size_t nn4 = w4.size() / 8;
            std::vector<float4> q4(nn4 * 4);
            for (size_t n = 0; n < nn4 && ok4; ++n) {
                const float* pl = reinterpret_cast<const float*>(&w4[8 * n]);        // 24 planes: child k at 6 k: x0 x1 y0 y1 z0 z1 (NaN: empty)
                const uint32_t* wd = reinterpret_cast<const uint32_t*>(&w4[8 * n + 6]);
                float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
                bool present[4];
                for (int k = 0; k < 4; ++k) {
                    present[k] = !(pl[6 * k] != pl[6 * k]);
                    if (!present[k]) continue;
                    for (int a = 0; a < 3; ++a) {
                        lo[a] = std::fmin(lo[a], pl[6 * k + 2 * a]);
                        hi[a] = std::fmax(hi[a], pl[6 * k + 2 * a + 1]);
                    }
                }
                float scale[3];
                uint8_t qb[24];
                std::memset(qb, 0, sizeof qb);
                for (int a = 0; a < 3; ++a) {
                    const double ext = (double)hi[a] - (double)lo[a];
                    int e = ext > 0.0 ? (int)std::ceil(std::log2(ext / 255.0)) : -120;
                    e = std::max(-120, std::min(120, e));
                    for (int tries = 0; tries < 4; ++tries) {
                        const double sc = std::ldexp(1.0, e);
                        bool fits = true;
                        for (int k = 0; k < 4 && fits; ++k) {
                            if (!present[k]) continue;
                            double ql = std::floor(((double)pl[6 * k + 2 * a] - (double)lo[a]) / sc), qh = std::ceil(((double)pl[6 * k + 2 * a + 1] - (double)lo[a]) / sc);
                            while (ql > 0.0 && (double)lo[a] + ql * sc > (double)pl[6 * k + 2 * a]) ql -= 1.0;
                            while ((double)lo[a] + qh * sc < (double)pl[6 * k + 2 * a + 1]) qh += 1.0;
                            if (ql < 0.0) ql = 0.0;
                            if (qh > 255.0) {
                                fits = false;
                                break;
                            }
                            qb[6 * k + 2 * a] = (uint8_t)ql;
                            qb[6 * k + 2 * a + 1] = (uint8_t)qh;
                        }
                        if (fits) break;
                        ++e;
                        if (tries == 3) ok4 = false;
                    }
                    scale[a] = (float)std::ldexp(1.0, e);
                }
                uint32_t dw[6];
                std::memcpy(dw, qb, sizeof dw);
                uint32_t words[4];
                for (int k = 0; k < 4; ++k) words[k] = present[k] ? wd[k] : kRefNone;
                q4[4 * n] = make_float4(lo[0], lo[1], lo[2], scale[0]);
                q4[4 * n + 1] = make_float4(__builtin_bit_cast(float, dw[0]), __builtin_bit_cast(float, dw[1]), __builtin_bit_cast(float, dw[2]), __builtin_bit_cast(float, dw[3]));
                q4[4 * n + 2] = make_float4(__builtin_bit_cast(float, dw[4]), __builtin_bit_cast(float, dw[5]), scale[1], scale[2]);
                q4[4 * n + 3] = make_float4(__builtin_bit_cast(float, words[0]), __builtin_bit_cast(float, words[1]), __builtin_bit_cast(float, words[2]), __builtin_bit_cast(float, words[3]));
            }
            if (ok4) {
                if (int rc = upload(ctx, s->d_acc_w4nodes, q4.data(), q4.size() * sizeof(float4))) return rc;
                s->wide_acc.w4nodes = (const float4*)s->d_acc_w4nodes.p;
                s->wide_acc.n_w4nodes = (uint32_t)nn4;
            }
        }
#else
        if (ok4 && 3 * (max_depth4 + 1) <= (uint32_t)kStack2Total) {
            if (int rc = upload(ctx, s->d_acc_w4nodes, w4.data(), w4.size() * sizeof(float4))) return rc;
            s->wide_acc.w4nodes = (const float4*)s->d_acc_w4nodes.p;
            s->wide_acc.n_w4nodes = (uint32_t)(w4.size() / 8);
        }
#endif
        clk.tick("accelerator: 4-wide nodes");
    }
    s->wide_acc.root_ref = 0;
    s->wide_acc.root_cnt = 0;
    s->wide_acc.leaf_tight = s->wide.leaf_tight;
    s->dev_acc.prims = (const float4*)s->d_acc_prims.p;
    s->hybrid_ok = true;
    clk.tick("accelerator: wnodes");
    return 0;
}

extern "C" {

int trhip_scene_new(trhip_ctx* ctx, trhip_scene** out) {
    if (!ctx || !out) return fail(ctx, TRHIP_ERR_INVALID, "null argument");
    auto s = new trhip_scene();
    s->ctx = ctx;
    *out = s;
    return 0;
}
void trhip_scene_free(trhip_scene* s) {
    if (!s) return;
    (void)hipSetDevice(s->ctx->device);
    release(s->d_nodes);
    release(s->d_prims);
    release(s->d_nrm);
    release(s->d_leaf_boxes);
    release(s->d_tan);
    release(s->d_shade);
    release(s->d_leaf_order);
    release(s->d_spheres);
    release(s->d_materials);
    release(s->d_lights);
    release(s->d_wnodes);
    release(s->d_occ_slots);
    release(s->d_occ_boxes);
    release(s->d_w8nodes);
    release(s->d_w8tris);
    release(s->d_acc_wnodes);
    release(s->d_acc_w4nodes);
    release(s->d_acc_prims);
    release(s->d_slot_boxes);
    release(s->d_sphere_boxes);
    release(s->d_sphere_slots);
    release(s->d_sphere_cert);
    release(s->d_acc_leaf_order);
    delete s;
}
int trhip_scene_add_material(trhip_scene* s, int kind, const float* params, int n_params, uint32_t* id_out) {
    if (!s || !params) return fail(s ? s->ctx : nullptr, TRHIP_ERR_INVALID, "null argument");
    MaterialRec m;
    if (build_material(kind, params, n_params, m)) return fail(s->ctx, TRHIP_ERR_INVALID, "bad material kind %d / parameter count %d", kind, n_params);
    if (s->materials.size() >= PRIM_NO_MATERIAL) return fail(s->ctx, TRHIP_ERR_INVALID, "too many materials");
    s->materials.push_back(m);
    if (id_out) *id_out = (uint32_t)s->materials.size() - 1;
    s->committed = false;
    return 0;
}
int trhip_scene_add_triangles(trhip_scene* s, const float* xyz, uint32_t n_verts, const uint32_t* idx, uint32_t n_tris, const float* normals, const uint32_t* mat, int flip,
                              uint32_t* first_out) {
    return trhip_scene_add_triangles_ex(s, xyz, n_verts, idx, n_tris, normals, nullptr, nullptr, mat, flip, first_out);
}
int trhip_scene_add_triangles_ex(trhip_scene* s, const float* xyz, uint32_t n_verts, const uint32_t* idx, uint32_t n_tris, const float* normals, const float* tangents, const float* uv,
                                 const uint32_t* mat, int flip, uint32_t* first_out) {
    if (!s || !xyz || !idx) return fail(s ? s->ctx : nullptr, TRHIP_ERR_INVALID, "null argument");
    const uint32_t first = (uint32_t)s->prims.size();
    // validate first (nothing is added when an index or a material is out of range), then fill the records on all cores
    std::atomic<uint32_t> bad_tri{0xffffffffu};
    parallel_for(n_tris, [&](size_t k0, size_t k1) {
        for (size_t k = k0; k < k1; ++k) {
            bool ok = true;
            for (int j = 0; j < 3; ++j) ok = ok && idx[3 * k + j] >= 1 && idx[3 * k + j] <= n_verts;
            if (mat && mat[k] != PRIM_NO_MATERIAL && mat[k] >= s->materials.size()) ok = false;
            if (!ok) {
                uint32_t cur = bad_tri.load();
                while ((uint32_t)k < cur && !bad_tri.compare_exchange_weak(cur, (uint32_t)k)) {
                }
                return;
            }
        }
    });
    if (bad_tri != 0xffffffffu) {
        const uint32_t k = bad_tri;
        for (int j = 0; j < 3; ++j) {
            const uint32_t vi = idx[3 * (size_t)k + j];
            if (vi < 1 || vi > n_verts) return fail(s->ctx, TRHIP_ERR_INVALID, "triangle %u: index %u outside 1..%u (indices are 1-based)", k, vi, n_verts);
        }
        return fail(s->ctx, TRHIP_ERR_INVALID, "triangle %u: material %u not defined", k, mat[k]);
    }
    s->prims.resize((size_t)first + n_tris);
    // the side arrays exist from the first mesh that brings tangents / (u, v)s on, one entry per primitive of the scene (zeros for the others)
    if (tangents || !s->prim_tan.empty()) s->prim_tan.resize(9 * ((size_t)first + n_tris), 0.0f);
    if (uv || !s->prim_uv.empty()) s->prim_uv.resize(7 * ((size_t)first + n_tris), 0.0f);
    parallel_for(n_tris, [&](size_t k0, size_t k1) {
        for (size_t k = k0; k < k1; ++k) {
            HostPrim p;
            std::memset(&p, 0, sizeof p);
            p.kind = 0;
            for (int j = 0; j < 3; ++j) {
                const uint32_t vi = idx[3 * k + j];
                std::memcpy(&p.v[3 * j], &xyz[3 * (size_t)(vi - 1)], 3 * sizeof(float));
                if (normals) std::memcpy(&p.n[3 * j], &normals[3 * (size_t)(vi - 1)], 3 * sizeof(float));
                if (tangents) std::memcpy(&s->prim_tan[9 * ((size_t)first + k) + 3 * j], &tangents[3 * (size_t)(vi - 1)], 3 * sizeof(float));
            }
            if (uv) {  // mesh.uv[t.i + j]: by corner position, not through the indices (triangle_mesh.jl:82)
                float* dst = &s->prim_uv[7 * ((size_t)first + k)];
                std::memcpy(dst, &uv[6 * k], 6 * sizeof(float));
                dst[6] = 1.0f;
            }
            const uint32_t m = mat ? mat[k] : PRIM_NO_MATERIAL;
            // is_degenerate (triangle_mesh.jl:65-68) depends on the triangle alone: evaluated here, once, in the kernels' arithmetic
            const f3 tv0 = mk3(p.v[0], p.v[1], p.v[2]), tv1 = mk3(p.v[3], p.v[4], p.v[5]), tv2 = mk3(p.v[6], p.v[7], p.v[8]);
            const f3 tn = cross(tv2 - tv0, tv1 - tv0);
            const bool degenerate = dot(tn, tn) == 0.0f;
            p.meta = (m & PRIM_MATERIAL_MASK) | (normals ? PRIM_HAS_NORMALS : 0u) | (tangents ? PRIM_HAS_TANGENTS : 0u) | (flip ? PRIM_FLIP : 0u) | (degenerate ? PRIM_DEGENERATE : 0u);
            s->prims[(size_t)first + k] = p;
        }
    });
    if (first_out) *first_out = first;
    s->committed = false;
    return 0;
}
static int add_sphere_rec(trhip_scene* s, const float* o2w, const float* o2w_inv, int reverse, SphereRec r, uint32_t material, uint32_t* prim_out) {
    if (material != PRIM_NO_MATERIAL && material >= s->materials.size()) return fail(s->ctx, TRHIP_ERR_INVALID, "material %u not defined", material);
    std::memcpy(r.o2w, o2w, sizeof r.o2w);
    std::memcpy(r.o2w_inv, o2w_inv, sizeof r.o2w_inv);
    const bool swaps = det3(o2w) < 0.0f;  // transformations.jl:161-163
    r.flip = ((reverse != 0) != swaps) ? 1u : 0u;
    r.never_clipped = (!(r.z_min > -r.radius) && !(r.z_max < r.radius) && r.phi_max >= 2.0f * kPi) ? 1u : 0u;
    if (!r.never_clipped) s->partial_spheres = true;
    HostPrim p;
    std::memset(&p, 0, sizeof p);
    p.kind = 1;
    p.sphere_id = (uint32_t)s->spheres.size();
    p.meta = (material & PRIM_MATERIAL_MASK) | PRIM_SPHERE;
    s->spheres.push_back(r);
    s->sphere_bounds.push_back(sphere_world_bound(r));
    if (prim_out) *prim_out = (uint32_t)s->prims.size();
    s->prims.push_back(p);
    s->committed = false;
    return 0;
}
int trhip_scene_add_sphere(trhip_scene* s, const float* o2w, const float* o2w_inv, int reverse, float radius, float z_min, float z_max, float phi_max_deg, uint32_t material,
                           uint32_t* prim_out) {
    if (!s || !o2w || !o2w_inv) return fail(s ? s->ctx : nullptr, TRHIP_ERR_INVALID, "null argument");
    SphereRec r;
    std::memset(&r, 0, sizeof r);
    r.radius = radius;  // Sphere ctor sphere.jl:13-26
    r.z_min = jclamp(jmin(z_min, z_max), -radius, radius);
    r.z_max = jclamp(jmax(z_min, z_max), -radius, radius);
    r.theta_min = tm_acosf(jclamp(jmin(z_min, z_max) / radius, -1.0f, 1.0f));
    r.theta_max = tm_acosf(jclamp(jmax(z_min, z_max) / radius, -1.0f, 1.0f));
    r.phi_max = deg2rad(jclamp(phi_max_deg, 0.0f, 360.0f));
    return add_sphere_rec(s, o2w, o2w_inv, reverse, r, material, prim_out);
}
int trhip_scene_add_sphere_fields(trhip_scene* s, const float* o2w, const float* o2w_inv, int reverse, float radius, float z_min, float z_max, float theta_min, float theta_max,
                                  float phi_max_rad, uint32_t material, uint32_t* prim_out) {
    if (!s || !o2w || !o2w_inv) return fail(s ? s->ctx : nullptr, TRHIP_ERR_INVALID, "null argument");
    SphereRec r;
    std::memset(&r, 0, sizeof r);
    r.radius = radius;
    r.z_min = z_min;
    r.z_max = z_max;
    r.theta_min = theta_min;
    r.theta_max = theta_max;
    r.phi_max = phi_max_rad;
    return add_sphere_rec(s, o2w, o2w_inv, reverse, r, material, prim_out);
}
static int add_light(trhip_scene* s, int kind, const float* l2w, const float* l2w_inv, const float* I, float total_deg, float falloff_deg, bool fields = false) {
    if (!s || !l2w || !l2w_inv || !I) return fail(s ? s->ctx : nullptr, TRHIP_ERR_INVALID, "null argument");
    LightRec l;
    std::memset(&l, 0, sizeof l);
    l.kind = kind;
    const f3 pos = xf_point(l2w, splat3(0.0f));  // light_to_world(Point3f(0)) point.jl:23, spot.jl:16
    l.position[0] = pos.x;
    l.position[1] = pos.y;
    l.position[2] = pos.z;
    std::memcpy(l.I, I, 3 * sizeof(float));
    if (kind == 1) {
        l.cos_total_width = fields ? total_deg : tm_cosf(deg2rad(total_deg));  // spot.jl:17
        l.cos_falloff_start = fields ? falloff_deg : tm_cosf(deg2rad(falloff_deg));
    }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            l.w2l[3 * r + c] = l2w_inv[4 * r + c];  // world_to_light = inv(light_to_world): .m = inv_m
            l.l2w[3 * r + c] = l2w[4 * r + c];
        }
    s->lights.push_back(l);
    s->committed = false;
    return 0;
}
int trhip_scene_add_point_light(trhip_scene* s, const float* l2w, const float* l2w_inv, const float* I) { return add_light(s, 0, l2w, l2w_inv, I, 0, 0); }
int trhip_scene_add_spot_light(trhip_scene* s, const float* l2w, const float* l2w_inv, const float* I, float total_deg, float falloff_deg) {
    return add_light(s, 1, l2w, l2w_inv, I, total_deg, falloff_deg);
}

int trhip_scene_add_spot_light_fields(trhip_scene* s, const float* l2w, const float* l2w_inv, const float* I, float cos_total, float cos_falloff) {
    return add_light(s, 1, l2w, l2w_inv, I, cos_total, cos_falloff, true);
}

// world_bound of every primitive in caller order (triangle_mesh.jl:97, Shape.jl:17-19)
static void primitive_bounds(const trhip_scene* s, std::vector<HostAABB>& pb) {
    pb.resize(s->prims.size());
    parallel_for(s->prims.size(), [&](size_t i0, size_t i1) {
        for (size_t i = i0; i < i1; ++i) {
            const HostPrim& p = s->prims[i];
            if (p.kind == 1) {
                pb[i] = s->sphere_bounds[p.sphere_id];
            } else {
                pb[i].reset();
                for (int j = 0; j < 3; ++j) pb[i].grow_point(&p.v[3 * j]);
            }
        }
    });
}

// The LIBRARY's tree over primitive bounds `pb_build` (binned SAH on the device from 64 Ki primitives, on the host below and for what the device builder hands back;
// mode 1: the linear BVH).  `mode` as option "bvh_builder" (2 / 4 are the caller's business).
static int build_library_tree(trhip_ctx* ctx, const std::vector<HostAABB>& pb_build, int max_node_primitives, int mode, bool want_chain, FlatBVH& out) {
    bool built = false;
    ctx->bvh_device_ms = 0.0;
    if ((mode == 3 || ((mode < 0 || mode == 4) && pb_build.size() >= (64u << 10))) && pb_build.size() > ctx->tiny_scene_prims) {
        // the host builder's binned SAH, on the device (th_sahb.h); scenes it hands back (TRHIP_ERR_UNSUPPORTED) go to the host builder below
        FlatBVH dev;
        const int rc = build_bvh_device_sah(ctx, pb_build, max_node_primitives, want_chain, dev, &ctx->bvh_device_ms);
        if (rc == 0) {
            out = std::move(dev);
            built = true;
        } else if (rc != TRHIP_ERR_UNSUPPORTED) {
            return rc;
        }
    }
    if (!built && mode == 1 && pb_build.size() > ctx->tiny_scene_prims) {
        FlatBVH dev;
        const int rc = build_bvh_device(ctx, pb_build, dev);
        if (rc == 0) {
            out = std::move(dev);
            built = true;
        } else if (rc != TRHIP_ERR_UNSUPPORTED) {
            return rc;
        }
    }
    if (!built) {
        BVHBuilder builder(pb_build, max_node_primitives, ctx->tiny_scene_prims, want_chain);  // traversal 4 wants one primitive per leaf
        out = builder.build();
    }
    return 0;
}

static void drop_accelerator(trhip_scene* s) {
    s->acc = FlatBVH();
    s->hybrid_ok = false;
    std::memset(&s->wide_acc, 0, sizeof s->wide_acc);
    s->wide_acc.root_ref = kRefNone;
    release(s->d_acc_wnodes);
    release(s->d_acc_w4nodes);
    release(s->d_acc_prims);
    release(s->d_slot_boxes);
    release(s->d_acc_leaf_order);
}

int trhip_scene_commit(trhip_scene* s, int max_node_primitives) {
    if (s) s->bvh_note.clear();
    CommitClock clk;
    if (!s) return fail(nullptr, TRHIP_ERR_INVALID, "null scene");
    HIP_TRY(s->ctx, hipSetDevice(s->ctx->device));
    std::vector<HostAABB> pb;
    primitive_bounds(s, pb);
    clk.tick("commit: primitive bounds");
    drop_accelerator(s);
    // Scenes with a few spheres beside a mesh: a chain root -> {sphere 1, {sphere 2, ... {sphere k, the triangles' subtree}}}.  Any BVH2 is a valid
    // BVHAccel (results depend on the topology only through exact-t ties); this one keeps the spheres — whose fp32
    // quadratic accepts rays far outside their box and can raise t_max (A.18) — out of the triangles' subtree, which the 8-wide
    // kernel then walks with conservative interior boxes (th_wide8.h).  The leaf-size hint is a hint (bvh.jl:159-165 decides by cost).
    std::vector<uint32_t> sph_ids, tri_ids;
    for (size_t i = 0; i < s->prims.size(); ++i) (s->prims[i].kind == 1 ? sph_ids : tri_ids).push_back((uint32_t)i);
    const int mode = s->ctx->bvh_builder;
    const bool want_chain = s->ctx->compose_spheres > 0 || (s->ctx->compose_spheres < 0 && s->ctx->traversal == 4);
    const bool compose = mode != 2 && want_chain && !sph_ids.empty() && sph_ids.size() <= (size_t)kW8MaxSpheres && tri_ids.size() >= 2 && pb.size() > s->ctx->tiny_scene_prims;
    // HYBRID (mode 4, and the default -1): the canonical tree is the reference's own construction — the answers are Trace.jl's, ray for ray — and the library's tree
    // rides along as the accelerator most rays walk instead (th_trace3c.h).  Not with the sphere chain (a layout only traversal 4 asks for).
    const bool want_hybrid = (mode == 4 || mode < 0) && !want_chain;
    s->ctx->bvh_device_ms = 0.0;
    FlatBVH lib_tree;  // a hybrid commit's library tree: the accelerator — or, when the reference's construction fails, the tree the default falls back on (built ONCE)
    int lib_rc = -1;   // -1: not built
    if (mode == 2 || want_hybrid) {
        // the reference's own construction, node for node (th_bvh_ref.h): also for scenes the library would commit as one leaf, never composed
        bool ok = true;
        std::string why;
        // the reference's builder is one host thread (0.6 s per million primitives); the library's tree — the accelerator of a hybrid commit, and the tree the default falls
        // back on when the reference's construction fails — is built meanwhile (on the device from 64 Ki primitives: th_sahb.h)
        auto ref_job = std::async(std::launch::async, [&pb, max_node_primitives]() {
            RefBVHBuilder rb(pb, max_node_primitives, (uint32_t)(kStackLds + kStackSpill));  // (gives up at depth 65: bvh.jl:222 could not walk that tree)
            return rb.build();
        });
        if (want_hybrid) lib_rc = build_library_tree(s->ctx, pb, max_node_primitives, mode, false, lib_tree);
        try {
            s->bvh = ref_job.get();
        } catch (const RefBVHBuilder::DepthExceeded& e) {
            ok = false;
            why = "BVH depth exceeds the 64-entry traversal stack (a node at depth " + std::to_string(e.depth) + "; bvh.jl:222 throws a BoundsError there)";
        } catch (const std::exception& e) {
            ok = false;
            why = e.what();
        }
        clk.tick("commit: reference tree (+ library tree)");
        if (ok) {
            s->literal_only = false;
            s->bvh_mode = 1;
            if (int rc = upload_scene(s)) return rc;
            if (!want_hybrid) return 0;
            if (lib_rc) return lib_rc;
            s->acc = std::move(lib_tree);
            if (int rc = upload_accelerator(s)) return rc;
            if (s->hybrid_ok) {
                s->bvh_mode = 2;
            } else {
                drop_accelerator(s);
                s->bvh_note = s->spheres.size() > kCertMaxSpheres ? "more than 32 spheres: every ray walks the canonical tree (th_trace3c.h kCertMaxSpheres)"
                                                                  : "no accelerator: the library's tree could not be conformed to the canonical leaves, or the canonical tree is a single leaf";
            }
            return 0;
        }
        s->bvh_note = "the reference's construction fails on this scene (" + why + "): there is no Trace.jl answer to reproduce, the library's tree alone";
        // Where the reference's own constructor fails (its recursion does not end, or its tree outgrows its 64-entry stack) there is no Trace.jl answer to reproduce:
        // an explicit request is an error, the default falls back to the library's tree alone
        if (mode == 2 || mode == 4) return fail(s->ctx, TRHIP_ERR_UNSUPPORTED, "bvh_builder %d: %s", mode, why.c_str());
    }
    std::vector<HostAABB> pb_sub;
    if (compose) {
        pb_sub.reserve(tri_ids.size());
        for (uint32_t id : tri_ids) pb_sub.push_back(pb[id]);
    }
    const std::vector<HostAABB>& pb_build = compose ? pb_sub : pb;
    if (lib_rc == 0 && !compose && !want_chain) {
        s->bvh = std::move(lib_tree);  // (the reference's construction failed: the tree built beside it, same arguments, is the one to keep)
    } else if (int rc = build_library_tree(s->ctx, pb_build, max_node_primitives, mode, want_chain, s->bvh)) {
        return rc;
    }
    clk.tick("commit: tree");
    if (compose) {
        // flat layout (bvh.jl:187-206): chain node i at 2 i = interior {leaf of sphere i at 2 i + 1, rest at 2 i + 2}; the triangles' subtree at 2 n_sph
        FlatBVH sub = std::move(s->bvh), out;
        const uint32_t n_sph = (uint32_t)sph_ids.size(), n_sub = (uint32_t)sub.a.size();
        std::vector<HostAABB> rest(n_sph + 1);
        std::memcpy(rest[n_sph].mn, &sub.bounds[0], 3 * sizeof(float));
        std::memcpy(rest[n_sph].mx, &sub.bounds[3], 3 * sizeof(float));
        for (uint32_t i = n_sph; i-- > 0;) {
            rest[i] = rest[i + 1];
            rest[i].grow(pb[sph_ids[i]]);
        }
        // one split axis for every chain node (it only decides whether a ray meets the sphere leaves before or after the triangles): where the
        // spheres' centre and the triangles' lie furthest apart
        HostAABB sall;
        sall.reset();
        for (uint32_t id : sph_ids) sall.grow(pb[id]);
        uint32_t axis = 0;
        float best = -1.0f;
        for (int a = 0; a < 3; ++a) {
            const float dc = std::fabs((0.5f * sall.mn[a] + 0.5f * sall.mx[a]) - (0.5f * rest[n_sph].mn[a] + 0.5f * rest[n_sph].mx[a]));
            if (dc > best) {
                best = dc;
                axis = (uint32_t)a;
            }
        }
        for (uint32_t i = 0; i < n_sph; ++i) {
            const HostAABB& sbx = pb[sph_ids[i]];
            out.bounds.insert(out.bounds.end(), {rest[i].mn[0], rest[i].mn[1], rest[i].mn[2], rest[i].mx[0], rest[i].mx[1], rest[i].mx[2]});
            out.a.push_back(2 * i + 2);
            out.flags.push_back(axis);
            out.bounds.insert(out.bounds.end(), {sbx.mn[0], sbx.mn[1], sbx.mn[2], sbx.mx[0], sbx.mx[1], sbx.mx[2]});
            out.a.push_back(i);
            out.flags.push_back((1u << 2) | 3u);
        }
        out.bounds.insert(out.bounds.end(), sub.bounds.begin(), sub.bounds.end());
        out.a.reserve(n_sub + 2 * n_sph);
        out.flags.reserve(n_sub + 2 * n_sph);
        for (uint32_t i = 0; i < n_sub; ++i) {
            out.a.push_back(sub.a[i] + ((sub.flags[i] & 3u) == 3u ? n_sph : 2 * n_sph));
            out.flags.push_back(sub.flags[i]);
        }
        out.order = sph_ids;
        out.order.reserve(pb.size());
        for (uint32_t k : sub.order) out.order.push_back(tri_ids[k]);
        out.max_depth = sub.max_depth + n_sph;
        s->bvh = std::move(out);
    }
    if (s->bvh.max_depth > (uint32_t)(kStackLds + kStackSpill))
        return fail(s->ctx, TRHIP_ERR_UNSUPPORTED, "BVH depth %u exceeds the 64-entry traversal stack (bvh.jl:222)", s->bvh.max_depth);
    s->literal_only = false;
    s->bvh_mode = 0;
    if (int rc = upload_scene(s)) return rc;
    // A DEFAULT commit that ended with the library's tree as its one (canonical) tree — the reference's construction failed on the scene: 10 M triangles — still gets the
    // four-wide certified walk: the same tree rides along as its own accelerator (mode 3).  The answers are the canonical tree's own walk's, as always; explicit requests
    // (bvh_builder 0 / 3) keep the single tree.
    if (mode < 0 && s->ctx->wide4 && !want_chain && !compose && s->wide_ok && s->wide.root_cnt == 0) {
        s->acc = s->bvh;
        if (int rc = upload_accelerator(s)) return rc;
        if (s->hybrid_ok && s->wide_acc.w4nodes)
            s->bvh_mode = 3;
        else
            drop_accelerator(s);
    }
    return 0;
}
int trhip_build_bvh_host(int builder, const float* prim_bounds, uint32_t n_prims, int max_node_primitives, float* node_bounds, uint32_t* node_a, uint32_t* node_flags, uint32_t* n_nodes_inout,
                         uint32_t* prim_order, uint32_t* max_depth_out) {
    if (!prim_bounds || !n_nodes_inout) return fail(nullptr, TRHIP_ERR_INVALID, "null argument");
    if (builder != 0 && builder != 2) return fail(nullptr, TRHIP_ERR_INVALID, "builder must be 0 (binned SAH, th_bvh.h) or 2 (the reference's construction, th_bvh_ref.h)");
    std::vector<HostAABB> pb(n_prims);
    static_assert(sizeof(HostAABB) == 6 * sizeof(float), "HostAABB layout");
    if (n_prims) std::memcpy(pb.data(), prim_bounds, (size_t)n_prims * 6 * sizeof(float));
    FlatBVH t;
    try {
        if (builder == 2) {
            RefBVHBuilder rb(pb, max_node_primitives);
            t = rb.build();
        } else {
            BVHBuilder b(pb, max_node_primitives, 0u, false);
            t = b.build();
        }
    } catch (const std::exception& e) {
        return fail(nullptr, TRHIP_ERR_UNSUPPORTED, "%s", e.what());
    }
    const uint32_t n_nodes = (uint32_t)t.a.size(), cap = *n_nodes_inout;
    *n_nodes_inout = n_nodes;
    if (max_depth_out) *max_depth_out = t.max_depth;
    if (!node_bounds && !node_a && !node_flags && !prim_order) return 0;  // size query
    if (cap < n_nodes) return fail(nullptr, TRHIP_ERR_INVALID, "the tree has %u nodes, the caller's arrays hold %u", n_nodes, cap);
    if (node_bounds) std::memcpy(node_bounds, t.bounds.data(), t.bounds.size() * sizeof(float));
    if (node_a) std::memcpy(node_a, t.a.data(), t.a.size() * sizeof(uint32_t));
    if (node_flags) std::memcpy(node_flags, t.flags.data(), t.flags.size() * sizeof(uint32_t));
    if (prim_order) std::memcpy(prim_order, t.order.data(), t.order.size() * sizeof(uint32_t));
    return 0;
}
int trhip_scene_bvh_size(const trhip_scene* s, uint32_t* n_nodes, uint32_t* n_prims) {
    if (!s) return TRHIP_ERR_INVALID;
    if (n_nodes) *n_nodes = (uint32_t)s->bvh.a.size();
    if (n_prims) *n_prims = (uint32_t)s->bvh.order.size();
    return 0;
}
int trhip_scene_get_bvh(const trhip_scene* s, float* bounds, uint32_t* a, uint32_t* flags, uint32_t* order) {
    if (!s) return TRHIP_ERR_INVALID;
    if (bounds) std::memcpy(bounds, s->bvh.bounds.data(), s->bvh.bounds.size() * sizeof(float));
    if (a) std::memcpy(a, s->bvh.a.data(), s->bvh.a.size() * sizeof(uint32_t));
    if (flags) std::memcpy(flags, s->bvh.flags.data(), s->bvh.flags.size() * sizeof(uint32_t));
    if (order) std::memcpy(order, s->bvh.order.data(), s->bvh.order.size() * sizeof(uint32_t));
    return 0;
}
int trhip_scene_bvh_note(const trhip_scene* s, char* buf, size_t n) {
    if (!s || !buf || n == 0) return TRHIP_ERR_INVALID;
    std::snprintf(buf, n, "%s", s->bvh_note.c_str());
    return 0;
}
int trhip_scene_bvh_mode(const trhip_scene* s, int* mode, uint32_t* accel_nodes, uint32_t* accel_depth) {
    if (!s) return TRHIP_ERR_INVALID;
    if (mode) *mode = s->bvh_mode;
    if (accel_nodes) *accel_nodes = s->hybrid_ok ? (uint32_t)s->acc.a.size() : 0u;
    if (accel_depth) *accel_depth = s->hybrid_ok ? s->acc.max_depth : 0u;
    return 0;
}
int trhip_scene_get_accelerator(const trhip_scene* s, float* bounds, uint32_t* a, uint32_t* flags, uint32_t* order) {
    if (!s) return TRHIP_ERR_INVALID;
    if (!s->hybrid_ok) return fail(s->ctx, TRHIP_ERR_INVALID, "the scene has no accelerator tree (trhip_scene_bvh_mode)");
    if (bounds) std::memcpy(bounds, s->acc.bounds.data(), s->acc.bounds.size() * sizeof(float));
    if (a) std::memcpy(a, s->acc.a.data(), s->acc.a.size() * sizeof(uint32_t));
    if (flags) std::memcpy(flags, s->acc.flags.data(), s->acc.flags.size() * sizeof(uint32_t));
    if (order) std::memcpy(order, s->acc.order.data(), s->acc.order.size() * sizeof(uint32_t));
    return 0;
}
int trhip_scene_set_bvh(trhip_scene* s, const float* bounds, const uint32_t* a, const uint32_t* flags, uint32_t n_nodes, const uint32_t* order, uint32_t n_prims) {
    if (!s || !bounds || !a || !flags || !order) return fail(s ? s->ctx : nullptr, TRHIP_ERR_INVALID, "null argument");
    if (n_nodes == 0) return fail(s->ctx, TRHIP_ERR_INVALID, "empty node array");
    for (uint32_t i = 0; i < n_prims; ++i)
        if (order[i] >= s->prims.size()) return fail(s->ctx, TRHIP_ERR_INVALID, "prim_order[%u] = %u out of range", i, order[i]);
    // The array must be ONE tree in the reference's depth-first layout (bvh.jl:187-206): the subtree of node i is the index range
    // [i, end): first child i + 1 .. a[i] - 1, second child a[i] .. end - 1.  Anything else (a[i] <= i + 1 closes a cycle: the
    // traversal kernels would never end) is rejected here; so is a tree deeper than the 64-entry stack, where the reference
    // throws a BoundsError (bvh.jl:222).
    struct Span {
        uint32_t node, end, depth;
    };
    std::vector<Span> todo;
    todo.push_back({0u, n_nodes, 1u});
    uint32_t max_depth = 0;
    bool nested = true;  // every child box inside its parent's, every primitive's bound inside its leaf box
    auto inside = [&](const float* in, const float* out) {
        return in[0] >= out[0] && in[1] >= out[1] && in[2] >= out[2] && in[3] <= out[3] && in[4] <= out[4] && in[5] <= out[5];
    };
    while (!todo.empty()) {
        const Span sp = todo.back();
        todo.pop_back();
        const uint32_t i = sp.node;
        max_depth = std::max(max_depth, sp.depth);
        if ((flags[i] & 3u) == 3u) {
            if (sp.end != i + 1) return fail(s->ctx, TRHIP_ERR_INVALID, "leaf %u is followed by nodes that belong to no subtree (not a depth-first layout)", i);
            const uint32_t cnt = flags[i] >> 2;
            if ((uint64_t)a[i] + cnt > n_prims) return fail(s->ctx, TRHIP_ERR_INVALID, "leaf %u references primitives outside the list", i);
            for (uint32_t k = a[i]; k < a[i] + cnt && nested; ++k) {
                const HostPrim& p = s->prims[order[k]];
                HostAABB pb;
                if (p.kind == 1) {
                    pb = s->sphere_bounds[p.sphere_id];
                } else {
                    pb.reset();
                    for (int j = 0; j < 3; ++j) pb.grow_point(&p.v[3 * j]);
                }
                const float pbox[6] = {pb.mn[0], pb.mn[1], pb.mn[2], pb.mx[0], pb.mx[1], pb.mx[2]};
                nested = inside(pbox, &bounds[6 * (size_t)i]);
            }
            continue;
        }
        if (a[i] <= i + 1 || a[i] >= sp.end)
            return fail(s->ctx, TRHIP_ERR_INVALID, "interior node %u: second child %u outside (%u, %u) — not the depth-first layout of bvh.jl:187-206", i, a[i], i + 1, sp.end);
        if ((flags[i] & 3u) > 2u) return fail(s->ctx, TRHIP_ERR_INVALID, "interior node %u: split axis %u", i, flags[i] & 3u);
        nested = nested && inside(&bounds[6 * (size_t)(i + 1)], &bounds[6 * (size_t)i]) && inside(&bounds[6 * (size_t)a[i]], &bounds[6 * (size_t)i]);
        todo.push_back({a[i], sp.end, sp.depth + 1});
        todo.push_back({i + 1, a[i], sp.depth + 1});
    }
    if (max_depth > (uint32_t)(kStackLds + kStackSpill))
        return fail(s->ctx, TRHIP_ERR_UNSUPPORTED, "BVH depth %u exceeds the 64-entry traversal stack (bvh.jl:222 throws a BoundsError there)", max_depth);
    s->bvh.bounds.assign(bounds, bounds + 6 * (size_t)n_nodes);
    s->bvh.a.assign(a, a + n_nodes);
    s->bvh.flags.assign(flags, flags + n_nodes);
    s->bvh.order.assign(order, order + n_prims);
    s->bvh.max_depth = max_depth;
    // The default kernels' shortcuts (tight slab clauses, largest-triangle pre-pass, wide nodes: th_trace2.h, th_trace8.h) are exact
    // only when boxes nest; a foreign tree that does not is walked by the literal kernels (the reference's loop, op for op).
    s->literal_only = !nested;
    HIP_TRY(s->ctx, hipSetDevice(s->ctx->device));
    drop_accelerator(s);
    s->bvh_mode = 1;
    if (int rc = upload_scene(s)) return rc;
    // The host's own tree is the canonical one (Trace.jl's BVHAccel through TraceHIP.jl): under the default / hybrid builder the library's tree over the same
    // primitives rides along as the accelerator (th_trace3c.h) — same answers, most rays on the cheaper tree.  Needs every primitive in the tree exactly once.
    const int mode = s->ctx->bvh_builder;
    const bool want_chain = s->ctx->compose_spheres > 0 || (s->ctx->compose_spheres < 0 && s->ctx->traversal == 4);
    if ((mode == 4 || mode < 0) && !want_chain && nested && n_prims == s->prims.size()) {
        std::vector<uint8_t> seen(n_prims, 0);
        bool perm = true;
        for (uint32_t i = 0; i < n_prims && perm; ++i) {
            perm = !seen[order[i]];
            seen[order[i]] = 1;
        }
        if (perm) {
            std::vector<HostAABB> pb;
            primitive_bounds(s, pb);
            if (int rc = build_library_tree(s->ctx, pb, 1, mode, false, s->acc)) return rc;
            if (int rc = upload_accelerator(s)) return rc;
            if (s->hybrid_ok)
                s->bvh_mode = 2;
            else
                drop_accelerator(s);
        }
    }
    return 0;
}

}  // extern "C"
