// orc_sppm.h — TEST INFRASTRUCTURE (see README.md): CPU restatement of integrators/sppm.jl (whole file) and of the helpers
// only SPPM uses: radical_inverse / Distribution1D / sample_discrete (sampler/sampling.jl:3-60), sample_le
// (lights/point.jl:60-69, lights/spot.jl:46-55), to_grid / hash (sppm.jl:479-501).
//
// Decisions this build makes where the reference is not reproducible (it draws the camera-pass numbers from the global RNG
// and adds photon contributions with unordered atomics):
//   * camera pass of iteration k uses the seeded sampler stream (seed, pixel, sample k-1): the iteration plays the role of
//     the sample number (`set_sample_number!(tile_sampler, iteration)` is commented out at sppm.jl:194);
//   * photons are followed one after another in photon-index order and a grid list is walked newest-node-first, i.e. the
//     single-threaded execution of the reference.  M, radius, N, Ld and the visible points do not depend on that order;
//     ϕ / τ do, at Float32 rounding level (the reference's own result varies the same way from run to run).
// Parity of the composite is UNPINNED (the reference has no SPPM test); the callees are pinned by tests/test_oracle_kat.py.
#pragma once
#include "orc_render.h"

namespace orc {

// ---- sampler/sampling.jl ---------------------------------------------------------------------------------------------------
// primes.jl: "First 1023 prime numbers (omitting 2)": PRIMES[1] = 3.
inline const std::vector<int64_t>& odd_primes() {
    static const std::vector<int64_t> table = [] {
        std::vector<int64_t> p;
        for (int64_t c = 3; p.size() < 1023; c += 2) {
            bool prime = true;
            for (int64_t d = 3; d * d <= c; d += 2)
                if (c % d == 0) {
                    prime = false;
                    break;
                }
            if (prime) p.push_back(c);
        }
        return p;
    }();
    return table;
}
inline uint32_t reverse_bits32(uint32_t n) {  // sampling.jl:62-68
    n = (n << 16) | (n >> 16);
    n = ((n & 0x00ff00ffu) << 8) | ((n & 0xff00ff00u) >> 8);
    n = ((n & 0x0f0f0f0fu) << 4) | ((n & 0xf0f0f0f0u) >> 4);
    n = ((n & 0x33333333u) << 2) | ((n & 0xccccccccu) >> 2);
    return ((n & 0x55555555u) << 1) | ((n & 0xaaaaaaaau) >> 1);
}
inline uint64_t reverse_bits64(uint64_t n) {  // sampling.jl:70-74
    const uint64_t n0 = reverse_bits32((uint32_t)((n << 32) >> 32)), n1 = reverse_bits32((uint32_t)(n >> 32));
    return (n0 << 32) | n1;
}
// sampling.jl:43-60.  Dimension 0 reverses bits (Float64 product, rounded to Float32 by the return type); dimension k >= 1
// uses PRIMES[k] (3, 5, 7, …) with Float32 inverse-base powers and a Float64 digit division.
inline float radical_inverse(int64_t base_index, uint64_t a) {
    if (base_index == 0) return (float)((double)reverse_bits64(a) * 5.4210108624275222e-20);
    const int64_t base = odd_primes()[(size_t)base_index - 1];
    const float inv_base = 1.0f / (float)base;
    uint64_t reversed_digits = 0;
    float inv_base_n = 1.0f;
    while (a > 0) {
        const uint64_t next = (uint64_t)std::floor((double)a / (double)base);
        const uint64_t digit = a - next * (uint64_t)base;
        reversed_digits = reversed_digits * (uint64_t)base + digit;
        inv_base_n *= inv_base;
        a = next;
    }
    return jl_min((float)reversed_digits * inv_base_n, 1.0f);
}

struct Distribution1D {  // sampling.jl:3-31
    std::vector<float> func, cdf;
    float func_int = 0;
    Distribution1D() = default;
    explicit Distribution1D(const std::vector<float>& f) : func(f), cdf(f.size() + 1) {
        const size_t n = func.size();
        cdf[0] = 0.0f;
        for (size_t i = 1; i <= n; ++i) cdf[i] = cdf[i - 1] + func[i - 1] / (float)n;
        func_int = cdf[n];
        if (func_int == 0.0f) {
            for (size_t i = 1; i <= n; ++i) cdf[i] = (float)((double)(i + 1) / (double)n);  // `i / n` with the 1-based i (Int / Int -> Float64)
        } else {
            for (size_t i = 1; i <= n; ++i) cdf[i] /= func_int;
        }
    }
};
struct DiscreteSample {
    int offset = 1;  // 1-based
    float pdf = 0, u_remapped = 0;
};
inline DiscreteSample sample_discrete(const Distribution1D& d, float u) {  // sampling.jl:33-41
    int offset = 0;
    for (int i = (int)d.cdf.size(); i >= 1; --i)  // findlast(i -> cdf[i] ≤ u)
        if (d.cdf[(size_t)i - 1] <= u) {
            offset = i;
            break;
        }
    const int n = (int)d.func.size();
    if (offset < 1) offset = 1;  // `nothing` would throw in the reference (u is never NaN or negative here)
    if (offset > n) offset = n;
    DiscreteSample s;
    s.offset = offset;
    s.pdf = d.func_int > 0 ? d.func[(size_t)offset - 1] / (d.func_int * (float)n) : 0.0f;
    s.u_remapped = (u - d.cdf[(size_t)offset - 1]) / (d.cdf[(size_t)offset] - d.cdf[(size_t)offset - 1]);
    return s;
}
inline Distribution1D compute_light_power_distribution(const Scene& scene) {  // sppm.jl:564-569
    std::vector<float> f;
    for (const Light& l : scene.lights) f.push_back(to_Y(light_power(l)));
    return Distribution1D(f);
}

// ---- lights: sample_le ------------------------------------------------------------------------------------------------------
struct LeSample {
    RGB le;
    Ray ray;
    V3 light_normal;
    float pdf_pos = 0, pdf_dir = 0;
};
inline LeSample sample_le(const Light& l, V2 u1) {  // point.jl:60-69, spot.jl:46-55 (u2 and time are ignored)
    LeSample s;
    if (l.kind == Light::POINT) {
        s.ray = Ray{l.position, uniform_sample_sphere(u1), INF32, 0.0f};
        s.light_normal = s.ray.d;
        s.pdf_pos = 1.0f;
        s.pdf_dir = uniform_sphere_pdf();
        s.le = l.i;
    } else {
        const V3 w = l.light_to_world.vec(uniform_sample_cone(u1, l.cos_total_width));
        s.ray = Ray{l.position, w, INF32, 0.0f};
        s.light_normal = s.ray.d;
        s.pdf_pos = 1.0f;
        s.pdf_dir = uniform_cone_pdf(l.cos_total_width);
        s.le = l.i * spot_falloff(l, s.ray.d);
    }
    return s;
}

// ---- sppm.jl:479-501 --------------------------------------------------------------------------------------------------------
inline V3 bounds_offset(const Bounds3& b, V3 p) {  // bounds.jl:134-143
    const V3 o = p - b.p_min;
    const bool g0 = b.p_max.x > b.p_min.x, g1 = b.p_max.y > b.p_min.y, g2 = b.p_max.z > b.p_min.z;
    if (!(g0 || g1 || g2)) return o;
    return {o.x / (g0 ? b.p_max.x - b.p_min.x : 1.0f), o.y / (g1 ? b.p_max.y - b.p_min.y : 1.0f), o.z / (g2 ? b.p_max.z - b.p_min.z : 1.0f)};
}
struct GridPoint {
    bool in_bounds = false;
    uint64_t g[3] = {0, 0, 0};
};
inline GridPoint to_grid(V3 p, const Bounds3& bounds, const int64_t res[3]) {  // sppm.jl:479-495
    const V3 po = bounds_offset(bounds, p);
    const float pf[3] = {po.x, po.y, po.z};
    GridPoint out;
    out.in_bounds = true;
    for (int a = 0; a < 3; ++a) {
        const int64_t gp = (int64_t)std::floor((float)res[a] * pf[a]);  // Int64 * Float32 -> Float32
        if (!(0 <= gp && gp < res[a])) out.in_bounds = false;
        out.g[a] = (uint64_t)(gp < 0 ? 0 : (gp > res[a] - 1 ? res[a] - 1 : gp));
    }
    return out;
}
inline uint64_t grid_hash(uint64_t p1, uint64_t p2, uint64_t p3, uint64_t hash_size) {  // sppm.jl:497-501, returned 0-based here
    return ((p1 * 73856093ull) ^ (p2 * 19349663ull) ^ (p3 * 83492791ull)) % hash_size;
}

// ---- sppm.jl:53-121 -----------------------------------------------------------------------------------------------------------
struct VisiblePoint {
    V3 p, wo;
    BSDF bsdf;  // bsdf.valid == false  <=>  `nothing`
    RGB beta{0.0f};
};
struct SPPMPixel {
    RGB Ld{0.0f};
    float phi[3] = {0, 0, 0};
    RGB tau{0.0f};
    float radius = 0;
    int64_t M = 0;
    double N = 0;
    VisiblePoint vp;
};
struct SPPMParams {
    float initial_search_radius = 1;
    int max_depth = 5;
    int64_t n_iterations = 1;
    int64_t photons_per_iteration = -1;  // <= 0: area(crop_bounds) (sppm.jl:121-124, the non-inclusive area of bounds.jl:87-90)
    uint64_t seed = 0;
    // How the reference spreads the work (Threads.@threads over tiles, sppm.jl:184, and over photons, sppm.jl:334, with
    // Threads.Atomic adds for ϕ and M, :398-399).  threads == 1 (the parity tests): the sequential order.
    int threads = 1;
    // Multi-process jobs (tests/test_sharding_gloo.py): this process traces photon indices [photon_begin, photon_end) of every
    // iteration (photon_end < 0: all) and `exchange`, if set, is called between the photon pass and _update_pixels! with the
    // per-pixel ϕ (3 floats each) and M to be summed over the processes in place.
    int64_t photon_begin = 0, photon_end = -1;
    void (*exchange)(void* user, float* phi3, int64_t* M, uint64_t n_pixels) = nullptr;
    void* exchange_user = nullptr;
};
struct SPPMState {
    int width = 0, height = 0;     // inclusive sides of crop_bounds
    std::vector<SPPMPixel> pixels;  // (y, x) row-major here; the reference's `for pixel in pixels` is column-major (y fastest)
    int64_t photons_per_iteration = 0;
    int64_t iteration = 0;  // iterations completed
    // snapshot of the last iteration, taken between the photon pass and _update_pixels!
    std::vector<int64_t> last_M;
    std::vector<float> last_phi, last_vp_p, last_vp_beta;
    Bounds3 grid_bounds;
    int64_t grid_res[3] = {1, 1, 1};
    bool grid_valid = false;
    uint64_t photon_hits = 0, grid_entries = 0;
    Counters totals;  // rays / visits of all threads
    SPPMPixel& at(int x1, int y1) { return pixels[(size_t)(y1 - 1) * width + (size_t)(x1 - 1)]; }  // pixels[y, x], 1-based
};

// sppm.jl:175-270
inline void sppm_camera_pass(Scene& scene, const PerspectiveCamera& cam, const Film& film, const SPPMParams& prm, SPPMState& st, int64_t iteration) {
    const Bounds2 pb = film.crop_bounds;
    const int tile_size = 16;
    const V2 extent{pb.p_max.x - pb.p_min.x, pb.p_max.y - pb.p_min.y};
    const long long width = (long long)std::floor((extent.x + tile_size) / tile_size), height = (long long)std::floor((extent.y + tile_size) / tile_size);
    const long long total_tiles = width * height - 1;
#pragma omp parallel num_threads(prm.threads > 0 ? prm.threads : 1)
    {
    counters() = Counters{};
#pragma omp for schedule(dynamic, 1)
    for (long long k = 0; k <= total_tiles; ++k) {
        const float tx = (float)(k % width), ty = (float)(k / width);
        SeededSampler smp(1, prm.seed, (uint32_t)(iteration - 1));  // deepcopy(sampler): UniformSampler(1)
        const V2 tb_min{pb.p_min.x + tx * tile_size, pb.p_min.y + ty * tile_size};
        const V2 tb_max{jl_min(tb_min.x + (tile_size - 1), pb.p_max.x), jl_min(tb_min.y + (tile_size - 1), pb.p_max.y)};
        for (float py = tb_min.y; py <= tb_max.y; py += 1.0f)
            for (float px = tb_min.x; px <= tb_max.x; px += 1.0f) {
                const V2 pixel_point{px, py};
                smp.start_pixel(pixel_point);
                const CameraSample cs = smp.get_camera_sample(pixel_point);
                Ray ray = generate_ray(cam, cs);  // weight 1; scale_differentials! touches dead data (A.10)
                RGB beta(1.0f);
                SPPMPixel& pixel = st.at((int)px, (int)py);
                bool specular_bounce = false;
                int depth = 1;
                while (depth <= prm.max_depth) {
                    SurfaceInteraction si;
                    if (!scene_intersect(scene, ray, si)) {
                        for (size_t l = 0; l < scene.lights.size(); ++l) pixel.Ld = pixel.Ld + beta * RGB(0.0f);  // le(light, ray) = 0 (light.jl:41)
                        break;
                    }
                    const BSDF bsdf = compute_scattering(scene, si, true);
                    if (!bsdf.valid) {
                        ray = spawn_ray_dir(si, ray.d);
                        continue;
                    }
                    const V3 wo = -ray.d;
                    if (depth == 1 || specular_bounce) pixel.Ld = pixel.Ld + beta * RGB(0.0f);  // le(si, wo) = 0 (surface_interaction.jl:149-152)
                    const uint32_t v = (uint32_t)(depth - 1);
                    pixel.Ld = pixel.Ld + uniform_sample_one_light(scene, si, bsdf, smp.u(ts_vertex_dim(v, TS_V_LIGHT_PICK)));  // no β (A.12)
                    const bool is_diffuse = bsdf.num_components(BSDF_DIFFUSE | BSDF_REFLECTION | BSDF_TRANSMISSION) > 0;
                    const bool is_glossy = bsdf.num_components(BSDF_GLOSSY | BSDF_REFLECTION | BSDF_TRANSMISSION) > 0;
                    if (is_diffuse || (is_glossy && depth == prm.max_depth)) {
                        pixel.vp.p = si.p;
                        pixel.vp.wo = wo;
                        pixel.vp.bsdf = bsdf;
                        pixel.vp.beta = beta;
                        break;
                    }
                    if (depth == prm.max_depth) {
                        depth += 1;
                        continue;
                    }
                    const V2 u{smp.u(ts_vertex_dim(v, TS_V_BSDF_U0)), smp.u(ts_vertex_dim(v, TS_V_BSDF_U1))};
                    const BSDFSample s = bsdf_sample_f(bsdf, wo, u, BSDF_ALL);
                    if (s.pdf == 0.0f || is_black(s.f)) break;  // `pdf ≈ 0f0` is `== 0` for Float32
                    specular_bounce = (s.sampled_type & BSDF_SPECULAR) != 0;
                    beta = beta * (s.f * std::fabs(dot(s.wi, si.sh_n)) / s.pdf);
                    const float by = to_Y(beta);
                    if (by < 0.25f) {
                        const float cont = jl_min(1.0f, by);
                        if (smp.u(ts_vertex_dim(v, TS_V_RR)) > cont) break;
                        beta = beta / cont;
                    }
                    ray = spawn_ray_dir(si, s.wi);
                    depth += 1;
                }
            }
    }
#pragma omp critical(orc_sppm_totals)
    {
        st.totals.closest += counters().closest, st.totals.shadow += counters().shadow, st.totals.nodes += counters().nodes, st.totals.prims += counters().prims;
    }
    }
}

struct SPPMGrid {
    std::vector<int64_t> head;  // per bucket: newest node or -1
    std::vector<int64_t> next;  // per node
    std::vector<uint32_t> node_pixel;
};
// sppm.jl:272-318
inline void sppm_populate_grid(SPPMState& st, SPPMGrid& grid, uint64_t n_pixels) {
    grid.head.assign((size_t)n_pixels, -1);
    grid.next.clear();
    grid.node_pixel.clear();
    Bounds3 gb;
    float max_radius = 0.0f;
    for (int x = 1; x <= st.width; ++x)
        for (int y = 1; y <= st.height; ++y) {  // column-major: y fastest
            const SPPMPixel& px = st.at(x, y);
            if (is_black(px.vp.beta)) continue;
            gb = bunion(gb, expand(Bounds3(px.vp.p), px.radius));
            max_radius = jl_max(max_radius, px.radius);
        }
    st.grid_valid = max_radius > 0.0f;
    if (!st.grid_valid) {  // no visible point at all: the reference throws (Int64(floor(-Inf / 0))); here the photons find nothing
        st.grid_bounds = gb;
        st.grid_res[0] = st.grid_res[1] = st.grid_res[2] = 1;
        st.grid_entries = 0;
        return;
    }
    const V3 diag = diagonal(gb);
    const float max_diag = jl_max(jl_max(diag.x, diag.y), diag.z);
    const int64_t base_res = (int64_t)std::floor(max_diag / max_radius);
    const float dg[3] = {diag.x, diag.y, diag.z};
    for (int a = 0; a < 3; ++a) {
        const int64_t r = (int64_t)std::floor((float)base_res * dg[a] / max_diag);
        st.grid_res[a] = r > 1 ? r : 1;
    }
    st.grid_bounds = gb;
    for (int x = 1; x <= st.width; ++x)
        for (int y = 1; y <= st.height; ++y) {
            const SPPMPixel& px = st.at(x, y);
            if (is_black(px.vp.beta)) continue;
            const float shift = px.radius;
            const GridPoint lo = to_grid(px.vp.p - V3(shift), gb, st.grid_res), hi = to_grid(px.vp.p + V3(shift), gb, st.grid_res);
            const uint32_t pid = (uint32_t)((size_t)(y - 1) * st.width + (size_t)(x - 1));
            for (uint64_t z = lo.g[2]; z <= hi.g[2]; ++z)
                for (uint64_t yy = lo.g[1]; yy <= hi.g[1]; ++yy)
                    for (uint64_t xx = lo.g[0]; xx <= hi.g[0]; ++xx) {
                        const uint64_t h = grid_hash(xx, yy, z, n_pixels);
                        grid.next.push_back(grid.head[(size_t)h]);
                        grid.node_pixel.push_back(pid);
                        grid.head[(size_t)h] = (int64_t)grid.next.size() - 1;
                    }
        }
    st.grid_entries = grid.next.size();
}

// sppm.jl:320-436
inline void sppm_trace_photons(Scene& scene, const SPPMParams& prm, SPPMState& st, const SPPMGrid& grid, const Distribution1D& light_distr, int64_t iteration,
                               uint64_t n_pixels) {
    const uint64_t halton_base = (uint64_t)(iteration - 1) * (uint64_t)st.photons_per_iteration;
    const int64_t p_begin = prm.photon_end < 0 ? 0 : prm.photon_begin, p_end = prm.photon_end < 0 ? st.photons_per_iteration : (prm.photon_end < st.photons_per_iteration ? prm.photon_end : st.photons_per_iteration);
    uint64_t hits_total = 0;
#pragma omp parallel num_threads(prm.threads > 0 ? prm.threads : 1) reduction(+ : hits_total)
    {
    counters() = Counters{};
#pragma omp for schedule(dynamic, 64)
    for (int64_t photon_index = p_begin; photon_index < p_end; ++photon_index) {
        const uint64_t hi = halton_base + (uint64_t)photon_index;
        int64_t dim = 0;
        const float light_sample = radical_inverse(dim, hi);
        dim += 1;
        const DiscreteSample ds = sample_discrete(light_distr, light_sample);
        const Light& light = scene.lights[(size_t)ds.offset - 1];
        const V2 u_light_0{radical_inverse(dim, hi), radical_inverse(dim + 1, hi)};
        dim += 5;  // u_light_1 (2) and the time (1) are drawn and ignored by the δ-lights
        const LeSample ls = sample_le(light, u_light_0);
        if (ls.pdf_pos == 0.0f || ls.pdf_dir == 0.0f || is_black(ls.le)) continue;
        Ray photon_ray = ls.ray;
        const RGB beta = std::fabs(dot(ls.light_normal, photon_ray.d)) * ls.le / (ds.pdf * ls.pdf_pos * ls.pdf_dir);
        if (is_black(beta)) continue;
        const float beta_y = to_Y(beta);
        int depth = 1;
        while (depth <= prm.max_depth) {
            SurfaceInteraction si;
            if (!scene_intersect(scene, photon_ray, si)) break;
            if (depth > 1 && st.grid_valid) {
                const GridPoint gp = to_grid(si.p, st.grid_bounds, st.grid_res);
                if (gp.in_bounds) {
                    hits_total++;
                    const uint64_t h = grid_hash(gp.g[0], gp.g[1], gp.g[2], n_pixels);
                    for (int64_t node = grid.head[(size_t)h]; node >= 0; node = grid.next[(size_t)node]) {
                        SPPMPixel& px = st.pixels[grid.node_pixel[(size_t)node]];
                        if (distance_squared(px.vp.p, si.p) > px.radius * px.radius) continue;
                        const RGB phi = beta * px.vp.bsdf.f(px.vp.wo, -photon_ray.d);  // β is the emission weight: never updated (A.13)
#pragma omp atomic
                        px.phi[0] += phi.x;  // Threads.Atomic{Float32} adds, sppm.jl:398
#pragma omp atomic
                        px.phi[1] += phi.y;
#pragma omp atomic
                        px.phi[2] += phi.z;
#pragma omp atomic
                        px.M += 1;
                    }
                }
            }
            const BSDF bsdf = compute_scattering(scene, si, true);  // TransportMode changes nothing (A.11)
            if (!bsdf.valid) {
                photon_ray = spawn_ray_dir(si, photon_ray.d);
                continue;
            }
            const V2 u{radical_inverse(dim, hi), radical_inverse(dim + 1, hi)};
            dim += 2;
            const BSDFSample s = bsdf_sample_f(bsdf, -photon_ray.d, u, BSDF_ALL);
            if (is_black(s.f) || s.pdf == 0.0f) break;
            const RGB beta_new = beta * s.f * std::fabs(dot(s.wi, si.sh_n)) / s.pdf;
            const float q = jl_max(0.0f, 1.0f - to_Y(beta_new) / beta_y);
            const float rr = radical_inverse(dim, hi);
            dim += 1;
            if (rr < q) break;
            photon_ray = spawn_ray_dir(si, s.wi);
            depth += 1;
        }
    }
#pragma omp critical(orc_sppm_totals)
    {
        st.totals.closest += counters().closest, st.totals.shadow += counters().shadow, st.totals.nodes += counters().nodes, st.totals.prims += counters().prims;
    }
    }
    st.photon_hits += hits_total;
}

// sppm.jl:438-459
inline void sppm_update_pixels(SPPMState& st, float gamma) {
    for (SPPMPixel& px : st.pixels) {
        if (px.M > 0) {
            const double n_new = px.N + (double)(gamma * (float)px.M);  // γ * M is Float32 * Int64 -> Float32
            const double radius_new = (double)px.radius * std::sqrt(n_new / (px.N + (double)px.M));
            const double ratio = radius_new / (double)px.radius, r2 = ratio * ratio;
            const RGB sum = px.tau + RGB(px.phi[0], px.phi[1], px.phi[2]);
            px.tau = RGB((float)((double)sum.x * r2), (float)((double)sum.y * r2), (float)((double)sum.z * r2));
            px.radius = (float)radius_new;
            px.N = n_new;
            px.phi[0] = px.phi[1] = px.phi[2] = 0.0f;
            px.M = 0;
        }
        px.vp.beta = RGB(0.0f);
        px.vp.bsdf = BSDF();
    }
}

// sppm.jl:461-472: image[y, x] as RGB triples, (height, width, 3) row-major
inline void sppm_to_image(const SPPMState& st, int64_t iteration, float* image) {
    const double Np = (double)(iteration * st.photons_per_iteration) * 3.141592653589793;
    for (size_t i = 0; i < st.pixels.size(); ++i) {
        const SPPMPixel& p = st.pixels[i];
        const RGB a = p.Ld / (float)iteration;
        const double den = Np * (double)(p.radius * p.radius);
        const RGB b((float)((double)p.tau.x / den), (float)((double)p.tau.y / den), (float)((double)p.tau.z / den));
        const RGB c = a + b;
        image[3 * i + 0] = c.x;
        image[3 * i + 1] = c.y;
        image[3 * i + 2] = c.z;
    }
}

// sppm.jl:132-173.  Returns false when the film's crop does not start at (1, 1) (the reference indexes `pixels[y, x]` with
// raster coordinates and would throw).
inline bool sppm_render(Scene& scene, const PerspectiveCamera& cam, const Film& film, const SPPMParams& prm, SPPMState& st, float* image) {
    const Bounds2 pb = film.crop_bounds;
    if (pb.p_min.x != 1.0f || pb.p_min.y != 1.0f) return false;
    st.width = (int)inclusive_side(pb.p_max.x, pb.p_min.x);
    st.height = (int)inclusive_side(pb.p_max.y, pb.p_min.y);
    const uint64_t n_pixels = (uint64_t)((float)st.width * (float)st.height);
    st.pixels.assign((size_t)st.width * st.height, SPPMPixel{});
    for (SPPMPixel& p : st.pixels) p.radius = prm.initial_search_radius;
    st.photons_per_iteration = prm.photons_per_iteration > 0 ? prm.photons_per_iteration : (int64_t)((pb.p_max.x - pb.p_min.x) * (pb.p_max.y - pb.p_min.y));
    const float gamma = 2.0f / 3.0f;
    const Distribution1D light_distr = compute_light_power_distribution(scene);
    SPPMGrid grid;
    st.totals = Counters{};
    std::vector<float> x_phi;
    std::vector<int64_t> x_M;
    for (int64_t it = 1; it <= prm.n_iterations; ++it) {
        sppm_camera_pass(scene, cam, film, prm, st, it);
        sppm_populate_grid(st, grid, n_pixels);
        if (!scene.lights.empty()) sppm_trace_photons(scene, prm, st, grid, light_distr, it, n_pixels);
        if (prm.exchange) {  // multi-process job: ϕ and M summed over the processes before _update_pixels!
            const size_t n = st.pixels.size();
            x_phi.resize(3 * n);
            x_M.resize(n);
            for (size_t i = 0; i < n; ++i) {
                for (int c = 0; c < 3; ++c) x_phi[3 * i + c] = st.pixels[i].phi[c];
                x_M[i] = st.pixels[i].M;
            }
            prm.exchange(prm.exchange_user, x_phi.data(), x_M.data(), (uint64_t)n);
            for (size_t i = 0; i < n; ++i) {
                for (int c = 0; c < 3; ++c) st.pixels[i].phi[c] = x_phi[3 * i + c];
                st.pixels[i].M = x_M[i];
            }
        }
        if (it == prm.n_iterations) {
            const size_t n = st.pixels.size();
            st.last_M.resize(n);
            st.last_phi.resize(3 * n);
            st.last_vp_p.resize(3 * n);
            st.last_vp_beta.resize(3 * n);
            for (size_t i = 0; i < n; ++i) {
                const SPPMPixel& p = st.pixels[i];
                st.last_M[i] = p.M;
                for (int c = 0; c < 3; ++c) st.last_phi[3 * i + c] = p.phi[c];
                st.last_vp_p[3 * i + 0] = p.vp.p.x, st.last_vp_p[3 * i + 1] = p.vp.p.y, st.last_vp_p[3 * i + 2] = p.vp.p.z;
                st.last_vp_beta[3 * i + 0] = p.vp.beta.x, st.last_vp_beta[3 * i + 1] = p.vp.beta.y, st.last_vp_beta[3 * i + 2] = p.vp.beta.z;
            }
        }
        sppm_update_pixels(st, gamma);
        st.iteration = it;
    }
    if (image) sppm_to_image(st, prm.n_iterations, image);
    return true;
}

}  // namespace orc
