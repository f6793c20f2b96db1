// oracle/orc_render.h — TEST INFRASTRUCTURE ONLY.
// Restates filter.jl, film.jl, camera/{camera,perspective}.jl, sampler/sampler.jl:129-151 (protocol; values come from
// include/trace_sampler.h because the reference sampler is unseeded), integrators/sampler.jl
// (driver + WhittedIntegrator) and defines PathIntegrator out of integrators/sppm.jl:208-266, 503-554.
#pragma once
#include "../include/trace_sampler.h"
#include "orc_scatter.h"

namespace orc {

// ---- filter.jl --------------------------------------------------------------------------------------------------------
struct LanczosSincFilter {
    V2 radius{1, 1};
    float tau = 3;
};
inline float sinc(float x) {  // filter.jl:12-17
    x = std::fabs(x);
    if (x < 1e-5f) return 1.0f;
    x *= PI_F;
    return tm_sinf(x) / x;
}
inline float windowed_sinc(float x, float r, float tau) {  // filter.jl:19-23
    x = std::fabs(x);
    if (x > r) return 0.0f;
    return sinc(x) * sinc(x / tau);
}
inline float filter_eval(const LanczosSincFilter& f, V2 p) { return windowed_sinc(p.x, f.radius.x, f.tau) * windowed_sinc(p.y, f.radius.y, f.tau); }  // :8-10

// ---- film.jl ----------------------------------------------------------------------------------------------------------
struct Pixel {  // film.jl:1-5
    V3 xyz;
    float filter_weight_sum = 0;
    V3 splat_xyz;
};
inline float inclusive_side(float b1, float b0) { return std::fabs(b1 - (b0 - 1.0f)); }  // bounds.jl:100-102
struct Film {
    V2 resolution;
    Bounds2 crop_bounds;
    LanczosSincFilter filter;
    int width = 0, height = 0;  // size(pixels) == (height, width)
    std::vector<Pixel> pixels;  // (y, x), stored row-major by y
    static constexpr int table_width = 16;
    float filter_table[16][16];  // (y, x)
    float scale = 1;
    Film() = default;
    // film.jl:34-61
    Film(V2 res, Bounds2 crop, LanczosSincFilter flt, float /*diagonal*/, float scale_) : resolution(res), filter(flt), scale(scale_) {
        crop_bounds.p_min = V2{std::ceil(res.x * crop.p_min.x) + 1.0f, std::ceil(res.y * crop.p_min.y) + 1.0f};
        crop_bounds.p_max = V2{std::ceil(res.x * crop.p_max.x), std::ceil(res.y * crop.p_max.y)};
        width = (int)inclusive_side(crop_bounds.p_max.x, crop_bounds.p_min.x);
        height = (int)inclusive_side(crop_bounds.p_max.y, crop_bounds.p_min.y);
        pixels.assign((size_t)width * height, Pixel{});
        const V2 r{filter.radius.x / (float)table_width, filter.radius.y / (float)table_width};
        for (int y = 0; y < table_width; ++y)
            for (int x = 0; x < table_width; ++x) filter_table[y][x] = filter_eval(filter, V2{((float)x + 0.5f) * r.x, ((float)y + 0.5f) * r.y});
    }
    Pixel& at(float x, float y) {  // get_pixel(f, p) film.jl:176-180
        const int px = (int)(x - crop_bounds.p_min.x + 1.0f), py = (int)(y - crop_bounds.p_min.y + 1.0f);
        return pixels[(size_t)(py - 1) * width + (px - 1)];
    }
};
// film.jl:68-73
inline Bounds2 get_sample_bounds(const Film& f) {
    return {{std::floor(f.crop_bounds.p_min.x + 0.5f - f.filter.radius.x), std::floor(f.crop_bounds.p_min.y + 0.5f - f.filter.radius.y)},
            {std::ceil(f.crop_bounds.p_max.x - 0.5f + f.filter.radius.x), std::ceil(f.crop_bounds.p_max.y - 0.5f + f.filter.radius.y)}};
}
struct FilmTilePixel {
    RGB contrib_sum;
    float filter_weight_sum = 0;
};
struct FilmTile {  // film.jl:92-125
    Bounds2 bounds;
    V2 filter_radius, inv_filter_radius;
    const Film* film = nullptr;
    int width = 0, height = 0;
    std::vector<FilmTilePixel> pixels;
    FilmTile(const Film& f, const Bounds2& sample_bounds) : film(&f) {
        const V2 p0{std::ceil(sample_bounds.p_min.x - 0.5f - f.filter.radius.x), std::ceil(sample_bounds.p_min.y - 0.5f - f.filter.radius.y)};
        const V2 p1{std::floor(sample_bounds.p_max.x - 0.5f + f.filter.radius.x) + 1.0f, std::floor(sample_bounds.p_max.y - 0.5f + f.filter.radius.y) + 1.0f};
        bounds = bintersect(Bounds2{p0, p1}, f.crop_bounds);
        filter_radius = f.filter.radius;
        inv_filter_radius = V2{1.0f / f.filter.radius.x, 1.0f / f.filter.radius.y};
        width = (int)inclusive_side(bounds.p_max.x, bounds.p_min.x);
        height = (int)inclusive_side(bounds.p_max.y, bounds.p_min.y);
        pixels.assign((size_t)width * height, FilmTilePixel{});
    }
    FilmTilePixel& at(float x, float y) {  // get_pixel(t, p) film.jl:169-172
        const int px = (int)(x - bounds.p_min.x + 1.0f), py = (int)(y - bounds.p_min.y + 1.0f);
        return pixels[(size_t)(py - 1) * width + (px - 1)];
    }
};
// film.jl:134-164 (quirks A.9: inclusive extra row/column, ceil for x but floor for y)
inline void add_sample(FilmTile& t, V2 point, RGB spectrum, float sample_weight = 1.0f) {
    const V2 dp{point.x - 0.5f, point.y - 0.5f};
    V2 p0{std::ceil(dp.x - t.filter_radius.x), std::ceil(dp.y - t.filter_radius.y)};
    V2 p1{std::floor(dp.x + t.filter_radius.x) + 1.0f, std::floor(dp.y + t.filter_radius.y) + 1.0f};
    p0 = V2{jl_max(p0.x, jl_max(t.bounds.p_min.x, 1.0f)), jl_max(p0.y, jl_max(t.bounds.p_min.y, 1.0f))};
    p1 = V2{jl_min(p1.x, t.bounds.p_max.x), jl_min(p1.y, t.bounds.p_max.y)};
    const float tw = (float)Film::table_width;
    for (float y = p0.y; y <= p1.y; y += 1.0f) {
        const float fy = std::fabs((y - dp.y) * t.inv_filter_radius.y * tw);
        const int oy = (int)jl_clamp(std::floor(fy), 1.0f, tw);
        for (float x = p0.x; x <= p1.x; x += 1.0f) {
            const float fx = std::fabs((x - dp.x) * t.inv_filter_radius.x * tw);
            const int ox = (int)jl_clamp(std::ceil(fx), 1.0f, tw);
            const float w = t.film->filter_table[oy - 1][ox - 1];
            FilmTilePixel& px = t.at(x, y);
            px.contrib_sum = px.contrib_sum + spectrum * sample_weight * w;
            px.filter_weight_sum += w;
        }
    }
}
// film.jl:182-193
inline void merge_film_tile(Film& f, FilmTile& ft) {
    for (float y = ft.bounds.p_min.y; y <= ft.bounds.p_max.y; y += 1.0f)
        for (float x = ft.bounds.p_min.x; x <= ft.bounds.p_max.x; x += 1.0f) {
            FilmTilePixel& tp = ft.at(x, y);
            Pixel& mp = f.at(x, y);
            mp.xyz = mp.xyz + RGB_to_XYZ(tp.contrib_sum);
            mp.filter_weight_sum += tp.filter_weight_sum;
        }
}
// film.jl:204-222 up to (not including) the PNG encoder: linear RGB in [0,1], rows NOT yet flipped.
inline void film_to_rgb(const Film& f, float* out_rgb /* height*width*3 */, float splat_scale = 1.0f) {
    for (int y = 0; y < f.height; ++y)
        for (int x = 0; x < f.width; ++x) {
            const Pixel& p = f.pixels[(size_t)y * f.width + x];
            V3 c = XYZ_to_RGB(p.xyz);
            if (p.filter_weight_sum != 0) {
                const float inv_w = 1.0f / p.filter_weight_sum;
                c = V3(jl_max(0.0f, c.x * inv_w), jl_max(0.0f, c.y * inv_w), jl_max(0.0f, c.z * inv_w));
            }
            const V3 s = XYZ_to_RGB(p.splat_xyz);
            c = c + splat_scale * s;
            c = c * f.scale;
            float* o = out_rgb + ((size_t)y * f.width + x) * 3;
            o[0] = jl_clamp(c.x, 0.0f, 1.0f);
            o[1] = jl_clamp(c.y, 0.0f, 1.0f);
            o[2] = jl_clamp(c.z, 0.0f, 1.0f);
        }
}

// ---- camera/perspective.jl ---------------------------------------------------------------------------------------------
struct CameraSample {  // camera.jl:10-27
    V2 film, lens;
    float time = 0;
};
struct PerspectiveCamera {
    Transformation camera_to_world, camera_to_screen, raster_to_camera, screen_to_raster, raster_to_screen;
    float shutter_open = 0, shutter_close = 1, lens_radius = 0, focal_distance = 1e6f;
    PerspectiveCamera() = default;
    // perspective.jl:11-40 + 58-80 (near = 0.01, far = 1000 hard-coded at :65)
    PerspectiveCamera(const Transformation& c2w, Bounds2 screen_window, float so, float sc, float lr, float fd, float fov, V2 film_resolution)
        : camera_to_world(c2w), shutter_open(so), shutter_close(sc), lens_radius(lr), focal_distance(fd) {
        camera_to_screen = perspective(fov, 0.01f, 1000.0f);
        screen_to_raster = scale(film_resolution.x, film_resolution.y, 1) *
                           scale(1.0f / (screen_window.p_max.x - screen_window.p_min.x), 1.0f / (screen_window.p_max.y - screen_window.p_min.y), 1) *
                           translate(V3(-screen_window.p_min.x, -screen_window.p_max.y, 0.0f));
        raster_to_screen = inv(screen_to_raster);
        raster_to_camera = inv(camera_to_screen) * raster_to_screen;
    }
};
// perspective.jl:85-114
inline Ray generate_ray(const PerspectiveCamera& cam, const CameraSample& s) {
    const V3 p_film(s.film.x, s.film.y, 0.0f);
    const V3 p_camera = cam.raster_to_camera.point(p_film);
    Ray ray{V3(0.0f), normalize(p_camera), INF32, 0.0f};
    if (cam.lens_radius > 0) {
        const V2 p_lens = cam.lens_radius * concentric_sample_disk(s.lens);
        const float t = cam.focal_distance / ray.d.z;
        const V3 p_focus = ray.at(t);
        ray.o = V3(p_lens.x, p_lens.y, 0.0f);
        ray.d = normalize(p_focus - ray.o);
    }
    ray.time = lerp(cam.shutter_open, cam.shutter_close, s.time);
    ray = cam.camera_to_world.ray(ray);
    ray.d = normalize(ray.d);
    return ray;
}

// ---- sampler protocol (sampler/sampler.jl:129-151) with the build's seeded values ---------------------------------------
struct SeededSampler {
    int64_t current_sample = 1;
    int64_t samples_per_pixel = 1;
    uint64_t seed = 0;
    uint32_t sample_offset = 0;  // first global sample index (multi-GPU sharding gives each rank its own range)
    int32_t px = 0, py = 0;
    uint64_t key = 0;
    SeededSampler(int64_t spp, uint64_t seed_, uint32_t offset = 0) : samples_per_pixel(spp), seed(seed_), sample_offset(offset) {}
    void rekey() { key = ts_stream_key(seed, px, py, sample_offset + (uint32_t)(current_sample - 1)); }
    void start_pixel(V2 p) {  // :147-149
        current_sample = 1;
        px = (int32_t)p.x;
        py = (int32_t)p.y;
        rekey();
    }
    bool has_next_sample() const { return current_sample <= samples_per_pixel; }  // :141-143
    void start_next_sample() {                                                     // :144-146
        current_sample += 1;
        rekey();
    }
    float u(uint32_t dim) const { return ts_uniform(key, dim); }
    // :135-139: p_film = p_raster + rand2; p_lens = rand2; time = rand
    CameraSample get_camera_sample(V2 p_raster) const {
        CameraSample cs;
        cs.film = V2{p_raster.x + u(TS_DIM_FILM_X), p_raster.y + u(TS_DIM_FILM_Y)};
        cs.lens = V2{u(TS_DIM_LENS_X), u(TS_DIM_LENS_Y)};
        cs.time = u(TS_DIM_TIME);
        return cs;
    }
};

// ---- integrators ---------------------------------------------------------------------------------------------------------
struct RenderStats {
    uint64_t camera_samples = 0, closest_rays = 0, shadow_rays = 0, nodes_visited = 0, prims_tested = 0;
};

// integrators/sampler.jl:58-101 + 103-143 + 145-199.  The get_2d() values Whitted draws are ignored by δ-lights and by
// single specular lobes (A.15), so none are consumed here.
inline RGB whitted_li(Scene& scene, Ray ray, int max_depth, int depth) {
    RGB l(0.0f);
    SurfaceInteraction si;
    if (!scene_intersect(scene, ray, si)) return l;  // Σ le(light, ray) = 0 (light.jl:41)
    const V3 n = si.sh_n;
    const V3 wo = si.wo;
    const BSDF bsdf = compute_scattering(scene, si, false);
    if (!bsdf.valid) return l;  // :77-80 calls a non-existent method; unreachable when every primitive has a material
    for (const Light& light : scene.lights) {
        const LightSample ls = sample_li(light, si.p, si.time);
        if (is_black(ls.radiance) || ls.pdf == 0.0f) continue;
        const RGB f = bsdf.f(wo, ls.wi);
        if (!is_black(f) && unoccluded(scene, ls)) l = l + f * ls.radiance * std::fabs(dot(ls.wi, n)) / ls.pdf;
    }
    if (depth + 1 <= max_depth) {
        for (int pass = 0; pass < 2; ++pass) {  // specular_reflect then specular_transmit
            const uint8_t type = (pass == 0 ? BSDF_REFLECTION : BSDF_TRANSMISSION) | BSDF_SPECULAR;
            const BSDFSample s = bsdf_sample_f(bsdf, wo, V2{0, 0}, type);
            const V3 ns = si.sh_n;
            if (!(s.pdf > 0.0f && !is_black(s.f) && std::fabs(dot(s.wi, ns)) != 0.0f)) continue;
            const Ray rd = spawn_ray_dir(si, s.wi);
            l = l + s.f * whitted_li(scene, rd, max_depth, depth + 1) * std::fabs(dot(s.wi, ns)) / s.pdf;
        }
    }
    return l;
}

// sppm.jl:503-554 for δ-lights (the only kind, F6)
inline RGB uniform_sample_one_light(Scene& scene, const SurfaceInteraction& si, const BSDF& bsdf, float u_pick) {
    const int n_lights = (int)scene.lights.size();
    if (n_lights == 0) return RGB(0.0f);
    int light_num = (int)std::ceil(u_pick * (float)n_lights);
    if (light_num > n_lights) light_num = n_lights;
    if (light_num < 1) light_num = 1;
    const float light_pdf = 1.0f / (float)n_lights;
    const Light& light = scene.lights[light_num - 1];
    // estimate_direct :520-554
    const uint8_t flags = BSDF_ALL & ~BSDF_SPECULAR;
    RGB Ld(0.0f);
    const LightSample ls = sample_li(light, si.p, si.time);
    RGB Li = ls.radiance;
    if (ls.pdf > 0 && !is_black(Li)) {
        const RGB f = bsdf.f(si.wo, ls.wi, flags) * std::fabs(dot(ls.wi, si.sh_n));
        if (!is_black(f)) {
            if (!unoccluded(scene, ls)) Li = RGB(0.0f);
            if (!is_black(Li)) Ld = Ld + f * Li / ls.pdf;
        }
    }
    return Ld / light_pdf;
}

// PathIntegrator.li — defined by this build (no reference counterpart: parity of the composite is unpinned, SURVEY.md
// F2/§8 a5).  It is the SPPM camera-pass loop (sppm.jl:208-266) without the visible-point early-out (:239-245), with β
// applied to the direct term (which :229 omits, A.12) and Russian roulette exactly as :257-263.
inline RGB path_li(Scene& scene, Ray ray, const SeededSampler& smp, int max_depth) {
    RGB L(0.0f), beta(1.0f);
    int depth = 1;
    while (depth <= max_depth) {
        SurfaceInteraction si;
        if (!scene_intersect(scene, ray, si)) break;  // background radiance is 0
        const BSDF bsdf = compute_scattering(scene, si, true);
        if (!bsdf.valid) {  // :219-222 (depth does not advance, A.12)
            ray = spawn_ray_dir(si, ray.d);
            continue;
        }
        const V3 wo = -ray.d;
        const uint32_t v = (uint32_t)(depth - 1);
        L = L + beta * uniform_sample_one_light(scene, si, bsdf, smp.u(ts_vertex_dim(v, TS_V_LIGHT_PICK)));
        if (depth == max_depth) break;  // :247 `depth == i.max_depth && (depth += 1; continue)`
        const V2 u{smp.u(ts_vertex_dim(v, TS_V_BSDF_U0)), smp.u(ts_vertex_dim(v, TS_V_BSDF_U1))};
        const BSDFSample s = bsdf_sample_f(bsdf, wo, u, BSDF_ALL);
        if (s.pdf == 0.0f || is_black(s.f)) break;
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
    return L;
}

enum IntegratorKind { INTEGRATOR_WHITTED = 0, INTEGRATOR_PATH = 1 };

// integrators/sampler.jl:12-56 — tiles of 16x16 sample-pixels in k order; tiles run sequentially here (the reference
// merges tiles from several threads without a lock, so its result is only defined up to summation order / the race).
// If sample_L is non-null it receives every sample's radiance after the NaN->0 rule, indexed
// [(s * n_pix) + (y - sb.min.y) * sb_width + (x - sb.min.x)] * 3.
inline void render(Scene& scene, const PerspectiveCamera& cam, Film& film, IntegratorKind kind, int64_t spp, int max_depth, uint64_t seed,
                   uint32_t sample_offset, float* sample_L, RenderStats* stats) {
    const Bounds2 sb = get_sample_bounds(film);
    const V2 extent{sb.p_max.x - sb.p_min.x, sb.p_max.y - sb.p_min.y};
    const int tile_size = 16;
    const long long width = (long long)std::floor((extent.x + tile_size) / tile_size), height = (long long)std::floor((extent.y + tile_size) / tile_size);
    const long long total_tiles = width * height - 1;
    const int sbw = (int)(sb.p_max.x - sb.p_min.x) + 1, sbh = (int)(sb.p_max.y - sb.p_min.y) + 1;
    const size_t n_pix = (size_t)sbw * sbh;
    counters() = Counters{};
    uint64_t n_samples = 0;
    for (long long k = 0; k <= total_tiles; ++k) {
        const float tx = (float)(k % width), ty = (float)(k / width);
        SeededSampler smp(spp, seed, sample_offset);  // deepcopy(i.sampler)
        const V2 tb_min{sb.p_min.x + tx * tile_size, sb.p_min.y + ty * tile_size};
        const V2 tb_max{jl_min(tb_min.x + (tile_size - 1), sb.p_max.x), jl_min(tb_min.y + (tile_size - 1), sb.p_max.y)};
        const Bounds2 tile_bounds{tb_min, tb_max};
        FilmTile tile(film, tile_bounds);
        for (float py = tb_min.y; py <= tb_max.y; py += 1.0f)
            for (float px = tb_min.x; px <= tb_max.x; px += 1.0f) {  // Bounds2 iteration is x-fastest (bounds.jl:39-47)
                const V2 pixel{px, py};
                smp.start_pixel(pixel);
                while (smp.has_next_sample()) {
                    const CameraSample cs = smp.get_camera_sample(pixel);
                    const Ray ray = generate_ray(cam, cs);  // the two extra differential rays are dead data (A.10)
                    RGB l = kind == INTEGRATOR_WHITTED ? whitted_li(scene, ray, max_depth, 1) : path_li(scene, ray, smp, max_depth);
                    if (has_nan(l)) l = RGB(0.0f);  // :46
                    if (sample_L) {
                        const size_t pix = (size_t)(py - sb.p_min.y) * sbw + (size_t)(px - sb.p_min.x);
                        float* o = sample_L + ((size_t)(smp.current_sample - 1) * n_pix + pix) * 3;
                        o[0] = l.x;
                        o[1] = l.y;
                        o[2] = l.z;
                    }
                    add_sample(tile, cs.film, l, 1.0f);
                    n_samples++;
                    smp.start_next_sample();
                }
            }
        merge_film_tile(film, tile);
    }
    if (stats) {
        stats->camera_samples = n_samples;
        stats->closest_rays = counters().closest;
        stats->shadow_rays = counters().shadow;
        stats->nodes_visited = counters().nodes;
        stats->prims_tested = counters().prims;
    }
}

}  // namespace orc
