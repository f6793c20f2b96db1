/* trace_sampler.h — the seeded, counter-based sampler of this build.
 *
 * The reference's only working sampler, `UniformSampler` (src/sampler/sampler.jl:129-151), draws from Julia's
 * global task-local RNG: no seed, not reproducible across thread counts.  "Fixed Sampler seed" therefore needs a
 * sampler of our own that keeps the reference's call protocol (start_pixel!, has_next_sample,
 * start_next_sample!, get_camera_sample, get_1d, get_2d, samples_per_pixel) but whose values are a pure function
 *
 *      u = ts_uniform(ts_stream_key(seed, pixel_x, pixel_y, sample_index), dimension)  in [0, 1)
 *
 * so that a breadth-first (wavefront, GPU) and a depth-first (recursive, CPU) evaluation of the same path
 * consume identical numbers.  `dimension` advances in the reference's consumption order:
 *
 *   camera sample (sampler/sampler.jl:135-139):  0,1 = film x,y   2,3 = lens x,y   4 = time
 *   path vertex v = 0,1,2,... (integrators/sppm.jl:509-514, 249-252, 260), base = 5 + 8 v:
 *      base+0 = light pick   base+1,2 = u_light   base+3,4 = u_scatter   base+5,6 = BSDF sample_f   base+7 = RR
 *
 * The generator is SplitMix64 used as a counter-based hash (state = key + (dim+1)*golden, output = mix(state)),
 * top 24 bits -> Float32 in [0,1) exactly representable.
 * This header is specification shared by oracle/, the HIP library and any host shim; it is not oracle code.
 */
#ifndef TRACE_SAMPLER_H
#define TRACE_SAMPLER_H

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define TS_HD __host__ __device__ inline
#else
#define TS_HD inline
#endif

#define TS_GOLDEN 0x9E3779B97F4A7C15ULL

enum {
    TS_DIM_FILM_X = 0,
    TS_DIM_FILM_Y = 1,
    TS_DIM_LENS_X = 2,
    TS_DIM_LENS_Y = 3,
    TS_DIM_TIME = 4,
    TS_DIM_VERTEX_BASE = 5,
    TS_DIM_VERTEX_STRIDE = 8,
    TS_V_LIGHT_PICK = 0,
    TS_V_LIGHT_U0 = 1,
    TS_V_LIGHT_U1 = 2,
    TS_V_SCATTER_U0 = 3,
    TS_V_SCATTER_U1 = 4,
    TS_V_BSDF_U0 = 5,
    TS_V_BSDF_U1 = 6,
    TS_V_RR = 7
};

TS_HD uint64_t ts_mix64(uint64_t z) {
    z ^= z >> 30;
    z *= 0xBF58476D1CE4E5B9ULL;
    z ^= z >> 27;
    z *= 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return z;
}

/* pixel_x / pixel_y are the reference's 1-based raster pixel coordinates (may be <= 0 inside the filter border,
 * film.jl:68-73); sample_index is 0-based (= UniformSampler.current_sample - 1). */
TS_HD uint64_t ts_stream_key(uint64_t seed, int32_t pixel_x, int32_t pixel_y, uint32_t sample_index) {
    const uint64_t pix = (uint64_t)(uint32_t)pixel_x | ((uint64_t)(uint32_t)pixel_y << 32);
    return ts_mix64(ts_mix64(seed ^ pix) + (uint64_t)sample_index * TS_GOLDEN);
}

TS_HD float ts_uniform(uint64_t key, uint32_t dim) {
    const uint64_t z = ts_mix64(key + (uint64_t)(dim + 1u) * TS_GOLDEN);
    return (float)(uint32_t)(z >> 40) * 5.9604644775390625e-08f; /* 2^-24 */
}

TS_HD uint32_t ts_vertex_dim(uint32_t vertex, uint32_t slot) {
    return (uint32_t)TS_DIM_VERTEX_BASE + (uint32_t)TS_DIM_VERTEX_STRIDE * vertex + slot;
}

#endif /* TRACE_SAMPLER_H */
