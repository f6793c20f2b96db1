// th_kernels.h — the gfx950 kernels of the wavefront engine.
//
//   k_raygen          camera samples -> ray queue            (sampler/sampler.jl:135-139, camera/perspective.jl:85-114)
//   k_trace<ANY,…>    BVH2 closest-hit / any-hit traversal    (accel/bvh.jl:212-299, bounds.jl:186-206, shapes/*.jl)
//   k_shade_path      PathIntegrator vertex: interaction, BSDF, light sample -> shadow queue, sample_f + RR -> next queue
//   k_film_gather     add_sample! + merge_film_tile! as a deterministic gather in the reference's summation order
//
// Execution model: persistent grid-stride kernels (no host round trip per bounce: queue sizes live in HBM),
// one ray per lane, 64-wide waves, per-lane traversal stack staged in LDS as stack[level][lane] (conflict-free: lane l
// and l+32 hit the same bank in different half-wave groups), live paths compacted per bounce with a 64-bit ballot +
// popcount prefix and ONE atomic per wave.  No MFMA anywhere: the path is branchy gather, not a contraction.
#pragma once
#include "th_device.h"

namespace th {

constexpr int kBlock = 256;
constexpr int kStackLds = 32;    // stack levels kept in LDS (8 KiB per wave)
constexpr int kStackSpill = 32;  // further levels in scratch; 64 in total like bvh.jl:222

struct PathQueue {
    float4* o;     // o.xyz, as_float(slot)
    float4* d;     // d.xyz, unused
    float4* beta;  // β.rgb, unused
};
struct ShadowQueue {
    float4* o;  // o.xyz, as_float(slot)
    float4* d;  // d.xyz, as_float(poison bits: bit c set = β[c] is not finite)
    float4* c;  // β·Ld to add when unoccluded
};
constexpr int kMaxDepth = 62;
// Queues are split into kSeg segments, each with its own fill counter (and its own k_trace2 work cursor): a single hot
// counter word serialises at ~88 returning atomics/us on MI355X, which at one atomic per wave per bounce was the whole
// run time of the shading kernel; 32 words on different L2 channels are not a bottleneck.
constexpr int kSeg = 32;
constexpr uint32_t kSegGran = 256;  // a segment's share of the flat work space is padded to a multiple of this
#ifndef TH_CTR_STRIDE
#define TH_CTR_STRIDE 32
#endif
constexpr uint32_t kCtrStride = TH_CTR_STRIDE;  // words between the counters of neighbouring segments: one 128-byte line each, so that the
                                     // atomics of different segments go to different L2 channels instead of queueing on one line
struct Counters {  // device-resident
    // per wavefront batch (zeroed by one memset at batch start): first index = path depth - 1, second = segment * kCtrStride
    uint32_t n_queue[kMaxDepth + 2][kSeg * kCtrStride];       // live paths entering that depth, per segment
    uint32_t n_shadow[kMaxDepth + 2][kSeg * kCtrStride];      // shadow rays emitted at that depth
    uint32_t work_closest[kMaxDepth + 2][kSeg * kCtrStride];  // k_trace2 dynamic ray-fetch cursors
    uint32_t work_shadow[kMaxDepth + 2][kSeg * kCtrStride];
    // per render call
    unsigned long long closest_total, shadow_total, nodes_closest, prims_closest, nodes_shadow, prims_shadow;
    unsigned long long fallback_total;  // closest-hit rays k_trace7 handed to the reference-order walk (th_trace7.h)
    unsigned long long fallback_why[4]; // … by reason, counted under "count_visits" (th_trace7.h)
    // hybrid mode under "count_visits": the visit counters split between the certified walk and the fallback walks (k_hybrid_count_mark, th_trace3c.h)
    unsigned long long nodes_seen, prims_seen, nodes_fallback, prims_fallback;
};
// How a kernel sees a queue: kSeg segments of `cap` physical entries with fill counts in HBM, or (counts == nullptr) one
// dense array of n_dense entries (kernel-level API entry points).
struct SegQueue {
    const uint32_t* counts;
    uint32_t cap;
    uint32_t n_dense;
    const uint32_t* indirect;  // optional: entry -> index into the ray arrays (k_any_occluders' survivor lists); null = the entry is the index
    uint32_t no_total;         // 1: the rays of this queue were already counted (k_trace8's fallback list): do not add them to the ray totals
};
struct SegView {  // per-block copy in LDS
    uint32_t count[kSeg];
    uint32_t prefix[kSeg + 1];  // padded prefix sums: the flat work space
};
// Block-wide: load the counts and build the padded prefix.  Contains a barrier.
TH_D void seg_load(const SegQueue& q, SegView& v) {
    if (threadIdx.x < kSeg) v.count[threadIdx.x] = q.counts ? min(q.counts[threadIdx.x * kCtrStride], q.cap) : (threadIdx.x == 0 ? q.n_dense : 0u);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t acc = 0;
        for (int s = 0; s < kSeg; ++s) {
            v.prefix[s] = acc;
            acc += (v.count[s] + kSegGran - 1) / kSegGran * kSegGran;
        }
        v.prefix[kSeg] = acc;
    }
    __syncthreads();
}
TH_D uint32_t seg_total(const SegView& v) {
    uint32_t t = 0;
    for (int s = 0; s < kSeg; ++s) t += v.count[s];
    return t;
}
// flat_base: wave-uniform multiple of 64 inside [0, prefix[kSeg]).  Returns the segment and the local index of lane 0.
TH_D void seg_locate(const SegView& v, uint32_t flat_base, uint32_t& seg, uint32_t& local_base) {
    uint32_t s = 0;
    while (s + 1 < (uint32_t)kSeg && flat_base >= v.prefix[s + 1]) ++s;
    seg = s;
    local_base = flat_base - v.prefix[s];
}
// … for a grid-stride loop, whose flat_base only grows: `seg` carries over from the iteration before (start it at 0), so the search is a step or two — and it is NOT unrolled:
// unrolled, the compiler hoists all kSeg prefix sums out of the caller's loop into 32 VGPRs (k_shade_path's FAST part: 119 -> 84 registers without them).
TH_D void seg_locate_from(const SegView& v, uint32_t flat_base, uint32_t& seg, uint32_t& local_base) {
    uint32_t s = seg;
#pragma unroll 1
    while (s + 1 < (uint32_t)kSeg && flat_base >= v.prefix[s + 1]) ++s;
    seg = s;
    local_base = flat_base - v.prefix[s];
}
TH_D uint32_t seg_phys(const SegQueue& q, uint32_t seg, uint32_t local) { return q.counts ? seg * q.cap + local : local; }

TH_D uint32_t lane_id() { return __lane_id(); }
// wave-level compaction: returns this lane's output index (valid when `alive`), one atomic per wave
TH_D uint32_t wave_compact(bool alive, uint32_t* counter) {
    const unsigned long long mask = __ballot(alive);
    const uint32_t n = (uint32_t)__popcll(mask);
    uint32_t base = 0;
    const uint32_t lane = lane_id();
    const int leader = __ffsll((long long)mask) - 1;
    if (n && (int)lane == leader) base = atomicAdd(counter, n);
    base = __shfl(base, leader < 0 ? 0 : leader);
    return base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}
// Two compactions at once: both returning atomics are in flight before either result is needed (one L2 round trip, not two).
TH_D void wave_compact2(bool alive_a, uint32_t* counter_a, bool alive_b, uint32_t* counter_b, uint32_t& idx_a, uint32_t& idx_b) {
    const unsigned long long ma = __ballot(alive_a), mb = __ballot(alive_b);
    const uint32_t na = (uint32_t)__popcll(ma), nb = (uint32_t)__popcll(mb);
    const uint32_t lane = lane_id();
    uint32_t base_a = 0, base_b = 0;
    if (lane == 0) {  // ballots are wave-wide: lane 0 of every wave in the loop is active
        if (na) base_a = atomicAdd(counter_a, na);
        if (nb) base_b = atomicAdd(counter_b, nb);
    }
    base_a = __shfl(base_a, 0);
    base_b = __shfl(base_b, 0);
    const unsigned long long below = (1ull << lane) - 1ull;
    idx_a = base_a + (uint32_t)__popcll(ma & below);
    idx_b = base_b + (uint32_t)__popcll(mb & below);
}
TH_D unsigned long long wave_sum(unsigned long long v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    return v;
}

// ---- sampler helpers ------------------------------------------------------------------------------------------------------
struct SlotInfo {
    int px, py;       // 1-based raster pixel (may be <= 0 in the filter border)
    uint32_t pix;     // linear sample-pixel index
    uint32_t sample;  // sample index within this render call
};
TH_D SlotInfo slot_info(const DeviceSensor& se, uint32_t slot) {
    const uint32_t npix = (uint32_t)(se.sb_w * se.band_rows);  // slots, like the radiance buffer, cover the band being rendered (the whole frame normally)
    SlotInfo r;
    r.sample = slot / npix;
    r.pix = slot - r.sample * npix;
    const uint32_t y = r.pix / (uint32_t)se.sb_w;
    r.px = se.sb_min[0] + (int)(r.pix - y * (uint32_t)se.sb_w);
    r.py = se.band_y0 + (int)y;
    return r;
}

// ---- film: where a camera sample's radiance record lives, and its splat descriptor (used by k_raygen and the film pass below) -------
// layout 0: sample-major (the integrators' order).  layout 1: pixel-group-major — [group of 64 consecutive sample pixels][sample][position], one 1 KB chunk per (group,
// sample) — with the position inside a chunk PERMUTED: pixels p = 0, 4, 8, ... first, then 1, 5, 9, ..., 2, 6, ..., 3, 7, ...  The film gather's lanes own BX adjacent film
// pixels each, so one of its load instructions wants every BX-th sample pixel: in pixel order that is 16 bytes out of every BX x 16 (half of every cache line fetched for
// 2 x 4 blocks, a quarter for 4 x 4 — the counters of round 3 showed 10x the algorithmic bytes); permuted, the pixels of one residue class mod 4 are 256 contiguous bytes,
// and BX = 1, 2 and 4 all read whole 128-byte lines.  Everything that touches the records goes through this function.
#ifndef TH_FILM_PERM
#define TH_FILM_PERM 4
#endif
TH_D uint32_t film_chunk_pos(uint32_t p) {  // p = pix & 63
#if TH_FILM_PERM == 4
    return ((p & 3u) << 4) | (p >> 2);
#elif TH_FILM_PERM == 2
    return ((p & 1u) << 5) | (p >> 1);
#else
    return p;
#endif
}
TH_D size_t film_index(uint32_t layout, uint32_t npix, uint32_t spp, uint32_t s, uint32_t pix) {
    return layout ? ((size_t)(pix >> 6) * spp + s) * 64u + film_chunk_pos(pix & 63u) : (size_t)s * npix + pix;
}
constexpr uint32_t kFilmPackSide = 0x80000000u;
constexpr uint32_t kFilmPackOverflow = 0xffffffffu;
constexpr uint32_t kFilmPackException = 0xffffffffu;  // what film_pack_desc returns for a descriptor that does not fit
struct FilmSideTable {
    uint4* desc;
    uint32_t* count;
    uint32_t cap;
};
TH_D uint32_t film_pack_word(const FilmSideTable& side, uint4 d, int px, int py);
TH_D uint4 film_splat_desc(const DeviceSensor& se, int px, int py, uint64_t key) {  // the SplatDesc of one camera sample (k_film_descriptors' arithmetic)
    const float rx = se.filter_radius[0], ry = se.filter_radius[1];
    const float inv_rx = 1.0f / rx, inv_ry = 1.0f / ry;
    const float pfx = (float)px + ts_uniform(key, TS_DIM_FILM_X), pfy = (float)py + ts_uniform(key, TS_DIM_FILM_Y);  // camera_sample.film
    const float dpx = pfx - 0.5f, dpy = pfy - 0.5f;
    const float p0x = __builtin_ceilf(dpx - rx), p0y = __builtin_ceilf(dpy - ry);
    const float p1x = __builtin_floorf(dpx + rx) + 1.0f, p1y = __builtin_floorf(dpy + ry) + 1.0f;
    const int nx = (int)(p1x - p0x) + 1, ny = (int)(p1y - p0y) + 1;
    uint32_t oxw = 0, oyw = 0;
    for (int c = 0; c < 8; ++c) {
        const float X = p0x + (float)c, Y = p0y + (float)c;
        oxw |= (uint32_t)((int)jclamp(__builtin_ceilf(fabs_((X - dpx) * inv_rx * 16.0f)), 1.0f, 16.0f) - 1) << (4 * c);   // ceil for x …
        oyw |= (uint32_t)((int)jclamp(__builtin_floorf(fabs_((Y - dpy) * inv_ry * 16.0f)), 1.0f, 16.0f) - 1) << (4 * c);  // … floor for y (A.9)
    }
    return make_uint4(((uint32_t)(int)p0x & 0xffffu) | ((uint32_t)(int)p0y << 16), (uint32_t)nx | ((uint32_t)ny << 8), oxw, oyw);
}
// 32-bit form: bit 0 / 1 = px - p0x / py - p0y (0 or 1), bits 2-3 / 4-5 = nx - 1 / ny - 1 (0..2), bits 6-17 = table index of columns 0..2, bits 18-29 = rows 0..2
TH_D uint32_t film_pack_desc(uint4 d, int px, int py) {
    const int p0x = (int)(short)(d.x & 0xffffu), p0y = (int)(short)(d.x >> 16);
    const uint32_t nx = d.y & 0xffu, ny = (d.y >> 8) & 0xffu;
    const int ox = px - p0x, oy = py - p0y;
    if (ox < 0 || ox > 1 || oy < 0 || oy > 1 || nx < 1u || nx > 3u || ny < 1u || ny > 3u) return kFilmPackException;
    return (uint32_t)ox | ((uint32_t)oy << 1) | ((nx - 1u) << 2) | ((ny - 1u) << 4) | ((d.z & 0xfffu) << 6) | ((d.w & 0xfffu) << 18);
}
TH_D uint32_t film_pack_word(const FilmSideTable& side, uint4 d, int px, int py) {
    const uint32_t w = film_pack_desc(d, px, py);
    if (w != kFilmPackException) return w;
    const uint32_t j = atomicAdd(side.count, 1u);
    if (j >= side.cap || j >= 0x7fffffffu) return kFilmPackOverflow;
    side.desc[j] = d;
    return kFilmPackSide | j;
}

// ---- camera (camera/perspective.jl:85-114) ----------------------------------------------------------------------------------
TH_D void generate_ray(const DeviceSensor& se, f2 film, f2 lens, float time_u, f3& o, f3& d, float& time) {
    const f3 p_camera = xf_point(se.raster_to_camera, mk3(film.x, film.y, 0.0f));
    o = splat3(0.0f);
    d = normalize(p_camera);
    if (se.lens_radius > 0.0f) {
        const f2 pl = concentric_sample_disk(lens);
        const f2 p_lens{se.lens_radius * pl.x, se.lens_radius * pl.y};
        const float t = se.focal_distance / d.z;
        const f3 p_focus = o + d * t;
        o = mk3(p_lens.x, p_lens.y, 0.0f);
        d = normalize(p_focus - o);
    }
    time = (1 - time_u) * se.shutter_open + time_u * se.shutter_close;  // lerp bounds.jl:126
    o = xf_point(se.camera_to_world, o);
    d = xf_vec(se.camera_to_world, d);
    d = normalize(d);
}

// Lf (optional, the path integrator's whole-frame launches): the sample's radiance record is initialised HERE, in the film pass's pixel-group-major layout
// (film_index layout 1) with the packed splat descriptor in its .w lane, and the path carries that record's index as its slot — the frame then needs neither the
// memset of L nor the pack / re-lay pass over it before the gather (2.9 ms of a 256-spp frame and a second copy of L)
template <int TH_ONE_COPY = 0> __global__ __launch_bounds__(kBlock) void k_raygen(const DeviceSensor* __restrict__ sep, uint32_t slot0, uint32_t n, uint64_t seed, uint32_t sample_offset,
                                                   PathQueue q, uint32_t cap, Counters* ctr, float4* __restrict__ Lf = nullptr, uint32_t spp_frame = 0, FilmSideTable side = FilmSideTable{nullptr, nullptr, 0}) {
    const DeviceSensor& se = *sep;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        uint32_t slot = slot0 + i;
        const SlotInfo si = slot_info(se, slot);
        const uint64_t key = ts_stream_key(seed, si.px, si.py, sample_offset + si.sample);
        if (Lf) {
            slot = (uint32_t)film_index(1u, (uint32_t)(se.sb_w * se.band_rows), spp_frame, si.sample, si.pix);
            Lf[slot] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(film_pack_word(side, film_splat_desc(se, si.px, si.py, key), si.px, si.py)));
        }
        const f2 film{(float)si.px + ts_uniform(key, TS_DIM_FILM_X), (float)si.py + ts_uniform(key, TS_DIM_FILM_Y)};
        const f2 lens{ts_uniform(key, TS_DIM_LENS_X), ts_uniform(key, TS_DIM_LENS_Y)};
        f3 o, d;
        float time;
        generate_ray(se, film, lens, ts_uniform(key, TS_DIM_TIME), o, d, time);
        d = check_direction(d);  // intersect!(bvh, ray) starts with check_direction! (bvh.jl:217)
        // wave w of the dense index space goes to segment w % kSeg
        const uint32_t w = i >> 6;
        const uint32_t phys = (w % kSeg) * cap + (w / kSeg) * 64u + (i & 63u);
        q.o[phys] = make_float4(o.x, o.y, o.z, __uint_as_float(slot));
        q.d[phys] = make_float4(d.x, d.y, d.z, __uint_as_float((uint32_t)key));  // the stream key travels with the path (k_shade_path)
        q.beta[phys] = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float((uint32_t)(key >> 32)));
    }
    if (blockIdx.x == 0 && threadIdx.x < kSeg) {
        const uint32_t sgm = threadIdx.x, W = (n + 63u) >> 6;
        uint32_t cnt = ((W + kSeg - 1 - sgm) / kSeg) * 64u;
        if (W > 0 && (W - 1) % kSeg == sgm && (n & 63u)) cnt -= 64u - (n & 63u);
        ctr->n_queue[0][sgm * kCtrStride] = cnt;
    }
}

// L[slot] += NaN on the channels a shading vertex noted in `poison` (ShadeStream::poison): NaN + x = NaN, so when the note is applied is immaterial
template <int TH_ONE_COPY = 0> __global__ __launch_bounds__(kBlock) void k_apply_poison(float4* __restrict__ L, const uint8_t* __restrict__ poison, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) {
        const uint32_t p = poison[i];
        if (p) {
            float4 l = L[i];
            const float nanv = __builtin_nanf("");
            if (p & 1u) l.x += nanv;
            if (p & 2u) l.y += nanv;
            if (p & 4u) l.z += nanv;
            L[i] = l;
        }
    }
}

// ---- traversal ------------------------------------------------------------------------------------------------------------------
struct Hit {
    float t;
    int prim;
    float b1, b2;
};
// bounds.jl:186-206 (`ty_max > tx_max && (tx_max = ty_max)` as written)
TH_D bool slab_test(float4 n0, float4 n1, f3 o, f3 inv_d, bool negx, bool negy, bool negz, float t_max) {
    float tx_min = ((negx ? n1.x : n0.x) - o.x) * inv_d.x;
    float tx_max = ((negx ? n0.x : n1.x) - o.x) * inv_d.x;
    const float ty_min = ((negy ? n1.y : n0.y) - o.y) * inv_d.y;
    const float ty_max = ((negy ? n0.y : n1.y) - o.y) * inv_d.y;
    if (tx_min > ty_max || ty_min > tx_max) return false;
    if (ty_min > tx_min) tx_min = ty_min;
    if (ty_max > tx_max) tx_max = ty_max;
    const float tz_min = ((negz ? n1.z : n0.z) - o.z) * inv_d.z;
    const float tz_max = ((negz ? n0.z : n1.z) - o.z) * inv_d.z;
    if (tx_min > tz_max || tz_min > tx_max) return false;
    if (tz_min > tx_min) tx_min = tz_min;
    if (tz_max < tx_max) tx_max = tz_max;
    return tx_min < t_max && tx_max > 0.0f;
}

// One ray through the BVH in the reference's order (near child first by dir_is_neg[split_axis], bvh.jl:239-246).
// ANY: intersect_p (returns at the first accepted primitive); otherwise closest hit with "later equal-t hit wins".
template <bool ANY, bool COUNT>
TH_D bool traverse(const DeviceScene& sc, f3 o, f3 d, float t_max, uint32_t (*stk)[kBlock], Hit* hit, uint32_t& n_nodes, uint32_t& n_prims) {
    const f3 inv_d = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    const bool negx = d.x < 0.0f, negy = d.y < 0.0f, negz = d.z < 0.0f;
    uint32_t spill[kStackSpill];
    int sp = 0;
    uint32_t cur = 0;
    bool found = false;
    if (sc.n_nodes == 0) return false;
    const uint32_t tid = threadIdx.x;
    while (true) {
        const float4 n0 = sc.nodes[2 * cur], n1 = sc.nodes[2 * cur + 1];
        if (COUNT) n_nodes++;
        bool pop = true;
        if (slab_test(n0, n1, o, inv_d, negx, negy, negz, t_max)) {
            const uint32_t a = __float_as_uint(n0.w), flags = __float_as_uint(n1.w);
            if ((flags & 3u) == 3u) {
                const uint32_t cnt = flags >> 2;
                for (uint32_t k = 0; k < cnt; ++k) {
                    const uint32_t slot = a + k;
                    const float4 p0 = sc.prims[3 * slot];
                    const uint32_t meta = __float_as_uint(p0.w);
                    if (COUNT) n_prims++;
                    if (meta & PRIM_SPHERE) {
                        SphereHit sh;
                        if (sphere_intersect<false>(sc.spheres[__float_as_uint(p0.x)], o, d, t_max, sh)) {
                            if (ANY) return true;
                            t_max = sh.t;  // primitive.jl:17 (unconditional)
                            found = true;
                            hit->prim = (int)slot;
                            hit->b1 = hit->b2 = 0.0f;
                        }
                    } else {
                        const float4 p1 = sc.prims[3 * slot + 1], p2 = sc.prims[3 * slot + 2];
                        TriTest tt;
                        if (tri_intersect<!ANY>(mk3(p0.x, p0.y, p0.z), mk3(p1.x, p1.y, p1.z), mk3(p2.x, p2.y, p2.z), o, d, t_max, &tt)) {
                            if (ANY) return true;
                            t_max = tt.t;
                            found = true;
                            hit->prim = (int)slot;
                            hit->b1 = tt.bary.x;
                            hit->b2 = tt.bary.y;
                        }
                    }
                }
            } else {
                const uint32_t axis = flags & 3u;
                const bool neg = axis == 0 ? negx : (axis == 1 ? negy : negz);
                const uint32_t far_child = neg ? cur + 1 : a;
                cur = neg ? a : cur + 1;
                if (sp < kStackLds)
                    stk[sp][tid] = far_child;
                else if (sp < kStackLds + kStackSpill)
                    spill[sp - kStackLds] = far_child;
                // deeper than 64: the reference would throw a BoundsError (bvh.jl:222); the entry is dropped
                sp++;
                pop = false;
            }
        }
        if (pop) {
            if (sp == 0) break;
            sp--;
            cur = sp < kStackLds ? stk[sp][tid] : (sp < kStackLds + kStackSpill ? spill[sp - kStackLds] : 0u);
        }
    }
    if (!ANY) hit->t = t_max;
    return found;
}

// Closest hit over a queue.  hits[phys] = {t or +Inf, slot or -1, b1, b2}.
template <bool COUNT>
__global__ __launch_bounds__(kBlock) void k_trace_closest(DeviceScene sc, SegQueue q, const float4* __restrict__ ro, const float4* __restrict__ rd, const float* __restrict__ tmax_or_null,
                                                          float4* __restrict__ hits, Counters* ctr) {
    __shared__ uint32_t stk[kStackLds][kBlock];
    __shared__ SegView sv;
    seg_load(q, sv);
    const uint32_t total = sv.prefix[kSeg];
    uint32_t nn = 0, np = 0;
    uint32_t seg = 0;  // (carried over the iterations: seg_locate_from)
    for (uint32_t flat = blockIdx.x * kBlock + threadIdx.x; flat < total; flat += gridDim.x * kBlock) {
        uint32_t lb;
        seg_locate_from(sv, flat & ~63u, seg, lb);
        const uint32_t local = lb + (flat & 63u);
        if (local >= sv.count[seg]) continue;
        const uint32_t i = seg_phys(q, seg, local);
        const float4 o4 = ro[i], d4 = rd[i];
        Hit h;
        h.prim = -1;
        h.b1 = h.b2 = 0.0f;
        const float t0 = tmax_or_null ? tmax_or_null[i] : kInf;
        const bool found = traverse<false, COUNT>(sc, mk3(o4.x, o4.y, o4.z), mk3(d4.x, d4.y, d4.z), t0, stk, &h, nn, np);
        hits[i] = make_float4(found ? h.t : kInf, __int_as_float(found ? h.prim : -1), h.b1, h.b2);
    }
    if (ctr) {
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&ctr->closest_total, (unsigned long long)seg_total(sv));
        if (COUNT) {
            const unsigned long long sn = wave_sum(nn), spr = wave_sum(np);
            if (lane_id() == 0) {
                atomicAdd(&ctr->nodes_closest, sn);
                atomicAdd(&ctr->prims_closest, spr);
            }
        }
    }
}

// Any-hit over the shadow queue; unoccluded rays add their contribution to the per-sample radiance buffer
// (estimate_direct, sppm.jl:536-541: `!unoccluded && (Li = 0)`).  If L is null, writes occluded[i] instead.
template <bool COUNT>
__global__ __launch_bounds__(kBlock) void k_trace_any(DeviceScene sc, SegQueue q, const float4* __restrict__ ro, const float4* __restrict__ rd, const float4* __restrict__ contrib,
                                                      const float* __restrict__ tmax_or_null, float4* __restrict__ L, uint8_t* __restrict__ occluded, Counters* ctr) {
    __shared__ uint32_t stk[kStackLds][kBlock];
    __shared__ SegView sv;
    seg_load(q, sv);
    const uint32_t total = sv.prefix[kSeg];
    uint32_t nn = 0, np = 0;
    uint32_t seg = 0;  // (carried over the iterations: seg_locate_from)
    for (uint32_t flat = blockIdx.x * kBlock + threadIdx.x; flat < total; flat += gridDim.x * kBlock) {
        uint32_t lb;
        seg_locate_from(sv, flat & ~63u, seg, lb);
        const uint32_t local = lb + (flat & 63u);
        if (local >= sv.count[seg]) continue;
        const uint32_t i = seg_phys(q, seg, local);
        const float4 o4 = ro[i], d4 = rd[i];
        const float t0 = tmax_or_null ? tmax_or_null[i] : kInf;
        const bool occ = traverse<true, COUNT>(sc, mk3(o4.x, o4.y, o4.z), mk3(d4.x, d4.y, d4.z), t0, stk, nullptr, nn, np);
        if (L) {
            const uint32_t slot = __float_as_uint(o4.w);
            if (!occ) {
                const float4 c = contrib[i];
                float4 l = L[slot];
                l.x += c.x;
                l.y += c.y;
                l.z += c.z;
                L[slot] = l;
            } else {
                const uint32_t poison = __float_as_uint(d4.w);  // β·0 is NaN where β is not finite
                if (poison) {
                    float4 l = L[slot];
                    const float nanv = __builtin_nanf("");
                    if (poison & 1u) l.x += nanv;
                    if (poison & 2u) l.y += nanv;
                    if (poison & 4u) l.z += nanv;
                    L[slot] = l;
                }
            }
        } else {
            occluded[i] = occ ? 1 : 0;
        }
    }
    if (ctr) {
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&ctr->shadow_total, (unsigned long long)seg_total(sv));
        if (COUNT) {
            const unsigned long long sn = wave_sum(nn), spr = wave_sum(np);
            if (lane_id() == 0) {
                atomicAdd(&ctr->nodes_shadow, sn);
                atomicAdd(&ctr->prims_shadow, spr);
            }
        }
    }
}

// ---- shading ---------------------------------------------------------------------------------------------------------------------
// Rebuild the SurfaceInteraction of a closest hit from the one hit primitive (re-running its intersection with
// t_max = Inf reproduces the accepted candidate's barycentrics / hit point bit-for-bit: they do not depend on t_max).
// With `bary` (hits written with TraceOut::bary_mode: {b2, prim, b0, b1}) the triangle test is not repeated: the stored
// barycentrics ARE the accepted candidate's.
// TAN = false: the scene has no mesh with vertex tangents (DeviceScene::tri_tan is null) — the hot shading kernels are instantiated both ways, because even the
// never-taken branch costs the general code 5 % (S-mesh shading 59.7 -> 62.7 ms, measured)
template <bool TRI_ONLY = false, bool TAN = true>
TH_D bool rebuild_shading(const DeviceScene& sc, int prim, f3 o, f3 d, Shading& sh, uint32_t& material, const float4* bary = nullptr, f3* fast_r = nullptr) {
    // all six 16-byte loads are issued up front (independent of the sphere / normals flags) so that their latencies overlap.  (Fetching them
    // in k_shade_path while it classifies the entry — two dependent trips instead of three — was measured: 24 more live registers, 368
    // instead of 208 bytes of scratch, S-mesh shading 92 -> 163 ms; fetching only the ray and the throughput there: 92 -> 107 ms.)
    const float4* rec = sc.shade + 8 * (size_t)prim;  // one 128-byte line: vertices and normals of the slot
    const float4 p0 = rec[0], p1 = rec[1], p2 = rec[2];
    const float4 na = rec[3], nb = rec[4], nc = rec[5];
    const float4 cn = rec[6], cs = rec[7];  // the triangle's geometric normal and unit ∂p∂u (k_shade_constants)
    const uint32_t meta = __float_as_uint(p0.w);
    material = meta & PRIM_MATERIAL_MASK;
    if (fast_r) *fast_r = mk3(na.w, nb.w, nc.w);  // PRIM_FAST: the single Lambert lobe's reflectance
    if (!TRI_ONLY && (meta & PRIM_SPHERE)) {
        const SphereRec& s = sc.spheres[__float_as_uint(p0.x)];
        SphereHit h;
        if (!sphere_intersect<true>(s, o, d, kInf, h)) return false;
        sh = shade_sphere(s, h, d);
        return true;
    }
    const f3 v0 = mk3(p0.x, p0.y, p0.z), v1 = mk3(p1.x, p1.y, p1.z), v2 = mk3(p2.x, p2.y, p2.z);
    TriTest tt;
    if (bary)
        tt.bary = mk3(bary->z, bary->w, bary->x);
    else if (!tri_intersect<true>(v0, v1, v2, o, d, kInf, &tt))
        return false;
    const bool has_n = (meta & PRIM_HAS_NORMALS) != 0;
    const TriConstants tc{mk3(cn.x, cn.y, cn.z), mk3(cs.x, cs.y, cs.z)};
    // a mesh with vertex tangents (no scene of the reference has one): the three records are fetched inside shade_triangle, where the normals are no longer live
    const float4* tan = (TAN && sc.tri_tan && (meta & PRIM_HAS_TANGENTS)) ? sc.tri_tan + 3 * (size_t)prim : nullptr;
    sh = shade_triangle(v0, v1, v2, has_n, mk3(na.x, na.y, na.z), mk3(nb.x, nb.y, nb.z), mk3(nc.x, nc.y, nc.z), (meta & PRIM_FLIP) != 0, tt.bary, d, &tc, tan);
    return true;
}
// Commit time: every slot's shading line (th_scene.h) from the two arrays the traversal kernels use — records 0-2 = prims, 3-5 = tri_nrm, 6 / 7 =
// triangle_constants of the vertices.
// uv: 2 float4 per slot {u0, v0, u1, v1}, {u2, v2, has_uv, 0} for scenes where some mesh carries (u, v)s, else null
template <int TH_ONE_COPY = 0> __global__ __launch_bounds__(kBlock) void k_shade_constants(float4* __restrict__ shade, const float4* __restrict__ prims, const float4* __restrict__ nrm, const float4* __restrict__ uv, uint32_t n_prims) {
    for (uint32_t k = blockIdx.x * kBlock + threadIdx.x; k < n_prims; k += gridDim.x * kBlock) {
        float4* rec = shade + 8 * (size_t)k;
        const float4 p0 = prims[3 * (size_t)k], p1 = prims[3 * (size_t)k + 1], p2 = prims[3 * (size_t)k + 2];
        rec[0] = p0;
        rec[1] = p1;
        rec[2] = p2;
        for (int j = 0; j < 3; ++j) rec[3 + j] = nrm[3 * (size_t)k + j];
        float4 c6 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), c7 = c6;
        if (!(__float_as_uint(p0.w) & PRIM_SPHERE)) {
            float uvs[6];
            bool has_uv = false;
            if (uv) {
                const float4 a = uv[2 * (size_t)k], b = uv[2 * (size_t)k + 1];
                uvs[0] = a.x, uvs[1] = a.y, uvs[2] = a.z, uvs[3] = a.w, uvs[4] = b.x, uvs[5] = b.y;
                has_uv = b.z != 0.0f;
            }
            const TriConstants tc = triangle_constants(mk3(p0.x, p0.y, p0.z), mk3(p1.x, p1.y, p1.z), mk3(p2.x, p2.y, p2.z), has_uv ? uvs : nullptr);
            c6 = make_float4(tc.n.x, tc.n.y, tc.n.z, 0.0f);
            c7 = make_float4(tc.ss.x, tc.ss.y, tc.ss.z, 0.0f);
        }
        rec[6] = c6;
        rec[7] = c7;
    }
}

// One PathIntegrator vertex (DESIGN.md "PathIntegrator"; sppm.jl:208-266 without the visible-point early-out, β on the
// direct term, RR as :257-263; uniform_sample_one_light / estimate_direct sppm.jl:503-554).
#ifndef TH_SHADE_WAVES
#define TH_SHADE_WAVES 4  // waves per SIMD the register allocator must leave room for: 128 VGPRs + 136 B scratch instead of 175 VGPRs at 2 waves; measured 37.0 -> 32.2 ms (S-cornell, 64 spp)
#endif
// STREAM (streaming wavefront, th_trace2.h): the queue of a ROUND holds paths of mixed depths — `row` is the round, every entry
// carries its depth tag (tags_in / tags_out), and what the path adds to its sample's radiance at depth d goes to the term slot
// L[(d - 1) * term_stride + slot] (k_fold_terms adds the slots in depth order afterwards).  Entries whose ray was suspended
// (hit primitive -2) are skipped like misses; they come back in a later round.  Classic: row = depth - 1, tags unused.
struct ShadeStream {
    const uint32_t* tags_in;
    uint32_t* tags_out;
    uint32_t term_stride;
    uint8_t* poison;  // classic wavefront: where a vertex notes "L += β · 0 with a non-finite β" (bit c = channel c) instead of touching L, which the
                      // shadow rays of the depth before may still be adding to on the other stream; k_apply_poison folds the notes in at the end
};
struct ShadeOut {  // what one vertex emits: a shadow ray and / or the continuation of the path
    bool want_shadow, want_next;
    float4 so4, sd4, sc4, no4, nd4, nb4;
    uint32_t next_depth;
};
// One PathIntegrator vertex for queue entry i (a real hit).  FAST: the caller has established that the hit is a triangle whose
// material is a single LambertianReflection lobe; the sphere path and the general BSDF code are then not even compiled in.
template <bool STREAM, bool FAST, bool TAN = true>
TH_D void shade_vertex(const DeviceScene& sc, const PathQueue& qin, const float4* __restrict__ hits, float4* __restrict__ L, uint32_t i, int depth_fixed, int max_depth,
                       uint32_t hits_have_bary, const ShadeStream& ss, ShadeOut& out) {
    const float4 h4 = hits[i];
    const int prim = __float_as_int(h4.y);
    const float4 o4 = qin.o[i], d4 = qin.d[i], b4 = qin.beta[i];
    const int depth = STREAM ? (int)ss.tags_in[i] : depth_fixed;
    // where this vertex's radiance terms go: the sample's slot, or (STREAM) its per-depth term slot
    const uint32_t slot = STREAM ? (uint32_t)(depth - 1) * ss.term_stride + __float_as_uint(o4.w) : __float_as_uint(o4.w);
    const f3 o = mk3(o4.x, o4.y, o4.z), d = mk3(d4.x, d4.y, d4.z);
    f3 beta = mk3(b4.x, b4.y, b4.z);
    Shading sh;
    uint32_t material;
    f3 fast_r;
    if (!(rebuild_shading<FAST, TAN>(sc, prim, o, d, sh, material, hits_have_bary ? &h4 : nullptr, &fast_r) && material != PRIM_NO_MATERIAL)) return;
    const LobeSet& bsdf = sc.materials[FAST ? 0u : material].set[1];  // compute_scattering!(si, ray, true); FAST never reads it
    const bool lambert = FAST || bsdf_is_single_lambert(bsdf);  // specialised evaluation of the same arithmetic (th_device.h)
    Lobe lam;  // the one lobe of a `lambert` vertex: FAST rebuilds it from the reflectance stored beside the normals
    if (FAST) {
        lam.kind = LOBE_LAMBERT_R;
        lam.type = BSDF_DIFFUSE | BSDF_REFLECTION;
        lam.r[0] = fast_r.x;
        lam.r[1] = fast_r.y;
        lam.r[2] = fast_r.z;
    } else if (lambert) {
        lam = bsdf.lobe[0];
    }
    // the sampler stream key of this camera sample rides in the queue (k_raygen): no slot -> pixel division, no re-hash
    const uint64_t key = ((uint64_t)__float_as_uint(b4.w) << 32) | (uint64_t)__float_as_uint(d4.w);
    const uint32_t v = (uint32_t)(depth - 1);
    const f3 wo = -d;  // sppm.jl:224
    const uint32_t poison = ((isnan_(beta.x) || isinf_(beta.x)) ? 1u : 0u) | ((isnan_(beta.y) || isinf_(beta.y)) ? 2u : 0u) | ((isnan_(beta.z) || isinf_(beta.z)) ? 4u : 0u);
    // ---- uniform_sample_one_light ----
    bool direct_added = false;
    if (sc.n_lights > 0) {
        const int nl = (int)sc.n_lights;
        int ln = (int)__builtin_ceilf(ts_uniform(key, ts_vertex_dim(v, TS_V_LIGHT_PICK)) * (float)nl);
        if (ln > nl) ln = nl;
        if (ln < 1) ln = 1;
        const float light_pdf = 1.0f / (float)nl;
        // one light (every scene of the reference): its record comes through scalar loads, off the vector-memory queue
        const LightRec light = nl == 1 ? uniform_load(sc.lights, 0u) : sc.lights[ln - 1];
        const LightSample ls = sample_li(light, sh.p);
        if (ls.pdf > 0.0f && !is_black(ls.radiance)) {
            const f3 f = (lambert ? lambert_bsdf_f(lam, sh, sh.wo, ls.wi) : bsdf_f(bsdf, sh, sh.wo, ls.wi, BSDF_ALL & ~BSDF_SPECULAR)) * fabs_(dot(ls.wi, sh.ns));
            if (!is_black(f)) {
                // x / 1 == x exactly: δ-lights have pdf 1, a single light has light_pdf 1 (6 correctly rounded divisions saved)
                const f3 fl = f * ls.radiance;
                const f3 Ld1 = splat3(0.0f) + (ls.pdf == 1.0f ? fl : fl / ls.pdf);
                const f3 Ld = light_pdf == 1.0f ? Ld1 : Ld1 / light_pdf;
                const f3 c = beta * Ld;
                const f3 lp = mk3(light.position[0], light.position[1], light.position[2]);
                const f3 dir = lp - sh.p;  // spawn_ray(p0, p1) Trace.jl:196-202
                const f3 org = sh.p + 1e-6f * dir;
                const f3 cd = check_direction(dir);
                out.so4 = make_float4(org.x, org.y, org.z, __uint_as_float(slot));
                out.sd4 = make_float4(cd.x, cd.y, cd.z, __uint_as_float(poison));
                out.sc4 = make_float4(c.x, c.y, c.z, 0.0f);
                out.want_shadow = true;
                direct_added = true;
            }
        }
    }
    if (!direct_added && poison && ss.poison) {
        ss.poison[slot] |= (uint8_t)poison;  // one vertex per sample and launch, launches in order: a plain read-modify-write of the sample's own byte
    } else if (!direct_added && poison) {  // L += β · 0 with a non-finite β
        float4 l = L[slot];
        const float nanv = __builtin_nanf("");
        if (poison & 1u) l.x += nanv;
        if (poison & 2u) l.y += nanv;
        if (poison & 4u) l.z += nanv;
        L[slot] = l;
    }
    // ---- continue the path ----
    if (depth < max_depth) {
        const f2 u{ts_uniform(key, ts_vertex_dim(v, TS_V_BSDF_U0)), ts_uniform(key, ts_vertex_dim(v, TS_V_BSDF_U1))};
        const BsdfSample bs = lambert ? lambert_bsdf_sample_f(lam, sh, wo, u) : bsdf_sample_f(bsdf, sh, wo, u, BSDF_ALL);
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
                const f3 org = sh.p + 1e-6f * bs.wi;  // spawn_ray(si, wi) Trace.jl:206-211
                const f3 nd = check_direction(bs.wi);
                out.no4 = make_float4(org.x, org.y, org.z, o4.w);
                out.nd4 = make_float4(nd.x, nd.y, nd.z, d4.w);
                out.nb4 = make_float4(beta.x, beta.y, beta.z, b4.w);
                out.next_depth = (uint32_t)(depth + 1);
                out.want_next = true;
            }
        }
    }
}

// The kernel proper.  Most vertices of most scenes are triangle hits on a matte surface; the rest (spheres, specular or
// multi-lobe materials) run several times as many instructions.  Mixed in one wave they serialise (S-cornell: 108 ms against
// 58 ms for the same box without its two spheres), so every wave shades its FAST entries at once and parks the indices of the
// others in an LDS ring; whenever 64 are parked they are shaded together by the general code, with all lanes busy.
// (Two launches instead — the FAST entries, then the others from an index list in HBM, each with a register allocation of its
// own — were measured: S-mesh frame 405 -> 415 ms, S-cornell 162 -> 173 ms; the list traffic and the second launch's tail cost
// more than the 144 bytes of scratch the FAST code gets rid of.)
// (Round 5 built the two-launch form again — the matte entries alone, restructured to 88 VGPRs without scratch at 5 waves per SIMD, then the rest from index lists: 2 ms per
// 64 spp SLOWER on S-mesh and S-cornell at 4, 5 and 6 waves alike; the matte path is bound neither by occupancy nor by scratch: profiles/r5/r5_shade_split_experiment.txt.)
template <bool STREAM, bool TAN = true>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(TH_SHADE_WAVES))) void k_shade_path(DeviceScene sc, const DeviceSensor* __restrict__ sep, PathQueue qin, PathQueue qout, ShadowQueue sq, uint32_t cap,
                                                       const float4* __restrict__ hits, float4* __restrict__ L, Counters* ctr, int row, int depth_fixed, int max_depth, uint32_t hits_have_bary,
                                                       ShadeStream ss) {
    __shared__ SegView sv;
    __shared__ uint32_t s_ring[kBlock / 64][128];
    const SegQueue qv{ctr->n_queue[row], cap, 0u};
    seg_load(qv, sv);
    const uint32_t total = sv.prefix[kSeg];  // multiple of kSegGran: whole waves stay in the loop, so ballots see every lane
    const uint32_t lane = lane_id(), wv = threadIdx.x >> 6;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    // every wave-iteration of a wave feeds the same output segment (the grid is a multiple of kSeg waves): at most cap entries each
    const uint32_t seg_out = ((blockIdx.x * kBlock + threadIdx.x) >> 6) % kSeg;
    uint32_t ring_head = 0, ring_cnt = 0;  // wave-uniform
    auto emit = [&](const ShadeOut& e) {   // called by all 64 lanes together
        uint32_t si, ni;
        wave_compact2(e.want_shadow, &ctr->n_shadow[row][seg_out * kCtrStride], e.want_next, &ctr->n_queue[row + 1][seg_out * kCtrStride], si, ni);
        si += seg_out * cap;
        ni += seg_out * cap;
        if (e.want_shadow) {
            sq.o[si] = e.so4;
            sq.d[si] = e.sd4;
            sq.c[si] = e.sc4;
        }
        if (e.want_next) {
            qout.o[ni] = e.no4;
            qout.d[ni] = e.nd4;
            qout.beta[ni] = e.nb4;
            if (STREAM) ss.tags_out[ni] = e.next_depth;
        }
    };
    uint32_t seg_in = 0;
    for (uint32_t flat = blockIdx.x * kBlock + threadIdx.x; flat < total; flat += gridDim.x * kBlock) {
        uint32_t lb;
        seg_locate_from(sv, flat & ~63u, seg_in, lb);
        const uint32_t local = lb + (flat & 63u);
        const uint32_t i = seg_in * cap + local;
        int cls = 0;  // 0: nothing to shade (padding, miss, suspended ray), 1: FAST, 2: general
        if (local < sv.count[seg_in]) {
            const int prim = __float_as_int(hits[i].y);
            if (prim >= 0) {
                cls = (__float_as_uint(sc.shade[8 * (size_t)prim].w) & PRIM_FAST) ? 1 : 2;  // flag set at upload: no material fetch to classify
            }
        }
        ShadeOut e;
        e.want_shadow = e.want_next = false;
        e.next_depth = 0;
        if (cls == 1) shade_vertex<STREAM, true, TAN>(sc, qin, hits, L, i, depth_fixed, max_depth, hits_have_bary, ss, e);
        emit(e);
        // park the others
        const unsigned long long m2 = __ballot(cls == 2);
        if (m2) {
            if (cls == 2) s_ring[wv][(ring_head + ring_cnt + (uint32_t)__popcll(m2 & lt_mask)) & 127u] = i;
            ring_cnt += (uint32_t)__popcll(m2);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (ring_cnt >= 64u) {
                const uint32_t j = s_ring[wv][(ring_head + lane) & 127u];
                ShadeOut g;
                g.want_shadow = g.want_next = false;
                g.next_depth = 0;
                shade_vertex<STREAM, false, TAN>(sc, qin, hits, L, j, depth_fixed, max_depth, hits_have_bary, ss, g);
                emit(g);
                ring_head = (ring_head + 64u) & 127u;
                ring_cnt -= 64u;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
        }
    }
    if (ring_cnt) {  // the last, partial batch
        ShadeOut g;
        g.want_shadow = g.want_next = false;
        g.next_depth = 0;
        if (lane < ring_cnt) shade_vertex<STREAM, false, TAN>(sc, qin, hits, L, s_ring[wv][(ring_head + lane) & 127u], depth_fixed, max_depth, hits_have_bary, ss, g);
        emit(g);
    }
}
// STREAM: per-sample radiance = its per-depth terms added in depth order, which is the order the classic wavefront (and the
// reference's loop) adds them in.  A depth that contributed nothing holds +0, and x + 0 == x for every x this sum can take.
template <int TH_ONE_COPY = 0> __global__ __launch_bounds__(kBlock) void k_fold_terms(const float4* __restrict__ terms, uint64_t n_slots, uint32_t n_depths, float4* __restrict__ L) {
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n_slots; i += (uint64_t)gridDim.x * kBlock) {
        float4 l = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        for (uint32_t dd = 0; dd < n_depths; ++dd) {
            const float4 t = terms[(uint64_t)dd * n_slots + i];
            l.x += t.x;
            l.y += t.y;
            l.z += t.z;
        }
        L[i] = l;
    }
}
// depth tag 1 for every camera ray of a streaming batch
template <int TH_ONE_COPY = 0> __global__ __launch_bounds__(kBlock) void k_fill_u32(uint32_t* __restrict__ p, uint64_t n, uint32_t v) {
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) p[i] = v;
}

// ---- film -------------------------------------------------------------------------------------------------------------------------
// add_sample! (film.jl:134-164) + merge_film_tile! (film.jl:182-193) for every sample of the frame, evaluated as a gather:
// film pixel (X, Y) visits the few sample-pixels whose filter footprint can reach it, tile by tile in k order
// (integrators/sampler.jl:24-31), inside a tile in Bounds2 iteration order (x fastest, bounds.jl:39-47), samples in order —
// i.e. exactly the reference's (single-threaded) summation order, with no atomics and no race (the reference's
// merge_film_tile! is unsynchronised).  out = xyz sums + filter_weight_sum.
// camera_sample.film of every slot (p_raster + get_2d, sampler/sampler.jl:135-139): written once so that the gather does not
// re-derive it from the sampler for each of the ~16 film pixels a sample reaches.
// Where sample s of sample-pixel pix lives in the film pass's inputs.  The integrators write L sample-major ([s][pix]: what the
// queues want); the gather walks ALL samples of a pixel before it moves to the next pixel (the reference's order), i.e. with a
// stride of npix * 16 bytes (16 MB at 1024²) between consecutive loads — a new page for every one.  launch_film therefore
// re-lays L (k_film_transpose) and writes p_film pixel-group-major: [group of 64 sample-pixels][s][lane]; consecutive samples
// are then 1 KB apart and a wave's load is one contiguous chunk as before.  layout 0 = sample-major (option film_transpose 0).
// film_index: defined with the camera-sample helpers above (k_raygen writes in that layout too)
template <int TH_ONE_COPY = 0> __global__ __launch_bounds__(kBlock) void k_film_transpose(const float4* __restrict__ L, uint32_t npix, uint32_t spp, float4* __restrict__ Lt) {
    const uint32_t groups = (npix + 63u) >> 6;
    const uint64_t chunks = (uint64_t)groups * spp;
    const uint32_t lane = threadIdx.x & 63u;
    for (uint64_t c = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) >> 6; c < chunks; c += ((uint64_t)gridDim.x * kBlock) >> 6) {
        const uint32_t s = (uint32_t)(c / groups), g = (uint32_t)(c - (uint64_t)s * groups);  // consecutive waves read consecutive chunks
        const uint32_t pix = g * 64u + lane;
        if (pix < npix) Lt[film_index(1u, npix, spp, s, pix)] = L[(size_t)s * npix + pix];
    }
}
template <int TH_ONE_COPY = 0> __global__ __launch_bounds__(kBlock) void k_film_positions(const DeviceSensor* __restrict__ sep, uint64_t n, uint64_t seed, uint32_t sample_offset, float2* __restrict__ pfilm, uint32_t layout,
                                                           uint32_t spp) {
    const DeviceSensor& se = *sep;
    for (uint64_t slot = (uint64_t)blockIdx.x * kBlock + threadIdx.x; slot < n; slot += (uint64_t)gridDim.x * kBlock) {
        const SlotInfo si = slot_info(se, (uint32_t)slot);
        const uint64_t key = ts_stream_key(seed, si.px, si.py, sample_offset + si.sample);
        pfilm[film_index(layout, (uint32_t)(se.sb_w * se.band_rows), spp, si.sample, si.pix)] = make_float2((float)si.px + ts_uniform(key, TS_DIM_FILM_X), (float)si.py + ts_uniform(key, TS_DIM_FILM_Y));
    }
}
template <int TH_ONE_COPY = 0> __global__ __launch_bounds__(kBlock) void k_film_gather(const DeviceSensor* __restrict__ sep, const float* __restrict__ table, const float4* __restrict__ L,
                                                        const float2* __restrict__ pfilm, uint32_t spp, uint32_t layout, float4* __restrict__ out) {
    const DeviceSensor& se = *sep;
    const uint32_t npx = (uint32_t)(se.film_w * se.film_h);
    const uint32_t npix = (uint32_t)(se.sb_w * se.band_rows);
    const float rx = se.filter_radius[0], ry = se.filter_radius[1];
    const float inv_rx = 1.0f / rx, inv_ry = 1.0f / ry;
    for (uint32_t idx = blockIdx.x * kBlock + threadIdx.x; idx < npx; idx += gridDim.x * kBlock) {
        const int fy = (int)(idx / (uint32_t)se.film_w), fx = (int)(idx - (uint32_t)fy * (uint32_t)se.film_w);
        const float X = se.crop_min[0] + (float)fx, Y = se.crop_min[1] + (float)fy;
        // sample-pixels that can reach (X, Y): sx in (X - 1.5 - r, X + r + 0.5]
        int sx_lo = (int)__builtin_floorf(X - 1.5f - rx), sx_hi = (int)__builtin_ceilf(X + rx + 0.5f);
        int sy_lo = (int)__builtin_floorf(Y - 1.5f - ry), sy_hi = (int)__builtin_ceilf(Y + ry + 0.5f);
        sx_lo = max(sx_lo, se.sb_min[0]);
        sy_lo = max(sy_lo, se.sb_min[1]);
        sx_hi = min(sx_hi, se.sb_max[0]);
        sy_hi = min(sy_hi, se.sb_max[1]);
        f3 xyz = splat3(0.0f);
        float wsum = 0.0f;
        if (se.accumulate) {  // a later band of the frame: its tiles are added, in k order, onto the tiles of the bands before
            const float4 prev = out[idx];
            xyz = mk3(prev.x, prev.y, prev.z);
            wsum = prev.w;
        }
        if (sx_lo <= sx_hi && sy_lo <= sy_hi) {
            const int ty_lo = max((sy_lo - se.sb_min[1]) >> 4, se.band_ty0), ty_hi = min((sy_hi - se.sb_min[1]) >> 4, se.band_ty1);
            const int tx_lo = (sx_lo - se.sb_min[0]) >> 4, tx_hi = (sx_hi - se.sb_min[0]) >> 4;
            for (int ty = ty_lo; ty <= ty_hi; ++ty)
                for (int tx = tx_lo; tx <= tx_hi; ++tx) {
                    // tile sample bounds (integrators/sampler.jl:29-31) and FilmTile bounds (film.jl:120-125)
                    const float tbx0 = (float)se.sb_min[0] + (float)tx * 16.0f, tby0 = (float)se.sb_min[1] + (float)ty * 16.0f;
                    const float tbx1 = jmin(tbx0 + 15.0f, (float)se.sb_max[0]), tby1 = jmin(tby0 + 15.0f, (float)se.sb_max[1]);
                    const float bx0 = jmax(__builtin_ceilf(tbx0 - 0.5f - rx), se.crop_min[0]), by0 = jmax(__builtin_ceilf(tby0 - 0.5f - ry), se.crop_min[1]);
                    const float bx1 = jmin(__builtin_floorf(tbx1 - 0.5f + rx) + 1.0f, se.crop_max[0]), by1 = jmin(__builtin_floorf(tby1 - 0.5f + ry) + 1.0f, se.crop_max[1]);
                    if (X < bx0 || X > bx1 || Y < by0 || Y > by1) continue;  // pixel not in this FilmTile: merge does not touch it
                    f3 csum = splat3(0.0f);
                    float fws = 0.0f;
                    const int y0 = max(sy_lo, (int)tby0), y1 = min(sy_hi, (int)tby1);
                    const int x0 = max(sx_lo, (int)tbx0), x1 = min(sx_hi, (int)tbx1);
                    for (int sy = y0; sy <= y1; ++sy)
                        for (int sx = x0; sx <= x1; ++sx) {
                            const uint32_t pix = (uint32_t)(sy - se.band_y0) * (uint32_t)se.sb_w + (uint32_t)(sx - se.sb_min[0]);
                            for (uint32_t s = 0; s < spp; ++s) {
                                const float2 pf = pfilm[film_index(layout, npix, spp, s, pix)];  // camera_sample.film, from k_film_positions
                                const float dpx = pf.x - 0.5f, dpy = pf.y - 0.5f;
                                float p0x = __builtin_ceilf(dpx - rx), p0y = __builtin_ceilf(dpy - ry);
                                float p1x = __builtin_floorf(dpx + rx) + 1.0f, p1y = __builtin_floorf(dpy + ry) + 1.0f;
                                p0x = jmax(p0x, jmax(bx0, 1.0f));
                                p0y = jmax(p0y, jmax(by0, 1.0f));
                                p1x = jmin(p1x, bx1);
                                p1y = jmin(p1y, by1);
                                if (X < p0x || X > p1x || Y < p0y || Y > p1y) continue;
                                const float ffx = fabs_((X - dpx) * inv_rx * 16.0f), ffy = fabs_((Y - dpy) * inv_ry * 16.0f);
                                const int ox = (int)jclamp(__builtin_ceilf(ffx), 1.0f, 16.0f);   // ceil for x …
                                const int oy = (int)jclamp(__builtin_floorf(ffy), 1.0f, 16.0f);  // … floor for y (A.9)
                                const float w = table[(oy - 1) * 16 + (ox - 1)];
                                const float4 l4 = L[film_index(layout, npix, spp, s, pix)];
                                f3 l = mk3(l4.x, l4.y, l4.z);
                                if (has_nan(l)) l = splat3(0.0f);  // integrators/sampler.jl:46
                                csum = csum + l * 1.0f * w;
                                fws += w;
                            }
                        }
                    xyz = xyz + rgb_to_xyz(csum);
                    wsum += fws;
                }
        }
        out[idx] = make_float4(xyz.x, xyz.y, xyz.z, wsum);
    }
}

// FilmTile bounds of sample tile (ty, tx) (integrators/sampler.jl:29-31 + film.jl:120-125).
TH_D void film_tile_bounds(const DeviceSensor& se, int ty, int tx, float rx, float ry, float& bx0, float& by0, float& bx1, float& by1) {
    const float tbx0 = (float)se.sb_min[0] + (float)tx * 16.0f, tby0 = (float)se.sb_min[1] + (float)ty * 16.0f;
    const float tbx1 = jmin(tbx0 + 15.0f, (float)se.sb_max[0]), tby1 = jmin(tby0 + 15.0f, (float)se.sb_max[1]);
    bx0 = jmax(__builtin_ceilf(tbx0 - 0.5f - rx), se.crop_min[0]);
    by0 = jmax(__builtin_ceilf(tby0 - 0.5f - ry), se.crop_min[1]);
    bx1 = jmin(__builtin_floorf(tbx1 - 0.5f + rx) + 1.0f, se.crop_max[0]);
    by1 = jmin(__builtin_floorf(tby1 - 0.5f + ry) + 1.0f, se.crop_max[1]);
}

// k_film_gather for a BX x BY block of film pixels per thread.  k_film_gather is bound by re-reading every sample for each of
// the ~(2r+3)^2 film pixels in whose reach it lies; here a sample is loaded once per block and offered to all BX*BY pixels
// (reach of the block: (BX + 2r + 2) x (BY + 2r + 2) sample pixels), and the filter index is computed once per row / column.
// Per film pixel nothing changes: tiles in k order, sample pixels in Bounds2 order, samples in order, one csum per tile; a
// tile or sample outside a pixel's own reach fails the same bounds tests as in k_film_gather and contributes nothing.
template <int BX, int BY>
__global__ __launch_bounds__(kBlock) void k_film_gather_block(const DeviceSensor* __restrict__ sep, const float* __restrict__ table, const float4* __restrict__ L,
                                                              const float2* __restrict__ pfilm, uint32_t spp, uint32_t layout, float4* __restrict__ out) {
    const DeviceSensor& se = *sep;
    // the 16 x 16 filter table in LDS: a lookup in global memory queues behind the sample loads in flight (vector loads return in
    // order), so every weight waited for the whole next batch
    __shared__ float s_table[256];
    for (uint32_t t = threadIdx.x; t < 256u; t += kBlock) s_table[t] = table[t];
    __syncthreads();
    const uint32_t npix = (uint32_t)(se.sb_w * se.band_rows);
    const float rx = se.filter_radius[0], ry = se.filter_radius[1];
    const float inv_rx = 1.0f / rx, inv_ry = 1.0f / ry;
    const uint32_t nbx = ((uint32_t)se.film_w + BX - 1) / BX, nby = ((uint32_t)se.film_h + BY - 1) / BY;
    for (uint32_t bidx = blockIdx.x * kBlock + threadIdx.x; bidx < nbx * nby; bidx += gridDim.x * kBlock) {
        const int fy0 = (int)(bidx / nbx) * BY, fx0 = (int)(bidx - (bidx / nbx) * nbx) * BX;
        float X[BX], Y[BY];
        for (int i = 0; i < BX; ++i) X[i] = se.crop_min[0] + (float)(fx0 + i);
        for (int j = 0; j < BY; ++j) Y[j] = se.crop_min[1] + (float)(fy0 + j);
        // union of the pixels' reaches (k_film_gather: sx in (X - 1.5 - r, X + r + 0.5])
        int sx_lo = max((int)__builtin_floorf(X[0] - 1.5f - rx), se.sb_min[0]), sx_hi = min((int)__builtin_ceilf(X[BX - 1] + rx + 0.5f), se.sb_max[0]);
        int sy_lo = max((int)__builtin_floorf(Y[0] - 1.5f - ry), se.sb_min[1]), sy_hi = min((int)__builtin_ceilf(Y[BY - 1] + ry + 0.5f), se.sb_max[1]);
        f3 xyz[BY][BX];
        float wsum[BY][BX];
        for (int j = 0; j < BY; ++j)
            for (int i = 0; i < BX; ++i) {
                xyz[j][i] = splat3(0.0f);
                wsum[j][i] = 0.0f;
                if (se.accumulate && fx0 + i < se.film_w && fy0 + j < se.film_h) {  // a later band: add onto the tiles of the bands before (k order)
                    const float4 prev = out[(size_t)(fy0 + j) * (size_t)se.film_w + (size_t)(fx0 + i)];
                    xyz[j][i] = mk3(prev.x, prev.y, prev.z);
                    wsum[j][i] = prev.w;
                }
            }
        if (sx_lo <= sx_hi && sy_lo <= sy_hi) {
            const int ty_lo = max((sy_lo - se.sb_min[1]) >> 4, se.band_ty0), ty_hi = min((sy_hi - se.sb_min[1]) >> 4, se.band_ty1);
            const int tx_lo = (sx_lo - se.sb_min[0]) >> 4, tx_hi = (sx_hi - se.sb_min[0]) >> 4;
            for (int ty = ty_lo; ty <= ty_hi; ++ty)
                for (int tx = tx_lo; tx <= tx_hi; ++tx) {
                    float bx0, by0, bx1, by1;
                    film_tile_bounds(se, ty, tx, rx, ry, bx0, by0, bx1, by1);
                    const float tbx0 = (float)se.sb_min[0] + (float)tx * 16.0f, tby0 = (float)se.sb_min[1] + (float)ty * 16.0f;
                    const float tbx1 = jmin(tbx0 + 15.0f, (float)se.sb_max[0]), tby1 = jmin(tby0 + 15.0f, (float)se.sb_max[1]);
                    bool in_tile[BY][BX];  // merge_film_tile! touches the pixel (film.jl:182-193)
                    bool any_in = false;
                    for (int j = 0; j < BY; ++j)
                        for (int i = 0; i < BX; ++i) {
                            in_tile[j][i] = !(X[i] < bx0 || X[i] > bx1 || Y[j] < by0 || Y[j] > by1);
                            any_in = any_in || in_tile[j][i];
                        }
                    if (!any_in) continue;
                    f3 csum[BY][BX];
                    float fws[BY][BX];
                    for (int j = 0; j < BY; ++j)
                        for (int i = 0; i < BX; ++i) {
                            csum[j][i] = splat3(0.0f);
                            fws[j][i] = 0.0f;
                        }
                    const int y0 = max(sy_lo, (int)tby0), y1 = min(sy_hi, (int)tby1);
                    const int x0 = max(sx_lo, (int)tbx0), x1 = min(sx_hi, (int)tbx1);
                    for (int sy = y0; sy <= y1; ++sy)
                        for (int sx = x0; sx <= x1; ++sx) {
                            const uint32_t pix = (uint32_t)(sy - se.band_y0) * (uint32_t)se.sb_w + (uint32_t)(sx - se.sb_min[0]);
                            // kFilmUnroll samples per trip: their p_film and L loads are issued together (the gather is a chain of
                            // dependent trips otherwise), then they are splatted one after the other, in sample order
                            auto splat = [&](float2 pf, float4 l4) {
                                const float dpx = pf.x - 0.5f, dpy = pf.y - 0.5f;
                                float p0x = __builtin_ceilf(dpx - rx), p0y = __builtin_ceilf(dpy - ry);
                                float p1x = __builtin_floorf(dpx + rx) + 1.0f, p1y = __builtin_floorf(dpy + ry) + 1.0f;
                                p0x = jmax(p0x, jmax(bx0, 1.0f));
                                p0y = jmax(p0y, jmax(by0, 1.0f));
                                p1x = jmin(p1x, bx1);
                                p1y = jmin(p1y, by1);
                                bool okx[BX], oky[BY];
                                bool anyx = false, anyy = false;
                                for (int i = 0; i < BX; ++i) {
                                    okx[i] = !(X[i] < p0x || X[i] > p1x);
                                    anyx = anyx || okx[i];
                                }
                                for (int j = 0; j < BY; ++j) {
                                    oky[j] = !(Y[j] < p0y || Y[j] > p1y);
                                    anyy = anyy || oky[j];
                                }
                                if (!(anyx && anyy)) return;
                                f3 l = mk3(l4.x, l4.y, l4.z);
                                if (has_nan(l)) l = splat3(0.0f);  // integrators/sampler.jl:46
                                int ox[BX], oy[BY];
                                for (int i = 0; i < BX; ++i) ox[i] = (int)jclamp(__builtin_ceilf(fabs_((X[i] - dpx) * inv_rx * 16.0f)), 1.0f, 16.0f) - 1;   // ceil for x …
                                for (int j = 0; j < BY; ++j) oy[j] = ((int)jclamp(__builtin_floorf(fabs_((Y[j] - dpy) * inv_ry * 16.0f)), 1.0f, 16.0f) - 1) * 16;  // … floor for y (A.9)
                                for (int j = 0; j < BY; ++j)
                                    for (int i = 0; i < BX; ++i)
                                        if (okx[i] && oky[j]) {
                                            const float w = s_table[oy[j] + ox[i]];
                                            csum[j][i] = csum[j][i] + l * 1.0f * w;
                                            fws[j][i] += w;
                                        }
                            };
#ifndef TH_FILM_UNROLL
#define TH_FILM_UNROLL 4
#endif
                            constexpr uint32_t kFilmUnroll = TH_FILM_UNROLL;
                            uint32_t s = 0;
                            for (; s + kFilmUnroll <= spp; s += kFilmUnroll) {
                                float2 pfv[kFilmUnroll];
                                float4 lv[kFilmUnroll];
#pragma unroll
                                for (uint32_t u = 0; u < kFilmUnroll; ++u) {
                                    const size_t at = film_index(layout, npix, spp, s + u, pix);
                                    pfv[u] = pfilm[at];
                                    lv[u] = L[at];
                                }
#pragma unroll
                                for (uint32_t u = 0; u < kFilmUnroll; ++u) splat(pfv[u], lv[u]);
                            }
                            for (; s < spp; ++s) splat(pfilm[film_index(layout, npix, spp, s, pix)], L[film_index(layout, npix, spp, s, pix)]);
                        }
                    for (int j = 0; j < BY; ++j)
                        for (int i = 0; i < BX; ++i)
                            if (in_tile[j][i]) {
                                xyz[j][i] = xyz[j][i] + rgb_to_xyz(csum[j][i]);
                                wsum[j][i] += fws[j][i];
                            }
                }
        }
        for (int j = 0; j < BY; ++j)
            for (int i = 0; i < BX; ++i)
                if (fx0 + i < se.film_w && fy0 + j < se.film_h) out[(size_t)(fy0 + j) * (size_t)se.film_w + (size_t)(fx0 + i)] = make_float4(xyz[j][i].x, xyz[j][i].y, xyz[j][i].z, wsum[j][i]);
    }
}

// ---- film gather from per-sample splat descriptors (film_block = 3, the default) ---------------------------------------------------
// k_film_gather_block spends ~300 VALU instructions per (sample, thread): every thread that a sample can reach recomputes the sample's
// pixel range (ceil / floor, film.jl:145-146) and, per pixel, the filter-table index (|x - dp| / r * 16, ceil for x, floor for y, clamp,
// film.jl:149-153).  All of that depends on the sample alone, so k_film_descriptors computes it ONCE per sample — with the same Float32
// operations — into 16 bytes: the first pixel column / row of the unclamped range (int16 each), the number of columns / rows (<= 8: filter
// radius <= 3), and one 4-bit table index per column and per row.  The gather then reads {descriptor, radiance} and, per film pixel, does
// two compares, two bit-field extracts, one LDS lookup and the three multiply-adds of add_sample! (film.jl:161-162) — in the same order
// (tile k, sample pixel, sample): the film is bit-identical to k_film_gather's.
struct SplatDesc {   // uint4
    uint32_t origin;  // int16 p0x | int16 p0y << 16  = ceil(dp - r), unclamped (film.jl:145)
    uint32_t count;   // nx | ny << 8               = floor(dp + r) + 1 - p0 + 1 (film.jl:146: the extra column / row of A.9 included)
    uint32_t ox;      // 4 bits per column: clamp(ceil(|x - dpx| / rx * 16), 1, 16) - 1
    uint32_t oy;      // 4 bits per row:    clamp(floor(|y - dpy| / ry * 16), 1, 16) - 1
};
template <int TH_ONE_COPY = 0> __global__ __launch_bounds__(kBlock) void k_film_descriptors(const DeviceSensor* __restrict__ sep, uint64_t n, uint64_t seed, uint32_t sample_offset, uint4* __restrict__ desc) {
    const DeviceSensor& se = *sep;
    const float rx = se.filter_radius[0], ry = se.filter_radius[1];
    const float inv_rx = 1.0f / rx, inv_ry = 1.0f / ry;
    for (uint64_t slot = (uint64_t)blockIdx.x * kBlock + threadIdx.x; slot < n; slot += (uint64_t)gridDim.x * kBlock) {
        const SlotInfo si = slot_info(se, (uint32_t)slot);
        const uint64_t key = ts_stream_key(seed, si.px, si.py, sample_offset + si.sample);
        const float pfx = (float)si.px + ts_uniform(key, TS_DIM_FILM_X), pfy = (float)si.py + ts_uniform(key, TS_DIM_FILM_Y);  // camera_sample.film
        const float dpx = pfx - 0.5f, dpy = pfy - 0.5f;
        const float p0x = __builtin_ceilf(dpx - rx), p0y = __builtin_ceilf(dpy - ry);
        const float p1x = __builtin_floorf(dpx + rx) + 1.0f, p1y = __builtin_floorf(dpy + ry) + 1.0f;
        const int nx = (int)(p1x - p0x) + 1, ny = (int)(p1y - p0y) + 1;
        uint32_t oxw = 0, oyw = 0;
        for (int c = 0; c < 8; ++c) {
            const float X = p0x + (float)c, Y = p0y + (float)c;
            oxw |= (uint32_t)((int)jclamp(__builtin_ceilf(fabs_((X - dpx) * inv_rx * 16.0f)), 1.0f, 16.0f) - 1) << (4 * c);   // ceil for x …
            oyw |= (uint32_t)((int)jclamp(__builtin_floorf(fabs_((Y - dpy) * inv_ry * 16.0f)), 1.0f, 16.0f) - 1) << (4 * c);  // … floor for y (A.9)
        }
        desc[slot] = make_uint4(((uint32_t)(int)p0x & 0xffffu) | ((uint32_t)(int)p0y << 16), (uint32_t)nx | ((uint32_t)ny << 8), oxw, oyw);
    }
}
template <int BY>
__global__ __launch_bounds__(kBlock) void k_film_gather_desc(const DeviceSensor* __restrict__ sep, const float* __restrict__ table, const float4* __restrict__ L,
                                                             const uint4* __restrict__ desc, uint32_t spp, float4* __restrict__ out) {
    const DeviceSensor& se = *sep;
    __shared__ float s_table[256];
    for (uint32_t t = threadIdx.x; t < 256u; t += kBlock) s_table[t] = table[t];
    __syncthreads();
    const uint32_t npix = (uint32_t)(se.sb_w * se.band_rows);
    const float rx = se.filter_radius[0], ry = se.filter_radius[1];
    const uint32_t nbx = (uint32_t)se.film_w, nby = ((uint32_t)se.film_h + BY - 1) / BY;
    for (uint32_t bidx = blockIdx.x * kBlock + threadIdx.x; bidx < nbx * nby; bidx += gridDim.x * kBlock) {
        const int fy0 = (int)(bidx / nbx) * BY, fx0 = (int)(bidx - (bidx / nbx) * nbx);
        const float X = se.crop_min[0] + (float)fx0;
        const int Xi = (int)X;
        float Y[BY];
        int Yi[BY];
        for (int j = 0; j < BY; ++j) {
            Y[j] = se.crop_min[1] + (float)(fy0 + j);
            Yi[j] = (int)Y[j];
        }
        // union of the pixels' reaches (k_film_gather: sx in (X - 1.5 - r, X + r + 0.5])
        const int sx_lo = max((int)__builtin_floorf(X - 1.5f - rx), se.sb_min[0]), sx_hi = min((int)__builtin_ceilf(X + rx + 0.5f), se.sb_max[0]);
        const int sy_lo = max((int)__builtin_floorf(Y[0] - 1.5f - ry), se.sb_min[1]), sy_hi = min((int)__builtin_ceilf(Y[BY - 1] + ry + 0.5f), se.sb_max[1]);
        f3 xyz[BY];
        float wsum[BY];
        for (int j = 0; j < BY; ++j) {
            xyz[j] = splat3(0.0f);
            wsum[j] = 0.0f;
            if (se.accumulate && fy0 + j < se.film_h) {  // a later band: add onto the tiles of the bands before (k order)
                const float4 prev = out[(size_t)(fy0 + j) * (size_t)se.film_w + (size_t)fx0];
                xyz[j] = mk3(prev.x, prev.y, prev.z);
                wsum[j] = prev.w;
            }
        }
        if (sx_lo <= sx_hi && sy_lo <= sy_hi) {
            const int ty_lo = max((sy_lo - se.sb_min[1]) >> 4, se.band_ty0), ty_hi = min((sy_hi - se.sb_min[1]) >> 4, se.band_ty1);
            const int tx_lo = (sx_lo - se.sb_min[0]) >> 4, tx_hi = (sx_hi - se.sb_min[0]) >> 4;
            for (int ty = ty_lo; ty <= ty_hi; ++ty)
                for (int tx = tx_lo; tx <= tx_hi; ++tx) {
                    float bx0, by0, bx1, by1;
                    film_tile_bounds(se, ty, tx, rx, ry, bx0, by0, bx1, by1);
                    const float tbx0 = (float)se.sb_min[0] + (float)tx * 16.0f, tby0 = (float)se.sb_min[1] + (float)ty * 16.0f;
                    const float tbx1 = jmin(tbx0 + 15.0f, (float)se.sb_max[0]), tby1 = jmin(tby0 + 15.0f, (float)se.sb_max[1]);
                    // merge_film_tile! touches the pixel (film.jl:182-193); add_sample! clamps its range to the tile's film bounds and to 1 (film.jl:147-148)
                    const bool in_x = !(X < bx0 || X > bx1), ok_x = in_x && !(X < 1.0f);
                    bool in_tile[BY], ok_y[BY];
                    bool any_in = false;
                    for (int j = 0; j < BY; ++j) {
                        const bool in_y = !(Y[j] < by0 || Y[j] > by1);
                        in_tile[j] = in_x && in_y;
                        ok_y[j] = in_y && !(Y[j] < 1.0f);
                        any_in = any_in || in_tile[j];
                    }
                    if (!any_in) continue;
                    f3 csum[BY];
                    float fws[BY];
                    for (int j = 0; j < BY; ++j) {
                        csum[j] = splat3(0.0f);
                        fws[j] = 0.0f;
                    }
                    const int y0 = max(sy_lo, (int)tby0), y1 = min(sy_hi, (int)tby1);
                    const int x0 = max(sx_lo, (int)tbx0), x1 = min(sx_hi, (int)tbx1);
                    for (int sy = y0; sy <= y1; ++sy)
                        for (int sx = x0; sx <= x1; ++sx) {
                            const uint32_t pix = (uint32_t)(sy - se.band_y0) * (uint32_t)se.sb_w + (uint32_t)(sx - se.sb_min[0]);
                            auto splat = [&](uint4 d, float4 l4) {
                                const int cx = Xi - (int)(short)(d.x & 0xffffu);
                                if (!(ok_x && (uint32_t)cx < (d.y & 0xffu))) return;
                                f3 l = mk3(l4.x, l4.y, l4.z);
                                if (has_nan(l)) l = splat3(0.0f);  // integrators/sampler.jl:46
                                const uint32_t oxi = (d.z >> (4 * cx)) & 15u;
                                const int p0y = (int)(short)(d.x >> 16);
                                const uint32_t ny = (d.y >> 8) & 0xffu;
                                for (int j = 0; j < BY; ++j) {
                                    const int cy = Yi[j] - p0y;
                                    if (ok_y[j] && (uint32_t)cy < ny) {
                                        const float w = s_table[((d.w >> (4 * cy)) & 15u) * 16u + oxi];
                                        csum[j] = csum[j] + l * w;   // contrib_sum += l * sample_weight (1) * w
                                        fws[j] += w;
                                    }
                                }
                            };
                            constexpr uint32_t kU = 4;
                            uint32_t s = 0;
                            for (; s + kU <= spp; s += kU) {
                                uint4 dv[kU];
                                float4 lv[kU];
#pragma unroll
                                for (uint32_t u = 0; u < kU; ++u) {
                                    const size_t at = (size_t)(s + u) * npix + pix;
                                    dv[u] = desc[at];
                                    lv[u] = L[at];
                                }
#pragma unroll
                                for (uint32_t u = 0; u < kU; ++u) splat(dv[u], lv[u]);
                            }
                            for (; s < spp; ++s) splat(desc[(size_t)s * npix + pix], L[(size_t)s * npix + pix]);
                        }
                    for (int j = 0; j < BY; ++j)
                        if (in_tile[j]) {
                            xyz[j] = xyz[j] + rgb_to_xyz(csum[j]);
                            wsum[j] += fws[j];
                        }
                }
        }
        for (int j = 0; j < BY; ++j)
            if (fy0 + j < se.film_h) out[(size_t)(fy0 + j) * (size_t)se.film_w + (size_t)fx0] = make_float4(xyz[j].x, xyz[j].y, xyz[j].z, wsum[j]);
    }
}

// ---- film gather from a 32-bit splat descriptor carried in the radiance record itself (film_block >= 4, the default for filter radii <= 1) --------------
// Counters of round 2 (profiles/r2): k_film_gather_block<1,4> moves 139 GB per 1024^2 x 256 spp frame (2 x FETCH_SIZE, coalesced 16 B / lane loads) in 24 ms:
// 5.8 TB/s — it is bound by HBM, by RE-READS: every sample (16 B radiance + 8 B film position) is fetched by each of the ~10 threads whose pixels it can
// reach, at re-read distances no cache covers, for 6.5 GB of distinct data.  Two levers, both taken here:
//   * bytes per sample 24 -> 16: what a film pixel needs of camera_sample.film is the sample's pixel range and filter-table indices (film.jl:145-153) — for
//     a radius <= 1 that is <= 4 columns x 4 rows: 1 + 1 bits of range origin (p0 - sample pixel in {-1, 0}), 2 + 2 bits of range length, 3 x 4 + 3 x 4 bits
//     of table indices = 30 bits, which ride in the unused .w lane of the sample's float4 radiance record (k_film_pack_w; the integrators only ever
//     read-modify-write whole records, so the lane survives).  A range of 4 columns or rows (p_film exactly on a half: ~1e-4 of the samples at 1024^2)
//     does not fit: marked, and the gather recomputes that sample's descriptor from the sampler.  No film-position buffer, no pass that writes one.
//   * a thread owns BX x BY film pixels with BX > 1: a 4 x 4 block re-reads 49 sample pixels for 16 film pixels (3.1x) instead of 28 for 4 (7x); what made
//     large blocks slower before — ~300 VALU instructions per (sample, thread) recomputing ceil / floor ranges and table indices for every pixel — is gone:
//     decode, two unsigned compares and a bit-field extract per column / row, one LDS lookup and the multiply-adds of add_sample! per pixel.
// Per film pixel nothing changes: tiles in k order, sample pixels in Bounds2 order, samples in order, one csum per tile (film.jl:134-193); the arithmetic of
// the descriptor is k_film_descriptors' (bit-identical to the reference's per-pixel computation: tests/test_gpu_parity.py, tools/soak_film.py).
// A sample whose range is 4 wide (or otherwise outside the encoding) keeps its full 16-byte descriptor in a side table: .w = kFilmPackSide | index.  The table
// holds total / 16 + 65536 entries (the rate is ~2.4e-4 at 1024^2); should it ever run full the word is kFilmPackOverflow and the gather recomputes that
// sample's descriptor from the sampler — in a cold copy of the loop, so that the hash and the descriptor arithmetic are not inlined into the unrolled hot one.
// writes the descriptor into L[slot].w; the other lanes of the record are not touched (4-byte stores)
template <int TH_ONE_COPY = 0> __global__ __launch_bounds__(kBlock) void k_film_pack_w(const DeviceSensor* __restrict__ sep, uint64_t n, uint64_t seed, uint32_t sample_offset, float4* __restrict__ L, FilmSideTable side) {
    const DeviceSensor& se = *sep;
    for (uint64_t slot = (uint64_t)blockIdx.x * kBlock + threadIdx.x; slot < n; slot += (uint64_t)gridDim.x * kBlock) {
        const SlotInfo si = slot_info(se, (uint32_t)slot);
        const uint64_t key = ts_stream_key(seed, si.px, si.py, sample_offset + si.sample);
        reinterpret_cast<uint32_t*>(L + slot)[3] = film_pack_word(side, film_splat_desc(se, si.px, si.py, key), si.px, si.py);
    }
}
// The same, and the records re-laid pixel-group-major on the way ([group of 64 sample pixels][sample][lane], film_index layout 1): the gather walks ALL samples of a
// sample pixel before the next pixel, which in the integrators' sample-major order is a 16 MB stride at 1024^2 — a new DRAM page and TLB entry for every 16-byte
// load.  Re-laid, a wave's loads for consecutive samples are consecutive 1 KB chunks.  This pass reads and writes every record once (the in-place pass touches
// every line too: same traffic), into a second buffer.
template <int TH_ONE_COPY = 0> __global__ __launch_bounds__(kBlock) void k_film_pack_transpose(const DeviceSensor* __restrict__ sep, uint32_t npix, uint32_t spp, uint64_t seed, uint32_t sample_offset,
                                                                const float4* __restrict__ L, float4* __restrict__ Lt, FilmSideTable side) {
    const DeviceSensor& se = *sep;
    const uint32_t groups = (npix + 63u) >> 6;
    const uint64_t chunks = (uint64_t)groups * spp;
    const uint32_t lane = threadIdx.x & 63u;
    for (uint64_t c = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) >> 6; c < chunks; c += ((uint64_t)gridDim.x * kBlock) >> 6) {
        const uint32_t smp = (uint32_t)(c / groups), g = (uint32_t)(c - (uint64_t)smp * groups);  // consecutive waves read consecutive chunks
        const uint32_t pix = g * 64u + lane;
        if (pix >= npix) continue;
        const uint32_t slot = smp * npix + pix;
        const SlotInfo si = slot_info(se, slot);
        const uint64_t key = ts_stream_key(seed, si.px, si.py, sample_offset + si.sample);
        float4 l = L[slot];
        l.w = __uint_as_float(film_pack_word(side, film_splat_desc(se, si.px, si.py, key), si.px, si.py));
        Lt[film_index(1u, npix, spp, smp, pix)] = l;
    }
}
#ifndef TH_FILM_PACKED_UNROLL
#define TH_FILM_PACKED_UNROLL 8
#endif
#ifndef TH_FILM_PACKED_PIPELINE
#define TH_FILM_PACKED_PIPELINE 1
#endif
template <int BX, int BY>
__global__ __launch_bounds__(kBlock) void k_film_gather_packed(const DeviceSensor* __restrict__ sep, const float* __restrict__ table, const float4* __restrict__ L, uint32_t spp, uint64_t seed,
                                                               uint32_t sample_offset, uint32_t layout, const uint4* __restrict__ side, float4* __restrict__ out, uint32_t xcd_bands) {
    const DeviceSensor& se = *sep;
    __shared__ float s_table[256];
    for (uint32_t t = threadIdx.x; t < 256u; t += kBlock) s_table[t] = table[t];
    __syncthreads();
    const uint32_t npix = (uint32_t)(se.sb_w * se.band_rows);
    const size_t sstride = layout ? (size_t)64 : (size_t)npix;  // records between consecutive samples of one sample pixel (film_index)
    const float rx = se.filter_radius[0], ry = se.filter_radius[1];
    const uint32_t nbx = ((uint32_t)se.film_w + BX - 1) / BX, nby = ((uint32_t)se.film_h + BY - 1) / BY;
    // option "film_swizzle" (A/B, off): workgroup ids are dealt to the 8 XCDs round-robin; with it XCD x gets the x-th contiguous eighth of the grid, so that vertically
    // adjacent strips share an L2.  Measured (profiles/r4): no fewer bytes leave L2 — a strip re-reads its neighbour's 3 halo rows ~4 row passes (hundreds of MB of
    // streaming per XCD) after the neighbour did, far beyond what the 4 MB L2 holds.
    const uint32_t vblock = (xcd_bands && gridDim.x % 8u == 0u) ? (blockIdx.x % 8u) * (gridDim.x / 8u) + blockIdx.x / 8u : blockIdx.x;
    for (uint32_t bidx = vblock * kBlock + threadIdx.x; bidx < nbx * nby; bidx += gridDim.x * kBlock) {
        const int fy0 = (int)(bidx / nbx) * BY, fx0 = (int)(bidx - (bidx / nbx) * nbx) * BX;
        float X[BX], Y[BY];
        int Xi[BX], Yi[BY];
        for (int i = 0; i < BX; ++i) {
            X[i] = se.crop_min[0] + (float)(fx0 + i);
            Xi[i] = (int)X[i];
        }
        for (int j = 0; j < BY; ++j) {
            Y[j] = se.crop_min[1] + (float)(fy0 + j);
            Yi[j] = (int)Y[j];
        }
        // union of the pixels' reaches.  A sample of sample pixel sx has p_film in [sx, sx + 1] (closed: px + u may round up), so with a radius <= 1 (this kernel's
        // precondition) its range is p0 = ceil(p_film - 0.5 - r) >= sx - 1 and p1 = floor(p_film - 0.5 + r) + 1 <= sx + 2 (film.jl:145-148; both bounds are exact in
        // Float32) — which is also all the 32-bit descriptor can say (origin sx - 1 or sx, at most 4 wide).  Film pixel X is reached from sx in [X - 2, X + 1]: a
        // block of BX x BY pixels reads (BX + 3) x (BY + 3) sample pixels.  (Until round 4 the loop ran over floor(X - 1.5 - r) .. ceil(X + r + 0.5), one more on
        // each side: 63 sample pixels instead of 35 for 2 x 4 blocks, every one of them 256 records that could not contribute.)
        const int sx_lo = max(Xi[0] - 2, se.sb_min[0]), sx_hi = min(Xi[BX - 1] + 1, se.sb_max[0]);
        const int sy_lo = max(Yi[0] - 2, se.sb_min[1]), sy_hi = min(Yi[BY - 1] + 1, se.sb_max[1]);
        f3 xyz[BY][BX];
        float wsum[BY][BX];
        for (int j = 0; j < BY; ++j)
            for (int i = 0; i < BX; ++i) {
                xyz[j][i] = splat3(0.0f);
                wsum[j][i] = 0.0f;
                if (se.accumulate && fx0 + i < se.film_w && fy0 + j < se.film_h) {  // a later band: add onto the tiles of the bands before (k order)
                    const float4 prev = out[(size_t)(fy0 + j) * (size_t)se.film_w + (size_t)(fx0 + i)];
                    xyz[j][i] = mk3(prev.x, prev.y, prev.z);
                    wsum[j][i] = prev.w;
                }
            }
        if (sx_lo <= sx_hi && sy_lo <= sy_hi) {
            const int ty_lo = max((sy_lo - se.sb_min[1]) >> 4, se.band_ty0), ty_hi = min((sy_hi - se.sb_min[1]) >> 4, se.band_ty1);
            const int tx_lo = (sx_lo - se.sb_min[0]) >> 4, tx_hi = (sx_hi - se.sb_min[0]) >> 4;
            for (int ty = ty_lo; ty <= ty_hi; ++ty)
                for (int tx = tx_lo; tx <= tx_hi; ++tx) {
                    float bx0, by0, bx1, by1;
                    film_tile_bounds(se, ty, tx, rx, ry, bx0, by0, bx1, by1);
                    const float tbx0 = (float)se.sb_min[0] + (float)tx * 16.0f, tby0 = (float)se.sb_min[1] + (float)ty * 16.0f;
                    const float tbx1 = jmin(tbx0 + 15.0f, (float)se.sb_max[0]), tby1 = jmin(tby0 + 15.0f, (float)se.sb_max[1]);
                    // merge_film_tile! touches the pixel (film.jl:182-193); add_sample! clamps its range to the tile's film bounds and to 1 (film.jl:147-148)
                    bool in_x[BX], ok_x[BX], in_y[BY], ok_y[BY];
                    bool any_x = false, any_y = false;
                    for (int i = 0; i < BX; ++i) {
                        in_x[i] = !(X[i] < bx0 || X[i] > bx1);
                        ok_x[i] = in_x[i] && !(X[i] < 1.0f);
                        any_x = any_x || in_x[i];
                    }
                    for (int j = 0; j < BY; ++j) {
                        in_y[j] = !(Y[j] < by0 || Y[j] > by1);
                        ok_y[j] = in_y[j] && !(Y[j] < 1.0f);
                        any_y = any_y || in_y[j];
                    }
                    if (!(any_x && any_y)) continue;
                    f3 csum[BY][BX];
                    float fws[BY][BX];
                    for (int j = 0; j < BY; ++j)
                        for (int i = 0; i < BX; ++i) {
                            csum[j][i] = splat3(0.0f);
                            fws[j][i] = 0.0f;
                        }
                    const int y0 = max(sy_lo, (int)tby0), y1 = min(sy_hi, (int)tby1);
                    const int x0 = max(sx_lo, (int)tbx0), x1 = min(sx_hi, (int)tbx1);
                    for (int sy = y0; sy <= y1; ++sy)
                        for (int sx = x0; sx <= x1; ++sx) {
                            const uint32_t pix = (uint32_t)(sy - se.band_y0) * (uint32_t)se.sb_w + (uint32_t)(sx - se.sb_min[0]);
                            // offsets of the block's pixels from the sample pixel: the descriptor's range is relative to it
                            int rxi[BX], ryi[BY];
                            for (int i = 0; i < BX; ++i) rxi[i] = Xi[i] - sx;
                            for (int j = 0; j < BY; ++j) ryi[j] = Yi[j] - sy;
                            // One sample against the block's pixels.  (A branch-free form — every lookup issued for every pixel, products selected to 0 — was
                            // measured: 21.6 ms against 19.0 for 1 x 4, 29.2 against 17.1 for 2 x 4: six in ten (sample, pixel) pairs do not contribute, and the
                            // branches skip their work for whole waves often enough.)
                            auto splat = [&](float4 l4, uint32_t s, auto cold) {
                                uint32_t d = __float_as_uint(l4.w);
                                int p0x, p0y;  // relative to (sx, sy)
                                uint32_t nx, ny, oxw, oyw;
                                if (d & kFilmPackSide) {  // a 4-wide range: the full descriptor from the side table (or, `cold` only, from the sampler)
                                    uint4 fd = make_uint4(0u, 0u, 0u, 0u);
                                    if (decltype(cold)::value && d == kFilmPackOverflow)
                                        fd = film_splat_desc(se, sx, sy, ts_stream_key(seed, sx, sy, sample_offset + s));
                                    else
                                        fd = side[d & 0x7fffffffu];
                                    p0x = (int)(short)(fd.x & 0xffffu) - sx;
                                    p0y = (int)(short)(fd.x >> 16) - sy;
                                    nx = fd.y & 0xffu;
                                    ny = (fd.y >> 8) & 0xffu;
                                    oxw = fd.z;
                                    oyw = fd.w;
                                } else {
                                    p0x = -(int)(d & 1u);
                                    p0y = -(int)((d >> 1) & 1u);
                                    nx = ((d >> 2) & 3u) + 1u;
                                    ny = ((d >> 4) & 3u) + 1u;
                                    oxw = (d >> 6) & 0xfffu;
                                    oyw = (d >> 18) & 0xfffu;
                                }
                                uint32_t cx[BX], cy[BY];
                                bool vx[BX], vy[BY];
                                bool anyx = false, anyy = false;
                                for (int i = 0; i < BX; ++i) {
                                    cx[i] = (uint32_t)(rxi[i] - p0x);
                                    vx[i] = ok_x[i] && cx[i] < nx;
                                    anyx = anyx || vx[i];
                                }
                                for (int j = 0; j < BY; ++j) {
                                    cy[j] = (uint32_t)(ryi[j] - p0y);
                                    vy[j] = ok_y[j] && cy[j] < ny;
                                    anyy = anyy || vy[j];
                                }
                                if (!(anyx && anyy)) return;
                                f3 l = mk3(l4.x, l4.y, l4.z);
                                if (has_nan(l)) l = splat3(0.0f);  // integrators/sampler.jl:46
                                uint32_t oxi[BX];
                                for (int i = 0; i < BX; ++i) oxi[i] = (oxw >> (4u * (cx[i] & 7u))) & 15u;
                                for (int j = 0; j < BY; ++j)
                                    if (vy[j]) {
                                        const uint32_t row = ((oyw >> (4u * (cy[j] & 7u))) & 15u) * 16u;
                                        for (int i = 0; i < BX; ++i)
                                            if (vx[i]) {
                                                const float w = s_table[row + oxi[i]];
                                                csum[j][i] = csum[j][i] + l * w;  // contrib_sum += l * sample_weight (1) * w
                                                fws[j][i] += w;
                                            }
                                    }
                            };
                            constexpr uint32_t kU = TH_FILM_PACKED_UNROLL;
                            const float4* sp = L + film_index(layout, npix, spp, 0u, pix);  // sample 0 of this sample pixel; the next sample is `sstride` records on
                            uint32_t s = 0;
#if TH_FILM_PACKED_PIPELINE
                            // two batches of kU records in flight: the next batch's loads are issued before the current one is consumed (the frame has 2 waves per SIMD
                            // here — 131 072 threads at 2 x 4 pixels each — so nothing else hides a load's latency, and registers are not what limits the occupancy)
                            if (spp >= 2u * kU) {
                                float4 la[kU], lb[kU];
#pragma unroll
                                for (uint32_t u = 0; u < kU; ++u) la[u] = sp[(size_t)u * sstride];
                                bool stop = false;
                                for (; s + 2u * kU <= spp && !stop; s += 2u * kU, sp += (size_t)(2u * kU) * sstride) {
#pragma unroll
                                    for (uint32_t u = 0; u < kU; ++u) lb[u] = sp[(size_t)(kU + u) * sstride];
                                    bool overflow = false;
#pragma unroll
                                    for (uint32_t u = 0; u < kU; ++u) overflow = overflow || __float_as_uint(la[u].w) == kFilmPackOverflow;
                                    if (__builtin_expect(overflow, 0)) break;  // never, unless the side table ran full: the cold loop below takes over at s
#pragma unroll
                                    for (uint32_t u = 0; u < kU; ++u) splat(la[u], s + u, std::false_type{});
                                    const bool more = s + 3u * kU <= spp;
                                    if (more) {
#pragma unroll
                                        for (uint32_t u = 0; u < kU; ++u) la[u] = sp[(size_t)(2u * kU + u) * sstride];
                                    }
#pragma unroll
                                    for (uint32_t u = 0; u < kU; ++u) overflow = overflow || __float_as_uint(lb[u].w) == kFilmPackOverflow;
                                    if (__builtin_expect(overflow, 0)) {  // the first half is done: the cold loop takes over at s + kU
                                        s += kU;
                                        sp += (size_t)kU * sstride;
                                        break;
                                    }
#pragma unroll
                                    for (uint32_t u = 0; u < kU; ++u) splat(lb[u], s + kU + u, std::false_type{});
                                    if (!more) {  // what is left is less than a batch, or one batch that was not prefetched: the loops below
                                        s += 2u * kU;
                                        sp += (size_t)(2u * kU) * sstride;
                                        stop = true;
                                        break;
                                    }
                                }
                            }
#endif
                            for (; s + kU <= spp; s += kU, sp += (size_t)kU * sstride) {
                                float4 lv[kU];
                                bool overflow = false;
#pragma unroll
                                for (uint32_t u = 0; u < kU; ++u) lv[u] = sp[(size_t)u * sstride];
#pragma unroll
                                for (uint32_t u = 0; u < kU; ++u) overflow = overflow || __float_as_uint(lv[u].w) == kFilmPackOverflow;
                                if (__builtin_expect(overflow, 0)) {  // never, unless the side table ran full: the cold loop below handles such a trip
                                    break;
                                }
#pragma unroll
                                for (uint32_t u = 0; u < kU; ++u) {
                                    splat(lv[u], s + u, std::false_type{});
                                }
                            }
#pragma unroll 1
                            for (; s < spp; ++s, sp += sstride) splat(*sp, s, std::true_type{});  // the remainder — and everything after an overflow marker
                        }
                    for (int j = 0; j < BY; ++j)
                        for (int i = 0; i < BX; ++i)
                            if (in_x[i] && in_y[j]) {
                                xyz[j][i] = xyz[j][i] + rgb_to_xyz(csum[j][i]);
                                wsum[j][i] += fws[j][i];
                            }
                }
        }
        for (int j = 0; j < BY; ++j)
            for (int i = 0; i < BX; ++i)
                if (fx0 + i < se.film_w && fy0 + j < se.film_h) out[(size_t)(fy0 + j) * (size_t)se.film_w + (size_t)(fx0 + i)] = make_float4(xyz[j][i].x, xyz[j][i].y, xyz[j][i].z, wsum[j][i]);
    }
}

// k_film_gather with the samples staged through LDS: one block = a 16x16 film tile; for every sample row (ascending) the
// block stages `cols` sample columns x `ns` samples {p_film, L} once (instead of each of the ~16 film pixels a sample
// reaches re-reading them from HBM/MALL), then the film pixels in reach accumulate from LDS.  The summation order per film
// pixel is unchanged: one accumulator per sample tile (at most 2x2 reach a pixel), inside a tile rows ascending, columns
// ascending, samples ascending; tiles merged in k order (film.jl:182-193).  LDS layout: five planes [s][col], so the lanes
// of a wave (different columns, same s) read consecutive banks and equal columns broadcast.
template <int TH_ONE_COPY = 0> __global__ __launch_bounds__(kBlock) void k_film_gather_tiled(const DeviceSensor* __restrict__ sep, const float* __restrict__ table, const float4* __restrict__ L,
                                                              const float2* __restrict__ pfilm, uint32_t spp, uint32_t layout, uint32_t cols, uint32_t ns_stage, float4* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float s_planes[];
    __shared__ float s_table[256];
    const DeviceSensor& se = *sep;
    const uint32_t npix = (uint32_t)(se.sb_w * se.sb_h);
    const float rx = se.filter_radius[0], ry = se.filter_radius[1];
    const float inv_rx = 1.0f / rx, inv_ry = 1.0f / ry;
    const int tid = (int)threadIdx.x, ltx = tid & 15, lty = tid >> 4;
    s_table[tid] = table[tid];  // Film.filter_table (16 x 16), kBlock == 256
    const int fx = (int)blockIdx.x * 16 + ltx, fy = (int)blockIdx.y * 16 + lty;
    const bool in_film = fx < se.film_w && fy < se.film_h;
    const float X = se.crop_min[0] + (float)fx, Y = se.crop_min[1] + (float)fy;
    // this pixel's reach (same arithmetic as k_film_gather)
    int sx_lo = max((int)__builtin_floorf(X - 1.5f - rx), se.sb_min[0]), sx_hi = min((int)__builtin_ceilf(X + rx + 0.5f), se.sb_max[0]);
    int sy_lo = max((int)__builtin_floorf(Y - 1.5f - ry), se.sb_min[1]), sy_hi = min((int)__builtin_ceilf(Y + ry + 0.5f), se.sb_max[1]);
    const int ty_lo = (sy_lo - se.sb_min[1]) >> 4, tx_lo = (sx_lo - se.sb_min[0]) >> 4;
    // block-wide reach
    const float Xb0 = se.crop_min[0] + (float)(blockIdx.x * 16), Yb0 = se.crop_min[1] + (float)(blockIdx.y * 16);
    const float Xb1 = jmin(Xb0 + 15.0f, se.crop_max[0]), Yb1 = jmin(Yb0 + 15.0f, se.crop_max[1]);
    const int SX0 = max((int)__builtin_floorf(Xb0 - 1.5f - rx), se.sb_min[0]), SX1 = min((int)__builtin_ceilf(Xb1 + rx + 0.5f), se.sb_max[0]);
    const int SY0 = max((int)__builtin_floorf(Yb0 - 1.5f - ry), se.sb_min[1]), SY1 = min((int)__builtin_ceilf(Yb1 + ry + 0.5f), se.sb_max[1]);
    const int NC = SX1 - SX0 + 1;
    float* p_fx = s_planes;
    float* p_fy = p_fx + (size_t)cols * ns_stage;
    float* p_r = p_fy + (size_t)cols * ns_stage;
    float* p_g = p_r + (size_t)cols * ns_stage;
    float* p_b = p_g + (size_t)cols * ns_stage;
    // accumulators of the (up to) 2 x 2 sample tiles that reach this pixel: [tile row][tile column]
    f3 c00 = splat3(0.0f), c01 = c00, c10 = c00, c11 = c00;
    float w00 = 0.0f, w01 = 0.0f, w10 = 0.0f, w11 = 0.0f;
    for (int sy = SY0; sy <= SY1; ++sy) {
        for (int c0 = 0; c0 < NC; c0 += (int)cols) {
            const int ncol = min((int)cols, NC - c0);
            for (uint32_t s0 = 0; s0 < spp; s0 += ns_stage) {
                const uint32_t nsn = min(ns_stage, spp - s0);
                // ---- stage ----
                for (uint32_t e = (uint32_t)tid; e < (uint32_t)ncol * nsn; e += kBlock) {
                    const uint32_t c = e % (uint32_t)ncol, sl = e / (uint32_t)ncol;
                    const uint32_t pix = (uint32_t)(sy - se.sb_min[1]) * (uint32_t)se.sb_w + (uint32_t)(SX0 + c0 + (int)c - se.sb_min[0]);
                    const size_t idx = film_index(layout, npix, spp, s0 + sl, pix);
                    const float2 pf = pfilm[idx];
                    const float4 l4 = L[idx];
                    f3 l = mk3(l4.x, l4.y, l4.z);
                    if (has_nan(l)) l = splat3(0.0f);  // integrators/sampler.jl:46
                    const uint32_t a = sl * cols + c;
                    p_fx[a] = pf.x;
                    p_fy[a] = pf.y;
                    p_r[a] = l.x;
                    p_g[a] = l.y;
                    p_b[a] = l.z;
                }
                __syncthreads();
                // ---- accumulate ----
                if (in_film && sy >= sy_lo && sy <= sy_hi) {
                    const int xa = max(sx_lo, SX0 + c0), xb = min(sx_hi, SX0 + c0 + ncol - 1);
                    const int tyi = ((sy - se.sb_min[1]) >> 4) - ty_lo;
                    for (int sx = xa; sx <= xb; ++sx) {
                        const int txi = ((sx - se.sb_min[0]) >> 4) - tx_lo;
                        float bx0, by0, bx1, by1;
                        film_tile_bounds(se, tyi + ty_lo, txi + tx_lo, rx, ry, bx0, by0, bx1, by1);
                        if (X < bx0 || X > bx1 || Y < by0 || Y > by1) continue;  // pixel not in this FilmTile
                        const bool r1 = tyi != 0, q1 = txi != 0;
                        f3 csum = r1 ? (q1 ? c11 : c10) : (q1 ? c01 : c00);
                        float fws = r1 ? (q1 ? w11 : w10) : (q1 ? w01 : w00);
                        const uint32_t c = (uint32_t)(sx - SX0 - c0);
                        const float lim_x0 = jmax(bx0, 1.0f), lim_y0 = jmax(by0, 1.0f);
#pragma unroll 4
                        for (uint32_t sl = 0; sl < nsn; ++sl) {  // branch-free body: the four unrolled iterations' LDS reads overlap
                            const uint32_t a = sl * cols + c;
                            const float dpx = p_fx[a] - 0.5f, dpy = p_fy[a] - 0.5f;
                            const f3 lrgb = mk3(p_r[a], p_g[a], p_b[a]);
                            const float p0x = jmax(__builtin_ceilf(dpx - rx), lim_x0), p0y = jmax(__builtin_ceilf(dpy - ry), lim_y0);
                            const float p1x = jmin(__builtin_floorf(dpx + rx) + 1.0f, bx1), p1y = jmin(__builtin_floorf(dpy + ry) + 1.0f, by1);
                            const bool reach = !(X < p0x || X > p1x || Y < p0y || Y > p1y);
                            const float ffx = fabs_((X - dpx) * inv_rx * 16.0f), ffy = fabs_((Y - dpy) * inv_ry * 16.0f);
                            const int ox = (int)jclamp(__builtin_ceilf(ffx), 1.0f, 16.0f);
                            const int oy = (int)jclamp(__builtin_floorf(ffy), 1.0f, 16.0f);
                            const float w = s_table[(oy - 1) * 16 + (ox - 1)];
                            const f3 cnew = csum + lrgb * 1.0f * w;
                            const float wnew = fws + w;
                            csum = reach ? cnew : csum;
                            fws = reach ? wnew : fws;
                        }
                        if (r1) {
                            if (q1) {
                                c11 = csum;
                                w11 = fws;
                            } else {
                                c10 = csum;
                                w10 = fws;
                            }
                        } else {
                            if (q1) {
                                c01 = csum;
                                w01 = fws;
                            } else {
                                c00 = csum;
                                w00 = fws;
                            }
                        }
                    }
                }
                __syncthreads();
            }
        }
    }
    if (in_film) {
        // merge_film_tile! in k order: tile rows ascending, tile columns ascending; a tile whose FilmTile does not contain the
        // pixel never touched it (its accumulator is still zero and is skipped, as the reference's merge loop skips the pixel)
        f3 xyz = splat3(0.0f);
        float wsum = 0.0f;
        const int ty_hi = (sy_hi - se.sb_min[1]) >> 4, tx_hi = (sx_hi - se.sb_min[0]) >> 4;
        if (sx_lo <= sx_hi && sy_lo <= sy_hi)
            for (int ty = ty_lo; ty <= ty_hi; ++ty)
                for (int tx = tx_lo; tx <= tx_hi; ++tx) {
                    float bx0, by0, bx1, by1;
                    film_tile_bounds(se, ty, tx, rx, ry, bx0, by0, bx1, by1);
                    if (X < bx0 || X > bx1 || Y < by0 || Y > by1) continue;
                    const bool r1 = ty != ty_lo, q1 = tx != tx_lo;
                    xyz = xyz + rgb_to_xyz(r1 ? (q1 ? c11 : c10) : (q1 ? c01 : c00));
                    wsum += r1 ? (q1 ? w11 : w10) : (q1 ? w01 : w00);
                }
        out[(size_t)fy * se.film_w + fx] = make_float4(xyz.x, xyz.y, xyz.z, wsum);
    }
}

// save(film) up to the encoder (film.jl:204-222)
template <int TH_ONE_COPY = 0> __global__ void k_film_to_rgb(const float4* __restrict__ xyzw, uint32_t n, float scale, float* __restrict__ rgb) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = xyzw[i];
    f3 c = xyz_to_rgb(mk3(p.x, p.y, p.z));
    if (p.w != 0.0f) {
        const float inv_w = 1.0f / p.w;
        c = mk3(jmax(0.0f, c.x * inv_w), jmax(0.0f, c.y * inv_w), jmax(0.0f, c.z * inv_w));
    }
    c = c + 1.0f * xyz_to_rgb(splat3(0.0f));
    c = c * scale;
    rgb[3 * i] = jclamp(c.x, 0.0f, 1.0f);
    rgb[3 * i + 1] = jclamp(c.y, 0.0f, 1.0f);
    rgb[3 * i + 2] = jclamp(c.z, 0.0f, 1.0f);
}

// ---- test / inspection kernels -----------------------------------------------------------------------------------------------------
template <int TH_ONE_COPY = 0> __global__ void k_hit_geometry(DeviceScene sc, const float4* __restrict__ ro, const float4* __restrict__ rd, const float4* __restrict__ hits, uint32_t n, float* __restrict__ out15) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float* g = out15 + 15 * (size_t)i;
    const int prim = __float_as_int(hits[i].y);
    Shading sh;
    uint32_t material;
    const float4 o4 = ro[i], d4 = rd[i];
    if (prim < 0 || !rebuild_shading(sc, prim, mk3(o4.x, o4.y, o4.z), mk3(d4.x, d4.y, d4.z), sh, material)) {
        for (int k = 0; k < 15; ++k) g[k] = 0.0f;
        return;
    }
    const float v[15] = {sh.p.x, sh.p.y, sh.p.z, sh.ng.x, sh.ng.y, sh.ng.z, sh.ns.x, sh.ns.y, sh.ns.z, sh.wo.x, sh.wo.y, sh.wo.z, sh.ss.x, sh.ss.y, sh.ss.z};
    for (int k = 0; k < 15; ++k) g[k] = v[k];
}
template <int TH_ONE_COPY = 0> __global__ void k_prepare_rays(const float* __restrict__ rays8, uint32_t n, float4* __restrict__ ro, float4* __restrict__ rd, float* __restrict__ tmax) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* r = rays8 + 8 * (size_t)i;
    const f3 d = check_direction(mk3(r[4], r[5], r[6]));  // bvh.jl:217 / :265
    ro[i] = make_float4(r[0], r[1], r[2], 0.0f);
    rd[i] = make_float4(d.x, d.y, d.z, 0.0f);
    tmax[i] = r[3];
}
template <int TH_ONE_COPY = 0> __global__ void k_generate_rays(const DeviceSensor* __restrict__ sep, const float* __restrict__ samples5, uint32_t n, float* __restrict__ out8) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* c = samples5 + 5 * (size_t)i;
    f3 o, d;
    float time;
    generate_ray(*sep, f2{c[0], c[1]}, f2{c[2], c[3]}, c[4], o, d, time);
    float* r = out8 + 8 * (size_t)i;
    r[0] = o.x;
    r[1] = o.y;
    r[2] = o.z;
    r[3] = kInf;
    r[4] = d.x;
    r[5] = d.y;
    r[6] = d.z;
    r[7] = time;
}
template <int TH_ONE_COPY = 0> __global__ void k_bsdf_query(DeviceScene sc, uint32_t material, int multi, int mode, int flags, const float* __restrict__ frame9, const float* __restrict__ dirs6, uint32_t n,
                             float* __restrict__ out8) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* fr = frame9 + 9 * (size_t)i;
    const float* dd = dirs6 + 6 * (size_t)i;
    Shading sh;
    sh.p = splat3(0.0f);
    sh.wo = splat3(0.0f);
    sh.ng = mk3(fr[0], fr[1], fr[2]);
    sh.ns = mk3(fr[3], fr[4], fr[5]);
    sh.ss = normalize(mk3(fr[6], fr[7], fr[8]));
    sh.ts = cross(sh.ns, sh.ss);
    const LobeSet& b = sc.materials[material].set[multi ? 1 : 0];
    const f3 wo = mk3(dd[0], dd[1], dd[2]);
    float* o = out8 + 8 * (size_t)i;
    if (mode == 0) {
        const f3 wi = mk3(dd[3], dd[4], dd[5]);
        const f3 f = bsdf_f(b, sh, wo, wi, flags);
        o[0] = f.x;
        o[1] = f.y;
        o[2] = f.z;
        o[3] = bsdf_pdf(b, sh, wo, wi, flags);
        o[4] = o[5] = o[6] = o[7] = 0.0f;
    } else {
        const BsdfSample s = bsdf_sample_f(b, sh, wo, f2{dd[3], dd[4]}, flags);
        o[0] = s.wi.x;
        o[1] = s.wi.y;
        o[2] = s.wi.z;
        o[3] = s.f.x;
        o[4] = s.f.y;
        o[5] = s.f.z;
        o[6] = s.pdf;
        o[7] = (float)s.sampled_type;
    }
}
// Per-sample radiance read-back: float4 L -> rgb with the NaN rule of integrators/sampler.jl:46
// layout 1: L is pixel-group-major (film_index; k_raygen's Lf mode); the export is sample-major either way
template <int TH_ONE_COPY = 0> __global__ void k_export_L(const float4* __restrict__ L, uint64_t n, float* __restrict__ out, uint32_t layout = 0, uint32_t npix = 1, uint32_t spp = 1) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 l = L[layout ? film_index(1u, npix, spp, (uint32_t)(i / npix), (uint32_t)(i % npix)) : (size_t)i];
    f3 c = mk3(l.x, l.y, l.z);
    if (has_nan(c)) c = splat3(0.0f);
    out[3 * i] = c.x;
    out[3 * i + 1] = c.y;
    out[3 * i + 2] = c.z;
}
template <int TH_ONE_COPY = 0> __global__ void k_import_L(const float* __restrict__ in, uint64_t n, float4* __restrict__ L) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    L[i] = make_float4(in[3 * i], in[3 * i + 1], in[3 * i + 2], 0.0f);
}

}  // namespace th
