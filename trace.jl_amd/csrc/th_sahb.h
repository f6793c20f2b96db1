// th_sahb.h — BVHAccel(primitives, max_node_primitives) built on the device with the binned SAH of th_bvh.h.
//
// Same decisions as the host builder (16 bins over the centroid bounds, all three axes, cost = area x count of the two sides, leaf-cost
// test at the leaf-size hint, median split past depth 40): the tree has the host builder's quality (the LBVH of th_lbvh.h costs
// 25-35 % more node visits), built in milliseconds instead of ~0.3 s per million primitives.  Where the host's std::partition /
// nth_element leave the order inside a side unspecified the two builders may order a leaf's primitives differently; every decision
// that shapes the tree depends on the SETS only.
//
// Two phases:
//   top   — nodes of more than kSahSmall primitives, one tree level per round, every primitive position handled by one thread:
//           bounds + centroid bounds per node (atomics on order-preserving encodings, aggregated per block in LDS when the block's
//           positions belong to one node), bins likewise, one thread per node evaluates the SAH, a device-wide exclusive scan of
//           the "goes left" flags places every primitive in its child's range (stable partition);
//   small — one thread per node of <= kSahSmall primitives runs th_bvh.h's recursion on its own range with an explicit stack.
// Nodes live in a pool in creation order; the reference's depth-first layout (first child = i + 1, bvh.jl:187-206) is a closed
// form: index = 2 x (leaves before the node's first leaf) + (left-child edges on its root path).
#pragma once
#include <hipcub/hipcub.hpp>

#include "th_bvh.h"
#include "th_sppm.h"  // enc_f32 / dec_f32, wave_min / wave_max

namespace th {

constexpr int kSahBins = TH_BVH_BINS;
constexpr uint32_t kSahSmall = 64;            // nodes of at most this many primitives finish in one thread
constexpr uint32_t kSahNone = 0xffffffffu;
constexpr uint32_t kSahChunk = kBlock * 8;    // positions per block in the per-position passes
constexpr int kSahBinWords = 3 * kSahBins * 7;  // per node: 3 axes x bins x (6 encoded bounds + count)
constexpr uint32_t kSahTopDepth = 39;         // the top phase gives up here (the host builder switches to median splits at 40)

struct SahBuild {
    const float* pb;     // n * 6
    float* cen;          // n * 3
    uint32_t* idx_in;    // position -> primitive (this round's input)
    uint32_t* idx_out;
    uint32_t* pos_in;    // position -> active slot of this round | kSahNone
    uint32_t* pos_out;
    // node pool
    float* nb;           // 6 per node
    uint32_t* n_lo;
    uint32_t* n_hi;
    uint32_t* n_left;    // kSahNone: leaf
    uint32_t* n_right;
    uint32_t* n_axis;
    uint32_t* n_depth;
    uint32_t* n_lefts;   // left-child edges on the root path
    uint32_t* counters;  // [0] nodes allocated  [1] active nodes of the next round  [2] small nodes  [3] give-up flag  [4] max depth
    uint32_t pool_cap;
    // per round
    uint32_t* act;       // slot -> node
    uint32_t* act_next;
    uint32_t* small;     // nodes for the small phase
    uint32_t* lvl_b;     // slot * 6 encoded bounds
    uint32_t* lvl_cb;    // slot * 6 encoded centroid bounds
    uint32_t* bins;      // slot * kSahBinWords
    uint32_t* split;     // slot -> axis | bin << 2, kSahNone = no split this round
    uint32_t* cslot;     // slot * 2 -> the children's slots in the next round | kSahNone
    uint32_t* flag;      // n + 1
    uint32_t* scan;      // n + 1
    uint32_t n;
    uint32_t n_active;
    int max_leaf;
    int split_coincident;
};

TH_D int sah_bin(float c, float c0, float scale) {
    int k = (int)((c - c0) * scale);
    return min(kSahBins - 1, max(0, k));
}
TH_D float sah_half_area(const float* mn, const float* mx) {
    const float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
    return dx * dy + dx * dz + dy * dz;
}

template <int TH_ONE_COPY = 0> __global__ __launch_bounds__(kBlock) void k_sah_init(SahBuild b) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < b.n; i += gridDim.x * kBlock) {
        b.idx_in[i] = i;
        b.pos_in[i] = b.n > kSahSmall ? 0u : kSahNone;
        for (int a = 0; a < 3; ++a) b.cen[3 * (size_t)i + a] = 0.5f * b.pb[6 * (size_t)i + a] + 0.5f * b.pb[6 * (size_t)i + 3 + a];  // bvh.jl:12
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        b.n_lo[0] = 0;
        b.n_hi[0] = b.n;
        b.n_left[0] = kSahNone;
        b.n_right[0] = kSahNone;
        b.n_axis[0] = 0;
        b.n_depth[0] = 1;
        b.n_lefts[0] = 0;
        b.counters[0] = 1;
        b.counters[1] = 0;
        b.counters[2] = b.n > kSahSmall ? 0u : 1u;
        b.counters[3] = 0;
        b.counters[4] = 1;
        b.act[0] = 0;
        b.small[0] = 0;
    }
}
template <int TH_ONE_COPY = 0> __global__ __launch_bounds__(kBlock) void k_sah_round_init(SahBuild b) {
    const uint32_t emin = enc_f32(kInf), emax = enc_f32(-kInf);
    const size_t words = (size_t)b.n_active * kSahBinWords;
    for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < words; i += (size_t)gridDim.x * kBlock) {
        const uint32_t w = (uint32_t)(i % 7);
        b.bins[i] = w < 3 ? emin : (w < 6 ? emax : 0u);
    }
    for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < (size_t)b.n_active * 6; i += (size_t)gridDim.x * kBlock) {
        const uint32_t v = (i % 6) < 3 ? emin : emax;
        b.lvl_b[i] = v;
        b.lvl_cb[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) b.counters[1] = 0;
}
// the block's positions [first, last) all belong to one active slot?  (node ranges are contiguous and disjoint)
TH_D uint32_t sah_block_slot(const SahBuild& b, uint32_t first, uint32_t last) {
    const uint32_t s0 = b.pos_in[first], s1 = b.pos_in[last - 1];
    return s0 == s1 ? s0 : kSahNone;
}
// node bounds and centroid bounds of every active node
template <int TH_ONE_COPY = 0> __global__ __launch_bounds__(kBlock) void k_sah_bounds(SahBuild b) {
    __shared__ uint32_t s_acc[12];
    const uint32_t first = blockIdx.x * kSahChunk, last = min(b.n, first + kSahChunk);
    if (first >= last) return;
    const uint32_t uni = sah_block_slot(b, first, last);
    if (uni != kSahNone) {
        if (threadIdx.x < 12) s_acc[threadIdx.x] = (threadIdx.x % 6) < 3 ? enc_f32(kInf) : enc_f32(-kInf);
        __syncthreads();
        float mn[3] = {kInf, kInf, kInf}, mx[3] = {-kInf, -kInf, -kInf}, cmn[3] = {kInf, kInf, kInf}, cmx[3] = {-kInf, -kInf, -kInf};
        for (uint32_t i = first + threadIdx.x; i < last; i += kBlock) {
            const uint32_t p = b.idx_in[i];
            for (int a = 0; a < 3; ++a) {
                mn[a] = fminf(mn[a], b.pb[6 * (size_t)p + a]);
                mx[a] = fmaxf(mx[a], b.pb[6 * (size_t)p + 3 + a]);
                const float c = b.cen[3 * (size_t)p + a];
                cmn[a] = fminf(cmn[a], c);
                cmx[a] = fmaxf(cmx[a], c);
            }
        }
        for (int a = 0; a < 3; ++a) {
            mn[a] = wave_min(mn[a]);
            mx[a] = wave_max(mx[a]);
            cmn[a] = wave_min(cmn[a]);
            cmx[a] = wave_max(cmx[a]);
        }
        if (lane_id() == 0)
            for (int a = 0; a < 3; ++a) {
                atomicMin(&s_acc[a], enc_f32(mn[a]));
                atomicMax(&s_acc[3 + a], enc_f32(mx[a]));
                atomicMin(&s_acc[6 + a], enc_f32(cmn[a]));
                atomicMax(&s_acc[9 + a], enc_f32(cmx[a]));
            }
        __syncthreads();
        if (threadIdx.x < 12) {
            uint32_t* dst = (threadIdx.x < 6 ? b.lvl_b : b.lvl_cb) + 6 * (size_t)uni + threadIdx.x % 6;
            if ((threadIdx.x % 6) < 3)
                atomicMin(dst, s_acc[threadIdx.x]);
            else
                atomicMax(dst, s_acc[threadIdx.x]);
        }
        return;
    }
    // a block that spans several nodes: a wave's 64 consecutive positions still belong to one, two or three of them (a node of this phase holds more than
    // kSahSmall = 64 primitives) — one reduction per node present in the wave, 12 atomics each, instead of 12 per primitive
    for (uint32_t base = first + (threadIdx.x & ~63u); base < last; base += kBlock) {
        const uint32_t i = base + (threadIdx.x & 63u);
        const bool in = i < last;
        const uint32_t slot = in ? b.pos_in[i] : kSahNone;
        float lo[3] = {kInf, kInf, kInf}, hi[3] = {-kInf, -kInf, -kInf}, c[3] = {0.0f, 0.0f, 0.0f};
        if (slot != kSahNone) {
            const uint32_t p = b.idx_in[i];
            for (int a = 0; a < 3; ++a) {
                lo[a] = b.pb[6 * (size_t)p + a];
                hi[a] = b.pb[6 * (size_t)p + 3 + a];
                c[a] = b.cen[3 * (size_t)p + a];
            }
        }
        unsigned long long todo = __ballot(slot != kSahNone);
        while (todo) {
            const uint32_t s = (uint32_t)__shfl((int)slot, __ffsll((long long)todo) - 1);
            const bool mine = slot == s;
            float r[12];
            for (int a = 0; a < 3; ++a) {
                r[a] = wave_min(mine && lo[a] == lo[a] ? lo[a] : kInf);
                r[3 + a] = wave_max(mine && hi[a] == hi[a] ? hi[a] : -kInf);
                r[6 + a] = wave_min(mine && c[a] == c[a] ? c[a] : kInf);
                r[9 + a] = wave_max(mine && c[a] == c[a] ? c[a] : -kInf);
            }
            if (lane_id() == 0)
                for (int a = 0; a < 3; ++a) {
                    atomicMin(&b.lvl_b[6 * (size_t)s + a], enc_f32(r[a]));
                    atomicMax(&b.lvl_b[6 * (size_t)s + 3 + a], enc_f32(r[3 + a]));
                    atomicMin(&b.lvl_cb[6 * (size_t)s + a], enc_f32(r[6 + a]));
                    atomicMax(&b.lvl_cb[6 * (size_t)s + 3 + a], enc_f32(r[9 + a]));
                }
            todo &= ~__ballot(mine);
        }
    }
}
TH_D void sah_bin_add(uint32_t* bin, const float* box) {
    for (int a = 0; a < 3; ++a) {
        if (box[a] == box[a]) atomicMin(&bin[a], enc_f32(box[a]));
        if (box[3 + a] == box[3 + a]) atomicMax(&bin[3 + a], enc_f32(box[3 + a]));
    }
    atomicAdd(&bin[6], 1u);
}
// every primitive into its bin on each axis
template <int TH_ONE_COPY = 0> __global__ __launch_bounds__(kBlock) void k_sah_bin(SahBuild b) {
    __shared__ uint32_t s_bins[kSahBinWords];
    const uint32_t first = blockIdx.x * kSahChunk, last = min(b.n, first + kSahChunk);
    if (first >= last) return;
    const uint32_t uni = sah_block_slot(b, first, last);
    if (uni != kSahNone) {
        for (int w = threadIdx.x; w < kSahBinWords; w += kBlock) s_bins[w] = (w % 7) < 3 ? enc_f32(kInf) : ((w % 7) < 6 ? enc_f32(-kInf) : 0u);
        __syncthreads();
        float c0[3], scale[3];
        bool on[3];
        for (int a = 0; a < 3; ++a) {
            c0[a] = dec_f32(b.lvl_cb[6 * (size_t)uni + a]);
            const float c1 = dec_f32(b.lvl_cb[6 * (size_t)uni + 3 + a]);
            on[a] = c1 > c0[a];
            scale[a] = (float)kSahBins / (c1 - c0[a]);
        }
        for (uint32_t i = first + threadIdx.x; i < last; i += kBlock) {
            const uint32_t p = b.idx_in[i];
            float box[6];
            for (int a = 0; a < 6; ++a) box[a] = b.pb[6 * (size_t)p + a];
            for (int a = 0; a < 3; ++a)
                if (on[a]) sah_bin_add(&s_bins[(a * kSahBins + sah_bin(b.cen[3 * (size_t)p + a], c0[a], scale[a])) * 7], box);
        }
        __syncthreads();
        uint32_t* dst = b.bins + (size_t)uni * kSahBinWords;
        for (int w = threadIdx.x; w < kSahBinWords; w += kBlock) {
            const uint32_t v = s_bins[w];
            if ((w % 7) < 3) {
                if (v != enc_f32(kInf)) atomicMin(&dst[w], v);
            } else if ((w % 7) < 6) {
                if (v != enc_f32(-kInf)) atomicMax(&dst[w], v);
            } else if (v) {
                atomicAdd(&dst[w], v);
            }
        }
        return;
    }
    // a block that spans several nodes: per wave, one pass per node present in it through the wave's own bins in LDS (see k_sah_bounds)
    __shared__ uint32_t s_wbins[kBlock / 64][kSahBinWords];
    uint32_t* wb = s_wbins[threadIdx.x >> 6];
    const uint32_t lane = lane_id();
    for (uint32_t base = first + (threadIdx.x & ~63u); base < last; base += kBlock) {
        const uint32_t i = base + lane;
        const bool in = i < last;
        const uint32_t slot = in ? b.pos_in[i] : kSahNone;
        uint32_t p = 0;
        float box[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}, cen[3] = {0.0f, 0.0f, 0.0f};
        if (slot != kSahNone) {
            p = b.idx_in[i];
            for (int a = 0; a < 6; ++a) box[a] = b.pb[6 * (size_t)p + a];
            for (int a = 0; a < 3; ++a) cen[a] = b.cen[3 * (size_t)p + a];
        }
        unsigned long long todo = __ballot(slot != kSahNone);
        while (todo) {
            const uint32_t s = (uint32_t)__shfl((int)slot, __ffsll((long long)todo) - 1);
            const bool mine = slot == s;
            for (int w = lane; w < kSahBinWords; w += 64) wb[w] = (w % 7) < 3 ? enc_f32(kInf) : ((w % 7) < 6 ? enc_f32(-kInf) : 0u);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (mine)
                for (int a = 0; a < 3; ++a) {
                    const float c0 = dec_f32(b.lvl_cb[6 * (size_t)s + a]), c1 = dec_f32(b.lvl_cb[6 * (size_t)s + 3 + a]);
                    if (!(c1 > c0)) continue;
                    const float scale = (float)kSahBins / (c1 - c0);
                    sah_bin_add(&wb[(a * kSahBins + sah_bin(cen[a], c0, scale)) * 7], box);
                }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            uint32_t* dst = b.bins + (size_t)s * kSahBinWords;
            for (int w = lane; w < kSahBinWords; w += 64) {
                const uint32_t v = wb[w];
                if ((w % 7) < 3) {
                    if (v != enc_f32(kInf)) atomicMin(&dst[w], v);
                } else if ((w % 7) < 6) {
                    if (v != enc_f32(-kInf)) atomicMax(&dst[w], v);
                } else if (v) {
                    atomicAdd(&dst[w], v);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            todo &= ~__ballot(mine);
        }
    }
}
// the SAH sweep of th_bvh.h over one axis' bins; bb = kSahBins x (min xyz, max xyz), cnt = counts
TH_D void sah_sweep(const float (*bb)[6], const uint32_t* cnt, int ax, float& best_cost, int& best_axis, int& best_bin) {
    float right_area[kSahBins];
    uint32_t right_cnt[kSahBins];
    float mn[3] = {kInf, kInf, kInf}, mx[3] = {-kInf, -kInf, -kInf};
    uint32_t c = 0;
    for (int k = kSahBins - 1; k > 0; --k) {
        for (int a = 0; a < 3; ++a) {
            mn[a] = fminf(mn[a], bb[k][a]);
            mx[a] = fmaxf(mx[a], bb[k][3 + a]);
        }
        c += cnt[k];
        right_area[k] = sah_half_area(mn, mx);
        right_cnt[k] = c;
    }
    for (int a = 0; a < 3; ++a) {
        mn[a] = kInf;
        mx[a] = -kInf;
    }
    c = 0;
    for (int k = 0; k < kSahBins - 1; ++k) {
        for (int a = 0; a < 3; ++a) {
            mn[a] = fminf(mn[a], bb[k][a]);
            mx[a] = fmaxf(mx[a], bb[k][3 + a]);
        }
        c += cnt[k];
        if (c == 0 || right_cnt[k + 1] == 0) continue;
        const float cost = sah_half_area(mn, mx) * (float)c + right_area[k + 1] * (float)right_cnt[k + 1];
        if (cost < best_cost) {
            best_cost = cost;
            best_axis = ax;
            best_bin = k;
        }
    }
}
// one thread per active node: its bounds into the pool, the best split, two child ids
template <int TH_ONE_COPY = 0> __global__ __launch_bounds__(kBlock) void k_sah_split(SahBuild b) {
    const uint32_t slot = blockIdx.x * kBlock + threadIdx.x;
    if (slot >= b.n_active) return;
    const uint32_t node = b.act[slot];
    for (int a = 0; a < 6; ++a) b.nb[6 * (size_t)node + a] = dec_f32(b.lvl_b[6 * (size_t)slot + a]);
    int best_axis = -1, best_bin = -1;
    float best_cost = kInf;
    for (int ax = 0; ax < 3; ++ax) {
        const float c0 = dec_f32(b.lvl_cb[6 * (size_t)slot + ax]), c1 = dec_f32(b.lvl_cb[6 * (size_t)slot + 3 + ax]);
        if (!(c1 > c0)) continue;
        float bb[kSahBins][6];
        uint32_t cnt[kSahBins];
        const uint32_t* src = b.bins + (size_t)slot * kSahBinWords + (size_t)ax * kSahBins * 7;
        for (int k = 0; k < kSahBins; ++k) {
            for (int a = 0; a < 6; ++a) bb[k][a] = dec_f32(src[k * 7 + a]);
            cnt[k] = src[k * 7 + 6];
        }
        sah_sweep(bb, cnt, ax, best_cost, best_axis, best_bin);
    }
    // no usable split of a large node (all centroids coincide) or a tree as deep as the host builder's median rule: the host builds this scene
    if (best_axis < 0 || b.n_depth[node] >= kSahTopDepth) {
        b.counters[3] = 1;
        b.split[slot] = kSahNone;
        return;
    }
    const uint32_t base = atomicAdd(&b.counters[0], 2u);
    if (base + 2 > b.pool_cap) {
        b.counters[3] = 1;
        b.split[slot] = kSahNone;
        return;
    }
    b.split[slot] = (uint32_t)best_axis | ((uint32_t)best_bin << 2);
    b.n_left[node] = base;
    b.n_right[node] = base + 1;
    b.n_axis[node] = (uint32_t)best_axis;
}
template <int TH_ONE_COPY = 0> __global__ __launch_bounds__(kBlock) void k_sah_flag(SahBuild b) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i <= b.n; i += gridDim.x * kBlock) {
        uint32_t f = 0;
        if (i < b.n) {
            const uint32_t slot = b.pos_in[i];
            if (slot != kSahNone) {
                const uint32_t sp = b.split[slot];
                if (sp != kSahNone) {
                    const int ax = (int)(sp & 3u), bin = (int)(sp >> 2);
                    const float c0 = dec_f32(b.lvl_cb[6 * (size_t)slot + ax]), c1 = dec_f32(b.lvl_cb[6 * (size_t)slot + 3 + ax]);
                    const float scale = (float)kSahBins / (c1 - c0);
                    f = sah_bin(b.cen[3 * (size_t)b.idx_in[i] + ax], c0, scale) <= bin ? 1u : 0u;
                }
            }
        }
        b.flag[i] = f;
    }
}
// one thread per split node: the children's ranges, and where they go next (another round, or the small phase)
template <int TH_ONE_COPY = 0> __global__ __launch_bounds__(kBlock) void k_sah_children(SahBuild b) {
    const uint32_t slot = blockIdx.x * kBlock + threadIdx.x;
    if (slot >= b.n_active) return;
    b.cslot[2 * (size_t)slot] = kSahNone;
    b.cslot[2 * (size_t)slot + 1] = kSahNone;
    if (b.split[slot] == kSahNone) return;
    const uint32_t node = b.act[slot], lo = b.n_lo[node], hi = b.n_hi[node];
    const uint32_t mid = lo + (b.scan[hi] - b.scan[lo]);
    const uint32_t depth = b.n_depth[node] + 1;
    atomicMax(&b.counters[4], depth);
    for (int c = 0; c < 2; ++c) {
        const uint32_t child = b.n_left[node] + c;
        const uint32_t clo = c ? mid : lo, chi = c ? hi : mid;
        b.n_lo[child] = clo;
        b.n_hi[child] = chi;
        b.n_left[child] = kSahNone;
        b.n_right[child] = kSahNone;
        b.n_axis[child] = 0;
        b.n_depth[child] = depth;
        b.n_lefts[child] = b.n_lefts[node] + (c ? 0u : 1u);
        if (chi - clo <= kSahSmall) {
            b.small[atomicAdd(&b.counters[2], 1u)] = child;
        } else {
            const uint32_t s = atomicAdd(&b.counters[1], 1u);
            b.act_next[s] = child;
            b.cslot[2 * (size_t)slot + c] = s;
        }
    }
}
template <int TH_ONE_COPY = 0> __global__ __launch_bounds__(kBlock) void k_sah_scatter(SahBuild b) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < b.n; i += gridDim.x * kBlock) {
        const uint32_t slot = b.pos_in[i], p = b.idx_in[i];
        if (slot == kSahNone || b.split[slot] == kSahNone) {
            b.idx_out[i] = p;
            b.pos_out[i] = kSahNone;
            continue;
        }
        const uint32_t node = b.act[slot], lo = b.n_lo[node], hi = b.n_hi[node];
        const uint32_t before = b.scan[i] - b.scan[lo], n_left = b.scan[hi] - b.scan[lo];
        const bool left = b.flag[i] != 0;
        const uint32_t dst = left ? lo + before : lo + n_left + (i - lo - before);
        b.idx_out[dst] = p;
        b.pos_out[dst] = b.cslot[2 * (size_t)slot + (left ? 0 : 1)];
    }
}

// ---- small phase: th_bvh.h's recursion, one thread per node of <= kSahSmall primitives ----
template <int TH_ONE_COPY = 0> __global__ __launch_bounds__(64) void k_sah_small(SahBuild b, uint32_t n_small) {
    const uint32_t t = blockIdx.x * 64 + threadIdx.x;
    if (t >= n_small) return;
    uint32_t stack[72];  // node ids; a subtree of <= 64 primitives is at most 63 interior nodes deep
    int sp = 0;
    stack[sp++] = b.small[t];
    uint32_t* idx = b.idx_in;
    while (sp > 0) {
        const uint32_t node = stack[--sp];
        const uint32_t lo = b.n_lo[node], hi = b.n_hi[node], n = hi - lo, depth = b.n_depth[node];
        float mn[3] = {kInf, kInf, kInf}, mx[3] = {-kInf, -kInf, -kInf}, cmn[3] = {kInf, kInf, kInf}, cmx[3] = {-kInf, -kInf, -kInf};
        for (uint32_t i = lo; i < hi; ++i) {
            const uint32_t p = idx[i];
            for (int a = 0; a < 3; ++a) {
                mn[a] = fminf(mn[a], b.pb[6 * (size_t)p + a]);
                mx[a] = fmaxf(mx[a], b.pb[6 * (size_t)p + 3 + a]);
                const float c = b.cen[3 * (size_t)p + a];
                cmn[a] = fminf(cmn[a], c);
                cmx[a] = fmaxf(cmx[a], c);
            }
        }
        for (int a = 0; a < 3; ++a) {
            b.nb[6 * (size_t)node + a] = mn[a];
            b.nb[6 * (size_t)node + 3 + a] = mx[a];
        }
        if (n == 1) continue;  // leaf (n_left stays kSahNone)
        int best_axis = -1, best_bin = -1;
        float best_cost = kInf;
        for (int ax = 0; ax < 3; ++ax) {
            const float c0 = cmn[ax], c1 = cmx[ax];
            if (!(c1 > c0)) continue;
            float bb[kSahBins][6];
            uint32_t cnt[kSahBins];
            for (int k = 0; k < kSahBins; ++k) {
                for (int a = 0; a < 3; ++a) {
                    bb[k][a] = kInf;
                    bb[k][3 + a] = -kInf;
                }
                cnt[k] = 0;
            }
            const float scale = (float)kSahBins / (c1 - c0);
            for (uint32_t i = lo; i < hi; ++i) {
                const uint32_t p = idx[i];
                const int k = sah_bin(b.cen[3 * (size_t)p + ax], c0, scale);
                for (int a = 0; a < 3; ++a) {
                    bb[k][a] = fminf(bb[k][a], b.pb[6 * (size_t)p + a]);
                    bb[k][3 + a] = fmaxf(bb[k][3 + a], b.pb[6 * (size_t)p + 3 + a]);
                }
                cnt[k]++;
            }
            sah_sweep(bb, cnt, ax, best_cost, best_axis, best_bin);
        }
        uint32_t mid;
        uint32_t axis_flag;
        if (best_axis < 0) {  // all centroids coincide (th_bvh.h: a leaf up to the hint, halves by index beyond it when asked to)
            if (!b.split_coincident || (int)n <= b.max_leaf || depth >= 60u || n > 255u) continue;
            mid = lo + n / 2;
            axis_flag = 0;
        } else {
            if ((int)n <= b.max_leaf) {
                const float area = sah_half_area(mn, mx);
                const float leaf_cost = (float)n * area;
                if (best_cost + 0.125f * area >= leaf_cost) continue;
            }
            const float c0 = cmn[best_axis], scale = (float)kSahBins / (cmx[best_axis] - cmn[best_axis]);
            // partition (the sets are th_bvh.h's; the order inside a side is this loop's)
            uint32_t i = lo, j = hi;
            while (i < j) {
                if (sah_bin(b.cen[3 * (size_t)idx[i] + best_axis], c0, scale) <= best_bin) {
                    ++i;
                } else {
                    --j;
                    const uint32_t tmp = idx[i];
                    idx[i] = idx[j];
                    idx[j] = tmp;
                }
            }
            mid = i;
            if (mid == lo || mid == hi || depth >= 40u) {  // median split: order the range by the centroid along the axis (<= 64 entries)
                for (uint32_t u = lo + 1; u < hi; ++u) {
                    const uint32_t pu = idx[u];
                    const float cu = b.cen[3 * (size_t)pu + best_axis];
                    uint32_t v = u;
                    while (v > lo && cu < b.cen[3 * (size_t)idx[v - 1] + best_axis]) {
                        idx[v] = idx[v - 1];
                        --v;
                    }
                    idx[v] = pu;
                }
                mid = lo + n / 2;
            }
            axis_flag = (uint32_t)best_axis;
        }
        const uint32_t base = atomicAdd(&b.counters[0], 2u);
        if (base + 2 > b.pool_cap || sp + 2 > 72) {
            b.counters[3] = 1;
            return;
        }
        b.n_left[node] = base;
        b.n_right[node] = base + 1;
        b.n_axis[node] = axis_flag;
        atomicMax(&b.counters[4], depth + 1);
        for (int c = 0; c < 2; ++c) {
            const uint32_t child = base + c;
            b.n_lo[child] = c ? mid : lo;
            b.n_hi[child] = c ? hi : mid;
            b.n_left[child] = kSahNone;
            b.n_right[child] = kSahNone;
            b.n_axis[child] = 0;
            b.n_depth[child] = depth + 1;
            b.n_lefts[child] = b.n_lefts[node] + (c ? 0u : 1u);
        }
        stack[sp++] = base + 1;
        stack[sp++] = base;
    }
}

// ---- depth-first layout ----
template <int TH_ONE_COPY = 0> __global__ __launch_bounds__(kBlock) void k_sah_mark_leaves(SahBuild b, uint32_t n_nodes) {
    for (uint32_t node = blockIdx.x * kBlock + threadIdx.x; node < n_nodes; node += gridDim.x * kBlock)
        if (b.n_left[node] == kSahNone) b.flag[b.n_lo[node]] = 1u;
}
struct SahFlat {
    float* bounds;
    uint32_t* a;
    uint32_t* flags;
};
template <int TH_ONE_COPY = 0> __global__ __launch_bounds__(kBlock) void k_sah_flatten(SahBuild b, uint32_t n_nodes, SahFlat f) {
    for (uint32_t node = blockIdx.x * kBlock + threadIdx.x; node < n_nodes; node += gridDim.x * kBlock) {
        const uint32_t lo = b.n_lo[node];
        const uint32_t dfs = 2 * b.scan[lo] + b.n_lefts[node];
        for (int a = 0; a < 6; ++a) f.bounds[6 * (size_t)dfs + a] = b.nb[6 * (size_t)node + a];
        if (b.n_left[node] == kSahNone) {
            f.a[dfs] = lo;
            f.flags[dfs] = ((b.n_hi[node] - lo) << 2) | 3u;
        } else {
            const uint32_t r = b.n_right[node];
            f.a[dfs] = 2 * b.scan[b.n_lo[r]] + b.n_lefts[r];
            f.flags[dfs] = b.n_axis[node];
        }
    }
}

}  // namespace th
