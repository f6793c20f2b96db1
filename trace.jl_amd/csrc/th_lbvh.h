// th_lbvh.h — BVHAccel(primitives, 1) built on the device: a linear BVH in the reference's flat layout.
//
// The reference builds its BVH on the host with a (quirky) SAH (accel/bvh.jl:55-206); the results of traversal do not depend on
// the topology except for exact-t ties (A.6), so any BVH2 over the same primitives in the same flat layout is a valid
// BVHAccel for this library, and the oracle walks whichever one the library built (trhip_scene_get_bvh).  The binned-SAH host
// builder (th_bvh.h) needs ~0.3 s per million primitives (its top levels run as parallel tasks); this one needs milliseconds:
//   centroid bounds -> 63-bit Morton keys -> radix sort (hipCUB) -> Karras' parallel hierarchy (one thread per internal
//   node) -> bottom-up bounds (second arrival at a node continues) -> depth-first numbering in closed form
//   (dfs(node) = 2 * first_leaf(node) + number of left-child edges on its root path), which yields the reference layout
//   directly: first child = i + 1, second child = i + 2 * leaves(first child), leaves in sorted order = ordered primitive slots.
// One primitive per leaf.  Quality is that of an LBVH (no SAH): see DESIGN.md for the measured traversal cost against th_bvh.h.
#pragma once
#include <hipcub/hipcub.hpp>

#include "th_bvh.h"
#include "th_sppm.h"  // enc_f32 / dec_f32, wave_min / wave_max

namespace th {

struct LbvhBuild {  // device arrays, n primitives, n - 1 internal nodes
    const float* prim_bounds;  // n * 6 (min xyz, max xyz)
    uint64_t* keys;            // sorted Morton keys
    uint32_t* sorted;          // sorted position -> primitive index
    uint32_t* left;            // per internal node: child ref, bit 31 = leaf (then the low bits are the sorted position)
    uint32_t* right;
    uint32_t* lo;              // first leaf (sorted position) of the internal node's range
    uint32_t* split;           // last leaf of the left child
    uint32_t* parent_int;      // per internal node: parent internal node | bit 31 = "I am the left child"; root: 0xffffffff
    uint32_t* parent_leaf;     // per leaf: the same
    uint32_t* visits;          // per internal node, zeroed
    float* ibounds;            // per internal node, 6 floats
    uint32_t* cbounds;         // 6 order-preserving encodings: centroid min xyz, max xyz
    uint32_t* max_depth;
    uint32_t n;
};

template <int TH_ONE_COPY = 0> __global__ __launch_bounds__(kBlock) void k_lbvh_centroid_bounds(LbvhBuild b) {
    float mn[3] = {kInf, kInf, kInf}, mx[3] = {-kInf, -kInf, -kInf};
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < b.n; i += gridDim.x * kBlock)
        for (int a = 0; a < 3; ++a) {
            const float c = 0.5f * b.prim_bounds[6 * (size_t)i + a] + 0.5f * b.prim_bounds[6 * (size_t)i + 3 + a];  // bvh.jl:12
            mn[a] = fminf(mn[a], c);
            mx[a] = fmaxf(mx[a], c);
        }
    for (int a = 0; a < 3; ++a) {
        mn[a] = wave_min(mn[a]);
        mx[a] = wave_max(mx[a]);
    }
    if (lane_id() == 0)
        for (int a = 0; a < 3; ++a) {
            atomicMin(&b.cbounds[a], enc_f32(mn[a]));
            atomicMax(&b.cbounds[3 + a], enc_f32(mx[a]));
        }
}
TH_D uint64_t spread21(uint64_t v) {  // 21 bits -> every third bit
    v &= 0x1fffffull;
    v = (v | v << 32) & 0x1f00000000ffffull;
    v = (v | v << 16) & 0x1f0000ff0000ffull;
    v = (v | v << 8) & 0x100f00f00f00f00full;
    v = (v | v << 4) & 0x10c30c30c30c30c3ull;
    v = (v | v << 2) & 0x1249249249249249ull;
    return v;
}
template <int TH_ONE_COPY = 0> __global__ __launch_bounds__(kBlock) void k_lbvh_keys(LbvhBuild b) {
    float cmin[3], inv[3];
    for (int a = 0; a < 3; ++a) {
        cmin[a] = dec_f32(b.cbounds[a]);
        const float ext = dec_f32(b.cbounds[3 + a]) - cmin[a];
        inv[a] = ext > 0.0f ? 2097152.0f / ext : 0.0f;
    }
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < b.n; i += gridDim.x * kBlock) {
        uint64_t q[3];
        for (int a = 0; a < 3; ++a) {
            const float c = 0.5f * b.prim_bounds[6 * (size_t)i + a] + 0.5f * b.prim_bounds[6 * (size_t)i + 3 + a];
            const float f = (c - cmin[a]) * inv[a];
            q[a] = (uint64_t)fminf(fmaxf(f, 0.0f), 2097151.0f);
        }
        b.keys[i] = spread21(q[0]) << 2 | spread21(q[1]) << 1 | spread21(q[2]);
        b.sorted[i] = i;
    }
}
// length of the common prefix of keys i and j (Karras 2012), ties broken by the index; -1 outside the array
TH_D int lbvh_delta(const uint64_t* __restrict__ keys, uint32_t n, int i, int j) {
    if (j < 0 || j >= (int)n) return -1;
    const uint64_t a = keys[i], c = keys[j];
    if (a != c) return __clzll((long long)(a ^ c));
    return 64 + __clz((int)((uint32_t)i ^ (uint32_t)j));
}
template <int TH_ONE_COPY = 0> __global__ __launch_bounds__(kBlock) void k_lbvh_hierarchy(LbvhBuild b) {
    const uint32_t n = b.n;
    for (uint32_t ii = blockIdx.x * kBlock + threadIdx.x; ii + 1 < n; ii += gridDim.x * kBlock) {
        const int i = (int)ii;
        const int d = lbvh_delta(b.keys, n, i, i + 1) - lbvh_delta(b.keys, n, i, i - 1) >= 0 ? 1 : -1;
        const int dmin = lbvh_delta(b.keys, n, i, i - d);
        int lmax = 2;
        while (lbvh_delta(b.keys, n, i, i + lmax * d) > dmin) lmax *= 2;
        int l = 0;
        for (int t = lmax / 2; t >= 1; t /= 2)
            if (lbvh_delta(b.keys, n, i, i + (l + t) * d) > dmin) l += t;
        const int j = i + l * d;
        const int dnode = lbvh_delta(b.keys, n, i, j);
        int s = 0;
        for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
            if (lbvh_delta(b.keys, n, i, i + (s + t) * d) > dnode) s += t;
            if (t == 1) break;
        }
        const int gamma = i + s * d + min(d, 0);
        const int first = min(i, j), last = max(i, j);
        const bool left_leaf = first == gamma, right_leaf = last == gamma + 1;
        b.left[ii] = left_leaf ? (0x80000000u | (uint32_t)gamma) : (uint32_t)gamma;
        b.right[ii] = right_leaf ? (0x80000000u | (uint32_t)(gamma + 1)) : (uint32_t)(gamma + 1);
        b.lo[ii] = (uint32_t)first;
        b.split[ii] = (uint32_t)gamma;
        if (left_leaf)
            b.parent_leaf[gamma] = ii | 0x80000000u;
        else
            b.parent_int[gamma] = ii | 0x80000000u;
        if (right_leaf)
            b.parent_leaf[gamma + 1] = ii;
        else
            b.parent_int[gamma + 1] = ii;
        if (ii == 0) b.parent_int[0] = 0xffffffffu;  // node 0 is the root (range [0, n - 1]); nobody writes its parent
    }
}
// bounds bottom-up: the second thread to arrive at a node unions its children's boxes and goes on
template <int TH_ONE_COPY = 0> __global__ __launch_bounds__(kBlock) void k_lbvh_refit(LbvhBuild b) {
    for (uint32_t k = blockIdx.x * kBlock + threadIdx.x; k < b.n; k += gridDim.x * kBlock) {
        uint32_t p = b.parent_leaf[k];
        while (true) {
            const uint32_t node = p & 0x7fffffffu;
            __threadfence();
            if (atomicAdd(&b.visits[node], 1u) == 0u) break;  // first arrival: the sibling subtree is not finished
            __threadfence();
            float box[6] = {kInf, kInf, kInf, -kInf, -kInf, -kInf};
            const uint32_t ch[2] = {b.left[node], b.right[node]};
            for (int c = 0; c < 2; ++c) {
                const float* src = (ch[c] & 0x80000000u) ? b.prim_bounds + 6 * (size_t)b.sorted[ch[c] & 0x7fffffffu] : b.ibounds + 6 * (size_t)ch[c];
                for (int a = 0; a < 3; ++a) {
                    box[a] = fminf(box[a], src[a]);
                    box[3 + a] = fmaxf(box[3 + a], src[3 + a]);
                }
            }
            for (int a = 0; a < 6; ++a) b.ibounds[6 * (size_t)node + a] = box[a];
            p = b.parent_int[node];
            if (p == 0xffffffffu) break;
        }
    }
}
// flat layout (bvh.jl:38-48, 187-206): node index = 2 * first leaf + left-child edges on the root path
struct LbvhFlat {
    float* bounds;     // (2n - 1) * 6
    uint32_t* a;       // leaf: ordered slot; interior: second child
    uint32_t* flags;   // leaf: 1 << 2 | 3; interior: split axis
    uint32_t* order;   // ordered slot -> primitive
};
template <int TH_ONE_COPY = 0> __global__ __launch_bounds__(kBlock) void k_lbvh_flatten(LbvhBuild b, LbvhFlat f) {
    const uint32_t n = b.n, total = 2 * n - 1;
    uint32_t deepest = 0;
    for (uint32_t t = blockIdx.x * kBlock + threadIdx.x; t < total; t += gridDim.x * kBlock) {
        const bool leaf = t >= n - 1;
        const uint32_t id = leaf ? t - (n - 1) : t;
        uint32_t p = leaf ? b.parent_leaf[id] : b.parent_int[id];
        uint32_t lefts = 0, depth = 1;
        while (p != 0xffffffffu) {
            lefts += p >> 31;
            depth++;
            p = b.parent_int[p & 0x7fffffffu];
        }
        deepest = max(deepest, depth);
        const uint32_t first = leaf ? id : b.lo[id];
        const uint32_t dfs = 2 * first + lefts;
        const float* src = leaf ? b.prim_bounds + 6 * (size_t)b.sorted[id] : b.ibounds + 6 * (size_t)id;
        for (int a = 0; a < 6; ++a) f.bounds[6 * (size_t)dfs + a] = src[a];
        if (leaf) {
            f.a[dfs] = id;
            f.flags[dfs] = (1u << 2) | 3u;
            f.order[id] = b.sorted[id];
        } else {
            const uint32_t s = b.split[id];
            f.a[dfs] = dfs + 2 * (s - first + 1);  // skip the first child's subtree: 2 * leaves - 1 nodes, + 1
            const uint64_t x = b.keys[s] ^ b.keys[s + 1];
            f.flags[dfs] = x ? (uint32_t)((63 - __clzll((long long)x)) % 3 == 2 ? 0 : ((63 - __clzll((long long)x)) % 3 == 1 ? 1 : 2)) : 0u;  // the axis of the highest differing key bit
        }
    }
    deepest = (uint32_t)wave_max((float)deepest);
    if (lane_id() == 0) atomicMax(b.max_depth, deepest);
}

}  // namespace th
