// th_bvh_ref.h — BVHAccel(primitives, max_node_primitives) with the REFERENCE's own construction (accel/bvh.jl:55-206, Trace.jl:128-137),
// node for node: option "bvh_builder" = 2, and the host-only entry point trhip_build_bvh_host.  Traversal results depend on the topology
// only where two primitives are accepted at (nearly) the same t — the later tested one wins (bvh.jl:229-237, triangle_mesh.jl:211-214) —
// so a host that must agree with Trace.jl on those rays too asks for this tree instead of the library's binned-SAH one (th_bvh.h).
// What is reproduced on purpose:
//   * 12 buckets whose bounds START as the point (0, 0, 0), not empty (bvh.jl:130): every bucket's box reaches the origin;
//   * the cost of splitting after bucket i weighs the two unions by the LENGTHS of the bucket ranges 1:i and (i+1):11 — not by primitive
//     counts — and the right range never includes bucket 12 (bvh.jl:141-156);
//   * leaf when !(n > max_node_primitives || cost < n) (bvh.jl:159-165); two primitives are split by the smaller centroid (bvh.jl:121-127);
//   * partition! never tests the first element of its range in place and returns the index of the first element it did not move, and
//     _init puts THAT element into the left child (Trace.jl:128-137, bvh.jl:166-185): the right child may be EMPTY — a leaf of 0
//     primitives with the invalid bounds (+Inf, -Inf), which no ray's box test passes (bounds.jl:186-200: tx_min = +Inf);
//   * nodes in depth-first order, first child at i + 1 (bvh.jl:187-206).
// Float32 throughout, in the reference's operation order: centroid = 0.5 min + 0.5 max (bvh.jl:12), offset = (c - min) / (max - min)
// (bounds.jl:137-145), bucket = floor(12 x offset) + 1 clamped to 12, surface area = 2 ((dx dy + dx dz) + dy dz) (bounds.jl:93-96),
// cost = 1 + (s1 + s2) / area.  argmin is Julia's: the first minimum, a NaN counts as the smallest.
#pragma once
#include <atomic>
#include <cmath>
#include <cstdint>
#include <future>
#include <stdexcept>
#include <vector>

#include "th_bvh.h"

namespace th {

class RefBVHBuilder {
   public:
    // depth_limit > 0: give up (DepthExceeded) as soon as a node lies deeper — the commit's use: a tree deeper than the reference's 64-entry traversal stack is refused anyway
    // (bvh.jl:222), and finishing it first cost 7 s on the 10 M-triangle scene
    RefBVHBuilder(const std::vector<HostAABB>& prim_bounds, int max_node_prims, uint32_t depth_limit = 0) : pb_(prim_bounds), max_leaf_(std::min(255, max_node_prims)), depth_limit_(depth_limit) {}
    struct DepthExceeded : std::runtime_error {
        uint32_t depth;
        explicit DepthExceeded(uint32_t d) : std::runtime_error("the reference's BVH is deeper than the limit"), depth(d) {}
    };

    // throws std::runtime_error when the recursion does not end within kMaxRecursion levels (the reference would overflow its stack)
    FlatBVH build() {
        const uint32_t n = (uint32_t)pb_.size();
        info_.resize(n);
        cen_.resize((size_t)n * 3);
        for (uint32_t i = 0; i < n; ++i) {
            info_[i] = i;
            for (int a = 0; a < 3; ++a) cen_[3 * (size_t)i + a] = 0.5f * pb_[i].mn[a] + 0.5f * pb_[i].mx[a];
        }
        FlatBVH out;
        reserve(out, n);
        if (n) node(out, 0, n, 1);
        return out;
    }

   private:
    static constexpr int kBuckets = 12;
    static constexpr uint32_t kMaxRecursion = 128;  // (a tree deeper than the 64-entry traversal stack is refused by the commit anyway; each level holds the 12 buckets on the host stack)

    static float surface_area(const HostAABB& b) {
        const float dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
        return 2.0f * (dx * dy + dx * dz + dy * dz);
    }
    // bucket of a centroid along `dim` (bvh.jl:135-138): offset() divides by the extent only where it is positive (bounds.jl:139-144)
    int bucket_of(const HostAABB& cb, uint32_t prim, int dim) const {
        const float o = cen_[3 * (size_t)prim + dim] - cb.mn[dim];
        const float off = cb.mx[dim] > cb.mn[dim] ? o / (cb.mx[dim] - cb.mn[dim]) : o;
        // a NaN / Inf centroid or extent (caller-supplied bounds, trhip_build_bvh_host): the reference's Int(floor(...)) throws InexactError, its bucket[b] a BoundsError —
        // and (int) of such a float is undefined behaviour here.  Refused before the conversion.
        const float fb = std::floor((float)kBuckets * off);
        if (!(fb >= 0.0f && fb <= (float)kBuckets)) throw std::runtime_error("the reference's BVH construction fails on this input: a primitive's bounds are not finite (bvh.jl:135-138)");
        int b = (int)fb + 1;
        if (b == kBuckets + 1) b -= 1;
        return b;
    }
    static void reserve(FlatBVH& out, uint32_t n_prims) {  // (a hint: one primitive per leaf gives 2 n - 1 nodes, the empty leaves of partition! a few more)
        out.order.reserve(n_prims);
        out.a.reserve(2 * (size_t)n_prims);
        out.flags.reserve(2 * (size_t)n_prims);
        out.bounds.reserve(12 * (size_t)n_prims);
    }
    static uint32_t emit(FlatBVH& out, const HostAABB& b, uint32_t a, uint32_t flags) {
        const uint32_t at = (uint32_t)out.a.size();
        out.bounds.insert(out.bounds.end(), {b.mn[0], b.mn[1], b.mn[2], b.mx[0], b.mx[1], b.mx[2]});
        out.a.push_back(a);
        out.flags.push_back(flags);
        return at;
    }
    uint32_t leaf(FlatBVH& out, uint32_t from, uint32_t to, const HostAABB& bounds) {  // _create_leaf bvh.jl:97-106
        const uint32_t first = (uint32_t)out.order.size();
        for (uint32_t i = from; i < to; ++i) out.order.push_back(info_[i]);
        return emit(out, bounds, first, ((to - from) << 2) | 3u);
    }
    // _init over info_[from, to) (bvh.jl:87-185); returns the node's index in the flat arrays
    // `out` holds the subtree's nodes in the reference's order (a node, its first child's subtree, its second child's) with indices LOCAL to it: the subtrees of a large node work on
    // disjoint ranges of info_ and are built concurrently, the second one into arrays of its own that are appended (indices shifted) behind the first — the same flat arrays as the
    // serial recursion, node for node
    uint32_t node(FlatBVH& out, uint32_t from, uint32_t to, uint32_t depth) {
        if (abort_.load(std::memory_order_relaxed)) throw Aborted();
        if (depth > kMaxRecursion) throw std::runtime_error("the reference's BVH construction does not terminate on this input (bvh.jl:166-185 recursion)");
        if (depth_limit_ && depth > depth_limit_) throw DepthExceeded(depth);
        out.max_depth = std::max(out.max_depth, depth);
        const uint32_t n = to - from;
        HostAABB bounds;
        bounds.reset();
        for (uint32_t i = from; i < to; ++i) bounds.grow(pb_[info_[i]]);
        if (n == 1) return leaf(out, from, to, bounds);
        HostAABB cb;
        cb.reset();
        for (uint32_t i = from; i < to; ++i) cb.grow_point(&cen_[3 * (size_t)info_[i]]);
        const float dx = cb.mx[0] - cb.mn[0], dy = cb.mx[1] - cb.mn[1], dz = cb.mx[2] - cb.mn[2];
        const int dim = (dx > dy && dx > dz) ? 0 : (dy > dz ? 1 : 2);  // maximum_extent bounds.jl:118-126
        bool valid = true;                                             // is_valid bounds.jl:30-32 (an empty range: the 0-primitive leaf)
        for (int a = 0; a < 3; ++a) valid = valid && cb.mn[a] != INFINITY && cb.mx[a] != -INFINITY;
        if (!valid || cb.mn[dim] == cb.mx[dim]) return leaf(out, from, to, bounds);
        uint32_t mid;  // LAST index of the left child (the reference's `mid`, 0-based here)
        if (n <= 2) {
            mid = (from + to - 1) / 2;
            // partialsort!(view, 1, by = centroid[dim]): the smaller centroid first (an insertion sort of two: swapped only when strictly smaller)
            if (cen_[3 * (size_t)info_[to - 1] + dim] < cen_[3 * (size_t)info_[from] + dim]) std::swap(info_[from], info_[to - 1]);
        } else {
            HostAABB bucket[kBuckets];
            for (auto& b : bucket)
                for (int a = 0; a < 3; ++a) b.mn[a] = b.mx[a] = 0.0f;  // Bounds3(Point3f(0f0)) bvh.jl:130
            for (uint32_t i = from; i < to; ++i) bucket[bucket_of(cb, info_[i], dim) - 1].grow(pb_[info_[i]]);
            const float sa = surface_area(bounds);
            float costs[kBuckets - 1];
            for (int i = 1; i <= kBuckets - 1; ++i) {
                HostAABB u = bucket[0];
                for (int b = 2; b <= i; ++b) u.grow(bucket[b - 1]);
                const float s1 = (float)i * surface_area(u);
                float s2 = 0.0f;
                const int len2 = (kBuckets - 1) - i;  // length((i+1):(n_buckets-1))
                if (len2 > 0) {
                    HostAABB v = bucket[i];
                    for (int b = i + 2; b <= kBuckets - 1; ++b) v.grow(bucket[b - 1]);
                    s2 = (float)len2 * surface_area(v);
                }
                costs[i - 1] = 1.0f + (s1 + s2) / sa;
            }
            int best = 1;  // argmin: first minimum; NaN is smaller than everything (Julia's findmin)
            for (int i = 1; i <= kBuckets - 1; ++i) {
                if (std::isnan(costs[i - 1])) {
                    best = i;
                    break;
                }
                if (costs[i - 1] < costs[best - 1]) best = i;
            }
            if (!((int)n > max_leaf_ || costs[best - 1] < (float)n)) return leaf(out, from, to, bounds);
            // partition! Trace.jl:128-137
            uint32_t left = from;
            for (uint32_t i = from; i < to; ++i)
                if (left != i && bucket_of(cb, info_[i], dim) <= best) {
                    std::swap(info_[i], info_[left]);
                    left += 1;
                }
            mid = left;
        }
        const uint32_t self = emit(out, bounds, 0u, (uint32_t)dim);  // bounds = left ∪ right = the union over the range (min / max are exact)
        uint32_t second;
        if (n >= kParallelMin && depth <= kParallelDepth) {
            std::future<FlatBVH> right = std::async(std::launch::async, [this, mid, to, depth] {
                FlatBVH r;
                reserve(r, to - (mid + 1));
                try {
                    node(r, mid + 1, to, depth + 1);
                } catch (...) {
                    abort_.store(true, std::memory_order_relaxed);  // the other subtrees stop at their next node
                    throw;
                }
                return r;
            });
            try {
                node(out, from, mid + 1, depth + 1);
            } catch (const Aborted&) {
                right.get();  // the second subtree's own failure, when that is what stopped this one
                throw;
            } catch (...) {
                abort_.store(true, std::memory_order_relaxed);
                right.wait();
                throw;
            }
            const FlatBVH r = right.get();  // (rethrows what the second subtree threw)
            second = (uint32_t)out.a.size();
            const uint32_t order_base = (uint32_t)out.order.size();
            out.bounds.insert(out.bounds.end(), r.bounds.begin(), r.bounds.end());
            out.flags.insert(out.flags.end(), r.flags.begin(), r.flags.end());
            out.order.insert(out.order.end(), r.order.begin(), r.order.end());
            out.a.reserve(out.a.size() + r.a.size());
            for (size_t i = 0; i < r.a.size(); ++i) out.a.push_back(r.a[i] + ((r.flags[i] & 3u) == 3u ? order_base : second));
            out.max_depth = std::max(out.max_depth, r.max_depth);
        } else {
            node(out, from, mid + 1, depth + 1);
            second = node(out, mid + 1, to, depth + 1);
        }
        out.a[self] = second;
        return self;
    }
    struct Aborted : std::runtime_error {
        Aborted() : std::runtime_error("the reference's BVH construction was abandoned (another subtree failed)") {}
    };
#ifndef TH_REF_PARALLEL_MIN
#define TH_REF_PARALLEL_MIN (1u << 15)
#endif
    static constexpr uint32_t kParallelMin = TH_REF_PARALLEL_MIN;  // primitives in a node whose two subtrees are worth a thread
#ifndef TH_REF_PARALLEL_DEPTH
#define TH_REF_PARALLEL_DEPTH 8
#endif
    static constexpr uint32_t kParallelDepth = TH_REF_PARALLEL_DEPTH;       // (at most 2^8 concurrent subtrees)

    const std::vector<HostAABB>& pb_;
    int max_leaf_;
    uint32_t depth_limit_;
    std::vector<uint32_t> info_;
    std::vector<float> cen_;
    std::atomic<bool> abort_{false};
};

}  // namespace th
