// oracle/orc_build.h — TEST INFRASTRUCTURE ONLY.
// Restates the construction half of accel/bvh.jl (:55-206) bug-for-bug and Trace.jl:128-137 partition!.
// Traversal results do not depend on this topology except for exact-t ties and slab-test edge cases, so the product
// builds its own BVH; this builder exists for the reference's BVH unit tests and for topology comparisons.
#pragma once
#include <algorithm>
#include <stdexcept>

#include "orc_shapes.h"

namespace orc {

struct BVHPrimitiveInfo {  // bvh.jl:3-15
    uint32_t primitive_number;
    Bounds3 bounds;
    V3 centroid;
    BVHPrimitiveInfo(uint32_t n, const Bounds3& b) : primitive_number(n), bounds(b), centroid(0.5f * b.p_min + 0.5f * b.p_max) {}
};
struct BuildNode {  // bvh.jl:17-36
    Bounds3 bounds;
    std::unique_ptr<BuildNode> children[2];
    uint8_t split_axis = 0;
    uint32_t offset = 0, n_primitives = 0;
};

struct RefBuilder {
    const std::vector<Primitive>& primitives;
    std::vector<BVHPrimitiveInfo> info;
    std::vector<Primitive> ordered;
    int total_nodes = 0;
    int max_node_primitives;

    RefBuilder(const std::vector<Primitive>& p, int mnp) : primitives(p), max_node_primitives(std::min(255, mnp)) {}

    int bucket_of(const Bounds3& cb, V3 centroid, int dim) const {  // :135-138, 166-170
        const int n_buckets = 12;
        int b = (int)std::floor((float)n_buckets * offset(cb, centroid)[dim - 1]) + 1;
        if (b == n_buckets + 1) b -= 1;
        return b;
    }

    // bvh.jl:87-185; from/to are 1-based inclusive.
    std::unique_ptr<BuildNode> init(int from, int to, int depth = 0) {
        total_nodes += 1;
        const int n_primitives = to - from + 1;
        Bounds3 bounds;
        for (int i = from; i <= to; ++i) bounds = bunion(bounds, info[i - 1].bounds);
        auto create_leaf = [&]() {
            auto node = std::make_unique<BuildNode>();
            node->offset = (uint32_t)ordered.size() + 1;
            for (int i = from; i <= to; ++i) ordered.push_back(primitives[info[i - 1].primitive_number - 1]);
            node->n_primitives = (uint32_t)n_primitives;
            node->bounds = bounds;
            return node;
        };
        if (n_primitives == 1) return create_leaf();
        Bounds3 cb;
        for (int i = from; i <= to; ++i) cb = bunion(cb, Bounds3(info[i - 1].centroid));
        const int dim = maximum_extent(cb);
        if (!is_valid(cb) || cb.p_min[dim - 1] == cb.p_max[dim - 1]) return create_leaf();
        int mid;
        if (n_primitives <= 2) {  // :121-127 partialsort!(view, 1): smallest centroid first, stable on ties
            mid = (from + to) / 2;
            if (info[to - 1].centroid[dim - 1] < info[from - 1].centroid[dim - 1]) std::swap(info[from - 1], info[to - 1]);
        } else {
            const int n_buckets = 12;
            Bounds3 bucket_bounds[12];
            for (auto& b : bucket_bounds) b = Bounds3(V3(0.0f));  // :130 buckets start as the point (0,0,0), not empty (A.6)
            for (int i = from; i <= to; ++i) {
                const int b = bucket_of(cb, info[i - 1].centroid, dim);
                bucket_bounds[b - 1] = bunion(bucket_bounds[b - 1], info[i - 1].bounds);
            }
            float costs[11];
            const float sa = surface_area(bounds);
            for (int i = 1; i <= n_buckets - 1; ++i) {  // :141-156: range LENGTHS instead of counts; right range stops at 11
                float s1 = 0, s2 = 0;
                {
                    Bounds3 u = bucket_bounds[0];
                    for (int b = 2; b <= i; ++b) u = bunion(u, bucket_bounds[b - 1]);
                    s1 = (float)i * surface_area(u);
                }
                const int len2 = (n_buckets - 1) - (i + 1) + 1;
                if (len2 > 0) {
                    Bounds3 u = bucket_bounds[i];
                    for (int b = i + 2; b <= n_buckets - 1; ++b) u = bunion(u, bucket_bounds[b - 1]);
                    s2 = (float)len2 * surface_area(u);
                }
                costs[i - 1] = 1.0f + (s1 + s2) / sa;
            }
            int min_cost_id = 1;  // argmin(costs) :158 — the first minimum; Julia's findmin treats a NaN as smaller than every number
            for (int i = 1; i <= n_buckets - 1; ++i) {
                if (costs[i - 1] != costs[i - 1]) {
                    min_cost_id = i;
                    break;
                }
                if (costs[i - 1] < costs[min_cost_id - 1]) min_cost_id = i;
            }
            const float leaf_cost = (float)n_primitives;
            if (!(n_primitives > max_node_primitives || costs[min_cost_id - 1] < leaf_cost)) return create_leaf();
            // partition! (Trace.jl:128-137): never tests the first element in place
            int left = from;
            for (int i = from; i <= to; ++i) {
                if (left != i && bucket_of(cb, info[i - 1].centroid, dim) <= min_cost_id) {
                    std::swap(info[i - 1], info[left - 1]);
                    left += 1;
                }
            }
            mid = left;  // mid == to gives the right child an empty range: a 0-primitive leaf with invalid bounds (A.6)
        }
        if (depth > 4096) throw std::runtime_error("reference BVH builder: runaway recursion (the reference would overflow its stack)");
        auto node = std::make_unique<BuildNode>();
        node->split_axis = (uint8_t)dim;
        node->children[0] = init(from, mid, depth + 1);
        node->children[1] = init(mid + 1, to, depth + 1);
        node->bounds = bunion(node->children[0]->bounds, node->children[1]->bounds);
        return node;
    }
    // bvh.jl:187-206
    uint32_t unroll(std::vector<LinearNode>& out, const BuildNode& node, uint32_t& offset) {
        const uint32_t l_offset = offset;
        offset += 1;
        LinearNode ln;
        ln.bounds = node.bounds;
        if (!node.children[0]) {
            ln.leaf = true;
            ln.primitives_offset = node.offset;
            ln.n_primitives = node.n_primitives;
            out[l_offset - 1] = ln;
            return l_offset + 1;
        }
        unroll(out, *node.children[0], offset);
        const uint32_t second = unroll(out, *node.children[1], offset) - 1;
        ln.leaf = false;
        ln.second_child_offset = second;
        ln.split_axis = node.split_axis;
        out[l_offset - 1] = ln;
        return l_offset + 1;
    }
};

// BVHAccel(primitives, max_node_primitives = 1)  bvh.jl:55-79
inline std::shared_ptr<BVHAccel> build_reference_bvh(const std::vector<Primitive>& prims, int max_node_primitives = 1) {
    auto bvh = std::make_shared<BVHAccel>();
    if (prims.empty()) return bvh;
    RefBuilder rb(prims, max_node_primitives);
    for (size_t i = 0; i < prims.size(); ++i) rb.info.emplace_back((uint32_t)(i + 1), world_bound(prims[i]));
    auto root = rb.init(1, (int)prims.size());
    bvh->nodes.resize((size_t)rb.total_nodes);
    uint32_t offset = 1;
    rb.unroll(bvh->nodes, *root, offset);
    bvh->primitives = std::move(rb.ordered);
    return bvh;
}

}  // namespace orc
