// th_bvh.h — host-side BVH2 construction for BVHAccel(primitives, max_node_primitives) (accel/bvh.jl:55-206).
// NOT a restatement of the reference builder (whose SAH has several quirks): traversal results do not
// depend on the topology except for exact-t ties, so this is a plain binned-SAH builder (16 bins, all three axes,
// median fallback) that emits nodes directly in the reference's depth-first layout (first child = i + 1).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <future>
#include <thread>
#include <vector>

namespace th {

struct HostAABB {
    float mn[3], mx[3];
    void reset() {
        for (int a = 0; a < 3; ++a) {
            mn[a] = INFINITY;
            mx[a] = -INFINITY;
        }
    }
    void grow(const HostAABB& b) {
        for (int a = 0; a < 3; ++a) {
            mn[a] = std::fmin(mn[a], b.mn[a]);
            mx[a] = std::fmax(mx[a], b.mx[a]);
        }
    }
    void grow_point(const float* p) {
        for (int a = 0; a < 3; ++a) {
            mn[a] = std::fmin(mn[a], p[a]);
            mx[a] = std::fmax(mx[a], p[a]);
        }
    }
    float half_area() const {
        const float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        return dx * dy + dx * dz + dy * dz;
    }
};

struct FlatBVH {
    std::vector<float> bounds;     // n_nodes * 6
    std::vector<uint32_t> a;       // leaf: first ordered slot | interior: second child
    std::vector<uint32_t> flags;   // leaf: n << 2 | 3 | interior: axis
    std::vector<uint32_t> order;   // ordered slot -> caller primitive index
    uint32_t max_depth = 0;
};

class BVHBuilder {
   public:
    // split_coincident: primitives whose centroids coincide (the two triangles of an axis-aligned quad) are still split down to the
    // leaf-size hint — one primitive per leaf for BVHAccel(prims, 1), what the 8-wide view needs (th_wide8.h); otherwise they share
    // a leaf like in the reference (bvh.jl:113-118), which the binary kernels walk faster (fewer nodes)
    BVHBuilder(const std::vector<HostAABB>& prim_bounds, int max_node_prims, uint32_t tiny_scene = 16, bool split_coincident = false)
        : pb_(prim_bounds), max_leaf_(std::max(1, std::min(255, max_node_prims))), tiny_(std::min(255u, tiny_scene)), split_coincident_(split_coincident) {}

    FlatBVH build() {
        const uint32_t n = (uint32_t)pb_.size();
        idx_.resize(n);
        cen_.resize((size_t)n * 3);
        for (uint32_t i = 0; i < n; ++i) {
            idx_[i] = i;
            for (int a = 0; a < 3; ++a) cen_[3 * (size_t)i + a] = 0.5f * pb_[i].mn[a] + 0.5f * pb_[i].mx[a];  // bvh.jl:12
        }
        out_.bounds.reserve((size_t)n * 12);
        out_.a.reserve((size_t)n * 2);
        out_.flags.reserve((size_t)n * 2);
        out_.order.reserve(n);
        // the top levels fan out into tasks (each subtree works on its own slice of idx_ and writes its own fragment, spliced in
        // depth-first order afterwards): the tree is the one the serial recursion builds, about cores / 3 times sooner
        int par = 0;
        for (unsigned t = std::max(1u, std::thread::hardware_concurrency()); t > 1 && par < 6; t >>= 1) ++par;
        if (n) recurse(0, n, 1, out_, n >= (1u << 16) ? par : 0);
        return std::move(out_);
    }

   private:
#ifndef TH_BVH_BINS
#define TH_BVH_BINS 16
#endif
    static constexpr int kBins = TH_BVH_BINS;
    const std::vector<HostAABB>& pb_;
    int max_leaf_;
    uint32_t tiny_;
    bool split_coincident_;
    std::vector<uint32_t> idx_;
    std::vector<float> cen_;
    FlatBVH out_;

    static uint32_t emit_node(FlatBVH& out, const HostAABB& b) {
        const uint32_t id = (uint32_t)out.a.size();
        out.bounds.insert(out.bounds.end(), {b.mn[0], b.mn[1], b.mn[2], b.mx[0], b.mx[1], b.mx[2]});
        out.a.push_back(0);
        out.flags.push_back(0);
        return id;
    }
    void make_leaf(FlatBVH& out, uint32_t node, uint32_t lo, uint32_t hi) {
        out.a[node] = (uint32_t)out.order.size();
        out.flags[node] = ((hi - lo) << 2) | 3u;
        for (uint32_t i = lo; i < hi; ++i) out.order.push_back(idx_[i]);
    }
    // append a subtree built on its own (node and slot indices relative to the fragment) behind what `out` holds
    static void splice(FlatBVH& out, const FlatBVH& frag) {
        const uint32_t noff = (uint32_t)out.a.size(), ooff = (uint32_t)out.order.size();
        out.bounds.insert(out.bounds.end(), frag.bounds.begin(), frag.bounds.end());
        out.flags.insert(out.flags.end(), frag.flags.begin(), frag.flags.end());
        out.order.insert(out.order.end(), frag.order.begin(), frag.order.end());
        out.a.reserve(out.a.size() + frag.a.size());
        for (size_t i = 0; i < frag.a.size(); ++i) out.a.push_back(frag.a[i] + ((frag.flags[i] & 3u) == 3u ? ooff : noff));
        out.max_depth = std::max(out.max_depth, frag.max_depth);
    }
    // `out` receives the subtree over idx_[lo, hi) in depth-first order; par > 0: the two children are built concurrently
    void recurse(uint32_t lo, uint32_t hi, uint32_t depth, FlatBVH& out, int par) {
        out.max_depth = std::max(out.max_depth, depth);
        HostAABB b, cb;
        b.reset();
        cb.reset();
        for (uint32_t i = lo; i < hi; ++i) {
            b.grow(pb_[idx_[i]]);
            cb.grow_point(&cen_[3 * (size_t)idx_[i]]);
        }
        const uint32_t node = emit_node(out, b);
        const uint32_t n = hi - lo;
        // A hierarchy over a handful of primitives only adds divergence: a wavefront that walks one leaf tests the same
        // primitive in every lane.  Scenes of ≤ tiny_ primitives (option tiny_scene_prims, default 16) become a single leaf (the leaf-size hint is a hint).
        if (n == 1 || (depth == 1 && n <= tiny_)) {
            make_leaf(out, node, lo, hi);
            return;
        }
        // choose the split: binned SAH over the three axes
        int best_axis = -1, best_bin = -1;
        float best_cost = INFINITY;
        for (int ax = 0; ax < 3; ++ax) {
            const float c0 = cb.mn[ax], c1 = cb.mx[ax];
            if (!(c1 > c0)) continue;
            HostAABB bb[kBins];
            uint32_t cnt[kBins];
            for (int k = 0; k < kBins; ++k) {
                bb[k].reset();
                cnt[k] = 0;
            }
            const float scale = (float)kBins / (c1 - c0);
            for (uint32_t i = lo; i < hi; ++i) {
                int k = (int)((cen_[3 * (size_t)idx_[i] + ax] - c0) * scale);
                k = std::min(kBins - 1, std::max(0, k));
                bb[k].grow(pb_[idx_[i]]);
                cnt[k]++;
            }
            float right_area[kBins];
            uint32_t right_cnt[kBins];
            HostAABB acc;
            acc.reset();
            uint32_t c = 0;
            for (int k = kBins - 1; k > 0; --k) {
                acc.grow(bb[k]);
                c += cnt[k];
                right_area[k] = acc.half_area();
                right_cnt[k] = c;
            }
            acc.reset();
            c = 0;
            for (int k = 0; k < kBins - 1; ++k) {
                acc.grow(bb[k]);
                c += cnt[k];
                if (c == 0 || right_cnt[k + 1] == 0) continue;
                const float cost = acc.half_area() * (float)c + right_area[k + 1] * (float)right_cnt[k + 1];
                if (cost < best_cost) {
                    best_cost = cost;
                    best_axis = ax;
                    best_bin = k;
                }
            }
        }
        if (best_axis < 0) {  // all centroids coincide: the reference makes a leaf (bvh.jl:113-118); so does this builder up to the leaf-size hint,
                              // beyond it the set is halved by index (same boxes on both sides) so that the hint holds: with
                              // max_node_primitives = 1 every leaf holds ONE primitive, what the 8-wide view needs (th_wide8.h)
            if (!split_coincident_ || (int)n <= max_leaf_ || depth >= 60 || n > 255u) {
                make_leaf(out, node, lo, hi);
                return;
            }
            const uint32_t mid = lo + n / 2;
            out.flags[node] = 0u;
            recurse(lo, mid, depth + 1, out, 0);
            out.a[node] = (uint32_t)out.a.size();
            recurse(mid, hi, depth + 1, out, 0);
            return;
        }
        if ((int)n <= max_leaf_) {
            const float leaf_cost = (float)n * b.half_area();
            if (best_cost + 0.125f * b.half_area() >= leaf_cost) {
                make_leaf(out, node, lo, hi);
                return;
            }
        }
        const float c0 = cb.mn[best_axis], c1 = cb.mx[best_axis];
        const float scale = (float)kBins / (c1 - c0);
        uint32_t* first = idx_.data() + lo;
        uint32_t* last = idx_.data() + hi;
        uint32_t* midp = std::partition(first, last, [&](uint32_t id) {
            int k = (int)((cen_[3 * (size_t)id + best_axis] - c0) * scale);
            k = std::min(kBins - 1, std::max(0, k));
            return k <= best_bin;
        });
        uint32_t mid = (uint32_t)(midp - idx_.data());
        // Traversal keeps the reference's 64-entry stack (bvh.jl:222): past depth 40 split at the median so the tree
        // cannot get deeper than 40 + log2(n) levels.
        if (mid == lo || mid == hi || depth >= 40) {
            mid = lo + n / 2;
            std::nth_element(first, idx_.data() + mid, last,
                             [&](uint32_t x, uint32_t y) { return cen_[3 * (size_t)x + best_axis] < cen_[3 * (size_t)y + best_axis]; });
        }
        out.flags[node] = (uint32_t)best_axis;
        if (par > 0 && n >= (1u << 14)) {
            FlatBVH left, right;
            auto task = std::async(std::launch::async, [&] { recurse(lo, mid, depth + 1, left, par - 1); });
            recurse(mid, hi, depth + 1, right, par - 1);
            task.get();
            splice(out, left);
            out.a[node] = (uint32_t)out.a.size();  // second child follows the whole first subtree
            splice(out, right);
            return;
        }
        recurse(lo, mid, depth + 1, out, 0);
        out.a[node] = (uint32_t)out.a.size();  // second child follows the whole first subtree
        recurse(mid, hi, depth + 1, out, 0);
    }
};

}  // namespace th
