// th_wide8.h — host side of traversal 4: the BVH2 of accel/bvh.jl collapsed into 8-wide nodes with quantised child boxes.
//
// WHY THIS IS STILL THE REFERENCE'S WALK.  intersect!(bvh, ray) (accel/bvh.jl:212-258) visits the LEAVES whose box test passes, in
// depth-first order with the near child first (dir_is_neg[split_axis], :239-246); interior boxes only decide which subtrees are
// skipped.  For a ray without a zero direction component the slab products are monotonic in the box planes, so a leaf box that
// passes the test implies that every ancestor box passes it (th_trace2.h, k_any_occluders, has the argument) — the set and the ORDER of
// the leaves the walk tests is therefore fixed by the leaf boxes, the split axes and the ray alone.  Traversal 4 keeps exactly that:
// leaf boxes are tested with the reference's arithmetic (slab_test2, recomputed from the triangle's vertices) at the moment the
// leaf is reached, with the t_max of that moment; the children of a wide node are visited in the order the binary walk would
// reach them (one precomputed permutation per direction octant, from the split axes of the collapsed interior nodes); and the
// interior boxes are replaced by CONSERVATIVE ones — 8-bit planes on a per-node grid, rounded outwards, tested on a box grown by
// twice the margin of the leaf test — which can only skip subtrees none of whose leaves would pass.  Rays the argument does not
// cover (a zero / denormal direction component, non-finite input, origins absurdly far away) are handed to k_trace3, and
// spheres (whose fp32 quadratic accepts rays far outside their box, and can RAISE t_max, A.18) never enter the wide tree: a scene
// with spheres is committed as a chain root -> {sphere 1, {sphere 2, ... {sphere k, the triangles' subtree}}} (trhip_scene_commit);
// the kernel tests the sphere leaves the walk reaches before the subtree, walks the subtree, then tests the ones it reaches after.
//
// Node (128-byte aligned, 104 bytes used; one L2 line):
//   dw 0-2   p = lower corner of the node's box (Float32)          dw 3      ex | ey << 8 | ez << 16 | ni << 24 | n << 28
//   dw 4     index of the first interior child (children ni)        dw 5      index of the first leaf child's triangle in `tris`
//   dw 6-17  qlo_x[8] qlo_y[8] qlo_z[8] qhi_x[8] qhi_y[8] qhi_z[8]  (bytes; plane = p + q * 2^(e - 127))
//   dw 18-25 one word per direction octant (negx | negy << 1 | negz << 2): 3 bits per slot = the slot's position in the visiting order
// Slots 0..ni-1 are interior children (wide nodes child_base + slot), slots ni..n-1 leaves (triangle tri_base + slot - ni).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "th_bvh.h"

namespace th {

constexpr int kW8NodeDwords = 32;
constexpr int kW8MaxDepth = 32;      // wide levels a stack may have to hold; deeper trees fall back to k_trace3
constexpr int kW8MaxSpheres = 8;     // spheres a scene may have beside its triangles for the 8-wide kernel to take it (more: k_trace3)
constexpr float kW8MaxCoord = 16777216.0f;  // 2^24: scenes beyond use k_trace3 (keeps every product of the quantised test finite)

struct Wide8Host {
    std::vector<uint32_t> nodes;  // kW8NodeDwords per node
    std::vector<float> tris;      // 12 floats per triangle: v0 | slot bits, v1 | meta bits, v2 | 0
    uint32_t depth = 0;
    bool ok = false;
};

// get_tri(slot, v[9], meta): the triangle in ordered primitive slot `slot`; returns false when the slot does not hold a triangle.
template <class GetTri>
Wide8Host build_wide8(const FlatBVH& bvh, uint32_t root, GetTri&& get_tri) {
    Wide8Host w;
    const uint32_t n_flat = (uint32_t)bvh.a.size();
    if (root >= n_flat || (bvh.flags[root] & 3u) == 3u) return w;  // the subtree must have a hierarchy
    auto is_leaf = [&](uint32_t f) { return (bvh.flags[f] & 3u) == 3u; };
    auto box = [&](uint32_t f) { return &bvh.bounds[6 * (size_t)f]; };
    auto area = [&](uint32_t f) {
        const float* b = box(f);
        const double dx = (double)b[3] - b[0], dy = (double)b[4] - b[1], dz = (double)b[5] - b[2];
        return dx * dy + dx * dz + dy * dz;
    };
    for (int a = 0; a < 6; ++a)
        if (!(std::fabs(box(root)[a]) < kW8MaxCoord)) return w;
    struct Job {
        uint32_t flat, wide, depth;
    };
    std::vector<Job> todo;
    w.nodes.assign(kW8NodeDwords, 0u);
    todo.push_back({root, 0u, 1u});
    struct TNode {
        uint32_t flat;
        int left, right;  // treelet children; -1: this treelet node is a child of the wide node
    };
    bool ok = true;
    while (!todo.empty() && ok) {
        const Job job = todo.back();
        todo.pop_back();
        w.depth = std::max(w.depth, job.depth);
        if (job.depth > (uint32_t)kW8MaxDepth) {
            ok = false;
            break;
        }
        // ---- collapse: open the interior child of largest surface area until 8 children ----
        TNode tn[15];
        int n_tn = 3, n_children = 2;
        tn[0] = {job.flat, 1, 2};
        tn[1] = {job.flat + 1, -1, -1};
        tn[2] = {bvh.a[job.flat], -1, -1};
        while (n_children < 8) {
            int best = -1;
            double best_area = -1.0;
            for (int i = 0; i < n_tn; ++i)
                if (tn[i].left < 0 && !is_leaf(tn[i].flat)) {
                    const double ar = area(tn[i].flat);
                    if (ar > best_area) {
                        best_area = ar;
                        best = i;
                    }
                }
            if (best < 0) break;
            const uint32_t f = tn[best].flat;
            tn[best].left = n_tn;
            tn[best].right = n_tn + 1;
            tn[n_tn++] = {f + 1, -1, -1};
            tn[n_tn++] = {bvh.a[f], -1, -1};
            n_children++;
        }
        // ---- slots: interior children first, then leaves, each group in the treelet's left-to-right order ----
        int seq[8], n_seq = 0;  // treelet nodes that are children, left to right
        {
            int stack[16], sp = 0;
            stack[sp++] = 0;
            while (sp) {
                const int i = stack[--sp];
                if (tn[i].left < 0) {
                    seq[n_seq++] = i;
                } else {
                    stack[sp++] = tn[i].right;
                    stack[sp++] = tn[i].left;
                }
            }
        }
        int slot_of[15];
        uint32_t child_flat[8];
        int ni = 0, n = 0;
        for (int k = 0; k < n_seq; ++k)
            if (!is_leaf(tn[seq[k]].flat)) {
                slot_of[seq[k]] = n;
                child_flat[n++] = tn[seq[k]].flat;
                ni++;
            }
        for (int k = 0; k < n_seq; ++k)
            if (is_leaf(tn[seq[k]].flat)) {
                slot_of[seq[k]] = n;
                child_flat[n++] = tn[seq[k]].flat;
            }
        // ---- the binary walk's visiting order per direction octant (bvh.jl:239-246: dir_is_neg[split_axis] -> second child first) ----
        uint32_t iperm[8];
        for (uint32_t oct = 0; oct < 8; ++oct) {
            uint32_t word = 0;
            int pos = 0;
            int stack[16], sp = 0;
            stack[sp++] = 0;
            while (sp) {
                const int i = stack[--sp];
                if (tn[i].left < 0) {
                    word |= (uint32_t)pos++ << (3 * slot_of[i]);
                } else {
                    const uint32_t axis = bvh.flags[tn[i].flat] & 3u;
                    const bool neg = (oct >> axis) & 1u;
                    const int first = neg ? tn[i].right : tn[i].left, second = neg ? tn[i].left : tn[i].right;
                    stack[sp++] = second;
                    stack[sp++] = first;
                }
            }
            for (int s = n; s < 8; ++s) word |= (uint32_t)pos++ << (3 * s);  // empty slots: positions behind the children (masked off by n)
            iperm[oct] = word;
        }
        // ---- quantisation grid: p = lower corner, 2^e per axis with 255 steps covering the node ----
        const float* nb = box(job.flat);
        uint32_t eb[3];
        double scale[3];
        for (int a = 0; a < 3; ++a) {
            const double ext = (double)nb[3 + a] - (double)nb[a];
            int e = -126;
            if (ext > 0.0) {
                e = std::max(-126, std::ilogb(ext / 255.0));
                while (std::ldexp(255.0, e) < ext) ++e;
            }
            if (e > 40) ok = false;
            eb[a] = (uint32_t)(e + 127);
            scale[a] = std::ldexp(1.0, e);
        }
        uint8_t qlo[3][8], qhi[3][8];
        for (int s = 0; s < 8; ++s)
            for (int a = 0; a < 3; ++a) {
                if (s >= n) {  // empty slot: inverted box (and masked off by n in the kernel)
                    qlo[a][s] = 255;
                    qhi[a][s] = 0;
                    continue;
                }
                const float* cb = box(child_flat[s]);
                const double lo = ((double)cb[a] - (double)nb[a]) / scale[a], hi = ((double)cb[3 + a] - (double)nb[a]) / scale[a];
                if (!(lo >= 0.0) || !(hi >= lo) || !(hi <= 255.0)) {  // a child box outside its parent's, or NaN: not a tree this kernel may walk
                    ok = false;
                    qlo[a][s] = 0;
                    qhi[a][s] = 255;
                    continue;
                }
                double ql = std::floor(lo), qh = std::ceil(hi);
                while (ql > 0.0 && (double)nb[a] + ql * scale[a] > (double)cb[a]) ql -= 1.0;        // outward, whatever the divisions rounded to
                while (qh < 255.0 && (double)nb[a] + qh * scale[a] < (double)cb[3 + a]) qh += 1.0;
                if ((double)nb[a] + qh * scale[a] < (double)cb[3 + a]) ok = false;
                qlo[a][s] = (uint8_t)ql;
                qhi[a][s] = (uint8_t)qh;
            }
        // ---- children: interior ones get consecutive wide nodes, leaves consecutive triangles ----
        const uint32_t child_base = (uint32_t)(w.nodes.size() / kW8NodeDwords);
        w.nodes.resize(w.nodes.size() + (size_t)ni * kW8NodeDwords, 0u);
        const uint32_t tri_base = (uint32_t)(w.tris.size() / 12);
        for (int s = ni; s < n && ok; ++s) {
            const uint32_t lf = child_flat[s];
            if ((bvh.flags[lf] >> 2) != 1u) {  // this kernel's leaves hold one primitive (every scene of the reference: BVHAccel(prims, 1))
                ok = false;
                break;
            }
            const uint32_t slot = bvh.a[lf];
            float v[9];
            uint32_t meta = 0;
            if (!get_tri(slot, v, meta)) {
                ok = false;
                break;
            }
            // the leaf's box must be the triangle's own bound (world_bound(::Triangle), triangle_mesh.jl:97): the kernel recomputes it from the vertices
            const float* lb = box(lf);
            for (int a = 0; a < 3; ++a) {
                const float mn = std::fmin(std::fmin(v[a], v[3 + a]), v[6 + a]), mx = std::fmax(std::fmax(v[a], v[3 + a]), v[6 + a]);
                if (!(mn == lb[a]) || !(mx == lb[3 + a])) ok = false;
            }
            float rec[12] = {v[0], v[1], v[2], 0.0f, v[3], v[4], v[5], 0.0f, v[6], v[7], v[8], 0.0f};
            std::memcpy(&rec[3], &slot, 4);
            std::memcpy(&rec[7], &meta, 4);
            w.tris.insert(w.tris.end(), rec, rec + 12);
        }
        if (tri_base + (uint32_t)(n - ni) >= (1u << 24) || child_base + (uint32_t)ni >= (1u << 30)) ok = false;
        uint32_t* d = &w.nodes[(size_t)job.wide * kW8NodeDwords];
        std::memcpy(&d[0], &nb[0], 12);
        d[3] = eb[0] | (eb[1] << 8) | (eb[2] << 16) | ((uint32_t)ni << 24) | ((uint32_t)n << 28);
        d[4] = child_base;
        d[5] = tri_base;
        for (int a = 0; a < 3; ++a) {
            std::memcpy(&d[6 + 2 * a], qlo[a], 8);
            std::memcpy(&d[12 + 2 * a], qhi[a], 8);
        }
        for (int o = 0; o < 8; ++o) d[18 + o] = iperm[o];
        for (int s = 0; s < ni; ++s) todo.push_back({child_flat[s], child_base + (uint32_t)s, job.depth + 1});
    }
    w.ok = ok;
    return w;
}

}  // namespace th
