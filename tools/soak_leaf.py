#!/usr/bin/env python
"""Soak test of the ONE-LEAF certified walk (k_trace_leaf_c / k_any_leaf_c, csrc/th_trace3c.h): random scenes of at most 16 primitives — the Cornell walls or a few of them,
random triangles (needles, axis-aligned, pairs sharing an edge, duplicates), full and clipped spheres, nested and overlapping —, committed with default options (two trees,
the accelerator a single leaf), the same rays through (a) the certified walk + the reference-order walk of what it hands back, (b) option hybrid = 0 (the canonical tree alone),
(c) the literal accel/bvh.jl loop (traversal 1).  Hits, barycentrics, occlusion and small frames must agree bit for bit.

    python tools/soak_leaf.py --scenes 200 --rays 60000 --frames 24"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as g

T = g.load_package()
import soak_parity as sp  # noqa: E402
import soak_hybrid as sh  # noqa: E402


def tiny_scene(rng, k):
    white = T.MatteMaterial(T.ConstantTexture(T.RGBSpectrum(0.8)), T.ConstantTexture(0.0))
    mirror = T.MirrorMaterial(T.ConstantTexture(T.RGBSpectrum(0.9)))
    glass = T.GlassMaterial(T.ConstantTexture(T.RGBSpectrum(1.0)), T.ConstantTexture(T.RGBSpectrum(1.0)), T.ConstantTexture(0.0), T.ConstantTexture(0.0), T.ConstantTexture(1.5), True)
    core = T.ShapeCore(T.translate([0, 0, 0]), False)
    walls, _ = T.scenes.cornell_primitives(spheres=False)
    prims = list(walls[: int(rng.integers(0, 6)) * 2])
    budget = 16 - len(prims)
    n_sph = int(rng.integers(0, min(5, budget) + 1))
    n_tri = int(rng.integers(1 if not prims else 0, budget - n_sph + 1))
    tris = []
    for j in range(n_tri):
        c = rng.random(3) + [0, 0, -3]
        size = 0.02 + 0.6 * rng.random() ** 2
        e1, e2 = rng.standard_normal(3), rng.standard_normal(3)
        kind = int(rng.integers(0, 6))
        if kind == 0:
            e2 = e1 + 2e-3 * e2  # a needle
        if kind == 1:
            a = int(rng.integers(0, 3))
            e1[a] = e2[a] = 0.0  # axis-aligned (a flat box)
        v = np.stack([c, c + size * e1, c + size * e2]).astype(np.float32)
        if kind == 2 and tris:
            v[:2] = tris[-1][1:][::-1]  # shares an edge with the one before
        if kind == 3 and tris:
            v = tris[-1].copy()  # a duplicate: every hit a tie
        tris.append(v)
    if tris:
        verts = np.concatenate(tris).astype(np.float32)
        prims.append(T.create_mesh_primitives(core, np.arange(verts.shape[0], dtype=np.uint32) + 1, verts, None, white))
    else:
        verts = np.zeros((0, 3), np.float32)
    for j in range(n_sph):
        p = rng.random(3) * 0.8 + [0.1, 0.1, -2.9]
        r = float(0.03 + 0.3 * rng.random() ** 2)
        mat = (white, mirror, glass)[int(rng.integers(0, 3))]
        tr = T.translate([float(p[0]), float(p[1]), float(p[2])])
        if rng.random() < 0.25:
            prims.append(T.GeometricPrimitive(T.Sphere(T.ShapeCore(tr, bool(rng.random() < 0.5)), r, -0.6 * r, 0.7 * r, float(rng.uniform(60, 359))), mat))
        else:
            prims.append(T.GeometricPrimitive(T.Sphere(T.ShapeCore(tr, False), r, 360.0), mat))
        if rng.random() < 0.3 and len(prims) < 16:  # one inside / overlapping the other
            prims.append(T.GeometricPrimitive(T.Sphere(T.ShapeCore(T.translate([float(p[0] + 0.3 * r), float(p[1]), float(p[2])]), False), 0.6 * r, 360.0), mat))
    return T.Scene(T.scenes.cornell_lights(), T.BVHAccel(prims, 1)), verts.reshape(-1, 3, 3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=200)
    ap.add_argument("--rays", type=int, default=60000)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--frames", type=int, default=24)
    a = ap.parse_args()
    ctx = T.default_context()
    bad_total = rays_total = one_leaf = fb_total = cl_total = 0
    for k in range(a.scenes):
        rng = np.random.default_rng(a.seed * 104729 + k)
        scene, tri = tiny_scene(rng, k)
        ctx.set_option("bvh_builder", -1)
        ctx.set_option("hybrid", 1)
        ctx.set_option("traversal", 3)
        flat = scene.flatten(ctx)
        mode, acc_nodes, _ = flat.bvh_mode()
        one_leaf += mode == 2 and acc_nodes == 1
        bnd = flat.bvh()[0][0]
        lo, hi = bnd[:3].copy() - 0.2, bnd[3:].copy() + 0.2
        tri_for_rays = tri if tri.shape[0] else np.zeros((1, 3, 3), np.float32) + np.float32([0.5, 0.5, -2.5])
        rays = np.concatenate([sp.rand_rays(rng, a.rays, lo, hi, tri_for_rays), sh.sphere_rays(rng, scene.aggregate.primitives, 1500)])
        h, o = flat.trace_closest(rays), flat.trace_any(rays)
        n_rays, n_fb = flat.last_fallback()
        bad = 0
        for opt, val in (("hybrid", 0), ("traversal", 1)):
            ctx.set_option(opt, val)
            h1, o1 = flat.trace_closest(rays), flat.trace_any(rays)
            bad += int((h["prim"] != h1["prim"]).sum())
            for f in ("t", "b1", "b2"):
                bad += int((h[f].view(np.uint32) != h1[f].view(np.uint32)).sum())
            bad += int((o != o1).sum())
        ctx.set_option("traversal", 3)
        ctx.set_option("hybrid", 1)
        fb = cl = 0
        if a.frames:
            cam = T.scenes.cornell_camera(a.frames)
            outs = []
            for hyb in (1, 0):
                ctx.set_option("hybrid", hyb)
                it = T.PathIntegrator(cam, T.SeededSampler(4, seed=100 + k), 6)
                film = it.render(scene, ctx).copy()
                outs.append((film, it.sample_radiance(scene).copy()))
                if hyb:
                    fb, cl = int(it.stats.fallback_rays), int(it.stats.closest_rays)
            for x, y in zip(outs[0], outs[1]):
                nan = np.isnan(x) & np.isnan(y)
                bad += int(((x.view(np.uint32) != y.view(np.uint32)) & ~nan).sum())
            ctx.set_option("hybrid", 1)
        rays_total += rays.shape[0]
        bad_total += bad
        fb_total += fb
        cl_total += cl
        if bad or k % 20 == 0:
            print(f"scene {k:3d}: mode {mode}, accelerator nodes {acc_nodes}, {flat.bvh()[3].size:2d} primitives, {rays.shape[0]} rays, hit {float((h['prim'] >= 0).mean()):.3f}, handed back {n_fb}/{n_rays}, "
                  f"frame {fb}/{cl}, mismatches {bad}", flush=True)
        flat.free()
        scene._flat = None
    print(f"total: {a.scenes} scenes ({one_leaf} with a one-leaf accelerator), {rays_total} rays x (closest + any) + {a.frames}^2 frames ({fb_total} of {cl_total} closest-hit rays handed back): "
          f"certified one-leaf walk against the canonical tree alone (k_trace3 and the literal loop): {bad_total} mismatches")
    sys.exit(1 if bad_total else 0)


if __name__ == "__main__":
    main()
