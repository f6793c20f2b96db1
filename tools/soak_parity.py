#!/usr/bin/env python
"""Soak test of the traversal shortcuts on the GPU: literal kernels (traversal 1: the reference's walk, loose box test, no pre-pass)
against k_trace3 / k_trace2 with every shortcut on (tight box clauses, largest-triangle pre-pass, clipped-sphere-free variants)
on random scenes — hits, barycentrics and occlusion must agree bit for bit.

    python tools/soak_parity.py --scenes 24 --rays 400000

Scenes: a closed or open box of large triangles around random small triangles (some needles, some axis-aligned), full and
clipped spheres, optionally a height field; rays: uniform in the bound, from far away, skimming, axis-parallel, through vertices.
Prints one line per scene and a total; exit code 1 on any mismatch."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g

T = g.load_package()


def rand_scene(rng, k):
    white = T.MatteMaterial(T.ConstantTexture(T.RGBSpectrum(0.8)), T.ConstantTexture(0.0))
    core = T.ShapeCore(T.translate([0, 0, 0]), False)
    prims = []
    closed = k % 3 != 2
    walls, _ = T.scenes.cornell_primitives(spheres=False)
    prims += walls if closed else walls[:4]
    n_tris = int(rng.integers(2000, 40000))
    c = rng.random((n_tris, 3), dtype=np.float32) + np.float32([0, 0, -3])
    size = (0.001 + 0.04 * rng.random((n_tris, 1), dtype=np.float32) ** 2).astype(np.float32)
    e1 = rng.standard_normal((n_tris, 3)).astype(np.float32)
    e2 = rng.standard_normal((n_tris, 3)).astype(np.float32)
    needle = rng.random(n_tris) < 0.2
    e2[needle] = (e1[needle] + np.float32(2e-3) * e2[needle]).astype(np.float32)
    ax = rng.integers(0, 9, n_tris)
    for a in range(3):
        e1[ax == a, a] = 0
        e2[ax == a, a] = 0
    verts = np.stack([c, c + size * e1, c + size * e2], axis=1).reshape(-1, 3).astype(np.float32)
    prims.append(T.create_mesh_primitives(core, np.arange(3 * n_tris, dtype=np.uint32) + 1, verts, None, white))
    if k % 2 == 0:
        hv, hi, hn = T.scenes.heightfield_mesh(int(rng.integers(20, 90)), seed=int(rng.integers(1, 1 << 30)))
        prims.append(T.create_mesh_primitives(core, hi, hv, hn, white))
    n_sph = int(rng.integers(0, 40)) if k % 4 < 2 else int(rng.integers(0, 7))  # half of the scenes have few enough spheres for the 8-wide kernel (<= 8)
    for _ in range(n_sph):
        p = rng.random(3) * 0.9 + [0.05, 0.05, -2.95]
        r = float(0.003 + 0.1 * rng.random() ** 2)
        if rng.random() < 0.3:  # clipped
            prims.append(T.GeometricPrimitive(T.Sphere(T.ShapeCore(T.translate([float(p[0]), float(p[1]), float(p[2])]), bool(rng.random() < 0.5)), r, -0.6 * r, 0.7 * r,
                                                       float(rng.uniform(60, 359))), white))
        else:
            prims.append(T.GeometricPrimitive(T.Sphere(T.ShapeCore(T.translate([float(p[0]), float(p[1]), float(p[2])]), False), r, 360.0), white))
    return T.Scene(T.scenes.cornell_lights(), T.BVHAccel(prims, 1)), verts.reshape(-1, 3, 3)


def make_rays(o, d):
    r = np.zeros((o.shape[0], 8), np.float32)
    r[:, 0:3] = o
    r[:, 3] = np.inf
    r[:, 4:7] = d
    return r


def rand_rays(rng, n, lo, hi, tri):
    parts = [T.scenes.incoherent_rays(n // 2, lo, hi, seed=int(rng.integers(1, 1 << 30)))]
    m = n // 8
    # towards the light from random points (what shadow rays look like), unnormalised
    o = (lo + (hi - lo) * rng.random((m, 3), dtype=np.float32)).astype(np.float32)
    parts.append(make_rays(o, (np.float32([0.5, 0.9, -2.5]) - o).astype(np.float32)))
    # far origins
    o = np.tile(np.float32([[0.5, 0.5, 50.0]]), (m, 1)) + rng.standard_normal((m, 3)).astype(np.float32)
    tgt = (lo + (hi - lo) * rng.random((m, 3), dtype=np.float32)).astype(np.float32)
    parts.append(make_rays(o, (tgt - o).astype(np.float32)))
    # skimming / axis-parallel
    o = (lo + (hi - lo) * rng.random((m, 3), dtype=np.float32)).astype(np.float32)
    d = rng.standard_normal((m, 3)).astype(np.float32)
    d[:, 1] *= np.float32(1e-3)
    z = rng.integers(0, 8, m)
    for a in range(3):
        d[z == a, a] = 0
    parts.append(make_rays(o, d))
    # through vertices / from vertices
    k = rng.integers(0, tri.shape[0], m)
    v = tri[k, rng.integers(0, 3, m)]
    o = (lo + (hi - lo) * rng.random((m, 3), dtype=np.float32)).astype(np.float32)
    parts.append(make_rays(o, (v - o).astype(np.float32)))
    rays = np.concatenate(parts)
    finite = rng.random(rays.shape[0]) < 0.3  # the kernel-level entry points take a t_max per ray: a third of the rays get a finite one
    rays[finite, 3] = (rng.random(int(finite.sum()), dtype=np.float32) * np.float32(1.5)).astype(np.float32)
    return rays


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=24)
    ap.add_argument("--rays", type=int, default=400000)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--wide", action="store_true", help="commit the scenes the way traversal 4 needs them (compose_spheres = 1) and compare the 8-wide kernel too")
    ap.add_argument("--frames", type=int, default=0, help="also render each scene at this resolution (4 spp, depth 6) with traversal 1 and 3 and compare the films")
    a = ap.parse_args()
    ctx = T.default_context()
    bad_total = 0
    rays_total = 0
    for k in range(a.scenes):
        rng = np.random.default_rng(a.seed * 1000 + k)
        scene, tri = rand_scene(rng, k)
        ctx.set_option("bvh_builder", (1 if ctx.has_option("bvh_builder", 1) else 0) if k % 5 == 4 else (3 if k % 5 == 3 else 0))  # host SAH, the same on the device, LBVH
        ctx.set_option("compose_spheres", 1 if a.wide else -1)
        flat = scene.flatten(ctx)
        bnd = flat.bvh()[0][0]
        rays = rand_rays(rng, a.rays, bnd[:3].copy(), bnd[3:].copy(), tri)
        ctx.set_option("traversal", 1)
        h1, o1 = flat.trace_closest(rays), flat.trace_any(rays)
        bad = 0
        for trav in ((3, 7, 2, 6, 4) if a.wide else (3, 7, 2, 6)):
            if not ctx.has_option("traversal", trav):  # (4 / 6 / 7: only in the EXPERIMENTS build of the library)
                continue
            h, o = flat.trace_closest(rays), flat.trace_any(rays)
            bad += int((h["prim"] != h1["prim"]).sum())
            for f in ("t", "b1", "b2"):
                bad += int((h[f].view(np.uint32) != h1[f].view(np.uint32)).sum())
            bad += int((o != o1).sum())
        ctx.set_option("traversal", 3)
        if a.frames:  # whole frames too: every bounce and shadow ray of a small render, literal walk against all shortcuts
            cam = T.scenes.cornell_camera(a.frames)
            films = []
            for trav in ((1, 3, 7, 6, 4) if a.wide else (1, 3, 7, 6)):
                if not ctx.has_option("traversal", trav):
                    continue
                films.append(T.PathIntegrator(cam, T.SeededSampler(4, seed=100 + k), 6).render(scene, ctx).copy())
            for other in films[1:]:
                fa, fb = films[0].view(np.uint32), other.view(np.uint32)
                nan = np.isnan(films[0]) & np.isnan(other)
                bad += int(((fa != fb) & ~nan).sum())
            ctx.set_option("traversal", 3)
        rays_total += rays.shape[0]
        bad_total += bad
        print(f"scene {k:3d}: {flat.bvh()[3].size:7d} primitives, {rays.shape[0]} rays, hit {float((h1['prim'] >= 0).mean()):.3f}, occluded {float(o1.mean()):.3f}, mismatches {bad}", flush=True)
        flat.free()
        scene._flat = None
    print(f"total: {a.scenes} scenes, {rays_total} rays x (closest + any) x (traversal {'3, 7, 2, 6, 4' if a.wide else '3, 7, 2, 6'}) against traversal 1: {bad_total} mismatches")
    sys.exit(1 if bad_total else 0)


if __name__ == "__main__":
    main()
