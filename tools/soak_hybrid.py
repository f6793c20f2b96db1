#!/usr/bin/env python
"""Soak test of the HYBRID mode's certificate (csrc/th_trace3c.h) on the GPU: random scenes committed with default options (canonical tree = the reference's own
construction, accelerator = the library's SAH tree), the same rays traced (a) through the certified walk on the accelerator + the fallback on the canonical tree,
(b) with option "hybrid" = 0: k_trace3 on the canonical tree alone, (c) with the literal accel/bvh.jl loop (traversal 1) on the canonical tree.  Hits, barycentrics and
occlusion must agree bit for bit; so must small frames (PathIntegrator, every bounce and shadow ray).

    python tools/soak_hybrid.py --scenes 24 --rays 400000 --frames 48

Scenes (tools/soak_parity.py's generator): a closed or open box of large triangles around 2-40 k random small triangles (needles, axis-aligned ones), full and clipped
spheres (0-39: more than 32 leave the scene without an accelerator, which is reported), optionally a height field; rays: uniform in the bound, towards the light, from
far away, skimming / axis-parallel, through vertices, from inside the spheres, a third with a finite t_max.  Exit code 1 on any mismatch."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as g

T = g.load_package()
import soak_parity as sp  # noqa: E402  (scene / ray generators)


def sphere_rays(rng, scene_prims, n):
    """Rays that start inside / on the scene's spheres."""
    out = []
    spheres = [p.shape for p in scene_prims if isinstance(getattr(p, "shape", None), T.Sphere)]
    for s in spheres[:16]:  # (the first ten have order bits in the primitive records: inside rays of the others go to the reference-order walk)
        c = s.core.object_to_world.point([0, 0, 0])
        u = rng.normal(size=(n, 3))
        u /= np.linalg.norm(u, axis=1, keepdims=True)
        o = (np.asarray(c, np.float64) + u * (float(s.radius) * rng.uniform(0.0, 1.02, (n, 1)))).astype(np.float32)
        out.append(sp.make_rays(o, rng.normal(size=(n, 3)).astype(np.float32)))
    return np.concatenate(out) if out else np.zeros((0, 8), np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=24)
    ap.add_argument("--rays", type=int, default=400000)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--frames", type=int, default=0, help="also render each scene at this resolution (4 spp, depth 6) in the three modes and compare films and per-sample radiance")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="library option set once at the start (e.g. leaf_queue=1)")
    a = ap.parse_args()
    ctx = T.default_context()
    for kv in a.opt:
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    bad_total = rays_total = fb_total = closest_total = 0
    hybrid_scenes = 0
    for k in range(a.scenes):
        rng = np.random.default_rng(a.seed * 7919 + k)
        scene, tri = sp.rand_scene(rng, k if k % 4 >= 2 or k % 8 < 4 else k + 2)  # (most scenes with <= 8 spheres)
        ctx.set_option("bvh_builder", -1)
        ctx.set_option("hybrid", 1)
        ctx.set_option("traversal", 3)
        try:
            flat = scene.flatten(ctx)
        except T.TraceHipError as e:  # the reference's construction can fail where the reference itself would (depth > 64): nothing to compare
            print(f"scene {k:3d}: commit refused ({e})", flush=True)
            scene._flat = None
            continue
        mode = flat.bvh_mode()[0]
        hybrid_scenes += mode == 2
        bnd = flat.bvh()[0][0]
        rays = np.concatenate([sp.rand_rays(rng, a.rays, bnd[:3].copy(), bnd[3:].copy(), tri), sphere_rays(rng, scene.aggregate.primitives, 2000)])
        h, o = flat.trace_closest(rays), flat.trace_any(rays)
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
            for hyb, trav in ((1, 3), (0, 3), (0, 1)):
                ctx.set_option("hybrid", hyb)
                ctx.set_option("traversal", trav)
                it = T.PathIntegrator(cam, T.SeededSampler(4, seed=100 + k), 6)
                film = it.render(scene, ctx).copy()
                outs.append((film, it.sample_radiance(scene).copy()))
                if hyb:
                    fb, cl = int(it.stats.fallback_rays), int(it.stats.closest_rays)
            for film, L in outs[1:]:
                for x, y in ((outs[0][0], film), (outs[0][1], L)):
                    nan = np.isnan(x) & np.isnan(y)
                    bad += int(((x.view(np.uint32) != y.view(np.uint32)) & ~nan).sum())
            ctx.set_option("traversal", 3)
            ctx.set_option("hybrid", 1)
        rays_total += rays.shape[0]
        bad_total += bad
        fb_total += fb
        closest_total += cl
        print(f"scene {k:3d}: mode {mode}, {flat.bvh()[3].size:7d} primitives, {rays.shape[0]} rays, hit {float((h['prim'] >= 0).mean()):.3f}, occluded {float(o.mean()):.3f}, "
              f"frame fallback {fb}/{cl}, mismatches {bad}", flush=True)
        flat.free()
        scene._flat = None
    print(f"total: {a.scenes} scenes ({hybrid_scenes} with an accelerator), {rays_total} rays x (closest + any), hybrid walk against the canonical tree alone (k_trace3 and the literal loop)"
          f"{', frames: ' + str(fb_total) + ' of ' + str(closest_total) + ' closest-hit rays fell back' if a.frames else ''}: {bad_total} mismatches")
    sys.exit(1 if bad_total else 0)


if __name__ == "__main__":
    main()
