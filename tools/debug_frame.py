#!/usr/bin/env python
"""Diagnostic: per-sample radiance of a small frame, GPU vs oracle, over traversal kernels / depths / BVH topologies."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft
T = graft.load_package()
import oracle_bridge as ob
ctx = T.default_context()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 181
res, spp = 64, 32
for compose in (1, 0):
    ctx.set_option("compose_spheres", compose)
    scene = T.scenes.mesh_scene(n)
    flat = scene.flatten(ctx)
    osc = ob.OracleScene.from_scene(scene, bvh=flat.bvh())
    cam = T.scenes.cornell_camera(res)
    for depth in (8, 16):
        ref_xyzw, ref_L, _ = osc.render(cam, "path", spp, depth, seed=0x5EED0001, threads=ob.lib().orc_num_threads(), want_samples=True)
        for trav in (1, 3, 4):
            ctx.set_option("traversal", trav)
            integ = T.PathIntegrator(cam, T.SeededSampler(spp, seed=0x5EED0001), depth)
            xyzw = integ.render(scene, ctx).copy()
            L = integ.sample_radiance(scene).copy()
            bad = L.view(np.uint32) != ref_L.view(np.uint32)
            where = np.argwhere(bad.any(-1))
            print(f"compose {compose} depth {depth} trav {trav} (ran {integ.stats.traversal}): {int(bad.sum())} values differ in {where.shape[0]} samples; first {where[:4].tolist()}", flush=True)
            for w in where[:3]:
                print("   gpu", L[tuple(w)], "ref", ref_L[tuple(w)])
ctx.set_option("compose_spheres", 1)
ctx.set_option("traversal", 4)
