#!/usr/bin/env python
"""Soak test of the SPPM integrator on the GPU against the oracle (oracle/orc_sppm.h) on random small scenes: matte walls, a few hundred to a
few thousand random matte / plastic / mirror / glass triangles and spheres, a point or a spot light; 40 x 40 film, 2-3 iterations.
M, N, radius, Ld, visible points, grid size and in-bounds photon hits must match bit for bit; phi / tau within the reordering tolerance
(tests/test_gpu_sppm.py: check_pair).  Run on the GPU box:  python tools/soak_sppm.py --scenes 40 --seed 1"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft

graft.build_library()
graft.build_oracle()
T = graft.load_package()
import oracle_bridge as ob
import test_gpu_sppm as ts


def rand_scene(rng, k):
    mats = [T.MatteMaterial(T.ConstantTexture(T.RGBSpectrum(0.8)), T.ConstantTexture(0.0)),
            T.PlasticMaterial(T.ConstantTexture(T.RGBSpectrum(0.5)), T.ConstantTexture(T.RGBSpectrum(0.4)), T.ConstantTexture(float(rng.uniform(0.02, 0.3))), True),
            T.MirrorMaterial(T.ConstantTexture(T.RGBSpectrum(0.9))),
            T.GlassMaterial(T.ConstantTexture(T.RGBSpectrum(1.0)), T.ConstantTexture(T.RGBSpectrum(1.0)), T.ConstantTexture(0.0), T.ConstantTexture(0.0), T.ConstantTexture(float(rng.uniform(1.2, 1.6))), True)]
    core = T.ShapeCore(T.translate([0, 0, 0]), False)
    prims, _ = T.scenes.cornell_primitives(spheres=False)
    for m in range(4):
        n_tris = int(rng.integers(20, 600))
        c = rng.random((n_tris, 3), dtype=np.float32) * np.float32(0.9) + np.float32([0.05, 0.0, -2.95])
        size = (0.01 + 0.12 * rng.random((n_tris, 1), dtype=np.float32) ** 2).astype(np.float32)
        e1 = rng.standard_normal((n_tris, 3)).astype(np.float32)
        e2 = rng.standard_normal((n_tris, 3)).astype(np.float32)
        verts = np.stack([c, c + size * e1, c + size * e2], axis=1).reshape(-1, 3).astype(np.float32)
        prims.append(T.create_mesh_primitives(core, np.arange(3 * n_tris, dtype=np.uint32) + 1, verts, None, mats[m]))
    for _ in range(int(rng.integers(0, 6))):
        p = rng.random(3) * 0.8 + [0.1, 0.1, -2.9]
        prims.append(T.GeometricPrimitive(T.Sphere(T.ShapeCore(T.translate([float(p[0]), float(p[1]), float(p[2])]), False), float(0.03 + 0.15 * rng.random()), 360.0), mats[int(rng.integers(0, 4))]))
    lights = T.scenes.cornell_lights() if k % 2 == 0 else [ts.spot_light(T)]
    return T.Scene(lights, T.BVHAccel(prims, 1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=20)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    ctx = T.default_context()
    cam = T.scenes.cornell_camera(40)
    bad = 0
    for k in range(a.scenes):
        rng = np.random.default_rng(a.seed * 1000 + k)
        scene = rand_scene(rng, k)
        iters = 2 + k % 2
        ctx.set_option("sppm_batch", k % 3)
        try:
            _, xyzw, got, ref = ts.run_pair(T, ob, ctx, scene, cam, float(rng.uniform(0.03, 0.1)), 5, iters, int(rng.integers(2000, 8000)), seed=100 + k)
            ts.check_pair(T, xyzw, got, ref, iters)
            print(f"scene {k:3d}: {iters} iterations, sum M {int(got['M'].sum())}, in-bounds photon hits {got['info']['photon_hits']}: equal", flush=True)
        except AssertionError as e:
            bad += 1
            print(f"scene {k:3d}: MISMATCH {str(e)[:300]}", flush=True)
        finally:
            ctx.set_option("sppm_batch", 0)
            scene._flat = None
    print(f"total: {a.scenes} scenes, {bad} with a mismatch")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
