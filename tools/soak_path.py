#!/usr/bin/env python
"""Soak test of the Path and Whitted integrators on the GPU against the oracle (oracle/orc_render.h) on random small scenes with all four
material kinds (tools/soak_sppm.py's generator): 48 x 48 film, 4 spp, depth 6; every traversal kernel.  Per-sample radiance, film pixels
and ray counts must match bit for bit.  Run on the GPU box:  python tools/soak_path.py --scenes 40 --seed 1"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import soak_sppm as ss

T, ob = ss.T, ss.ob


def same_bits(a, b):
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    na, nb = np.isnan(a), np.isnan(b)
    return np.array_equal(na, nb) and not ((a.view(np.uint32) != b.view(np.uint32)) & ~na).any()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=20)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    ctx = T.default_context()
    cam = T.scenes.cornell_camera(48)
    bad = 0
    for k in range(a.scenes):
        rng = np.random.default_rng(a.seed * 1000 + k)
        scene = ss.rand_scene(rng, k)
        flat = scene.flatten(ctx)
        osc = ob.OracleScene.from_scene(scene, bvh=flat.bvh())
        ref_p, ref_L, st = osc.render(cam, "path", 4, 6, seed=500 + k, want_samples=True)
        ref_w, _, _ = osc.render(cam, "whitted", 2, 5, seed=500 + k)
        msgs = []
        for trav in (3, 7, 6, 2, 1):
            if not ctx.has_option("traversal", trav):  # (6 / 7: only in the EXPERIMENTS build of the library)
                continue
            try:
                integ = T.PathIntegrator(cam, T.SeededSampler(4, seed=500 + k), 6)
                film = integ.render(scene, ctx).copy()
                L = integ.sample_radiance(scene).copy()
                if not same_bits(L, ref_L):
                    msgs.append(f"traversal {trav}: per-sample radiance differs")
                if not same_bits(film, ref_p):
                    msgs.append(f"traversal {trav}: path film differs")
                if (integ.stats.closest_rays, integ.stats.shadow_rays) != (st.closest_rays, st.shadow_rays):
                    msgs.append(f"traversal {trav}: ray counts {integ.stats.closest_rays}/{integ.stats.shadow_rays} vs {st.closest_rays}/{st.shadow_rays}")
                if trav == 3:
                    w = T.WhittedIntegrator(cam, T.SeededSampler(2, seed=500 + k), 5).render(scene, ctx)
                    if not same_bits(w, ref_w):
                        msgs.append("whitted film differs")
            finally:
                ctx.set_option("traversal", 3)
        bad += 1 if msgs else 0
        print(f"scene {k:3d}: {st.closest_rays} closest + {st.shadow_rays} shadow rays: {'equal' if not msgs else 'MISMATCH ' + '; '.join(msgs)}", flush=True)
        scene._flat = None
    print(f"total: {a.scenes} scenes x (path with every traversal the loaded library carries of 3, 7, 6, 2, 1 + whitted), {bad} with a mismatch")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
