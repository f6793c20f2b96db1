#!/usr/bin/env python
"""Film gather variants on one frame: python tools/film_probe.py [--spp 64]   (prints ms_film per film_block mode)."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
T = g.load_package()
ap = argparse.ArgumentParser()
ap.add_argument("--spp", type=int, default=64)
a = ap.parse_args()
scene, cam = T.scenes.cornell_scene(), T.scenes.cornell_camera(1024)
ctx = T.default_context()
ap2 = [(m, 1) for m in (5, 7, 8, 4, 10, 6, 12, 13, 9, 11)] + [(5, 0), (2, 0)]
for mode, relayout in ap2:
    ctx.set_option("film_block", mode)
    ctx.set_option("film_relayout", relayout)
    integ = T.PathIntegrator(cam, T.SeededSampler(a.spp, seed=1), 8)
    integ.render(scene, ctx)
    integ.render(scene, ctx)
    print("film_block", mode, "relayout", relayout, "ms_film", round(integ.stats.ms_film, 3), "ms_total", round(integ.stats.ms_total, 2))
