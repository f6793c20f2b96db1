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
for mode in (2, 3, 4, 5, 6, 7, 8, 9):
    ctx.set_option("film_block", mode)
    integ = T.PathIntegrator(cam, T.SeededSampler(a.spp, seed=1), 8)
    integ.render(scene, ctx)
    integ.render(scene, ctx)
    print("film_block", mode, "ms_film", round(integ.stats.ms_film, 3), "ms_total", round(integ.stats.ms_total, 2))
