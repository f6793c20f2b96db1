#!/usr/bin/env python
"""Times the steps of the fresh-context streaming + two-stream test (tests/test_gpu_parity.py) — diagnostic."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
g.build()
T = g.load_package()
t0 = time.time()
def lap(msg):
    global t0
    t = time.time()
    print(f"{msg}: {t - t0:.2f} s", flush=True)
    t0 = t
ctx = T.default_context()
scene, cam = T.scenes.mesh_scene(24), T.scenes.cornell_camera(32)
ctx.set_option("compose_spheres", 1)
T.PathIntegrator(cam, T.SeededSampler(4, seed=21), 6).render(scene, ctx)
lap("main context, classic")
fresh = T.Context(0)
lap("fresh context")
fresh.set_option("compose_spheres", 1)
scene.flatten(fresh)
lap("flatten on fresh")
for opts in ({"streaming": 1, "overlap": 0}, {"streaming": 1, "overlap": 1}, {"streaming": 1, "overlap": 1, "stream_budget_min": 2}, {"streaming": 0, "overlap": 1}, {"streaming": 1, "overlap": 1}):
    for k, v in opts.items():
        fresh.set_option(k, v)
    integ = T.PathIntegrator(cam, T.SeededSampler(4, seed=21), 6)
    integ.render(scene, fresh)
    lap(f"render {opts} ms_total {integ.stats.ms_total:.2f}")
    fresh.set_option("streaming", 0)
    fresh.set_option("stream_budget_min", 2048)
scene._flat.free(); scene._flat = None
fresh.close()
lap("close")
