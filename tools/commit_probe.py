#!/usr/bin/env python
"""Where scene commit spends its time:   TRHIP_COMMIT_TIMING=1 python tools/commit_probe.py --workload mesh_10m
(the library prints its stages on stderr; this prints the host side around it)."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
import bench
T = g.load_package()
ap = argparse.ArgumentParser()
ap.add_argument("--workload", nargs="+", default=["mesh_1m"])
ap.add_argument("--builders", type=int, nargs="+", default=[-1])
a = ap.parse_args()
ctx = T.default_context()
for wl in a.workload:
    for builder in a.builders:
        ctx.set_option("bvh_builder", builder)
        t0 = time.time()
        scene, cam, desc = bench.build_workload(T, wl, 1024)
        t1 = time.time()
        flat = scene.flatten(ctx)
        t2 = time.time()
        print(f"{wl} builder {builder}: scene objects {t1 - t0:.2f} s, flatten (add + commit) {t2 - t1:.3f} s, {flat.bvh()[1].size} nodes", flush=True)
        flat.free()
        scene._flat = None
ctx.set_option("bvh_builder", -1)
