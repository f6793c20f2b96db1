#!/usr/bin/env python
"""Frame time of a bench workload under sets of context options (one render each, after one warm-up of the first set):

    python tools/option_sweep.py --workload mesh_1m --spp 256 --set traversal=2 --set traversal=3,overlap=0 --set streaming=1

The first line is always the defaults.  Prints total / per-kernel-class HIP-event times from trhip_stats."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
import bench
T = g.load_package()
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="mesh_1m")
ap.add_argument("--spp", type=int, default=256)
ap.add_argument("--res", type=int, default=1024)
ap.add_argument("--depth", type=int, default=8)
ap.add_argument("--repeat", type=int, default=2)
ap.add_argument("--set", action="append", default=[], help="comma-separated name=value options")
a = ap.parse_args()
scene, cam, desc = bench.build_workload(T, a.workload, a.res)


def run(opts):
    ctx = T.Context(0)   # a fresh context per set: options that act at scene commit (bvh_builder, tiny_scene_prims) take effect
    for k, v in opts.items():
        ctx.set_option(k, v)
    best = None
    for _ in range(a.repeat):
        integ = T.PathIntegrator(cam, T.SeededSampler(a.spp, seed=1), a.depth)
        integ.render(scene, ctx)
        s = integ.stats
        if best is None or s.ms_total < best.ms_total:
            best = s
    s = best
    rays = s.closest_rays + s.shadow_rays
    scene._flat.free()
    scene._flat = None
    ctx.close()
    print(f"{str(opts):60s} total {s.ms_total:8.1f} ms {rays / s.ms_total / 1e3:8.1f} Mray/s  closest {s.ms_trace_closest:7.1f}  any {s.ms_trace_any:7.1f}  shade {s.ms_shade:6.1f}  film {s.ms_film:5.1f}", flush=True)


run({})
for spec in a.set:
    run({kv.split("=")[0]: int(kv.split("=")[1]) for kv in spec.split(",") if kv})
