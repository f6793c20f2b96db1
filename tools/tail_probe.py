#!/usr/bin/env python
"""How much of a frame's traversal time is the straggler tail?  Renders the workload with the DIAGNOSTIC option
debug_trace_budget (rays abandoned after N node fetches: wrong image, bulk timing) for several budgets.
    python tools/tail_probe.py --workload mesh_1m --spp 256"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
import bench
T = g.load_package()
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="mesh_1m")
ap.add_argument("--spp", type=int, default=256)
ap.add_argument("--budgets", type=int, nargs="+", default=[0, 20000, 5000, 2000, 500])
a = ap.parse_args()
scene, cam, desc = bench.build_workload(T, a.workload, 1024)
ctx = T.default_context()
for b in a.budgets:
    ctx.set_option("debug_trace_budget", b)
    integ = T.PathIntegrator(cam, T.SeededSampler(a.spp, seed=1), 8)
    integ.render(scene, ctx)
    s = integ.stats
    print(f"budget {b:6d}: total {s.ms_total:8.1f} ms  closest {s.ms_trace_closest:8.1f}  any {s.ms_trace_any:8.1f}  shade {s.ms_shade:6.1f}  rays {s.closest_rays + s.shadow_rays}")
ctx.set_option("debug_trace_budget", 0)
