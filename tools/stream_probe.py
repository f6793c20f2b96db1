#!/usr/bin/env python
"""Streaming wavefront tuning: python tools/stream_probe.py --workload mesh_1m --spp 256"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
import bench
T = g.load_package()
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="mesh_1m")
ap.add_argument("--spp", type=int, default=256)
a = ap.parse_args()
scene, cam, desc = bench.build_workload(T, a.workload, 1024)
ctx = T.default_context()
for streaming, bmin, shift in [(0, 0, 0), (1, 2048, 12), (1, 2048, 14), (1, 2048, 16), (1, 1024, 31), (1, 4096, 31), (1, 16384, 31)]:
    ctx.set_option("streaming", streaming)
    if streaming:
        ctx.set_option("stream_budget_min", bmin)
        ctx.set_option("stream_budget_shift", shift)
    integ = T.PathIntegrator(cam, T.SeededSampler(a.spp, seed=1), 8)
    integ.render(scene, ctx)
    s = integ.stats
    print(f"streaming {streaming} min {bmin:6d} shift {shift:2d}: total {s.ms_total:8.1f} ms  closest {s.ms_trace_closest:8.1f}  any {s.ms_trace_any:8.1f}  shade {s.ms_shade:6.1f} film {s.ms_film:5.1f}", flush=True)
