#!/usr/bin/env python
"""Fixed cost of a frame (what strong scaling over 8 GPUs is left with):   [TRHIP_FRAME_TIMING=1] python tools/frame_overhead.py [--workload mesh_1m] [--spp 8 32]
Prints, per spp: wall time per frame around the render call, the library's own GPU time (HIP events) and the sum of its kernel classes."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
import bench
import torch
T = g.load_package()
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="mesh_1m")
ap.add_argument("--spp", type=int, nargs="+", default=[8, 32])
ap.add_argument("--frames", type=int, default=8)
ap.add_argument("--opt", action="append", default=[])
a = ap.parse_args()
ctx = T.default_context()
for kv in a.opt:
    k, v = kv.split("=")
    ctx.set_option(k, int(v))
scene, cam, desc = bench.build_workload(T, a.workload, 1024)
scene.flatten(ctx)
h, w = cam.film.size
film = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
for spp in a.spp:
    integ = T.PathIntegrator(cam, T.SeededSampler(spp, seed=1), 8)
    for _ in range(3):
        integ.render(scene, ctx, device_out=film.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gpu = cls = 0.0
    for _ in range(a.frames):
        integ.render(scene, ctx, device_out=film.data_ptr())
        s = integ.stats
        gpu += s.ms_total
        cls += s.ms_raygen + s.ms_trace_closest + s.ms_shade + s.ms_trace_any + s.ms_film
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3 / a.frames
    print(f"spp {spp:4d}: wall {wall:8.3f} ms per frame, GPU (events) {gpu / a.frames:8.3f}, kernel classes {cls / a.frames:8.3f}, host outside the GPU span {wall - gpu / a.frames:6.3f}", flush=True)
