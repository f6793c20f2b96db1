#!/usr/bin/env python
"""Host SAH builder vs device LBVH builder vs device SAH builder: commit time and frame time.   python tools/bvh_probe.py --workload mesh_1m --spp 64"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
import bench
T = g.load_package()
ap = argparse.ArgumentParser()
ap.add_argument("--workload", nargs="+", default=["mesh_1m"])
ap.add_argument("--spp", type=int, default=64)
ap.add_argument("--depth", type=int, default=8)
ap.add_argument("--builders", type=int, nargs="+", default=[0, 1, 3])
a = ap.parse_args()
ctx = T.default_context()
for wl in a.workload:
    for builder in a.builders:
        ctx.set_option("bvh_builder", builder)
        scene, cam, desc = bench.build_workload(T, wl, 1024)
        t0 = time.time()
        flat = scene.flatten(ctx)
        tb = time.time() - t0
        import ctypes as C
        dms = C.c_double()
        T.lib().trhip_last_bvh_build_ms(ctx._h, C.byref(dms))
        nn = flat.bvh()[1].size
        integ = T.PathIntegrator(cam, T.SeededSampler(a.spp, seed=1), a.depth)
        ctx.set_option("count_visits", 1)
        integ.render(scene, ctx)
        s = integ.stats
        ctx.set_option("count_visits", 0)
        integ.render(scene, ctx)
        s2 = integ.stats
        flat.free()
        scene._flat = None
        print(f"{wl} builder {builder}: commit {tb:6.2f} s (device build {dms.value:7.2f} ms, {nn} nodes)  frame {s2.ms_total:8.1f} ms  closest {s2.ms_trace_closest:8.1f}  nodes/ray {s.nodes_visited / s.closest_rays:6.1f} prims/ray {s.prims_tested / s.closest_rays:5.1f}  Mray/s {(s2.closest_rays + s2.shadow_rays) / s2.ms_total / 1e3:.1f}", flush=True)
ctx.set_option("bvh_builder", -1)
