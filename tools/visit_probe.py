#!/usr/bin/env python
"""Which rays are the traversal tail?  Needs a DIAGNOSTIC build of the library (per-ray interior-fetch counts in hits[].x):

    python -c "import __graft_entry__ as g; g.build_library(extra_flags=['-DTH_DIAG_RAY_VISITS'], out_name='libtracehip_diag.so')"
    TRHIP_LIB=$PWD/trace.jl_amd/libtracehip_diag.so python tools/visit_probe.py --workload mesh_1m

Traces the bounce ray set of tools/trace_bench.py (cosine-distributed directions leaving the primary hit points) with
k_trace3 and prints the distribution of node fetches per ray and the worst rays (origin, direction, fetches, hit primitive).
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="mesh_1m")
    ap.add_argument("--rays", type=int, default=1 << 22)
    ap.add_argument("--res", type=int, default=1024)
    ap.add_argument("--top", type=int, default=24)
    ap.add_argument("--bvh-builder", type=int, default=-1)
    args = ap.parse_args()
    import torch
    import __graft_entry__ as graft
    T = graft.load_package()
    import bench
    ctx = T.Context(0)
    ctx.set_option("bvh_builder", args.bvh_builder)
    scene, cam, desc = bench.build_workload(T, args.workload, args.res)
    flat = scene.flatten(ctx)
    L = T.lib()
    n = args.rays
    samples = T.scenes.camera_sample_grid(cam, max(1, -(-n // ((args.res + 2) ** 2))), seed=3)[:n]
    sn = cam.sensor()
    primary = np.empty((samples.shape[0], 8), np.float32)
    ctx.check(L.trhip_generate_rays(ctx._h, C.byref(sn), T._ffi.fptr(samples), samples.shape[0], T._ffi.fptr(primary)))
    geom = flat.hit_geometry(primary)
    hit = np.abs(geom[:, 6:9]).sum(axis=1) > 0
    p, ns = geom[hit, 0:3], geom[hit, 6:9]
    rng = np.random.default_rng(1)
    u = rng.random((p.shape[0], 2), dtype=np.float32)
    r, phi = np.sqrt(u[:, 0]), 2 * np.pi * u[:, 1]
    loc = np.stack([r * np.cos(phi), r * np.sin(phi), np.sqrt(np.maximum(0, 1 - u[:, 0]))], axis=1).astype(np.float32)
    a = np.where(np.abs(ns[:, :1]) > 0.9, np.array([[0, 1, 0]], np.float32), np.array([[1, 0, 0]], np.float32))
    t1 = np.cross(ns, a)
    t1 /= np.linalg.norm(t1, axis=1, keepdims=True)
    t2 = np.cross(ns, t1)
    d = (loc[:, :1] * t1 + loc[:, 1:2] * t2 + loc[:, 2:3] * ns).astype(np.float32)
    bounce = np.zeros((p.shape[0], 8), np.float32)
    bounce[:, 0:3] = p + np.float32(1e-6) * d
    bounce[:, 3] = np.inf
    bounce[:, 4:7] = d
    ctx.set_option("traversal", 3)
    for name, rays in (("primary", primary), ("bounce", bounce)):
        d_rays = torch.from_numpy(np.ascontiguousarray(rays)).cuda()
        nr = rays.shape[0]
        d_hits = torch.empty((nr, 4), dtype=torch.float32, device="cuda")
        ms = C.c_double()
        ctx.check(L.trhip_trace_closest_device(ctx._h, flat._h, C.c_void_p(d_rays.data_ptr()), nr, C.c_void_p(d_hits.data_ptr()), 1, C.byref(ms)))
        h = d_hits.cpu().numpy()
        v = h[:, 0]
        prim = h[:, 1].view(np.int32)
        q = np.quantile(v, [0.5, 0.9, 0.99, 0.999, 0.9999, 1.0])
        print(json.dumps({"rays": name, "n": nr, "ms": round(ms.value, 2), "mean": round(float(v.mean()), 1), "quantiles(50,90,99,99.9,99.99,100)": [float(x) for x in q],
                          "share_of_fetches_above_p99.9": round(float(v[v > q[3]].sum() / v.sum()), 4)}), flush=True)
        order = np.argsort(-v)[:args.top]
        for i in order:
            print(f"  fetches {int(v[i]):8d}  prim {int(prim[i]):9d}  o = ({rays[i,0]:.6f}, {rays[i,1]:.6f}, {rays[i,2]:.6f})  d = ({rays[i,4]:.6f}, {rays[i,5]:.6f}, {rays[i,6]:.6f})", flush=True)


if __name__ == "__main__":
    main()
