#!/usr/bin/env python
"""Traversal micro-benchmark: closest-hit / any-hit kernels alone on device-resident ray sets.

    python tools/trace_bench.py --workload mesh_1m [--rays 4194304] [--traversal 1 2]

Ray sets: primary (camera rays of the scene), bounce (cosine-distributed directions leaving the primary hit points:
what depth-2 of the path tracer traces), incoherent (uniform origins in the scene bound, uniform directions).
Prints one JSON line per (ray set, kernel) with Mray/s, visit counts and algorithmic GB/s.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="mesh_1m")
    ap.add_argument("--rays", type=int, default=1 << 22)
    ap.add_argument("--repeat", type=int, default=5)
    ap.add_argument("--traversal", type=int, nargs="+", default=[1, 2, 3])
    ap.add_argument("--res", type=int, default=1024)
    ap.add_argument("--sorted-sets", action="store_true", help="also time the bounce / incoherent rays grouped by octant and origin cell")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="library option set before the scene is committed (e.g. bvh_builder=0: the library's tree alone)")
    ap.add_argument("--post-opt", action="append", default=[], metavar="NAME=VALUE", help="library option set after the commit (e.g. hybrid=0: every ray on the canonical tree)")
    ap.add_argument("--sets", nargs="+", default=None, help="ray sets to run (default: all)")
    ap.add_argument("--kinds", nargs="+", default=["closest", "any"])
    args = ap.parse_args()
    import torch
    import __graft_entry__ as graft
    graft.build_library()
    T = graft.load_package()
    import bench
    ctx = T.Context(0)
    scene, cam, desc = bench.build_workload(T, args.workload, args.res)
    for kv in args.opt:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    t0 = time.time()
    flat = scene.flatten(ctx)
    for kv in args.post_opt:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    print(json.dumps({"workload": args.workload, "desc": desc, "bvh_build_upload_s": round(time.time() - t0, 3), "nodes": int(flat.bvh()[1].size), "bvh_mode": flat.bvh_mode()[0]}), flush=True)
    L = T.lib()
    n = args.rays
    # ---- ray sets ----
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
    bnd = flat.bvh()[0][0]
    incoherent = T.scenes.incoherent_rays(n, bnd[:3], bnd[3:])
    sets = {"primary": primary, "bounce": bounce, "incoherent": incoherent}
    if args.sets:
        sets = {k: v for k, v in sets.items() if k in args.sets}
    if args.sorted_sets:
        # what would ray reordering buy?  the same rays grouped by direction octant, and by octant + origin cell
        def octant(r):
            return ((r[:, 4] < 0).astype(np.int64) << 2) | ((r[:, 5] < 0).astype(np.int64) << 1) | (r[:, 6] < 0).astype(np.int64)

        def cell(r, bits):
            lo, hi = bnd[:3], bnd[3:]
            q = np.clip(((r[:, 0:3] - lo) / np.maximum(hi - lo, 1e-20) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
            key = np.zeros(r.shape[0], np.int64)
            for b in range(bits - 1, -1, -1):
                for a in range(3):
                    key = (key << 1) | ((q[:, a] >> b) & 1)
            return key
        for nm in ("bounce", "incoherent"):
            r = sets[nm]
            sets[nm + "_oct"] = r[np.argsort(octant(r), kind="stable")]
            sets[nm + "_oct_cell"] = r[np.argsort((octant(r) << 15) | cell(r, 5), kind="stable")]
            sets[nm + "_cell_oct"] = r[np.argsort((cell(r, 5) << 3) | octant(r), kind="stable")]
    counts = np.zeros(4, np.uint64)
    for name, rays in sets.items():
        d_rays = torch.from_numpy(np.ascontiguousarray(rays)).cuda()
        nr = rays.shape[0]
        d_hits = torch.empty((nr, 4), dtype=torch.float32, device="cuda")
        d_occ = torch.empty(nr, dtype=torch.uint8, device="cuda")
        ref = None
        for trav in args.traversal:
            ctx.set_option("traversal", trav)
            for kind in args.kinds:
                fn = L.trhip_trace_closest_device if kind == "closest" else L.trhip_trace_any_device
                outp = d_hits.data_ptr() if kind == "closest" else d_occ.data_ptr()
                ctx.set_option("count_visits", 1)
                ms = C.c_double()
                ctx.check(fn(ctx._h, flat._h, C.c_void_p(d_rays.data_ptr()), nr, C.c_void_p(outp), 1, C.byref(ms)))
                ctx.check(L.trhip_last_visit_counts(ctx._h, counts.ctypes.data_as(C.POINTER(C.c_uint64))))
                nodes, prims = (int(counts[0]), int(counts[1])) if kind == "closest" else (int(counts[2]), int(counts[3]))
                ctx.set_option("count_visits", 0)
                ctx.check(fn(ctx._h, flat._h, C.c_void_p(d_rays.data_ptr()), nr, C.c_void_p(outp), 2, C.byref(ms)))  # warm
                ctx.check(fn(ctx._h, flat._h, C.c_void_p(d_rays.data_ptr()), nr, C.c_void_p(outp), args.repeat, C.byref(ms)))
                res = d_hits.cpu().numpy().copy() if kind == "closest" else d_occ.cpu().numpy().copy()
                key = (name, kind)
                same = None
                if ref is None:
                    ref = {}
                if key in ref:
                    same = bool(np.array_equal(ref[key].view(np.uint8), res.view(np.uint8)))
                else:
                    ref[key] = res
                node_bytes = 64 if trav == 2 else 32
                # trav 2 counts node BOXES tested (2 per 64-byte fetch); trav 1 counts 32-byte nodes fetched
                fetch_bytes = nodes * 96 if trav == 4 else ((nodes // 2) * 64 if trav in (2, 3, 6, 7) else nodes * 32)
                alg = 32 * nr + (16 if kind == "closest" else 1) * nr + fetch_bytes + 48 * prims
                print(json.dumps({"rays": name, "n": nr, "kernel": kind, "traversal": trav, "ms": round(ms.value, 3), "Mray_s": round(nr / ms.value / 1e3, 1),
                                  "nodes_per_ray": round(nodes / nr, 2), "prims_per_ray": round(prims / nr, 2), "alg_GBps": round(alg / ms.value / 1e6, 1),
                                  "hit_frac": round(float((res[:, 1].view(np.int32) >= 0).mean()) if kind == "closest" else float(res.mean()), 4), "same_as_first": same}), flush=True)


if __name__ == "__main__":
    main()
