#!/usr/bin/env python
"""Hybrid mode A/B: the same frame on (a) the library's SAH tree alone (bvh_builder 0 / 3: fast, not Trace.jl's tie-breaks), (b) the reference's
tree alone (hybrid 0: exact, slow), (c) both (default: the certified walk on the SAH tree + the flagged rays on the reference's tree).

Prints per mode: commit seconds, frame / closest-hit / any-hit ms, fallback rays; and checks (b) == (c) bit for bit (film + per-sample radiance).

    python tools/hybrid_probe.py --workload mesh_1m --res 1024 --spp 32 --depth 8 [--check-spp 4]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="mesh_1m")
    ap.add_argument("--res", type=int, default=1024)
    ap.add_argument("--spp", type=int, default=32)
    ap.add_argument("--check-spp", type=int, default=4)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x5EED0001)
    ap.add_argument("--count", action="store_true", help="also report boxes / primitives per closest-hit ray (an instrumented pass)")
    ap.add_argument("--skip-library", action="store_true")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="library option set once at the start (e.g. leaf_queue=1)")
    args = ap.parse_args()
    graft.build()
    T = graft.load_package()
    ctx = T.default_context()
    for kv in args.opt:
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    scene, cam, desc = bench.build_workload(T, args.workload, args.res)
    out = {"workload": f"{args.workload}: {desc}; {args.res}x{args.res}, depth {args.depth}", "modes": {}}
    films = {}
    modes = [("hybrid (default)", -1, 1), ("reference tree alone (hybrid 0)", -1, 0)]
    if not args.skip_library:
        modes.append(("library SAH tree alone (bvh_builder 0)", 0, 1))
    last_builder = None
    flat = None
    for tag, builder, hyb in modes:
        ctx.set_option("bvh_builder", builder)
        ctx.set_option("hybrid", hyb)
        info = {}
        if builder != last_builder:
            scene._flat = None
            if flat is not None:
                flat.free()
            t0 = time.time()
            flat = scene.flatten(ctx)
            info["commit_s"] = round(time.time() - t0, 2)
            last_builder = builder
        mode, acc_nodes, acc_depth = flat.bvh_mode()
        info.update({"bvh_mode": mode, "canonical_nodes": int(flat.bvh()[1].size), "accelerator_nodes": acc_nodes})
        it = T.PathIntegrator(cam, T.SeededSampler(args.check_spp, seed=args.seed), args.depth)
        films[tag] = (it.render(scene, ctx).copy(), it.sample_radiance(scene).copy())
        it = T.PathIntegrator(cam, T.SeededSampler(args.spp, seed=args.seed), args.depth)
        it.render(scene, ctx)
        it.render(scene, ctx)
        st = it.stats
        info.update({"traversal": int(st.traversal), "frame_ms": round(st.ms_total, 2), "closest_ms": round(st.ms_trace_closest, 2), "any_ms": round(st.ms_trace_any, 2),
                     "shade_ms": round(st.ms_shade, 2), "closest_rays": int(st.closest_rays), "fallback_rays": int(st.fallback_rays),
                     "fallback_fraction": round(st.fallback_rays / max(1, st.closest_rays), 5), "launches_closest": int(st.launches_trace_closest)})
        if args.count:
            ctx.set_option("count_visits", 1)
            it2 = T.PathIntegrator(cam, T.SeededSampler(min(args.spp, 8), seed=args.seed), args.depth)
            it2.render(scene, ctx)
            s2 = it2.stats
            info["boxes_per_closest_ray"] = round(s2.nodes_visited / max(1, s2.closest_rays), 2)
            info["prims_per_closest_ray"] = round(s2.prims_tested / max(1, s2.closest_rays), 3)
            info["fallback_why (direction, sphere, tie/guard, unknown entry)"] = [int(x) for x in s2.count_sub]
            info["fallback_fraction_counted_pass"] = round(s2.fallback_rays / max(1, s2.closest_rays), 5)
            ctx.set_option("count_visits", 0)
        out["modes"][tag] = info
    ctx.set_option("bvh_builder", -1)
    ctx.set_option("hybrid", 1)
    a, b = films["hybrid (default)"], films["reference tree alone (hybrid 0)"]
    out["hybrid_equals_reference_tree"] = {"film_values_differing": int((a[0].view(np.uint32) != b[0].view(np.uint32)).sum()),
                                           "sample_values_differing": int(((a[1].view(np.uint32) != b[1].view(np.uint32)) & ~(np.isnan(a[1]) & np.isnan(b[1]))).sum()),
                                           "samples": int(a[1].size // 3)}
    if not args.skip_library:
        c = films["library SAH tree alone (bvh_builder 0)"]
        out["library_tree_vs_reference_tree"] = {"film_values_differing": int((c[0].view(np.uint32) != b[0].view(np.uint32)).sum()),
                                                 "sample_values_differing": int(((c[1].view(np.uint32) != b[1].view(np.uint32)) & ~(np.isnan(c[1]) & np.isnan(b[1]))).sum())}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
