#!/usr/bin/env python
"""How much of a frame depends on the BVH's TOPOLOGY — measured on the REAL rays of a frame.  (VERDICT r3 "next round" 1)

The reference's answer depends on the visiting order where (a) a ray starts inside a sphere (sphere.jl:137-138 returns the far root whatever t_max is), (b) two
candidates lie at (nearly) the same t, (c) a leaf box is grazed so closely that rounding decides `tx_min < t_max`.  Two renders of the same workload:

  reference answers   the default commit (hybrid: the reference's own tree is canonical, the rays walk the library's SAH tree under the certificate of
                      csrc/th_trace3c.h; bit-equal to a walk of the reference's tree, tests/test_gpu_hybrid.py) — with its per-reason ray counters
                      (option "count_visits"): every closest-hit ray of every bounce of the frame is classified, no probe rays
  library tree alone  option "bvh_builder" = 0: ties and inside-sphere rays resolve in the SAH tree's order

and what differs between them: camera rays (hit primitive / t), per-sample radiance, film pixels; frame time of both and of the reference's tree alone.
(Round 3's version of this tool classified `p + 1e-4 d` probe rays leaving the camera hits instead of the frame's own rays, and had no notion of (a).)

    python tools/tie_exposure.py --workload mesh_1m --res 1024 --spp 16 --depth 8 [--time-spp 256]
    python tools/tie_exposure.py --workload caustic --sppm --iterations 10
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
    ap.add_argument("--spp", type=int, default=16)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--time-spp", type=int, default=0)
    ap.add_argument("--sppm", action="store_true")
    ap.add_argument("--iterations", type=int, default=10)
    ap.add_argument("--radius", type=float, default=0.075)
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x5EED0001)
    args = ap.parse_args()
    graft.build()
    import oracle_bridge as ob
    T = graft.load_package()
    ctx = T.default_context()
    scene, cam, desc = bench.build_workload(T, args.workload, args.res)
    out = {"workload": f"{args.workload}: {desc}; {args.res}x{args.res}", "trees": {}}
    res = {}
    for tag, builder in (("library (binned SAH)", 0), ("reference (bvh.jl:87-206)", -1)):
        ctx.set_option("bvh_builder", builder)
        ctx.set_option("hybrid", 1)
        scene._flat = None
        t0 = time.time()
        flat = scene.flatten(ctx)
        t_build = time.time() - t0
        b, a, f, order = flat.bvh()
        leaf = (f & 3) == 3
        info = {"nodes": int(a.size), "leaves": int(leaf.sum()), "empty_leaves": int((leaf & ((f >> 2) == 0)).sum()), "build_upload_s": round(t_build, 2)}
        r = {"order": order}
        if not args.sppm:
            # ---- camera rays (real rays of the frame: one per sample pixel) ----
            cam_rays = ob.generate_rays(cam, T.scenes.camera_sample_grid(cam, 1, seed=3))
            r["cam"] = flat.trace_closest(cam_rays)
            integ = T.PathIntegrator(cam, T.SeededSampler(args.spp, seed=args.seed), args.depth)
            r["film"] = integ.render(scene, ctx).copy()
            r["L"] = integ.sample_radiance(scene).copy()
            info["frame_ms"] = round(integ.stats.ms_total, 2)
            info["rays"] = int(integ.stats.closest_rays + integ.stats.shadow_rays)
            info["bvh_mode"] = int(flat.bvh_mode()[0])
            if builder < 0 and flat.bvh_mode()[0] == 2:
                # every closest-hit ray of the frame, classified by the certified walk itself
                ctx.set_option("count_visits", 1)
                ic = T.PathIntegrator(cam, T.SeededSampler(args.spp, seed=args.seed), args.depth)
                ic.render(scene, ctx)
                ctx.set_option("count_visits", 0)
                c, n = [int(x) for x in ic.stats.count_sub], int(ic.stats.closest_rays)
                info["frame_rays_classified"] = {
                    "closest_hit_rays": n,
                    "start_inside_a_sphere__certified_through_the_order_word": c[3],
                    "to_the_reference_order_walk": {"total": int(ic.stats.fallback_rays), "zero_or_near_axis_parallel_direction": c[0], "sphere_clipped_or_inside_two": c[1],
                                                    "candidate_within_2dt_of_the_incumbent_or_before_its_leaf_box (ties, grazed boxes)": c[2]},
                    "fractions": {"inside_sphere": c[3] / max(1, n), "fallback": ic.stats.fallback_rays / max(1, n), "ties_and_grazes": c[2] / max(1, n)}}
                if args.time_spp:
                    ctx.set_option("hybrid", 0)
                    it = T.PathIntegrator(cam, T.SeededSampler(args.time_spp, seed=args.seed), args.depth)
                    it.render(scene, ctx)
                    it.render(scene, ctx)
                    info[f"reference_tree_alone_frame_ms_{args.time_spp}spp"] = round(it.stats.ms_total, 2)
                    ctx.set_option("hybrid", 1)
            if args.time_spp:
                it = T.PathIntegrator(cam, T.SeededSampler(args.time_spp, seed=args.seed), args.depth)
                it.render(scene, ctx)
                it.render(scene, ctx)
                info[f"frame_ms_{args.time_spp}spp"] = round(it.stats.ms_total, 2)
                info[f"closest_ms_{args.time_spp}spp"] = round(it.stats.ms_trace_closest, 2)
        else:
            integ = T.SPPMIntegrator(cam, args.radius, args.depth, args.iterations, -1, seed=args.seed)
            integ.render(scene, ctx)
            r["film"] = integ.render(scene, ctx).copy()
            st = integ.state()
            r["M"], r["N"], r["Ld"] = st["M"].copy(), st["N"].copy(), st["Ld"].copy()
            info["run_ms"] = round(integ.stats.ms_total, 2)
        out["trees"][tag] = info
        res[tag] = r
    ctx.set_option("bvh_builder", -1)
    A, B = res["library (binned SAH)"], res["reference (bvh.jl:87-206)"]

    def ray_diff(name):
        ha, hb = A[name], B[name]
        ca = np.where(ha["prim"] >= 0, A["order"][np.maximum(ha["prim"], 0)], -1)  # ordered slot -> caller primitive
        cb = np.where(hb["prim"] >= 0, B["order"][np.maximum(hb["prim"], 0)], -1)
        dp = ca != cb
        same_t = ha["t"].view(np.uint32) == hb["t"].view(np.uint32)
        dt = ~same_t & ~(np.isinf(ha["t"]) & np.isinf(hb["t"]))
        return {"rays": int(ha.size), "primitive_differs": int(dp.sum()), "of_those_exact_t_ties": int((dp & same_t).sum()), "t_differs": int(dt.sum()),
                "hit_vs_miss": int(((ca < 0) != (cb < 0)).sum())}

    d = {}
    if not args.sppm:
        d["camera_rays"] = ray_diff("cam")
        La, Lb = A["L"], B["L"]
        sd = (La.view(np.uint32) != Lb.view(np.uint32)).any(-1) & ~(np.isnan(La).any(-1) & np.isnan(Lb).any(-1))
        d["samples"] = {"n": int(sd.size), "differ": int(sd.sum()), "fraction": float(sd.mean()), "spp": args.spp, "depth": args.depth}
    fa, fb = A["film"], B["film"]
    pd = (fa.view(np.uint32) != fb.view(np.uint32)).any(-1)
    w = np.maximum(fa[..., 3:4], 1e-20)
    rgb_a, rgb_b = fa[..., :3] / w, fb[..., :3] / np.maximum(fb[..., 3:4], 1e-20)
    d["film"] = {"pixels": int(pd.size), "differ": int(pd.sum()), "fraction": float(pd.mean()), "max_abs_diff_xyz_over_weight": float(np.abs(rgb_a - rgb_b).max()),
                 "relative_rmse": float(np.sqrt(np.mean((rgb_a - rgb_b) ** 2)) / max(1e-30, float(np.mean(np.abs(rgb_a)))))}
    if args.sppm:
        d["sppm"] = {"iterations": args.iterations, "pixels_M_differs": int((A["M"] != B["M"]).sum()), "pixels_N_differs": int((A["N"] != B["N"]).sum()),
                     "pixels_Ld_differs": int((A["Ld"].view(np.uint32) != B["Ld"].view(np.uint32)).any(-1).sum())}
    out["difference"] = d
    print(json.dumps(out))


if __name__ == "__main__":
    main()
