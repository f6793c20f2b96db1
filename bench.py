#!/usr/bin/env python
"""bench.py — the reference's headline metric on MI355X: Mray/s (all bounces, closest-hit + shadow rays) at 1024x1024,
depth 8 (BASELINE.json), PathIntegrator on the north star's 1 M-triangle synthetic scene (S-mesh) by default.

    python bench.py --gpus N --steps K --warmup W [--workload mesh_1m|blob_870k|cornell|shadows|mesh_10m|caustic|caustic_sppm]
                    [--spp S] [--res R] [--depth D] [--scaling weak|strong]

One "step" = one full render of the workload.  N > 1 (one process per GPU, torch.distributed.run): the frame is sharded by
global sample index and the film accumulators are sum-reduced to rank 0 by trhip_film_reduce (RCCL inside libtracehip.so).
`--scaling strong` (default for N > 1): the `spp` samples of ONE frame are split over the ranks — what BASELINE configs[4] asks
(`--workload mesh_10m --res 4096 --spp 1024 --depth 16`) and the only mode in which "x N" is not true by construction; `weak`:
every rank renders `spp` samples per pixel (per-GPU work fixed).  For N > 1 the line also carries the other mode, measured right
after, as `"weak_scaling"` / `"strong_scaling"`, the number of ranks the library's RCCL communicator has (`rccl_ranks`: an N-rank
line cannot be produced without it — a failed trhip_comm_init ends the run with a non-zero exit) and the film reduce's time.
`--workload mesh_10m --res 4096 --spp 128 --depth 16` on ONE GPU is one rank's share of BASELINE configs[4].
Inputs (scene, BVH) are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
KERNEL_CLASSES = ("raygen", "trace_closest", "shade", "trace_any", "film")
TRAVERSAL_KERNEL = {1: "k_trace_closest", 2: "k_trace2", 3: "k_trace3", 4: "k_trace8", 5: "k_trace_leaf", 6: "k_trace4", 7: "k_trace7", 9: "k_trace3c"}
BVH_MODE = {0: "library-sah", 1: "reference", 2: "hybrid", 3: "library-sah + itself four-wide as accelerator"}  # trhip_scene_bvh_mode: which tree(s) the scene holds (include/tracehip.h)
L2_PLUS_MALL_BYTES = (32 + 256) << 20  # 8 x 4 MiB L2 + 256 MiB Infinity Cache (MI355X_MICROARCH.md): a scene below this is served from cache, not HBM
TRAVERSAL_KERNEL_ANY = {1: "k_trace_any", 2: "k_trace2", 3: "k_trace3", 4: "k_trace8", 5: "k_trace_leaf", 6: "k_trace4", 7: "k_trace3"}


def build_workload(T, name: str, res: int):
    if name == "cornell":
        return T.scenes.cornell_scene(), T.scenes.cornell_camera(res), "S-cornell: 2 spheres + 10 triangles, PointLight"
    if name == "cornell_walls":  # profiling aid: one material, one shape kind
        return T.scenes.cornell_scene(False), T.scenes.cornell_camera(res), "S-cornell without the spheres: 10 matte triangles, PointLight"
    if name == "shadows":
        return T.scenes.shadows_scene(), T.scenes.shadows_camera(res), "docs/src/shadows.md scene: 4 spheres + 4 triangles, PointLight"
    if name in T.scenes.MESH_N:
        n = T.scenes.MESH_N[name]
        return T.scenes.mesh_scene(n), T.scenes.cornell_camera(res), f"S-mesh: Cornell box (10 wall triangles, mirror + glass sphere) + {2 * n * n} triangle height field"
    if name == "blob_870k":
        return T.scenes.blob_scene(270), T.scenes.cornell_camera(res), "S-blob: Cornell walls + a closed bumpy object of ~870 k triangles (stand-in for configs[2], Dragon in Cornell box)"
    if name == "caustic":
        return T.scenes.caustic_scene(caustic_model()), T.scenes.caustic_camera(res), "S-caustic: docs/code/caustic_glass.jl (caustic-glass.ply, 88 064 triangles, on a plastic floor, SpotLight)"
    raise SystemExit(f"unknown workload {name}")


def caustic_model() -> str:
    """The reference's own mesh asset (tests/golden/caustic-glass.ply, placed there by make_caustic_ply.py); the procedural goblet if absent."""
    p = os.path.join(ROOT, "tests", "golden", "caustic-glass.ply")
    return p if os.path.exists(p) else ""


def kernel_bytes(st, film_px: int):
    """Algorithmic bytes per kernel class for one frame, from the records the kernels touch:
    traversal: per ray 32 B (o, d) in + 16 B hit out (any-hit: 16 B contribution in, 16 B radiance read-modify-write counted once)
        + node_bytes per counted node + 48 B per primitive fetched; a one-leaf scene (traversal 5) reads its primitives through
        scalar loads: ray + hit only.
    shade: per path vertex 64 B (o, d, beta, hit) + the hit slot's 128-byte shading record (vertices, normals, triangle constants) read,
        48 B next ray + 48 B shadow ray written = 288 B;
    raygen: 48 B (o, d, beta) written per camera sample; film: 24 B (radiance + film position) per camera sample + 16 B per film pixel."""
    nb = int(st.node_bytes)
    leaf = int(st.traversal) == 5 or (int(st.traversal) == 9 and nb == 0)
    return {
        "trace_closest": st.closest_rays * 48 + (0 if leaf else st.nodes_visited * nb + st.prims_tested * 48),
        # hybrid mode: the part of trace_closest that is the fallback walks' (k_trace3 on the canonical tree: its rays read again, its nodes are always 32 B per box)
        "trace_fallback": st.fallback_rays * 48 + st.nodes_visited_fallback * 32 + st.prims_tested_fallback * 48,
        "trace_any": st.shadow_rays * 64 + (0 if leaf else st.nodes_visited_shadow * nb + st.prims_tested_shadow * 48),
        "shade": st.closest_rays * 288,
        "film": st.camera_samples * 24 + film_px * 16,
        "raygen": st.camera_samples * 48,
    }


def pmc_child_runs(args, kernel_prefix: str, want_any: bool, passes):
    """Counters of the dominant kernel from separate `rocprofv3 --pmc` child runs of this same command (one step, no baselines; a PMC pass
    never shares a run with a trace domain).  `passes` = lists of counter names, one child run each.  Returns {counter: average per
    dispatch of the kernels whose name contains kernel_prefix}, or None for a pass that failed / when rocprofv3 is absent."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None
    totals = {}
    for counters in passes:
        out = tempfile.mkdtemp(prefix="trhip_pmc_", dir="/tmp")
        cmd = [exe, "--pmc"] + list(counters) + ["--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", "1", "--warmup", "0",
               "--workload", args.workload, "--res", str(args.res), "--spp", str(args.spp), "--depth", str(args.depth), "--seed", str(args.seed), "--traversal", str(args.traversal),
               "--iterations", str(args.iterations), "--radius", str(args.radius),
               "--no-cpu-baseline", "--no-traffic", "--no-micro", "--no-visits", "--no-modes"] + [a for kv in args.opt for a in ("--opt", kv)]
        try:
            res = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=900)
            if res.returncode != 0:
                err = res.stderr.decode(errors="replace")
                keep = [l for l in err.splitlines() if any(k in l for k in ("rror", "Traceback", "trhip", "HIP", "assert", "File \""))]
                raise RuntimeError(f"rc {res.returncode}: " + " | ".join(keep[-12:])[-1500:])
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            acc, n = {}, {}
            for r in csv.DictReader(open(files[0])):
                name = r["Kernel_Name"]
                if kernel_prefix not in name:
                    continue
                if kernel_prefix.startswith("k_trace") and kernel_prefix not in ("k_trace_closest", "k_trace_any") and (("<true" in name) != want_any):
                    continue
                c = r["Counter_Name"]
                acc[c] = acc.get(c, 0.0) + float(r["Counter_Value"])
                n[c] = n.get(c, 0) + 1
            for c in acc:
                totals[c] = acc[c] / n[c]
        except Exception as e:  # the line then carries traffic / valu = null; say why on stderr
            print(f"[bench] PMC pass {list(counters)} failed: {type(e).__name__}: {str(e)[-700:]}", file=sys.stderr, flush=True)
        finally:
            shutil.rmtree(out, ignore_errors=True)
    return totals or None


# bytes that left L2 per unit of FETCH_SIZE (KiB) for the dominant kernel's access pattern.  MI355X_MICROARCH.md gives x2 for 16 B / lane coalesced
# streaming reads (128-byte requests tallied at 64 B) and leaves other widths to the user; tools/calib/fetch_calib.hip measured the per-lane 64- /
# 48- / 128-byte record gathers of the traversal and shading kernels on a 1 GiB table read exactly once (profiles/r3/fetch_calib_summary.txt).
# Measured (profiles/r3/r3a_fetch_calib_summary.txt): bytes / (FETCH_SIZE x 1024) = 2.000 for 16 B / lane streams and 128-byte records, 1.000 for 64-byte record gathers
# (one 64-byte request each: FETCH_SIZE tallies REQUESTS at 64 B), 0.590 for 48-byte records (they straddle request boundaries).  The traversal kernels' reads are
# 64-byte node gathers (1 510 of ~1 760 B per ray on S-mesh), 48-byte primitive records and a 32-byte ray stream: factor 1.0.  When the request-size counters
# (TCC_EA0_RDREQ_{32B,64B,128B}) are available the bench uses their exact byte count instead of any factor.
FETCH_FACTOR = {"default": 2.0, "k_trace3": 1.0, "k_trace3c": 1.0, "k_trace3c4": 1.0, "k_trace2": 1.0, "k_trace7": 1.0, "k_trace4": 1.0, "k_trace_closest": 1.0, "k_sppm_gather": 1.0}


def measure_counters(args, kernel_prefix: str, want_any: bool):
    """roofline.traffic (HBM-side bytes per launch of the dominant kernel: FETCH_SIZE and WRITE_SIZE need separate passes — TCC slots) and
    roofline.valu (lanes per VALU instruction = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU, VALU busy) from three PMC child runs."""
    t = pmc_child_runs(args, kernel_prefix, want_any, [["FETCH_SIZE"], ["WRITE_SIZE"], ["SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_INSTS_VALU", "VALUBusy"],
                                                       ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"]])
    if not t:
        return None, None
    traffic = None
    if "FETCH_SIZE" in t and "WRITE_SIZE" in t:
        f = FETCH_FACTOR.get(kernel_prefix, FETCH_FACTOR["default"])
        traffic = {"bytes": int(f * t["FETCH_SIZE"] * 1024.0 + t["WRITE_SIZE"] * 1024.0), "fetch_factor": f, "FETCH_SIZE_KiB": round(t["FETCH_SIZE"], 1), "WRITE_SIZE_KiB": round(t["WRITE_SIZE"], 1)}
        n = [t.get("TCC_EA0_RDREQ_32B_sum"), t.get("TCC_EA0_RDREQ_64B_sum"), t.get("TCC_EA0_RDREQ_128B_sum")]
        if all(v is not None for v in n) and t.get("TCC_EA0_RDREQ_sum"):
            # the L2's memory-side read requests by size: exact read bytes whatever the access pattern (no factor).  On parts where the 64-byte counter is the total
            # (32- and 128-byte requests counted in it too) the split does not add up to the total and the factor stays
            split = n[0] + n[1] + n[2]
            if 0.98 <= split / t["TCC_EA0_RDREQ_sum"] <= 1.02:
                rd = 32.0 * n[0] + 64.0 * n[1] + 128.0 * n[2]
                traffic.update({"bytes": int(rd + t["WRITE_SIZE"] * 1024.0), "read_requests": {"32B": int(n[0]), "64B": int(n[1]), "128B": int(n[2])}, "fetch_factor": round(rd / (t["FETCH_SIZE"] * 1024.0), 3),
                                "source": "TCC_EA0_RDREQ_{32B,64B,128B} x size + WRITE_SIZE"})
    valu = None
    if t.get("SQ_ACTIVE_INST_VALU"):
        valu = {"lanes_per_valu_inst": round(t["SQ_THREAD_CYCLES_VALU"] / t["SQ_ACTIVE_INST_VALU"], 2), "of": 64,
                "wave_valu_insts_per_launch": int(t.get("SQ_INSTS_VALU", 0))}
        if t.get("VALUBusy") is not None:  # rocprofv3's derived metric (gfx94x formula: ROCm 7.2 ships no gfx950 section), in percent
            valu["valu_busy"] = round(min(1.0, t["VALUBusy"] / 100.0), 3)
    return traffic, valu


def classify_bound(roofline):
    """Which resource the evidence says bounds the dominant kernel.  `frac` stays SURVEY §8(d)'s request-rate figure; the label comes from counters:
    VALU >= 80 % busy -> "valu-issue" (with lanes per instruction beside it); else HBM-side bytes >= 60 % of peak -> "hbm"; else "latency"."""
    v = roofline.get("valu") or {}
    if v.get("valu_busy", 0.0) >= 0.8:
        return "valu-issue"
    if roofline.get("frac_counters", 0.0) >= 0.6:
        return "hbm"
    if v or roofline.get("frac_counters") is not None:
        return "latency"
    return "hbm"  # no counters available: SURVEY §8(d)'s nominal roofline


def micro_benchmark(args, T, ctx, flat, osc):
    """BASELINE.md §2 leg 3 / SURVEY.md §8d: 2^24 incoherent rays (origins uniform in the scene bound, directions uniform on the
    sphere, t_max = Inf, seed 0x5EED0002) against the same BVH: closest-hit and any-hit kernels alone on device-resident rays, and
    the CPU restatement (all host cores) on a 2^21-ray subset of the same set."""
    import ctypes as C
    import numpy as np
    import torch
    n = 1 << args.micro_log2
    bnd = flat.bvh()[0][0]
    rays = T.scenes.incoherent_rays(n, bnd[:3], bnd[3:])
    d_rays = torch.from_numpy(rays).cuda()
    d_hits = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    d_occ = torch.empty(n, dtype=torch.uint8, device="cuda")
    L = T.lib()
    out = {"rays": n, "set": "incoherent: uniform origins in the scene bound, uniform directions, t_max = Inf, seed 0x5EED0002"}
    ms = C.c_double()
    for name, fn, buf in (("closest", L.trhip_trace_closest_device, d_hits), ("any", L.trhip_trace_any_device, d_occ)):
        ctx.check(fn(ctx._h, flat._h, C.c_void_p(d_rays.data_ptr()), n, C.c_void_p(buf.data_ptr()), 1, C.byref(ms)))  # warm-up
        ctx.check(fn(ctx._h, flat._h, C.c_void_p(d_rays.data_ptr()), n, C.c_void_p(buf.data_ptr()), 3, C.byref(ms)))
        out[f"gpu_{name}_Mray_s"] = round(n / (ms.value * 1e-3) / 1e6, 1)
        out[f"gpu_{name}_ms"] = round(ms.value, 3)
    if osc is not None:
        m = min(n, 1 << 21)
        sub = rays[:m]
        t1 = time.perf_counter()
        t, prim, _, counts = osc.trace_closest(sub)
        dt = time.perf_counter() - t1
        out["cpu_closest_Mray_s"] = round(m / dt / 1e6, 3)
        out["cpu_nodes_per_ray"] = round(float(counts[0]) / m, 1)
        out["cpu_prims_per_ray"] = round(float(counts[1]) / m, 2)
        t1 = time.perf_counter()
        occ, _ = osc.trace_any(sub)
        out["cpu_any_Mray_s"] = round(m / (time.perf_counter() - t1) / 1e6, 3)
        out["cpu_rays"] = m
        hits = d_hits[:m].cpu().numpy()
        out["gpu_equals_cpu_on_subset"] = bool(np.array_equal(hits[:, 1].view(np.int32), prim) and np.array_equal(hits[:, 0].view(np.uint32), t.view(np.uint32))
                                               and np.array_equal(d_occ[:m].cpu().numpy(), occ))
    return out


def run_sppm(args, T, ctx, graft, rank, world, comm_ok):
    """BASELINE.json configs[3]: docs/code/caustic_glass.jl with SPPMIntegrator, 1024x1024, 100 iterations, depth 8.  A step = one whole
    SPPMIntegrator call; rays = camera + shadow + photon rays.  N > 1: the photon pass is sharded inside the library (strong scaling)."""
    import torch
    import torch.distributed as dist
    scene, cam = T.scenes.caustic_scene(caustic_model()), T.scenes.caustic_camera(args.res)
    t0 = time.time()
    flat = scene.flatten(ctx)
    t_build = time.time() - t0
    integ = T.SPPMIntegrator(cam, args.radius, args.depth, args.iterations, -1, seed=args.seed)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        integ.render(scene, ctx)
    sync()
    t_start = time.perf_counter()
    rays, replicated, ms, launches = 0, 0, {k: 0.0 for k in KERNEL_CLASSES}, {}
    sub_ms, sub_launches = [0.0] * 4, [0] * 4
    for _ in range(args.steps):
        integ.render(scene, ctx)
        st = integ.stats
        rays += st.closest_rays + st.shadow_rays
        replicated += st.replicated_rays
        for k in ms:
            ms[k] += getattr(st, "ms_" + k)
            launches[k] = launches.get(k, 0) + getattr(st, "launches_" + k)
        for j in range(4):
            sub_ms[j] += st.ms_sub[j]
            sub_launches[j] += st.launches_sub[j]
    sync()
    elapsed = time.perf_counter() - t_start
    tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    # only the photon pass is sharded: every rank repeats the camera pass (closest-hit + shadow rays).  Useful rays of the job = the ranks'
    # own (photon) rays + the camera pass ONCE (rank 0's count; identical on every rank)
    cnt = torch.tensor([float(rays - replicated)], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    elapsed, rays = float(tmax.item()), float(cnt.item()) + float(replicated)
    if rank != 0:
        return None
    ctx.set_option("count_visits", 1)
    if world == 1:
        integ.render(scene, ctx)
    sv = integ.stats
    ctx.set_option("count_visits", 0)
    # the per-class HIP events the roofline needs are ~12 records per iteration on the launching stream: what the same frames take without them (not `value`: the timed
    # region above is the one the roofline's launch durations come from)
    untimed = None
    if world == 1 and not any(kv.replace(" ", "").startswith("timing=") for kv in args.opt):
        ctx.set_option("timing", 0)
        try:
            integ.render(scene, ctx)
            sync()
            t_u = time.perf_counter()
            n_u = max(1, min(3, args.steps))
            for _ in range(n_u):
                integ.render(scene, ctx)
            sync()
            untimed = {"ms_per_step": round((time.perf_counter() - t_u) / n_u * 1e3, 3), "steps": n_u, "note": "option timing = 0: the library records no per-class events (the default for a caller that passes no stats)"}
        finally:
            ctx.set_option("timing", 1)
    kb = kernel_bytes(sv, 0)
    per_step = {"raygen+photon_gen": ms["raygen"] / args.steps, "trace_closest": ms["trace_closest"] / args.steps, "photon_gather": sub_ms[0] / args.steps,
                "camera+photon_shading": sub_ms[1] / args.steps, "grid+bin+scan": sub_ms[2] / args.steps, "fold+pixel_update": sub_ms[3] / args.steps,
                "trace_any": ms["trace_any"] / args.steps, "image": ms["film"] / args.steps}
    dominant = max((k for k in per_step if k != "trace_any"), key=lambda k: per_step[k])
    n_px = args.res * args.res
    if dominant == "photon_gather":
        # k_sppm_gather + k_sppm_gather_hot of one iteration = one "launch" (DESIGN.md §10).  Algorithmic bytes: per pixel with a visible point 16 B position +
        # 4 B radius + 16 B beta + 80 B frame (wo, ng, ns, ss, ts) read, 16 B (phi, M) written; 8 B of bucket bounds per cell visited;
        # 16 B per candidate distance-tested (sorted photon position + record index); 32 B per accepted pair (photon direction + beta)
        c = sv.count_sub
        it = max(1, args.iterations)
        dom_bytes = ((c[3] * 132 + c[0] * 16 + c[1] * 32) / it) if c[0] else 0
        dom_ms = sub_ms[0] / max(1, sub_launches[0])
        dom_launches = sub_launches[0]
        kname, kprefix = "k_sppm_gather + k_sppm_gather_hot", "k_sppm_gather"
        extra = {"per_iteration": {"visible_points": int(c[3] / it), "candidates": int(c[0] / it), "accepted_pairs": int(c[1] / it)} if c[0] else None}
    else:
        dom_bytes = kb["trace_closest"] * (args.steps if world == 1 else 0) / max(1, launches["trace_closest"])
        dom_ms = ms["trace_closest"] / max(1, launches["trace_closest"])
        dom_launches = launches["trace_closest"]
        kprefix = flat.closest_kernel_name()  # (the library's own launch decision for this scene: a one-leaf accelerator runs k_trace_leaf_c, wide4 = 0 k_trace3c)
        kname = kprefix + "<closest>"
        extra = {}
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    roofline = {"bound": "hbm", "kernel": kname, "dominant_class": dominant, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                "avg_launch_ms": round(dom_ms, 4), "launches": dom_launches, "algorithmic_bytes_per_launch": int(dom_bytes),
                "visits_per_ray": {"closest_nodes": round(sv.nodes_visited / max(1, sv.closest_rays), 2), "closest_prims": round(sv.prims_tested / max(1, sv.closest_rays), 2),
                                   "node_bytes": int(sv.node_bytes)},
                "kernel_ms_per_step": {k: round(v, 3) for k, v in per_step.items()}}
    roofline.update(extra)
    if world == 1 and not args.no_traffic:
        traffic, valu = measure_counters(args, kprefix, False)
        if traffic:
            n_k = 2 if dominant == "photon_gather" else 1  # two kernels per "launch" of the gather class
            roofline["traffic"] = traffic["bytes"] * n_k
            roofline["traffic_detail"] = traffic
            roofline["achieved_counters"] = round(roofline["traffic"] / (dom_ms * 1e-3) / 1e9, 2)
            roofline["frac_counters"] = round(roofline["achieved_counters"] / HBM_PEAK_GBS, 5)
        if valu:
            roofline["valu"] = valu
    roofline["bound"] = classify_bound(roofline)
    cpu = None
    if world == 1 and not args.no_cpu_baseline:  # the oracle with its photon loop threaded like sppm.jl:334, on a bounded number of iterations
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        graft.build_oracle()
        import oracle_bridge as ob
        osc = ob.OracleScene.from_scene(scene, bvh=flat.bvh())
        threads = ob.lib().orc_num_threads()
        n_it = max(1, min(args.iterations, 2))
        t1 = time.perf_counter()
        r = osc.sppm(cam, args.radius, args.depth, n_it, -1, seed=args.seed, threads=threads)
        dt = time.perf_counter() - t1
        cpu = {"value": round((r["stats"].closest_rays + r["stats"].shadow_rays) / dt / 1e6, 3), "unit": "Mray/s", "cores": threads, "kind": "port",
               "sample": f"{n_it} of {args.iterations} iterations of the same configuration ({dt:.1f} s)"}
    info = integ.state()["info"]
    # ---- full-size witness (C4 at its literal size): the same run with every ray on the canonical tree in the reference's order (option hybrid = 0) against the default's.
    #      M, N, radius, Ld and the visible points are integers / Float32 values that do not depend on the order of the photon atomics: they must be EQUAL; phi / tau / the image
    #      are Float32 sums of the same terms in an order that differs from run to run (the reference's own are unordered atomics, sppm.jl:398-399): largest relative difference ----
    witness = None
    if world == 1 and not args.no_modes and flat.bvh_mode()[0] == 2:
        import numpy as np
        img_h = integ.render(scene, ctx).copy()
        st_h = {k: v.copy() for k, v in integ.state().items() if k != "info"}
        ctx.set_option("hybrid", 0)
        try:
            ref_i = T.SPPMIntegrator(cam, args.radius, args.depth, args.iterations, -1, seed=args.seed)
            img_r = ref_i.render(scene, ctx)
            st_r = ref_i.state()
        finally:
            ctx.set_option("hybrid", 1)
        exact = {k: bool(np.array_equal(st_h[k].view(np.uint8), st_r[k].view(np.uint8))) for k in ("M", "N", "radius", "Ld", "vp_p", "vp_beta")}
        rel = lambda a, b: float(np.max(np.abs(a - b)) / max(1e-30, float(np.max(np.abs(b)))))
        witness = {"vs": "the same run with option hybrid = 0 (every ray on the reference's tree in the reference's order), at this line's full size",
                   "equal_bit_for_bit": exact, "all_equal": all(exact.values()),
                   "largest_relative_difference": {"phi": rel(st_h["phi"], st_r["phi"]), "tau": rel(st_h["tau"], st_r["tau"]), "image": rel(img_h, img_r)},
                   "tolerance_of_the_tests": {"phi": 2e-5, "tau": 5e-5, "image": 1e-4}}
    result = {"metric": "Mray/s (all bounces)", "value": round(rays / elapsed / 1e6, 2), "unit": "Mray/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
              "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong" if world > 1 else "weak", "vs_baseline": None,
              "dtype": "f32 (f64 pixel update)", "data": "synthetic",
              "config": {"workload": f"caustic_sppm: docs/code/caustic_glass.jl ({'caustic-glass.ply' if caustic_model() else 'procedural goblet'}, {flat.bvh()[3].size} primitives, SpotLight), "
                                     f"SPPMIntegrator, {args.res}x{args.res}, {args.iterations} iterations, {info['photons_per_iteration']} photons per iteration, max depth {args.depth}, "
                                     f"radius {args.radius}, seed {args.seed:#x}",
                         "rays_per_step": int(rays / args.steps), "ms_per_iteration": round(elapsed / args.steps / args.iterations * 1e3, 3), "bvh_build_upload_s": round(t_build, 3),
                         "bvh": BVH_MODE.get(flat.bvh_mode()[0], "?"), "fallback_fraction": round(sv.fallback_rays / max(1, sv.closest_rays), 5), "traversal": int(sv.traversal),
                         "parallelism": f"photon indices sharded x{world}, camera pass replicated (its rays counted once), one RCCL all-reduce of phi / M per iteration inside libtracehip" if world > 1 else "single GPU",
                         "rccl_ranks": ctx.comm_rank()[1]},
              "roofline": roofline, "cpu_baseline": cpu,
              "parity": {"full_size_witness": witness,
                         "where": "tests/test_gpu_sppm.py, tests/test_gpu_baseline_configs.py (C4 at its literal size by properties), tools/soak_sppm.py; the oracle is too slow for 100 iterations at 1024^2"}}
    if untimed:
        result["without_class_timers"] = untimed
    print(json.dumps(result), flush=True)
    return result


def require_communicator(T, ctx, rank, world, device):
    """The job's communicator, inside the library (include/tracehip.h "multi-GPU"): rank 0 makes the RCCL id, torch.distributed carries it.  An N-rank bench line must
    come from the library's own RCCL communicator with N ranks (trhip_comm_rank), or not at all: when it is not up on EVERY rank the ranks agree on that (one MIN
    all-reduce over the launcher's group), say why on stderr and leave with exit code 3 — no torch.distributed stand-in, no line.  Returns True when world > 1 and
    the communicator is up (tests/test_sharding_gloo.py drives this function with two CPU processes and a context whose communicator has one rank)."""
    if world <= 1:
        return False
    import torch
    import torch.distributed as dist
    comm_ok, err = False, ""
    try:
        job = T.parallel.Job(ctx, rank, world)
        comm_ok = job.ok and ctx.comm_rank()[1] == world
    except Exception as e:  # noqa: BLE001
        err = str(e)
    flag = torch.tensor([1 if comm_ok else 0], device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if not bool(flag.item()):
        sys.stderr.write(f"[bench] rank {rank}: the library's RCCL communicator is not up on every rank ({err or ('it has ' + str(ctx.comm_rank()[1]) + ' rank(s)' if not comm_ok else 'another rank failed')}); no bench line\n")
        dist.barrier()
        dist.destroy_process_group()
        raise SystemExit(3)
    return True


def hbm_resident_record(args):
    """The roofline statement on the one workload where HBM is real: the 10.5 M-triangle scene — ~1.2 GB of nodes and primitives, far beyond L2 + the
    256 MiB MALL — rendered by a CHILD run of this script (2 steps, its own PMC passes), after this process has released the GPU.  Returns the child's roofline figures
    (request rate and counter rate against the HBM peak, lanes per VALU instruction, VALU busy) or {"error": ...}."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--workload", "mesh_10m", "--steps", "2", "--warmup", "1", "--res", str(args.res), "--spp", str(args.spp), "--depth", str(args.depth),
           "--seed", str(args.seed), "--no-cpu-baseline", "--no-micro", "--no-modes", "--no-hbm-resident"] + [a for kv in args.opt for a in ("--opt", kv)]
    t0 = time.perf_counter()
    try:
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
        line = [l for l in res.stdout.decode(errors="replace").splitlines() if l.startswith("{\"metric\"")]
        if res.returncode != 0 or not line:
            return {"error": f"child run rc {res.returncode}: " + res.stderr.decode(errors="replace")[-400:]}
        d = json.loads(line[-1])
        r = d.get("roofline") or {}
        return {"workload": d["config"]["workload"], "why": "nodes + primitives exceed L2 + MALL: the request rate and the counter rate are both HBM figures here",
                "value_Mray_s": d["value"], "ms_per_step": d["ms_per_step"], "steps": d["steps"], "bvh": d["config"].get("bvh"), "bvh_note": d["config"].get("bvh_note"),
                "kernel": r.get("kernel"), "avg_launch_ms": r.get("avg_launch_ms"), "scene_bytes": r.get("scene_bytes"), "scene_fits_l2_plus_mall": r.get("scene_fits_l2_plus_mall"),
                "frac_requests": r.get("frac_requests"), "frac_counters": r.get("frac_counters"), "achieved_requests_GBps": round((r.get("frac_requests") or 0.0) * HBM_PEAK_GBS, 1),
                "achieved_counters_GBps": r.get("achieved_counters"), "valu": r.get("valu"), "valu_frac": r.get("valu_frac"), "bound": r.get("bound"),
                "kernel_ms_per_step": r.get("kernel_ms_per_step"), "child_run_s": round(time.perf_counter() - t0, 1)}
    except Exception as e:
        return {"error": f"{type(e).__name__}: {str(e)[-300:]}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="mesh_1m")
    ap.add_argument("--res", type=int, default=1024)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x5EED0001)
    ap.add_argument("--scaling", choices=("weak", "strong"), default=None, help="default: strong when --gpus > 1 (one frame's samples split over the ranks), weak (= the frame) on one GPU")
    ap.add_argument("--traversal", type=int, default=0, help="traversal kernel (0 = the library's default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-spp", type=int, default=0, help="spp of the bounded CPU-baseline sample (0 = auto)")
    ap.add_argument("--no-traffic", action="store_true", help="skip the two rocprofv3 --pmc child runs that fill roofline.traffic")
    ap.add_argument("--no-micro", action="store_true", help="skip the 2^24-incoherent-ray traversal micro-benchmark")
    ap.add_argument("--no-visits", action="store_true", help="skip the untimed instrumented pass (visit counts; no roofline then)")
    ap.add_argument("--micro-log2", type=int, default=24)
    ap.add_argument("--iterations", type=int, default=100, help="caustic_sppm: SPPM iterations per step")
    ap.add_argument("--radius", type=float, default=0.075, help="caustic_sppm: initial search radius")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="library option (trhip_set_option) set before the scene is committed; repeatable")
    ap.add_argument("--no-hbm-resident", action="store_true", help="skip the short child run on the 10.5 M-triangle scene (`hbm_resident`: the one workload whose nodes + primitives exceed L2 + MALL)")
    ap.add_argument("--no-modes", action="store_true", help="skip the two short comparison runs on the library's tree alone and on the reference's tree alone (`bvh_modes`)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.scaling is None:
        args.scaling = "strong" if world > 1 else "weak"
    if world != args.gpus and world == 1 and args.gpus > 1:
        raise SystemExit("launch multi-GPU runs with torch.distributed.run (one process per GPU)")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    import __graft_entry__ as graft
    if rank == 0:
        graft.build_library()  # only rank 0 builds (the others wait at the barrier): no write race on the shared object
    if world > 1:
        dist.barrier()
    T = graft.load_package()
    ctx = T.Context(local_rank)
    if args.traversal:
        ctx.set_option("traversal", args.traversal)
    for kv in args.opt:
        name, _, value = kv.partition("=")
        ctx.set_option(name, int(value))

    comm_ok = require_communicator(T, ctx, rank, world, "cuda")

    if args.workload == "caustic_sppm":
        r = run_sppm(args, T, ctx, graft, rank, world, comm_ok)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return r
    scene, cam, desc = build_workload(T, args.workload, args.res)
    t0 = time.time()
    flat = scene.flatten(ctx)  # BVH build + upload: outside the timed region
    t_build = time.time() - t0
    h, w = cam.film.size
    film = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")

    reduce_s = [0.0]  # wall time this rank spent inside trhip_film_reduce (includes waiting for the slowest rank)

    def shard(mode):
        """(spp of this rank, first global sample index) — weak: every rank `spp`; strong: the frame's `spp` split over the ranks."""
        if mode == "weak" or world == 1:
            return args.spp, T.parallel.shard_sample_offset(rank, args.spp)
        return T.parallel.shard_samples(args.spp, rank, world)

    def make_step(mode):
        spp_r, off = shard(mode)
        integ = T.PathIntegrator(cam, T.SeededSampler(max(1, spp_r), seed=args.seed, sample_offset=off), args.depth)

        def step():
            if spp_r > 0:
                integ.render(scene, ctx, device_out=film.data_ptr())
            else:
                film.zero_()
            if world > 1:  # Film pixels are additive (film.jl:161-162, 190-191): one sum-reduce ends the frame (ncclReduce inside the library; blocking)
                t_r = time.perf_counter()
                ctx.film_reduce(film.data_ptr(), h * w, 0)
                reduce_s[0] += time.perf_counter() - t_r
            return integ.stats
        return step, integ

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(mode, steps, warmup):
        step, integ = make_step(mode)
        for _ in range(warmup):
            step()
        sync()
        t_start = time.perf_counter()
        agg = {"samples": 0, "ms": {k: 0.0 for k in KERNEL_CLASSES}, "launches": {}, "closest": 0, "shadow": 0, "fallback": 0, "ms_fallback": 0.0, "launches_fallback": 0}
        reduce_s[0] = 0.0
        for _ in range(steps):
            st = step()
            agg["closest"] += st.closest_rays
            agg["shadow"] += st.shadow_rays
            agg["samples"] += st.camera_samples
            agg["fallback"] += st.fallback_rays
            agg["ms_fallback"] += st.ms_fallback
            agg["launches_fallback"] += st.launches_fallback
            for k in agg["ms"]:
                agg["ms"][k] += getattr(st, "ms_" + k)
                agg["launches"][k] = agg["launches"].get(k, 0) + getattr(st, "launches_" + k)
        sync()
        elapsed = time.perf_counter() - t_start
        agg["film_reduce_ms_per_step"] = reduce_s[0] / max(1, steps) * 1e3
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        counts = torch.tensor([agg["closest"] + agg["shadow"], agg["samples"]], dtype=torch.float64, device="cuda")
        if world > 1:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dist.all_reduce(counts, op=dist.ReduceOp.SUM)
        return float(tmax.item()), float(counts[0].item()), float(counts[1].item()), agg, integ

    elapsed, total_rays, total_samples, agg, integ = timed(args.scaling, args.steps, args.warmup)
    other = None
    if world > 1:  # the other scaling mode, right after (fewer steps): both values in one line
        omode = "strong" if args.scaling == "weak" else "weak"
        osteps = max(1, min(args.steps, 5))
        oe, orays, osamples, oagg, _ = timed(omode, osteps, 1)
        other = {"scaling": omode, "value": round(orays / oe / 1e6, 2), "unit": "Mray/s", "steps": osteps, "ms_per_step": round(oe / osteps * 1e3, 3), "spp_per_gpu": shard(omode)[0],
                 "Msample_per_s": round(osamples / oe / 1e6, 3), "film_reduce_ms_per_step": round(oagg["film_reduce_ms_per_step"], 3)}

    result = None
    if rank == 0:
        steps = args.steps
        roofline = None
        sv = integ.stats
        bvh_mode, n_acc_nodes, _ = flat.bvh_mode()
        bvh_note = flat.bvh_note()
        _bvh = flat.bvh()
        n_canonical_nodes = int(_bvh[1].size)
        n_scene_bytes = 32 * (n_acc_nodes if bvh_mode in (2, 3) else n_canonical_nodes) + 48 * int(_bvh[3].size)  # what the dominant walk reads: its tree's boxes + the primitive records
        modes = None
        if not args.no_visits:
            # ---- roofline of the dominant kernel (rank 0's launches): live HIP-event durations from the timed region, bytes from an
            #      untimed instrumented pass of the same frame (node / primitive visit counts) ----
            ctx.set_option("count_visits", 1)
            integ.render(scene, ctx, device_out=film.data_ptr())
            sv = integ.stats
            ctx.set_option("count_visits", 0)
            per_step = kernel_bytes(sv, h * w)
            hybrid = int(sv.traversal) == 9
            fb_bytes = per_step.pop("trace_fallback")
            # the shadow rays of depth d run on a second, low-priority stream beside the closest-hit rays of depth d+1: their HIP-event
            # time is wall time under contention, not the kernel's own — never the dominant kernel of these workloads; left out of the choice
            dominant = max((k for k in agg["ms"] if k != "trace_any"), key=lambda k: agg["ms"][k])
            dom_ms = agg["ms"][dominant] / max(1, agg["launches"][dominant])
            dom_bytes = per_step[dominant] * steps / max(1, agg["launches"][dominant])
            # (the closest-hit kernel's name comes from the library — launch_trace's own decision for this scene under the current options —, not from the option string)
            kname = {"trace_closest": flat.closest_kernel_name(), "shade": "k_shade_path", "film": "k_film_gather", "raygen": "k_raygen"}[dominant]
            if hybrid and dominant == "trace_closest":
                # a hybrid closest-hit launch = the certified walk on the accelerator tree (k_trace3c4 or k_trace3c, the dominant kernel) + the reference-order walk of the rays it hands back
                # (k_trace3 on the canonical tree); the library times the hand-over inside every launch (trhip_stats.ms_fallback): the roofline is k_trace3c's alone
                dom_ms = (agg["ms"]["trace_closest"] - agg["ms_fallback"]) / max(1, agg["launches"]["trace_closest"])
                dom_bytes = (per_step["trace_closest"] - fb_bytes) * steps / max(1, agg["launches"]["trace_closest"])
            achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
            kprefix = kname
            gbps = {k: round(per_step[k] / (agg["ms"][k] / steps * 1e-3) / 1e9, 1) if agg["ms"][k] > 0 else None for k in per_step}
            roofline = {"bound": "hbm", "kernel": kname + ("<closest>" if dominant == "trace_closest" else ""), "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None, "avg_launch_ms": round(dom_ms, 4), "launches": agg["launches"][dominant],
                        "algorithmic_bytes_per_launch": int(dom_bytes),
                        "frac_requests": round(achieved / HBM_PEAK_GBS, 5),
                        "frac_note": "frac_requests = ALGORITHMIC bytes (what the kernel asks the memory system for: rays, hits, every node and primitive fetch — SURVEY 8(d)) / launch time / HBM "
                                     "peak: a request rate, L2 and the 256 MiB MALL serve part of it; frac_counters = bytes that left L2 (rocprofv3 --pmc) / launch time / HBM peak.  `frac` is the "
                                     "counter figure whenever the scene's nodes + primitives fit L2 + MALL (then the request rate says nothing about HBM), else the request rate",
                        "visits_per_ray": {"closest_nodes": round(sv.nodes_visited / max(1, sv.closest_rays), 2), "closest_prims": round(sv.prims_tested / max(1, sv.closest_rays), 2),
                                           "shadow_nodes": round(sv.nodes_visited_shadow / max(1, sv.shadow_rays), 2), "shadow_prims": round(sv.prims_tested_shadow / max(1, sv.shadow_rays), 2),
                                           "node_bytes": int(sv.node_bytes), "traversal": int(sv.traversal)},
                        "kernel_ms_per_step": {k: round(v / steps, 3) for k, v in agg["ms"].items()},
                        "kernel_ms_note": "HIP-event time per kernel class on its own stream; option overlap (default 1) puts the shadow rays of depth d on a second stream beside the closest-hit rays of depth d + 1: class times are wall times under that contention and add up to more than the frame",
                        "kernel_GBps_note": "algorithmic bytes / class time; the traversal classes count REQUESTS (32 B per box tested, 48 B per primitive fetched): L2 and MALL serve part of them, so they may exceed the HBM peak — achieved_counters / frac_counters is what left L2",
                        "kernel_GBps": gbps}
            if hybrid:
                roofline["hybrid"] = {"certified_walk_ms_per_step": round((agg["ms"]["trace_closest"] - agg["ms_fallback"]) / steps, 3), "fallback_walk_ms_per_step": round(agg["ms_fallback"] / steps, 3),
                                      "fallback_launches": agg["launches_fallback"], "fallback_nodes_per_fallback_ray": round(sv.nodes_visited_fallback / max(1, sv.fallback_rays), 1),
                                      "fallback_why": dict(zip(("direction", "sphere", "near_tie_or_guard"), [int(x) for x in sv.count_sub[:3]])),
                                      "note": "closest-hit launch = the certified walk (k_trace3c4, four-wide; k_trace3c under wide4 = 0) on the accelerator tree (the library's SAH tree, walked under the order-independence certificate) + k_trace3 on the "
                                              "canonical tree (the reference's own) over the rays handed back; kernel_ms_per_step.trace_closest is both"}
            over = [k for k, v in gbps.items() if v is not None and v > HBM_PEAK_GBS and not k.startswith("trace")]
            # no line may carry a fraction above 1 without saying so: a request rate above the HBM peak means L2 / MALL serve part of the requests
            roofline["byte_models_within_peak"] = not over and roofline["frac_requests"] <= 1.0
            if over:
                sys.stderr.write(f"[bench] byte model exceeds the HBM peak for {over}: those bytes are not being moved\n")
            if dominant.startswith("trace"):
                # the compulsory-traffic lower bound beside the algorithmic figure — every ray in and its hit out, the scene once
                bvh = flat.bvh()
                rays_per_launch = sv.closest_rays / max(1, agg["launches"][dominant] // max(1, steps))
                compulsory = rays_per_launch * 48 + 32 * int(bvh[1].size) + 48 * int(bvh[3].size)
                roofline["compulsory_bytes_per_launch"] = int(compulsory)
                roofline["achieved_compulsory"] = round(compulsory / (dom_ms * 1e-3) / 1e9, 2) if dom_ms > 0 else 0.0
            if sv.fallback_rays or int(sv.traversal) == 7:
                roofline["fallback_rays_per_step"] = int(agg["fallback"] / steps)
                roofline["fallback_fraction_of_closest_rays"] = round(agg["fallback"] / max(1, agg["closest"]), 5)
            want_counters = world == 1 and not args.no_traffic  # measured last (below): the child runs need the HBM this process holds
            roofline["bound"] = classify_bound(roofline)
            roofline["bound_note"] = ("from counters: VALU busy >= 80 % -> valu-issue (lanes per VALU instruction in `valu`); else HBM-side bytes >= 60 % of peak -> hbm; else latency. "
                                      "`frac` stays SURVEY 8(d)'s algorithmic-bytes figure against the HBM peak")
        # ---- CPU baseline: the oracle (faithful restatement, OpenMP over the reference's 16x16 tiles) on a bounded sample ----
        cpu, micro = None, None
        osc = None
        if world == 1 and not (args.no_cpu_baseline and args.no_micro):
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            graft.build_oracle()
            import oracle_bridge as ob
            osc = ob.OracleScene.from_scene(scene, bvh=flat.bvh())
        if world == 1 and not args.no_cpu_baseline:
            threads = ob.lib().orc_num_threads()
            cpu_spp = args.cpu_spp
            cpu_cam, cpu_res = cam, args.res
            if args.res > 1024:  # one sample pass of a 4096^2 frame is already minutes of CPU work: the bounded sample is the same scene and depth at 1024^2
                cpu_res = 1024
                cpu_cam = build_workload(T, args.workload, cpu_res)[1]
            if cpu_spp <= 0:  # calibrate on one pass, then size the sample for ~15 s
                t1 = time.perf_counter()
                osc.render(cpu_cam, "path", 1, args.depth, seed=args.seed, threads=threads)
                one = time.perf_counter() - t1
                cpu_spp = int(max(1, min(args.spp, round(15.0 / max(one, 1e-3)))))
            t1 = time.perf_counter()
            _, _, cst = osc.render(cpu_cam, "path", cpu_spp, args.depth, seed=args.seed, threads=threads)
            dt = time.perf_counter() - t1
            cpu = {"value": round((cst.closest_rays + cst.shadow_rays) / dt / 1e6, 3), "unit": "Mray/s", "cores": threads, "kind": "port",
                   "sample": f"same scene, {cpu_res}x{cpu_res}, depth {args.depth}, {cpu_spp} spp of {args.spp} ({cst.camera_samples} camera samples, {dt:.1f} s)",
                   "Msample_per_s": round(cst.camera_samples / dt / 1e6, 4)}
        if world == 1 and not args.no_micro:
            micro = micro_benchmark(args, T, ctx, flat, osc if not args.no_cpu_baseline else None)
        rccl_ranks = ctx.comm_rank()[1]
        full_frame_equal = None
        if world == 1 and bvh_mode in (2, 3) and not args.no_modes:
            # the same frame on either tree alone, a few steps each: what the hybrid default is measured against
            modes = {}
            m_steps = max(1, min(steps, 3))

            def frame_ms(n):
                it = T.PathIntegrator(cam, T.SeededSampler(args.spp, seed=args.seed), args.depth)
                it.render(scene, ctx, device_out=film.data_ptr())
                t1 = time.perf_counter()
                rays = 0
                for _ in range(n):
                    it.render(scene, ctx, device_out=film.data_ptr())
                    rays += it.stats.closest_rays + it.stats.shadow_rays
                torch.cuda.synchronize()
                dt = time.perf_counter() - t1
                return {"ms_per_step": round(dt / n * 1e3, 3), "Mray_s": round(rays / dt / 1e6, 2), "closest_ms": round(it.stats.ms_trace_closest, 3), "steps": n}

            def film_witness(reference_film):
                """The film now on the device against `reference_film`, bit for bit, at the frame's FULL size: (equal, values that differ, 64-bit checksum of the bit patterns)."""
                a, b = film.view(torch.int32), reference_film.view(torch.int32)
                differ = int((a != b).sum().item())
                return differ == 0, differ, int(a.to(torch.int64).sum().item()) & 0xFFFFFFFFFFFFFFFF

            # the hybrid default's own frame (same sampler, same seed: what the timed region rendered), kept on the device as the thing the other two are held against
            frame_ms(1)
            hybrid_film = film.clone()
            hybrid_sum = film_witness(hybrid_film)[2]
            ctx.set_option("hybrid", 0)  # every ray on the canonical (reference) tree, in the reference's order (accel/bvh.jl:212-258): the frame the default claims to reproduce
            m = frame_ms(m_steps)
            eq, differ, csum = film_witness(hybrid_film)
            modes["reference_tree_alone"] = dict(m, exact=eq, film_values_that_differ_from_hybrid=differ, film_checksum=f"{csum:#018x}",
                                                 note="option hybrid = 0: k_trace3 on the reference's own tree; `exact` is MEASURED in this run: its film against the hybrid default's, "
                                                      f"all {h}x{w}x4 Float32 bit patterns on the device")
            full_frame_equal = eq
            ctx.set_option("hybrid", 1)
            if bvh_mode == 2:
                ctx.set_option("bvh_builder", 0)
                scene._flat = None
                flat.free()
                flat = scene.flatten(ctx)
                m = frame_ms(m_steps)
                eq, differ, csum = film_witness(hybrid_film)
                modes["library_tree_alone"] = dict(m, exact=eq, film_values_that_differ_from_hybrid=differ, film_checksum=f"{csum:#018x}",
                                                   note="option bvh_builder = 0: the library's SAH tree alone — rays whose answer depends on the visiting order (ties, a sphere entered "
                                                        "from inside) resolve in ITS order, not Trace.jl's; `exact` is measured the same way")
            else:  # mode 3: the canonical tree IS the library's (the reference's construction fails on this scene): "reference_tree_alone" above is that tree walked in the reference's order
                modes["reference_tree_alone"]["note"] += " — on this scene the canonical tree is the library's own (trhip_scene_bvh_note): the comparison is the four-wide certified walk against the reference-ORDER walk of the same tree"
            modes["hybrid_film_checksum"] = f"{hybrid_sum:#018x}"
            del hybrid_film
            ctx.set_option("bvh_builder", -1)
        if roofline and want_counters:
            # the PMC child runs render the same workload in their own process: this one's wavefront buffers are sized to what was free (0.85 of HBM for a
            # 4096^2 frame) and would leave them nothing — release the scene and the context first
            flat.free()
            scene._flat = None
            ctx.close()
            traffic, valu = measure_counters(args, kprefix, False)
            roofline["traffic_note"] = ("bytes per launch = fetch_factor x FETCH_SIZE + WRITE_SIZE (rocprofv3 --pmc, separate child runs of this command with --steps 1); fetch_factor: "
                                        "MI355X_MICROARCH.md's gfx950 correction, re-measured for this kernel's access pattern by tools/calib/fetch_calib.hip")
            if traffic:
                roofline["traffic"] = traffic["bytes"]
                roofline["traffic_detail"] = traffic
                roofline["achieved_counters"] = round(roofline["traffic"] / (dom_ms * 1e-3) / 1e9, 2)
                roofline["frac_counters"] = round(roofline["achieved_counters"] / HBM_PEAK_GBS, 5)
            if valu:
                roofline["valu"] = valu
                if valu.get("valu_busy") is not None:
                    # the share of the SIMDs' lane-cycles that did arithmetic: VALU busy x active lanes / 64 — what bounds a kernel whose HBM-side fraction says nothing
                    # (a scene that lives in L2 + MALL): 1.0 = every lane of every SIMD issuing every cycle
                    roofline["valu_frac"] = round(valu["valu_busy"] * valu["lanes_per_valu_inst"] / 64.0, 4)
            roofline["bound"] = classify_bound(roofline)
        if roofline:
            # which fraction the line leads with: a scene that fits L2 + MALL is not served from HBM, so its request rate is no HBM fraction — the counter figure is
            scene_bytes = n_scene_bytes
            roofline["scene_bytes"] = int(scene_bytes)
            roofline["scene_fits_l2_plus_mall"] = bool(scene_bytes <= L2_PLUS_MALL_BYTES)
            if roofline.get("frac_counters") is not None and (scene_bytes <= L2_PLUS_MALL_BYTES or roofline["frac_requests"] > 1.0):
                roofline["frac"] = roofline["frac_counters"]
                roofline["achieved"] = roofline["achieved_counters"]
                roofline["frac_is"] = "counters (bytes that left L2)"
            else:
                roofline["frac_is"] = "requests (algorithmic bytes)"
                if roofline["frac"] > 1.0:  # no counters in this run: a request rate above the peak is not a bandwidth
                    roofline["frac"] = None
        spp_r = shard(args.scaling)[0]
        result = {
            "metric": "Mray/s (all bounces)", "value": round(total_rays / elapsed / 1e6, 2), "unit": "Mray/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "Msample_per_s": round(total_samples / elapsed / 1e6, 3),
            "config": {"workload": f"{args.workload}: {desc}; {args.res}x{args.res}, {spp_r} spp per GPU" + (f" ({args.spp} per frame)" if args.scaling == "strong" and world > 1 else "")
                                   + f", max depth {args.depth}, PathIntegrator, seed {args.seed:#x}",
                       "rays_per_step": int(total_rays / args.steps), "samples_per_step": int(total_samples / args.steps), "bvh_build_upload_s": round(t_build, 3),
                       "traversal": int(sv.traversal),
                       "bvh": BVH_MODE.get(bvh_mode, str(bvh_mode)), "bvh_note": bvh_note or None, "bvh_nodes": {"canonical": n_canonical_nodes, "accelerator": n_acc_nodes},
                       "fallback_fraction": round(agg["fallback"] / max(1, agg["closest"]), 5),
                       "parallelism": f"sample-index sharding x{world} + film sum-reduce over RCCL (trhip_film_reduce)" if world > 1 else "single GPU",
                       "rccl_ranks": rccl_ranks, "film_reduce_ms_per_step": round(agg["film_reduce_ms_per_step"], 3) if world > 1 else 0.0},
            "roofline": roofline, "cpu_baseline": cpu,
            "parity": {"vs": "the CPU oracle walking the reference's own tree (accel/bvh.jl:55-206, restated in oracle/orc_build.h)" if bvh_mode in (1, 2) else
                             "the CPU oracle walking the library's tree (NOT Trace.jl's tie-breaks: option bvh_builder selects it)",
                       # measured, not asserted: what traversal_micro found in THIS run (None: --no-micro or more than one rank: nothing was compared)
                       "bits": {True: "equal", False: "differ", None: "not checked in this run"}[(micro or {}).get("gpu_equals_cpu_on_subset")],
                       "tolerance": "0 ulp (the tests demand bit equality; SURVEY 8(d)'s 1e-3 radiance tolerance is not used)",
                       "checked_in_this_run": (micro or {}).get("gpu_equals_cpu_on_subset"),
                       # the headline frame at its own size: the hybrid default's film == the film of every ray walking the reference's tree in the reference's order (bvh_modes;
                       # None: not compared in this run — more than one rank, --no-modes, or a scene without two trees)
                       "full_frame_equal": full_frame_equal,
                       "where": "tests/test_gpu_hybrid.py, tests/test_gpu_scale.py, tools/soak_hybrid.py (-m gpu); traversal_micro compares 2^21 rays in this run",
                       "witness": "the oracle is a restatement pinned to the reference's unit-test vectors per callee (ray / shape / BSDF / film / BVH construction); the PathIntegrator composite "
                                  "does not exist in the reference (SURVEY F2: defined here as the SPPM camera-pass loop) and has no reference-produced output to compare with — the only "
                                  "end-to-end artefact of the reference, its SPPM golden PNG, is matched radiometrically (tests/test_gpu_golden_radiometry.py)"},
        }
        if modes:
            result["bvh_modes"] = modes
        if micro:
            result["traversal_micro"] = micro
        if world == 1 and args.workload == "mesh_1m" and roofline and not args.no_hbm_resident and not args.no_traffic:
            result["hbm_resident"] = hbm_resident_record(args)
        if other:
            result[other["scaling"] + "_scaling"] = other
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
