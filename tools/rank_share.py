#!/usr/bin/env python
"""One rank's share of a strong-scaled frame, measured on one GPU (docs/design/07): the frame of BASELINE's metric configuration split over N ranks by global sample index —
rank r of N renders spp / N samples per pixel starting at sample r * spp / N — with option overlap 0 and 1.  Prints one JSON object: per (spp share, overlap) the frame's
wall time (ms, HIP work + host loop, mean of --frames after one warm-up) and the predicted N-rank strong-scaling factor T(256) / (T(256 / N) + reduce), reduce = the
16.8 MB film sum-reduce priced at --reduce-ms (unmeasured: no second GPU)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="mesh_1m")
    ap.add_argument("--res", type=int, default=1024)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--ranks", default="1,2,4,8")
    ap.add_argument("--reduce-ms", type=float, default=0.5, help="price of the film reduce per frame (16.8 MB over xGMI at ~50 GB/s effective + launch): an assumption, stated in the output")
    args = ap.parse_args()
    import torch
    graft.build()
    T = graft.load_package()
    ctx = T.default_context()
    scene, cam, desc = bench.build_workload(T, args.workload, args.res)
    h, w = cam.film.size
    film = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    out = {"workload": f"{args.workload}: {desc}; {args.res}x{args.res}, {args.spp} spp per frame, depth {args.depth}", "frames": args.frames, "shares": {}}
    for overlap in (1, 0):
        ctx.set_option("overlap", overlap)
        for n in [int(x) for x in args.ranks.split(",")]:
            spp_r, off = T.parallel.shard_samples(args.spp, n - 1, n)  # the LAST rank's share (its sample offset is the largest)
            integ = T.PathIntegrator(cam, T.SeededSampler(spp_r, seed=0x5EED0001, sample_offset=off), args.depth)
            integ.render(scene, ctx, device_out=film.data_ptr())
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.frames):
                integ.render(scene, ctx, device_out=film.data_ptr())
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / args.frames * 1e3
            out["shares"][f"overlap={overlap},ranks={n}"] = {"spp_of_this_rank": spp_r, "sample_offset": off, "frame_ms": round(ms, 3), "closest_ms": round(integ.stats.ms_trace_closest, 3),
                                                            "any_ms": round(integ.stats.ms_trace_any, 3), "shade_ms": round(integ.stats.ms_shade, 3)}
    pred = {}
    for overlap in (1, 0):
        t1 = out["shares"][f"overlap={overlap},ranks=1"]["frame_ms"]
        for n in [int(x) for x in args.ranks.split(",")]:
            tn = out["shares"][f"overlap={overlap},ranks={n}"]["frame_ms"]
            pred[f"overlap={overlap},ranks={n}"] = round(t1 / (tn + (args.reduce_ms if n > 1 else 0.0)), 3)
    out["predicted_strong_scaling"] = pred
    out["predicted_note"] = (f"PREDICTED, not measured: T(1 rank) / (T(this rank's share on one GPU) + {args.reduce_ms} ms assumed for the film reduce); every rank of a real job holds its own "
                             "copy of the scene and its own GPU, so a share's time on one GPU is its time in the job up to the reduce")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
