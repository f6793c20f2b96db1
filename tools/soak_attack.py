#!/usr/bin/env python
"""The certificate under attack, long form (tests/attack_scenes.py says what and why; tests/test_gpu_certificate_attack.py is the short form in the -m gpu suite).

Per family and seed: hits (primitive, t, barycentrics) and occlusion of every attack ray with the hybrid walk against option "hybrid" = 0 (k_trace3 on the canonical tree alone),
bit for bit; the share of rays handed to the reference-order walk per ray kind; a PathIntegrator frame from inside the box with hybrid on / off.  Exit code 1 on any mismatch.

    python tools/soak_attack.py --seeds 6 --rays 400000 > profiles/r5/r5_soak_attack.txt
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g  # noqa: E402

T = g.load_package()
import attack_scenes as A  # noqa: E402


def u32(a):
    return np.ascontiguousarray(a).view(np.uint32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=4)
    ap.add_argument("--rays", type=int, default=400000)
    ap.add_argument("--family", action="append", default=[])
    args = ap.parse_args()
    ctx = T.default_context()
    bad = 0
    total_rays = 0
    print(f"{'family':28s} {'seed':>4s} {'rays':>9s} {'mismatches':>10s}  fallback share per ray kind | frame")
    for fam in [f[0] for f in A.FAMILIES if not args.family or f[0] in args.family]:
        for seed in range(1, args.seeds + 1):
            scene, tri, lo, hi, spheres = A.make_family(T, fam, seed=100 + seed)
            flat = scene.flatten(ctx)
            if flat.bvh_mode()[0] != 2:
                print(f"{fam:28s} {seed:4d} one tree only: {flat.bvh_note()}")
                continue
            parts = A.attack_rays(np.random.default_rng(seed), args.rays, lo, hi, tri, spheres)
            shares, mism, n = [], 0, 0
            for kind, rays in parts.items():
                got = flat.trace_closest(rays)
                nr, nf = flat.last_fallback()
                occ = flat.trace_any(rays)
                ctx.set_option("hybrid", 0)
                ref = flat.trace_closest(rays)
                occ_ref = flat.trace_any(rays)
                ctx.set_option("hybrid", 1)
                m = np.zeros(rays.shape[0], bool)
                for k in ("prim", "t", "b1", "b2"):
                    m |= u32(got[k]) != u32(ref[k])
                m |= occ != occ_ref
                mism += int(m.sum())
                n += rays.shape[0]
                shares.append(f"{kind} {nf / max(1, nr):.4f}")
            cam = A.attack_camera(T, lo, hi, 64)
            it = T.PathIntegrator(cam, T.SeededSampler(4, seed=seed), 8)
            film = it.render(scene, ctx).copy()
            L = it.sample_radiance(scene).copy()
            st = it.stats
            ctx.set_option("hybrid", 0)
            it0 = T.PathIntegrator(cam, T.SeededSampler(4, seed=seed), 8)
            film0 = it0.render(scene, ctx)
            L0 = it0.sample_radiance(scene)
            ctx.set_option("hybrid", 1)
            fm = int((~((u32(L) == u32(L0)) | (np.isnan(L) & np.isnan(L0)))).sum()) + int((~((u32(film) == u32(film0)) | (np.isnan(film) & np.isnan(film0)))).sum())
            bad += mism + fm
            total_rays += n + int(st.closest_rays + st.shadow_rays)
            print(f"{fam:28s} {seed:4d} {n:9d} {mism + fm:10d}  " + ", ".join(shares) + f" | frame fallback {st.fallback_rays / max(1, st.closest_rays):.4f} of {st.closest_rays}", flush=True)
            scene._flat = None
            flat.free()
    print(f"total: {total_rays} rays, {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
