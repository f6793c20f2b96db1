#!/usr/bin/env python
"""Soak test of the film pass (add_sample! + merge_film_tile!, film.jl:134-193) on the GPU against the oracle: random film sizes (not
multiples of the 16-pixel tiles), crop windows, anisotropic Lanczos filters of radius 0.5-3.5, 1-5 spp, every gather variant (film_block
0-3, film_tiled).  The oracle renders the shadows scene (any radiance will do) and keeps its per-sample radiance;
trhip_film_accumulate must turn those samples into the oracle's film bit for bit.  Run on the GPU box: python tools/soak_film.py --cases 60"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft

graft.build_library()
graft.build_oracle()
T = graft.load_package()
import oracle_bridge as ob


def same_bits(a, b):
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    na, nb = np.isnan(a), np.isnan(b)
    return np.array_equal(na, nb) and not ((a.view(np.uint32) != b.view(np.uint32)) & ~na).any()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=30)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    ctx = T.default_context()
    scene = T.scenes.shadows_scene()
    osc = ob.OracleScene.from_scene(scene)
    bad = 0
    for k in range(a.cases):
        rng = np.random.default_rng(a.seed * 1000 + k)
        res = [int(rng.integers(9, 90)), int(rng.integers(9, 90))]
        lo = rng.uniform(0.0, 0.4, 2) if k % 3 == 1 else np.zeros(2)
        hi = rng.uniform(0.6, 1.0, 2) if k % 3 == 1 else np.ones(2)
        rmax = 1.0 if k % 2 == 0 else 3.5  # half of the cases inside the packed descriptor's range (radius <= 1: the default path), half beyond it
        flt = T.LanczosSincFilter([float(rng.uniform(0.3, rmax)), float(rng.uniform(0.3, rmax))], float(rng.uniform(1.0, 4.0)))
        film = T.Film(res, T.Bounds2([float(lo[0]), float(lo[1])], [float(hi[0]), float(hi[1])]), flt, 1.0, float(rng.uniform(0.5, 2.0)), "")
        cam = T.PerspectiveCamera(T.look_at([0, 15, 50], [0, 0, -2], [0, 1, 0]), T.Bounds2([-1.0, -1.0], [1.0, 1.0]), 0.0, 1.0, 0.0, 1e6, 90.0, film)
        spp = int(rng.integers(1, 6))
        seed = 900 + k
        ref_xyzw, ref_L, _ = osc.render(cam, "path", spp, 3, seed=seed, want_samples=True)
        if k % 4 == 0:  # NaN samples are zeroed (integrators/sampler.jl:46): poison a few in both
            idx = rng.integers(0, ref_L.size // 3, 5)
            ref_L.reshape(-1, 3)[idx, int(rng.integers(0, 3))] = np.nan
            # the oracle film for the poisoned samples: accumulate on the CPU side through the oracle is not exposed, so only compare the variants with each other here
            ref_xyzw = None
        sn = cam.sensor()
        outs = {}
        for name, opts in (("default", {}), ("block0", {"film_block": 0}), ("block1", {"film_block": 1}), ("block2", {"film_block": 2}), ("block3", {"film_block": 3}), ("packed1x4", {"film_block": 4}),
                           ("packed2x4", {"film_block": 5}), ("packed2x2", {"film_block": 7}), ("packed4x2", {"film_block": 8}), ("packed8x4", {"film_block": 9}), ("tiled", {"film_tiled": 1})):
            for o, v in opts.items():
                ctx.set_option(o, v)
            try:
                out = np.empty((cam.film.size[0], cam.film.size[1], 4), np.float32)
                ctx.check(T.lib().trhip_film_accumulate(ctx._h, C.byref(sn), spp, seed, 0, T._ffi.fptr(ref_L), T._ffi.fptr(out)))
                outs[name] = out
            finally:
                for o in opts:
                    ctx.set_option(o, {"film_block": 5, "film_tiled": 0}[o])
        msgs = []
        base = ref_xyzw if ref_xyzw is not None else outs["default"]
        for name, out in outs.items():
            if not same_bits(out, base):
                msgs.append(name)
        bad += 1 if msgs else 0
        print(f"case {k:3d}: film {res[0]} x {res[1]}, crop {'yes' if k % 3 == 1 else 'no '}, radius ({flt.radius[0]:.2f}, {flt.radius[1]:.2f}), {spp} spp, "
              f"{'vs oracle' if ref_xyzw is not None else 'NaN samples, variants vs default'}: {'equal' if not msgs else 'MISMATCH ' + ', '.join(msgs)}", flush=True)
    print(f"total: {a.cases} cases x 11 gather variants, {bad} with a mismatch")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
