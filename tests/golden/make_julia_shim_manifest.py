#!/usr/bin/env python
"""Writes tests/golden/julia_shim_calls.json: (1) every ccall signature of trace.jl_amd/julia/TraceHIP.jl, (2) the call sequences
the shim's flatten() issues for the scenes of the GPU replay test (tests/test_gpu_julia_replay.py), derived WITHOUT a GPU by
walking the same object graph with tests/julia_replay.ShimReplay in dry mode: function, array shapes, scalar arguments.

    python tests/golden/make_julia_shim_manifest.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft  # noqa: E402
import julia_replay as jr  # noqa: E402


class DryReplay(jr.ShimReplay):
    """Logs the calls without making them (material / scene handles are counters)."""

    def __init__(self, T):
        self.T = T
        self.ccalls = jr.parse_ccalls()
        import ctypes as C
        self.ctx = C.c_void_p(1)
        self.log = []
        self._keep = []
        self._next_material = 0

    def call(self, fn, *args, which=0):
        desc = [jr.describe(a) for a in args]
        self.log.append([fn] + desc)
        if fn == "trhip_scene_add_material":
            args[-1]._obj.value = self._next_material
            self._next_material += 1
        return 0


def scenes(T):
    nested_inner = T.BVHAccel(T.scenes.cornell_primitives()[0][:6], 1)
    nested = T.Scene(T.scenes.cornell_lights(), T.BVHAccel([nested_inner] + T.scenes.cornell_primitives()[0][6:], 1))
    return {"shadows": T.scenes.shadows_scene(), "caustic_glass_ply": T.scenes.caustic_scene(os.path.join(HERE, "caustic-glass.ply")), "nested_bvh_cornell": nested,
            "tangent_uv_mesh": jr.tangent_uv_scene(T)}


if __name__ == "__main__":
    T = graft.load_package()
    out = {"shim": "trace.jl_amd/julia/TraceHIP.jl", "ccalls": {fn: [[ret, args] for ret, args in sigs] for fn, sigs in sorted(jr.parse_ccalls().items())}, "sequences": {}}
    for name, scene in scenes(T).items():
        r = DryReplay(T)
        r.flatten(scene)
        out["sequences"][name] = r.summary()
    json.dump(out, open(os.path.join(HERE, "julia_shim_calls.json"), "w"), indent=1)
    for k, v in out["sequences"].items():
        print(k, [(c[0].replace("trhip_scene_", ""), c[1]) for c in v])
