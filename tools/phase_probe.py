#!/usr/bin/env python
"""Where k_trace3<closest> spends its wave cycles, and with how many lanes.  Needs the DIAGNOSTIC build:

    python -c "import __graft_entry__ as g; g.build_library(extra_flags=['-DTH_DIAG_PHASES'], out_name='libtracehip_phases.so')"
    TRHIP_LIB=$PWD/trace.jl_amd/libtracehip_phases.so python tools/phase_probe.py --workload mesh_1m --spp 64

Phases: refill (idle lanes take new rays), pop (stack pops of lanes whose node is done), node (interior step: one 64-byte node,
two boxes), leaf (primitive tests).  cycles = wave cycles inside the phase summed over all waves; lanes = lanes with work at entry."""
import argparse, ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
import bench
T = g.load_package()
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="mesh_1m")
ap.add_argument("--spp", type=int, default=64)
ap.add_argument("--no-spheres", action="store_true", help="mesh_* workloads: the Cornell box without its two spheres")
ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="library option set before the commit (bvh_builder=0: the library's tree alone -> k_trace3)")
ap.add_argument("--certified", action="store_true", help="read k_trace3c's counters (the hybrid mode's walk on the accelerator tree) instead of k_trace3's")
a = ap.parse_args()
L = T.lib()
fn = L.trhip_debug_phases_c if a.certified else L.trhip_debug_phases
fn.restype = C.c_int
fn.argtypes = [C.POINTER(C.c_uint64), C.c_int]
scene, cam, desc = bench.build_workload(T, a.workload, 1024)
if a.no_spheres:
    prims, _ = T.scenes.cornell_primitives(spheres=False)
    grey = T.MatteMaterial(T.ConstantTexture(T.RGBSpectrum(0.8)), T.ConstantTexture(0.0))
    verts, idx, nrm = T.scenes.heightfield_mesh(T.scenes.MESH_N[a.workload])
    prims = prims + [T.create_mesh_primitives(T.ShapeCore(T.translate([0, 0, 0]), False), idx, verts, nrm, grey)]
    scene = T.Scene(T.scenes.cornell_lights(), T.BVHAccel(prims, 1))
ctx = T.default_context()
for kv in a.opt:
    k_, v_ = kv.split("=")
    ctx.set_option(k_, int(v_))
out = np.zeros(13, np.uint64)
integ = T.PathIntegrator(cam, T.SeededSampler(a.spp, seed=1), 8)
integ.render(scene, ctx)
fn(out.ctypes.data_as(C.POINTER(C.c_uint64)), 1)
integ.render(scene, ctx)
assert fn(out.ctypes.data_as(C.POINTER(C.c_uint64)), 1) == 0
tot = float(sum(out[0:12:3]))
print(f"{a.workload}, {a.spp} spp: closest-hit {integ.stats.ms_trace_closest:.1f} ms, {integ.stats.closest_rays} rays")
for k, name in enumerate(("refill", "pop", "node", "leaf")):
    cyc, lan, cnt = float(out[3 * k]), float(out[3 * k + 1]), float(out[3 * k + 2])
    print(f"  {name:7s} {100 * cyc / tot:5.1f} % of the instrumented wave cycles, {cnt:.3e} entries, {cyc / max(cnt, 1):7.1f} cycles each, {lan / max(cnt, 1):5.1f} lanes with work")
print(f"  of the leaf cycles, {100 * float(out[12]) / max(float(out[9]), 1):.0f} % pass before the primitive's records have arrived")
