#!/usr/bin/env python
"""How k_sppm_gather's tiles fare on C4 (caustic-glass.ply).  Needs the DIAGNOSTIC build:
    hipcc <HIPCC_FLAGS> -DTH_DIAG_PHASES -o _diag/libtracehip_phases.so trace.jl_amd/csrc/tracehip.hip
    TRHIP_LIB=$PWD/_diag/libtracehip_phases.so python tools/sppm_tile_probe.py"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
import bench
T = g.load_package()
fn = T.lib().trhip_debug_sppm
fn.restype = C.c_int
fn.argtypes = [C.POINTER(C.c_uint64), C.c_int]
scene, cam = T.scenes.caustic_scene(bench.caustic_model()), T.scenes.caustic_camera(1024)
ctx = T.default_context()
out = np.zeros(16, np.uint64)
for its in (1, 100):
    fn(out.ctypes.data_as(C.POINTER(C.c_uint64)), 1)
    integ = T.SPPMIntegrator(cam, 0.075, 8, its, -1)
    integ.render(scene, ctx)
    fn(out.ctypes.data_as(C.POINTER(C.c_uint64)), 1)
    o = [int(x) for x in out]
    t = max(o[0], 1)
    print(f"{its} iterations: {o[0] / its:.0f} tiles with visible points per iteration; to the cursor path: box > 3 cells {o[1] / its:.1f}, > 64 cells {o[2] / its:.1f}, equal hashes {o[3] / its:.1f};")
    print(f"   tile path: {o[4] / t:.1f} listed cells and {o[5] / t:.1f} candidates per tile; heavy tiles {o[6] / its:.0f} per iteration with {o[7] / max(o[6], 1):.0f} candidates each")
