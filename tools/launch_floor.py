#!/usr/bin/env python
"""Floor of one traversal launch: a persistent-grid kernel over 64 rays (what the last depths of a frame and the fallback lists cost)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as g
T = g.load_package()
ctx = T.Context(0)
scene = T.scenes.mesh_scene(181)
flat = scene.flatten(ctx)
L = T.lib()
for n in (64, 4096, 1 << 16, 1 << 20):
    rays = T.scenes.incoherent_rays(n, [0, 0, -3], [1, 1, -2])
    d_rays = torch.from_numpy(rays).cuda()
    d_hits = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    d_occ = torch.empty(n, dtype=torch.uint8, device="cuda")
    for trav in (3, 1):
        ctx.set_option("traversal", trav)
        ms = C.c_double()
        out = []
        for fn, buf in ((L.trhip_trace_closest_device, d_hits), (L.trhip_trace_any_device, d_occ)):
            ctx.check(fn(ctx._h, flat._h, C.c_void_p(d_rays.data_ptr()), n, C.c_void_p(buf.data_ptr()), 2, C.byref(ms)))
            ctx.check(fn(ctx._h, flat._h, C.c_void_p(d_rays.data_ptr()), n, C.c_void_p(buf.data_ptr()), 20, C.byref(ms)))
            out.append(round(ms.value * 1e3, 1))
        print(f"rays {n:8d} traversal {trav}: closest {out[0]} us, any {out[1]} us per launch", flush=True)
