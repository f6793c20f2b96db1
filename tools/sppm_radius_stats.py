#!/usr/bin/env python
"""C4 (caustic-glass.ply under SPPM): how the per-pixel search radius compares with the reference's grid cell after N iterations —
what a finer level of the photon grid could save.  Run on the GPU box."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
import bench
T = g.load_package()
scene, cam = T.scenes.caustic_scene(bench.caustic_model()), T.scenes.caustic_camera(1024)
ctx = T.default_context()
for its in (1, 10, 30, 100):
    integ = T.SPPMIntegrator(cam, 0.075, 8, its, -1)
    integ.render(scene, ctx)
    st = integ.state()
    rad, M, beta = st["radius"].ravel(), st["M"].ravel(), st["vp_beta"].reshape(-1, 3)
    have = (beta != 0).any(axis=1)
    r = rad[have]
    cell = float(rad.max())  # the grid's cell edge is (about) the largest radius
    q = np.quantile(r, [0.01, 0.1, 0.25, 0.5, 0.75, 0.9, 0.99])
    print(f"iterations {its}: {int(have.sum())} visible points, grid {st['info']['grid_res']}, max radius {cell:.4f}; radius / max quantiles 1/10/25/50/75/90/99 %: " + " ".join(f"{x / cell:.3f}" for x in q))
    print(f"   last iteration: sum M {int(M.sum())}, pixels with M > 0: {int((M > 0).sum())}; M-weighted median radius / max: {float(np.median(np.repeat(rad[M > 0], np.minimum(M[M > 0], 50)))) / cell:.3f}; in-bounds photon hits (all iterations) {st['info']['photon_hits']}; {integ.stats.ms_total:.1f} ms")
