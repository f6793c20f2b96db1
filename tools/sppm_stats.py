import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
T = g.load_package()
scene = T.scenes.caustic_scene("")
cam = T.scenes.caustic_camera(1024, "")
ctx = T.default_context()
for its in (1, 10, 100):
    integ = T.SPPMIntegrator(cam, 0.075, 8, its, -1)
    integ.render(scene, ctx)
    st = integ.state()
    M = st["M"]; rad = st["radius"]
    info = st["info"]
    print("iterations", its, "sum M last", int(M.sum()), "max M", int(M.max()), "pixels with M>0", int((M > 0).sum()), "photon_hits total", info["photon_hits"], "grid", info["grid_res"], "entries", info["grid_entries"],
          "radius min/median/max", float(rad.min()), float(np.median(rad)), float(rad.max()), "ms", integ.stats.ms_total)
    hot = M > 192
    print("   pixels with M > 192:", int(hot.sum()), "their M sum", int(M[hot].sum()), "radius median of those", float(np.median(rad[hot])) if hot.any() else None)
