import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import soak_parity as sp
T = sp.T
ctx = T.default_context()
seed = int(sys.argv[1]); n = int(sys.argv[2])
for k in range(int(sys.argv[3])):
    rng = np.random.default_rng(seed * 1000 + k)
    scene, tri = sp.rand_scene(rng, k)
    ctx.set_option("bvh_builder", 1 if k % 5 == 4 else 0)
    t = time.time(); flat = scene.flatten(ctx); print(k, "flatten", round(time.time() - t, 3), flat.bvh()[3].size, flush=True)
    bnd = flat.bvh()[0][0]
    rays = sp.rand_rays(rng, n, bnd[:3].copy(), bnd[3:].copy(), tri)
    for trav in (1, 3, 2):
        ctx.set_option("traversal", trav)
        t = time.time(); h = flat.trace_closest(rays); print(k, "trav", trav, "closest", round(time.time() - t, 3), flush=True)
        t = time.time(); o = flat.trace_any(rays); print(k, "trav", trav, "any", round(time.time() - t, 3), flush=True)
    cam = T.scenes.cornell_camera(64)
    for trav in (1, 3):
        ctx.set_option("traversal", trav)
        t = time.time(); T.PathIntegrator(cam, T.SeededSampler(4, seed=100 + k), 6).render(scene, ctx); print(k, "frame trav", trav, round(time.time() - t, 3), flush=True)
    t = time.time(); flat.free(); scene._flat = None; print(k, "free", round(time.time() - t, 3), flush=True)
