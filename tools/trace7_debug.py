#!/usr/bin/env python
"""k_trace7 (traversal 7) against k_trace3 / the oracle on one scene, ray by ray — diagnostic.
    python tools/trace7_debug.py shadows|mesh64|mesh_1m [n_random_rays]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
g.build()
T = g.load_package()
import oracle_bridge as ob
which = sys.argv[1]
n_rand = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
ctx = T.default_context()
ctx.set_option("tiny_scene_prims", 0)
if which == "shadows":
    ctx.set_option("compose_spheres", 1)
    scene, cam, box = T.scenes.shadows_scene(), T.scenes.shadows_camera(64), ([-1.2, -0.3, -3.2], [1.3, 1.2, 1.0])
elif which == "mesh64":
    scene, cam, box = T.scenes.mesh_scene(64), T.scenes.cornell_camera(64), ([0, 0, -3], [1, 1, -2])
else:
    scene, cam, box = T.scenes.mesh_scene(T.scenes.MESH_N[which]), T.scenes.cornell_camera(512), ([0, 0, -3], [1, 1, -2])
flat = scene.flatten(ctx)
b, a, f, order = flat.bvh()
print("nodes", a.size, "prims", order.size, flush=True)
rays = np.concatenate([ob.generate_rays(cam, T.scenes.camera_sample_grid(cam, 1, seed=3)), T.scenes.incoherent_rays(n_rand, np.float32(box[0]), np.float32(box[1]), seed=17)])
# a generation of bounce rays spawned like spawn_ray does (origin + 1e-6 * direction) from the hits
ctx.set_option("traversal", 3)
t0 = time.time(); h3 = flat.trace_closest(rays); t3 = time.time() - t0
hit = h3["prim"] >= 0
p = rays[hit, 0:3] + h3["t"][hit, None] * rays[hit, 4:7]
rng = np.random.default_rng(7)
d = rng.normal(size=(p.shape[0], 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
br = np.empty((p.shape[0], 8), np.float32); br[:, 0:3], br[:, 3], br[:, 4:7], br[:, 7] = p + np.float32(1e-6) * d, np.inf, d, 0.0
rays = np.concatenate([rays, br])
h3 = flat.trace_closest(rays)
print("traversal 3:", rays.shape[0], "rays", flush=True)
for cheap in (0, 1):
    ctx.set_option("traversal", 7); ctx.set_option("trace7_cheap", cheap)
    t0 = time.time(); h7 = flat.trace_closest(rays); t7 = time.time() - t0
    bad = (h3["prim"] != h7["prim"]) | (h3["t"].view(np.uint32) != h7["t"].view(np.uint32)) | (h3["b1"].view(np.uint32) != h7["b1"].view(np.uint32)) | (h3["b2"].view(np.uint32) != h7["b2"].view(np.uint32))
    print(f"traversal 7 (cheap {cheap}): {int(bad.sum())} of {rays.shape[0]} rays differ from traversal 3   [{t7:.2f} s vs {t3:.2f} s incl. transfers]", flush=True)
    kinds = np.array([1 if isinstance(pp, T.GeometricPrimitive) and isinstance(getattr(pp, 'shape', None), T.Sphere) else 0 for pp in T.api.splice_nested(scene.aggregate.primitives)]) if which == "shadows" else None
    for i in np.nonzero(bad)[0][:12]:
        r = rays[i]
        k3 = k7 = "?"
        if kinds is not None:
            k3 = "sphere" if h3["prim"][i] >= 0 and kinds[order[h3["prim"][i]]] else "tri"
            k7 = "sphere" if h7["prim"][i] >= 0 and kinds[order[h7["prim"][i]]] else "tri"
        print(f"  ray {i}: o {r[0:3]} d {r[4:7]}  t3 {h3['t'][i]!r} prim {h3['prim'][i]} ({k3})  |  t7 {h7['t'][i]!r} prim {h7['prim'][i]} ({k7})", flush=True)
# fallback fraction from a small frame's stats
ctx.set_option("traversal", 7); ctx.set_option("trace7_cheap", 1)
integ = T.PathIntegrator(cam, T.SeededSampler(2, seed=5), 6)
integ.render(scene, ctx)
st = integ.stats
print("frame: closest rays", st.closest_rays, "fallback", st.fallback_rays, "fraction", st.fallback_rays / max(1, st.closest_rays), "ms closest", round(st.ms_trace_closest, 3), flush=True)
ctx.set_option("count_visits", 1)
integ.render(scene, ctx)
st = integ.stats
ctx.set_option("count_visits", 0)
print("   reasons [direction, sphere, own t_max, near tie]:", list(st.count_sub), " nodes / ray", round(st.nodes_visited / max(1, st.closest_rays), 2), " prims / ray", round(st.prims_tested / max(1, st.closest_rays), 2), flush=True)
ctx.set_option("traversal", 3)
integ.render(scene, ctx)
print("frame traversal 3: ms closest", round(integ.stats.ms_trace_closest, 3), flush=True)
ctx.set_option("count_visits", 1)
integ.render(scene, ctx)
ctx.set_option("count_visits", 0)
print("   traversal 3: nodes / ray", round(integ.stats.nodes_visited / max(1, integ.stats.closest_rays), 2), " prims / ray", round(integ.stats.prims_tested / max(1, integ.stats.closest_rays), 2), flush=True)
