"""The box test of k_trace2 / k_trace3 (th_trace2.h, slab_test2) adds the two slab clauses the reference's intersect_p lost
(bounds.jl:190 keeps the larger of the x and y exits), on boxes grown by a margin.  It must change NOTHING but the number of
boxes visited: hits, barycentrics, occlusion and whole frames are compared bit for bit with the reference's loose test
(option slab_margin_log2 = 0), with the literal kernels (traversal 1, which only know the loose test) and with the oracle,
on ray sets built to sit on the clauses' edges: far origins, near-horizontal rays skimming a height field, axis-parallel
rays, rays inside triangle planes, needle triangles, small spheres seen from far away (whose fp32 quadratic "hits" beyond
the sphere: their subtrees keep the loose test).
"""
import numpy as np
import pytest

from test_gpu_parity import assert_bits_equal

pytestmark = pytest.mark.gpu


def stress_scene(T, n_tris=6000, n_spheres=200, seed=7):
    """Cornell walls + random triangles (a third of them needles, a few axis-aligned) + small spheres, in [0,1] x [0,1] x [-3,-2]."""
    rng = np.random.default_rng(seed)
    prims, white = T.scenes.cornell_primitives(spheres=False)
    core = T.ShapeCore(T.translate([0, 0, 0]), False)
    c = rng.random((n_tris, 3), dtype=np.float32) * np.float32([1, 1, 1]) + np.float32([0, 0, -3])
    size = (0.002 + 0.03 * rng.random((n_tris, 1), dtype=np.float32)).astype(np.float32)
    e1 = rng.standard_normal((n_tris, 3)).astype(np.float32)
    e2 = rng.standard_normal((n_tris, 3)).astype(np.float32)
    e1 /= np.linalg.norm(e1, axis=1, keepdims=True)
    e2 /= np.linalg.norm(e2, axis=1, keepdims=True)
    needle = rng.random(n_tris) < 0.33
    e2[needle] = (e1[needle] + np.float32(1e-3) * e2[needle]).astype(np.float32)   # nearly collinear edges: aspect ~ 1000
    flat_axis = rng.integers(0, 12, n_tris)                                         # some triangles inside an axis plane: zero-thickness boxes
    for a in range(3):
        m = flat_axis == a
        e1[m, a] = 0
        e2[m, a] = 0
    verts = np.stack([c, c + size * e1, c + size * e2], axis=1).reshape(-1, 3).astype(np.float32)
    idx = (np.arange(3 * n_tris, dtype=np.uint32) + 1)
    prims = prims + [T.create_mesh_primitives(core, idx, verts, None, white)]
    for k in range(n_spheres):
        p = rng.random(3) * [0.9, 0.9, 0.9] + [0.05, 0.05, -2.95]
        prims.append(T.GeometricPrimitive(T.Sphere(T.ShapeCore(T.translate([float(p[0]), float(p[1]), float(p[2])]), False), float(0.004 + 0.03 * rng.random()), 360.0), white))
    return T.Scene(T.scenes.cornell_lights(), T.BVHAccel(prims, 1)), verts.reshape(-1, 3, 3)


def make_rays(o, d):
    r = np.zeros((o.shape[0], 8), np.float32)
    r[:, 0:3] = o
    r[:, 3] = np.inf
    r[:, 4:7] = d
    return r


def edge_ray_sets(T, ob, cam, flat, tri_verts, seed=11):
    rng = np.random.default_rng(seed)
    bnd = flat.bvh()[0][0]
    lo, hi = bnd[:3], bnd[3:]
    sets = {}
    sets["camera (far origin)"] = ob.generate_rays(cam, T.scenes.camera_sample_grid(cam, 2, 3))
    sets["incoherent"] = T.scenes.incoherent_rays(60000, lo, hi)
    # near-horizontal rays from surface points (the loose test's worst case: the y slab never bounds anything)
    geom = flat.hit_geometry(sets["camera (far origin)"])
    hit = np.abs(geom[:, 6:9]).sum(axis=1) > 0
    p = geom[hit, 0:3]
    phi = rng.random(p.shape[0]) * 2 * np.pi
    dy = (rng.random(p.shape[0]) - 0.3) * 4e-3
    d = np.stack([np.cos(phi), dy, np.sin(phi)], axis=1).astype(np.float32)
    sets["skimming"] = make_rays(p + np.float32(1e-6) * d, d)
    # axis-parallel rays: one or two direction components exactly zero (1/d = Inf; 0 * Inf = NaN on a face)
    n = 30000
    o = (lo + (hi - lo) * rng.random((n, 3), dtype=np.float32)).astype(np.float32)
    d = rng.standard_normal((n, 3)).astype(np.float32)
    z = rng.integers(0, 6, n)
    for a in range(3):
        d[z == a, a] = 0
        d[z == a + 3, a] = 0
        d[z == a + 3, (a + 1) % 3] = 0
    snap = rng.random(n) < 0.5                       # origins snapped to vertex coordinates: rays inside box faces
    v = tri_verts.reshape(-1, 3)
    o[snap] = v[rng.integers(0, v.shape[0], int(snap.sum()))]
    sets["axis-parallel"] = make_rays(o, d)
    # rays inside triangle planes: origin in the plane but outside the triangle, direction in the plane (edge-on)
    k = rng.integers(0, tri_verts.shape[0], n)
    a0, a1, a2 = tri_verts[k, 0], tri_verts[k, 1], tri_verts[k, 2]
    s, t = rng.uniform(-30, 30, (n, 1)).astype(np.float32), rng.uniform(-30, 30, (n, 1)).astype(np.float32)
    o = (a0 + s * (a1 - a0) + t * (a2 - a0)).astype(np.float32)
    tgt = (a0 + rng.random((n, 1), dtype=np.float32) * (a1 - a0) + rng.random((n, 1), dtype=np.float32) * (a2 - a0)).astype(np.float32)
    sets["edge-on"] = make_rays(o, (tgt - o).astype(np.float32))
    # rays through vertices and edge midpoints from far and near
    o = np.where(rng.random((n, 1)) < 0.5, np.float32([0.5, 0.5, 50.0]), (lo + (hi - lo) * rng.random((n, 3), dtype=np.float32))).astype(np.float32)
    tgt = np.where(rng.random((n, 1)) < 0.5, a0, (np.float32(0.5) * a1 + np.float32(0.5) * a2)).astype(np.float32)
    sets["through vertices / edges"] = make_rays(o, (tgt - o).astype(np.float32))
    return sets


def compare_all(T, ob, ctx, scene, flat, sets, oracle_rays=6000):
    osc = ob.OracleScene.from_scene(scene, bvh=flat.bvh())
    L = T.lib()
    counts = np.zeros(4, np.uint64)
    import ctypes as C
    fewer = 0
    for name, rays in sets.items():
        ctx.set_option("traversal", 1)
        ref_hits, ref_occ = flat.trace_closest(rays), flat.trace_any(rays)
        sub = np.random.default_rng(5).choice(rays.shape[0], min(oracle_rays, rays.shape[0]), replace=False)
        t_ref, prim_ref, _, _ = osc.trace_closest(rays[sub])
        assert np.array_equal(ref_hits["prim"][sub], prim_ref), name
        assert_bits_equal(ref_hits["t"][sub], t_ref, f"{name}: literal kernel vs oracle")
        visits = {}
        from conftest import supported
        for trav in supported(ctx, "traversal", (7, 6, 4, 3, 2)):  # 6: two rays per lane; 4: the 8-wide kernel (margin 0 hands its scenes to k_trace3); 7, 6, 4: EXPERIMENTS build
            ctx.set_option("traversal", trav)
            for margin in (0, 14, 16):
                ctx.set_option("slab_margin_log2", margin)
                ctx.set_option("count_visits", 1)
                got = flat.trace_closest(rays)
                ctx.check(L.trhip_last_visit_counts(ctx._h, counts.ctypes.data_as(C.POINTER(C.c_uint64))))  # of the last trace call
                ctx.set_option("count_visits", 0)
                occ = flat.trace_any(rays)
                if trav == 3:  # the any-hit pre-pass on the scene's largest triangles (k_any_occluders) on and off
                    ctx.set_option("occluder_pretest", 0)
                    assert np.array_equal(flat.trace_any(rays), occ), f"{name}: occluder pre-pass changes occlusion (margin 2^-{margin})"
                    ctx.set_option("occluder_pretest", 1)
                visits[(trav, margin)] = int(counts[0])
                what = f"{name}: traversal {trav}, margin 2^-{margin}"
                assert np.array_equal(got["prim"], ref_hits["prim"]), what
                for f in ("t", "b1", "b2"):
                    assert_bits_equal(got[f], ref_hits[f], f"{what}, {f}")
                assert np.array_equal(occ, ref_occ), what
        assert visits[(3, 14)] <= visits[(3, 0)] and visits[(3, 16)] <= visits[(3, 14)]
        fewer += visits[(3, 14)] < visits[(3, 0)]
    ctx.set_option("traversal", 3)
    ctx.set_option("slab_margin_log2", 14)
    return fewer


def test_tight_slab_changes_no_result_stress_scene(T, ob, ctx):
    scene, tri_verts = stress_scene(T)
    flat = scene.flatten(ctx)
    assert flat.bvh()[1].size > 1
    cam = T.scenes.cornell_camera(96)
    sets = edge_ray_sets(T, ob, cam, flat, tri_verts)
    assert compare_all(T, ob, ctx, scene, flat, sets) >= 4   # and it does prune


def test_tight_slab_changes_no_result_height_field(T, ob, ctx):
    scene = T.scenes.mesh_scene(120)   # 28 800 triangles + the Cornell box with its two spheres
    flat = scene.flatten(ctx)
    cam = T.scenes.cornell_camera(128)
    verts, idx, _ = T.scenes.heightfield_mesh(120)
    tri_verts = verts[idx.reshape(-1, 3).astype(np.int64) - 1]
    sets = edge_ray_sets(T, ob, cam, flat, tri_verts)
    assert compare_all(T, ob, ctx, scene, flat, sets) >= 4


def test_tight_slab_frames_identical(T, ctx):
    """Whole frames (all bounces, shadow rays, both integrators' traversal calls) with and without the added clauses, and with and
    without the any-hit pre-pass on the largest triangles."""
    scene = T.scenes.mesh_scene(181)
    cam = T.scenes.cornell_camera(160)
    films, rays = {}, {}
    for margin, pre in ((0, 0), (14, 1), (14, 0)):
        ctx.set_option("slab_margin_log2", margin)
        ctx.set_option("occluder_pretest", pre)
        integ = T.PathIntegrator(cam, T.SeededSampler(8, seed=21), 8)
        films[(margin, pre)] = integ.render(scene, ctx).copy()
        rays[(margin, pre)] = (integ.stats.closest_rays, integ.stats.shadow_rays)
    ctx.set_option("slab_margin_log2", 14)
    ctx.set_option("occluder_pretest", 1)
    assert rays[(0, 0)] == rays[(14, 1)] == rays[(14, 0)]
    assert_bits_equal(films[(14, 1)], films[(0, 0)], "film, tight box test + occluder pre-pass vs the reference's walk")
    assert_bits_equal(films[(14, 0)], films[(0, 0)], "film, tight vs loose box test")
