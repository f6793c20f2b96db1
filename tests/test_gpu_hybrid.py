"""Row g1 (run with -m gpu): the DEFAULT configuration returns Trace.jl's answers.

A scene committed with default options holds two trees (trhip_scene_bvh_mode == 2): the canonical one is the reference's own construction
(accel/bvh.jl:55-206 — equal, node for node, to the oracle's restatement oracle/orc_build.h), the library's binned-SAH tree rides along as an accelerator.
Closest-hit rays walk the accelerator under the order-independence certificate of csrc/th_trace3c.h; the rays it cannot certify — a sphere entered from
inside (sphere.jl:137-138), near ties, grazed leaf boxes, zero direction components — are re-walked on the canonical tree in the reference's order.  So:

* hits, occlusion, per-sample radiance and films must equal, BIT FOR BIT, the oracle walking ITS OWN reference tree — which none of the rays walked
  unless flagged (trhip_stats.fallback_rays counts those; the tests bound their share);
* the same scene with option "hybrid" = 0 (every ray on the canonical tree) must give the same bits on millions of rays: the certificate's claim itself;
* a host's own tree handed over with trhip_scene_set_bvh (the Julia shim's EXACT_TREE route) gets an accelerator too.
"""
import os

import numpy as np
import pytest
from conftest import experiments

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bits_equal(a, b, what):
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    na, nb = np.isnan(a), np.isnan(b)
    assert np.array_equal(na, nb), f"{what}: NaN pattern differs"
    bad = (bits(a) != bits(b)) & ~na
    assert not bad.any(), f"{what}: {int(bad.sum())} of {a.size} values differ, first at {np.argwhere(bad)[0]}"


@pytest.fixture
def hyb_ctx(ctx):
    ctx.set_option("bvh_builder", -1)
    ctx.set_option("hybrid", 1)
    ctx.set_option("traversal", 3)
    yield ctx
    ctx.set_option("bvh_builder", -1)
    ctx.set_option("hybrid", 1)
    ctx.set_option("traversal", 3)
    ctx.set_option("count_visits", 0)


def cases(T):
    yield "shadows", T.scenes.shadows_scene(), T.scenes.shadows_camera(48), ([-1.2, -0.3, -3.2], [1.3, 1.2, 1.0])
    yield "cornell", T.scenes.cornell_scene(), T.scenes.cornell_camera(40), ([0, 0, -3], [1, 1, -2])
    yield "mesh64", T.scenes.mesh_scene(64), T.scenes.cornell_camera(48), ([0, 0, -3], [1, 1, -2])
    # a closed object whose SAH leaves straddle the reference's leaves: the accelerator is regrouped by canonical leaf at commit (tu_scene.hip conform_accelerator)
    yield "blob24", T.scenes.blob_scene(24), T.scenes.cornell_camera(48), ([0, 0, -3], [1, 1, -2])
    ply = os.path.join(GOLDEN, "caustic-glass.ply")
    if os.path.exists(ply):
        yield "caustic-glass.ply", T.scenes.caustic_scene(ply), T.scenes.caustic_camera(32), None


def special_rays(T, lo, hi, n, seed):
    """Rays the certificate must flag or survive: origins inside the Cornell spheres, zero direction components, finite t_max, surface-spawned rays."""
    f32 = np.float32
    rng = np.random.default_rng(seed)
    lo, hi = np.asarray(lo, f32), np.asarray(hi, f32)

    def rays_from(o, d, tmax=np.inf):
        r = np.empty((o.shape[0], 8), f32)
        r[:, 0:3], r[:, 3], r[:, 4:7], r[:, 7] = o, tmax, d, 0.0
        return r

    parts = []
    o = rng.uniform(lo, hi, (n, 3)).astype(f32)
    d = rng.normal(size=(n, 3)).astype(f32)
    parts.append(rays_from(o, d, rng.uniform(0.0, 0.8, n).astype(f32)))  # finite t_max
    d2 = d.copy()
    d2[np.arange(n), rng.integers(0, 3, n)] = 0.0  # a zero component
    d2[: n // 4, 1] = -0.0
    parts.append(rays_from(o, d2))
    for c, r in (([0.3, 0.25, -2.7], 0.25), ([0.7, 0.2, -2.35], 0.2), ([-0.3, 0.1, -2.0], 0.3)):  # inside / on spheres of the Cornell / shadows scenes (elsewhere: plain rays)
        u = rng.normal(size=(n // 2, 3))
        u /= np.linalg.norm(u, axis=1, keepdims=True)
        oo = (np.asarray(c) + u * (r * rng.uniform(0.0, 1.02, (n // 2, 1)))).astype(f32)
        parts.append(rays_from(oo, rng.normal(size=(n // 2, 3)).astype(f32)))
    return np.concatenate(parts)


def test_default_commit_is_hybrid_and_equals_the_oracle_on_its_own_reference_tree(T, ob, hyb_ctx):
    ctx = hyb_ctx
    for name, scene, cam, box in cases(T):
        osc = ob.OracleScene.from_scene(scene)  # bvh=None: the oracle builds the reference's tree itself (orc_build.h)
        rb, ra, rf, ro = osc.get_bvh()
        flat = scene.flatten(ctx)
        mode, acc_nodes, acc_depth = flat.bvh_mode()
        assert mode == 2 and acc_nodes >= 1, f"{name}: default commit is not hybrid (mode {mode})"
        b, a, f, o = flat.bvh()
        assert a.size == ra.size and np.array_equal(o, ro) and np.array_equal(a, ra) and np.array_equal(f, rf), f"{name}: canonical tree differs from the oracle's reference tree"
        assert np.array_equal(bits(b), bits(rb)), f"{name}: node bounds differ"
        ab, aa, af, ao = flat.accelerator()
        assert aa.size == acc_nodes and sorted(ao.tolist()) == list(range(o.size)), f"{name}: the accelerator does not hold every primitive once"
        assert acc_nodes == 1 or not (aa.size == a.size and np.array_equal(aa, a) and np.array_equal(ao, o)), f"{name}: the accelerator IS the reference tree: the test proves nothing"
        wb = osc.world_bound()
        lo, hi = (wb[:3], wb[3:]) if box is None else (np.float32(box[0]), np.float32(box[1]))
        rays = np.concatenate([ob.generate_rays(cam, T.scenes.camera_sample_grid(cam, 1, seed=3)), T.scenes.incoherent_rays(40000, lo, hi, seed=17), special_rays(T, lo, hi, 6000, 5)])
        t_ref, prim_ref, _, _ = osc.trace_closest(rays)
        occ_ref, _ = osc.trace_any(rays)
        hits = flat.trace_closest(rays)
        assert np.array_equal(hits["prim"], prim_ref), f"{name}: primitives differ from the oracle on its reference tree in {int((hits['prim'] != prim_ref).sum())} rays"
        assert np.array_equal(bits(hits["t"]), bits(t_ref)), f"{name}: t differs"
        assert np.array_equal(flat.trace_any(rays), occ_ref), f"{name}: occlusion differs"
        ctx.set_option("hybrid", 0)  # barycentrics: against the canonical tree's own walk (the oracle's entry point does not return them)
        ref_hits = flat.trace_closest(rays)
        ctx.set_option("hybrid", 1)
        for k in ("prim", "t", "b1", "b2"):
            assert np.array_equal(np.ascontiguousarray(hits[k]).view(np.uint32), np.ascontiguousarray(ref_hits[k]).view(np.uint32)), f"{name}: {k} differs between hybrid on / off"
        ref_film, ref_L, _ = osc.render(cam, "path", 3, 8, seed=31, want_samples=True)
        integ = T.PathIntegrator(cam, T.SeededSampler(3, seed=31), 8)
        film = integ.render(scene, ctx)
        st = integ.stats
        assert int(st.traversal) == 9, f"{name}: the frame did not run the hybrid walk (traversal {st.traversal})"
        assert st.fallback_rays <= 0.25 * st.closest_rays, f"{name}: {st.fallback_rays} of {st.closest_rays} closest-hit rays went to the canonical tree"
        assert_bits_equal(integ.sample_radiance(scene), ref_L, f"{name}: per-sample radiance")
        assert_bits_equal(film, ref_film, f"{name}: film")
        # Whitted through the same launches
        ref_w, _, _ = osc.render(cam, "whitted", 2, 5, seed=4)
        assert_bits_equal(T.WhittedIntegrator(cam, T.SeededSampler(2, seed=4), 5).render(scene, ctx), ref_w, f"{name}: Whitted film")
        scene._flat = None
        flat.free()


def test_hybrid_on_equals_hybrid_off_on_a_large_mesh(T, ob, hyb_ctx):
    """The certificate's claim on 2.6 M rays of a 131 k-triangle height field with the two spheres: the accelerator walk + fallback returns what the canonical tree alone returns."""
    ctx = hyb_ctx
    scene, cam = T.scenes.mesh_scene(256), T.scenes.cornell_camera(512)
    flat = scene.flatten(ctx)
    assert flat.bvh_mode()[0] == 2
    lo, hi = np.float32([0, 0, -3]), np.float32([1, 1, -2])
    rays = np.concatenate([ob.generate_rays(cam, T.scenes.camera_sample_grid(cam, 1, seed=3)), T.scenes.incoherent_rays(1 << 21, lo, hi, seed=23), special_rays(T, lo, hi, 60000, 11)])
    got = flat.trace_closest(rays)
    occ = flat.trace_any(rays)
    ctx.set_option("hybrid", 0)
    ref = flat.trace_closest(rays)
    occ_ref = flat.trace_any(rays)
    ctx.set_option("hybrid", 1)
    for k in ("prim", "t", "b1", "b2"):
        assert np.array_equal(np.ascontiguousarray(got[k]).view(np.uint32), np.ascontiguousarray(ref[k]).view(np.uint32)), f"{k} differs in {int((got[k] != ref[k]).sum())} rays"
    assert np.array_equal(occ, occ_ref)
    # … and a frame: same film, and the bulk of the rays stayed on the accelerator
    integ = T.PathIntegrator(cam, T.SeededSampler(2, seed=9), 8)
    film = integ.render(scene, ctx)
    L = integ.sample_radiance(scene)
    st = integ.stats
    assert int(st.traversal) == 9 and 0 < st.fallback_rays < 0.1 * st.closest_rays, f"fallback {st.fallback_rays} of {st.closest_rays}"
    ctx.set_option("hybrid", 0)
    integ2 = T.PathIntegrator(cam, T.SeededSampler(2, seed=9), 8)
    film2 = integ2.render(scene, ctx)
    assert int(integ2.stats.traversal) == 3 and integ2.stats.fallback_rays == 0
    assert_bits_equal(L, integ2.sample_radiance(scene), "per-sample radiance, hybrid on vs off")
    assert_bits_equal(film, film2, "film, hybrid on vs off")
    ctx.set_option("hybrid", 1)
    # the oracle on a subsample (it walks the reference tree it built itself)
    osc = ob.OracleScene.from_scene(scene)
    sub = np.random.default_rng(5).choice(rays.shape[0], 1 << 16, replace=False)
    t_ref, prim_ref, _, _ = osc.trace_closest(rays[sub])
    assert np.array_equal(got["prim"][sub], prim_ref) and np.array_equal(bits(got["t"][sub]), bits(t_ref))
    scene._flat = None
    flat.free()


def test_host_tree_through_set_bvh_gets_an_accelerator(T, ob, hyb_ctx):
    """TraceHIP.jl's EXACT_TREE route: the host's BVHAccel is the canonical tree, the library adds its own as the accelerator."""
    ctx = hyb_ctx
    scene, cam = T.scenes.mesh_scene(40), T.scenes.cornell_camera(40)
    osc = ob.OracleScene.from_scene(scene)
    tree = osc.get_bvh()
    ctx.set_option("bvh_builder", 0)
    flat = scene.flatten(ctx)  # the library's tree alone first
    assert flat.bvh_mode()[0] == 0
    ctx.set_option("bvh_builder", -1)
    flat.set_bvh(*tree)
    assert flat.bvh_mode()[0] == 2
    rays = np.concatenate([ob.generate_rays(cam, T.scenes.camera_sample_grid(cam, 1, seed=5)), T.scenes.incoherent_rays(20000, np.float32([0, 0, -3]), np.float32([1, 1, -2]), seed=9)])
    t_ref, prim_ref, _, _ = osc.trace_closest(rays)
    hits = flat.trace_closest(rays)
    assert np.array_equal(hits["prim"], prim_ref) and np.array_equal(bits(hits["t"]), bits(t_ref))
    ref_film, _, _ = osc.render(cam, "path", 2, 5, seed=8)
    integ = T.PathIntegrator(cam, T.SeededSampler(2, seed=8), 5)
    film = integ.render(scene, ctx)
    assert int(integ.stats.traversal) == 9
    assert_bits_equal(film, ref_film, "film through set_bvh(reference tree) + accelerator")
    scene._flat = None
    flat.free()


def test_explicit_builders_keep_their_single_tree(T, ob, hyb_ctx):
    ctx = hyb_ctx
    scene = T.scenes.mesh_scene(24)
    for builder, mode in ((0, 0), (2, 1), (4, 2), (-1, 2)):
        ctx.set_option("bvh_builder", builder)
        flat = scene.flatten(ctx)
        assert flat.bvh_mode()[0] == mode, f"bvh_builder {builder}: mode {flat.bvh_mode()[0]}"
        assert flat.bvh_note() == ""
        scene._flat = None
        flat.free()
    ctx.set_option("bvh_builder", -1)


def test_sppm_in_hybrid_mode(T, ob, hyb_ctx):
    ply = os.path.join(GOLDEN, "caustic-glass.ply")
    scene = T.scenes.caustic_scene(ply if os.path.exists(ply) else "")
    cam = T.scenes.caustic_camera(24)
    osc = ob.OracleScene.from_scene(scene)
    flat = scene.flatten(hyb_ctx)
    assert flat.bvh_mode()[0] == 2 and np.array_equal(flat.bvh()[1], osc.get_bvh()[1])
    integ = T.SPPMIntegrator(cam, 0.075, 5, 2, 20000, seed=11)
    integ.render(scene, hyb_ctx)
    got = integ.state()
    ref = osc.sppm(cam, 0.075, 5, 2, 20000, seed=11)
    assert np.array_equal(got["M"], ref["M"]) and np.array_equal(bits(got["radius"]), bits(ref["radius"])) and np.array_equal(got["N"], ref["N"])
    assert_bits_equal(got["Ld"], ref["Ld"], "Ld")
    scene._flat = None
    flat.free()


def _sphere_scene(T, n_spheres):
    prims, _ = T.scenes.cornell_primitives(spheres=False)
    white = T.MatteMaterial(T.ConstantTexture(T.RGBSpectrum(0.7)), T.ConstantTexture(0.0))
    glass = T.GlassMaterial(T.ConstantTexture(T.RGBSpectrum(1.0)), T.ConstantTexture(T.RGBSpectrum(1.0)), T.ConstantTexture(0.0), T.ConstantTexture(0.0), T.ConstantTexture(1.5), True)
    for k in range(n_spheres):
        c = [0.1 + 0.1 * (k % 9) + 0.02 * (k // 9), 0.1 + 0.25 * (k % 3) + 0.05 * (k // 9), -2.8 + 0.2 * (k % 4) + 0.04 * (k // 12)]
        prims.append(T.GeometricPrimitive(T.Sphere(T.ShapeCore(T.translate(c), False), 0.09 if k < 9 else 0.04, 360.0), glass if k % 2 else white))
    return T.Scene(T.scenes.cornell_lights(), T.BVHAccel(prims, 1))


@pytest.mark.parametrize("n_spheres", [9, 14])
def test_more_than_ten_spheres_stay_on_the_accelerator(T, ob, hyb_ctx, n_spheres):
    """The order word of a primitive record has bits for the scene's first ten spheres (th_trace3c.h kCertOrderSpheres); a scene of up to 32 spheres (kCertMaxSpheres) still commits
    with both trees: every ray is tested against every sphere before its walk, and a ray that starts INSIDE one of the later spheres (glass: every refracted ray does) is handed to
    the reference-order walk.  Film and per-sample radiance: the oracle's on the reference's tree, bit for bit."""
    scene = _sphere_scene(T, n_spheres)
    flat = scene.flatten(hyb_ctx)
    assert flat.bvh_mode()[0] == 2 and flat.bvh_note() == "", flat.bvh_note()
    osc = ob.OracleScene.from_scene(scene)
    assert np.array_equal(flat.bvh()[1], osc.get_bvh()[1])
    cam = T.scenes.cornell_camera(32)
    ref, ref_L, _ = osc.render(cam, "path", 2, 6, seed=3, want_samples=True)
    integ = T.PathIntegrator(cam, T.SeededSampler(2, seed=3), 6)
    assert_bits_equal(integ.render(scene, hyb_ctx), ref, f"film ({n_spheres} spheres, hybrid)")
    assert_bits_equal(integ.sample_radiance(scene), ref_L, f"per-sample radiance ({n_spheres} spheres, hybrid)")
    assert integ.stats.traversal == 9 and 0 < integ.stats.fallback_rays < integ.stats.closest_rays
    scene._flat = None
    flat.free()


def test_more_than_32_spheres_keep_the_canonical_tree_alone(T, ob, hyb_ctx):
    """The certified walk tests every sphere for every ray (th_trace3c.h kCertMaxSpheres = 32): a scene with more commits without an accelerator (mode 1) and every ray
    walks the reference's tree — same bits as the oracle."""
    scene = _sphere_scene(T, 36)
    flat = scene.flatten(hyb_ctx)
    assert flat.bvh_mode()[0] == 1 and "spheres" in flat.bvh_note()  # (trhip_scene_bvh_note says why there is one tree)
    osc = ob.OracleScene.from_scene(scene)
    assert np.array_equal(flat.bvh()[1], osc.get_bvh()[1])
    cam = T.scenes.cornell_camera(24)
    ref, _, _ = osc.render(cam, "path", 2, 5, seed=3)
    integ = T.PathIntegrator(cam, T.SeededSampler(2, seed=3), 5)
    assert_bits_equal(integ.render(scene, hyb_ctx), ref, "film (36 spheres, canonical tree alone)")
    assert integ.stats.fallback_rays == 0 and integ.stats.traversal == 3
    scene._flat = None
    flat.free()


@pytest.mark.parametrize("opts", [{"hybrid": 0}, pytest.param({"leaf_queue": 1}, marks=experiments("leaf_queue", 1)), {"any_on_accelerator": 1}, {"any_on_accelerator": 0}, {"node_layout": 1}, {"overlap": 0},
                                  {"pipelines": 2}, pytest.param({"traversal": 7}, marks=experiments("traversal", 7)), {"traversal": 1}, {"slab_margin_log2": 0}, {"count_visits": 1},
                                  pytest.param({"leaf_queue": 1, "overlap": 0}, marks=experiments("leaf_queue", 1))],
                         ids=lambda o: ",".join(f"{k}={v}" for k, v in o.items()))
def test_options_that_change_how_a_frame_is_computed_leave_the_frame_alone(T, opts):
    """A two-tree scene under the switches of the hybrid mode and of the frame loop: film and per-sample radiance bit-equal to the default configuration's
    (tools/option_exactness.py runs the long list on four scenes)."""
    scene, cam = T.scenes.mesh_scene(48), T.scenes.cornell_camera(64)
    out = []
    for o in ({}, opts):
        c = T.Context(0)
        try:
            for k, v in o.items():
                c.set_option(k, v)
            integ = T.PathIntegrator(cam, T.SeededSampler(3, seed=5), 6)
            film = integ.render(scene, c).copy()
            out.append((film, integ.sample_radiance(scene).copy(), scene._flat.bvh_mode()[0]))
        finally:
            if scene._flat is not None:
                scene._flat.free()
                scene._flat = None
            c.close()
    assert out[0][2] == 2 and out[1][2] == 2
    assert_bits_equal(out[1][1], out[0][1], f"per-sample radiance under {opts}")
    assert_bits_equal(out[1][0], out[0][0], f"film under {opts}")


def test_a_scene_whose_reference_construction_fails_keeps_the_library_tree(T, hyb_ctx):
    """240 triangles whose sizes and positions grow geometrically (x 1.45 each, from 1e-8): accel/bvh.jl:87-185 peels them off a few per level and ends more than 64 levels deep — deeper than the
    64-entry stack bvh.jl:222 walks with, so Trace.jl itself cannot trace this scene.  The default commit says so (trhip_scene_bvh_note), keeps the library's tree as the canonical
    one (mode 0) or, when three times its four-wide depth fits the stack, ALSO as its own four-wide accelerator (mode 3: what the 10.5 M-triangle scene of test_gpu_scale.py gets);
    either way the default kernels return what the literal accel/bvh.jl loop returns on that tree, bit for bit."""
    import attack_scenes as A
    n, base, s0 = 240, 1.45, 1e-8
    rng = np.random.default_rng(7)
    tris = np.empty((n, 3, 3), np.float64)
    for i in range(n):
        sc = s0 * base ** i
        tris[i] = np.array([sc, 0.37 * sc, -0.2 * sc]) + 0.05 * sc * rng.uniform(-1.0, 1.0, (3, 3))
    pb = np.concatenate([tris.min(axis=1), tris.max(axis=1)], axis=1).astype(np.float32)
    assert T._ffi.build_bvh_host(pb, 1, builder=2)[4] > 64  # (the host-only entry builds it to the end)
    scene, tri32, _, _ = A.build_scene(T, tris.reshape(-1, 3), [], 1.0, (0.0, 0.0, 0.0))
    flat = scene.flatten(hyb_ctx)
    mode = flat.bvh_mode()[0]
    assert mode in (0, 3) and "depth" in flat.bvh_note(), (mode, flat.bvh_note())
    # rays from a point off the chain towards points on triangles 40 … 120 (coordinates 3e-2 … 2e11) and into empty space
    k = rng.integers(40, 120, 4000)
    w = rng.dirichlet([1.0, 1.0, 1.0], 4000)
    target = np.einsum("nk,nkc->nc", w, tri32[k].astype(np.float64))
    origin = np.array([-3.0, 2.0, 5.0]) + rng.uniform(-1.0, 1.0, (4000, 3))
    d = target - origin
    d[3000:] = rng.normal(size=(1000, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = A.rays8(origin.astype(np.float32), d.astype(np.float32))
    got = {}
    for trav in (3, 1):
        hyb_ctx.set_option("traversal", trav)
        try:
            got[trav] = (flat.trace_closest(rays), flat.trace_any(rays))
        finally:
            hyb_ctx.set_option("traversal", 3)
    assert (got[1][0]["prim"] >= 0).sum() > 1000
    assert np.array_equal(got[3][0]["prim"], got[1][0]["prim"])
    for f in ("t", "b1", "b2"):
        assert_bits_equal(got[3][0][f], got[1][0][f], f"closest-hit {f} on a scene the reference cannot build")
    assert np.array_equal(got[3][1], got[1][1])
    scene._flat = None
    flat.free()


def test_the_library_names_the_closest_hit_kernel_it_launches(T):
    """trhip_closest_kernel_name: launch_trace's own decision for the scene under the context's current options — what bench.py's roofline and its PMC filter name.  The
    frame's statistics (trhip_stats.traversal) must agree with the name."""
    cases = [({}, T.scenes.mesh_scene(48), "k_trace3c4", 9), ({"wide4": 0}, T.scenes.mesh_scene(48), "k_trace3c", 9), ({}, T.scenes.cornell_scene(), "k_trace_leaf_c", 9),
             ({"bvh_builder": 0}, T.scenes.mesh_scene(48), "k_trace3", 3), ({"bvh_builder": 0}, T.scenes.cornell_scene(), "k_trace_leaf", 5)]
    for opts, scene, name, trav in cases:
        c = T.Context(0)
        try:
            for k, v in opts.items():
                c.set_option(k, v)
            flat = scene.flatten(c)
            assert flat.closest_kernel_name() == name, (opts, flat.closest_kernel_name())
            integ = T.PathIntegrator(T.scenes.cornell_camera(32), T.SeededSampler(2, seed=3), 4)
            integ.render(scene, c)
            assert int(integ.stats.traversal) == trav, (opts, integ.stats.traversal)
            if name.startswith("k_trace3c"):
                c.set_option("hybrid", 0)  # the same scene, every ray on the canonical tree
                assert flat.closest_kernel_name() == "k_trace3"
                c.set_option("hybrid", 1)
                c.set_option("traversal", 1)  # the literal loop walks the canonical tree
                assert flat.closest_kernel_name() == "k_trace_closest"
        finally:
            if scene._flat is not None:
                scene._flat.free()
                scene._flat = None
            c.close()
