"""Edge cases of the hot path on the GPU, against the oracle: degenerate inputs, limits and the combinations the main parity
tests do not reach (cropped films, scenes without lights or without geometry, zero-area triangles, several lights of both kinds
in the path integrator, depth 1, the single-leaf threshold)."""
import numpy as np
import pytest

from test_gpu_parity import assert_bits_equal

pytestmark = pytest.mark.gpu


def cornell_tris(T, material=None):
    prims, white = T.scenes.cornell_primitives(spheres=False)
    return prims, white


def test_scene_without_lights_and_depth_one(T, ob, ctx):
    prims, _ = T.scenes.cornell_primitives()
    scene = T.Scene([], T.BVHAccel(prims, 1))
    cam = T.scenes.cornell_camera(24)
    osc = ob.OracleScene.from_scene(scene, bvh=scene.flatten(ctx).bvh())
    for depth in (1, 4):
        ref, _, st = osc.render(cam, "path", 2, depth, seed=3)
        integ = T.PathIntegrator(cam, T.SeededSampler(2, seed=3), depth)
        film = integ.render(scene, ctx)
        assert_bits_equal(film, ref, f"film without lights, depth {depth}")
        assert integ.stats.shadow_rays == 0 == st.shadow_rays and integ.stats.closest_rays == st.closest_rays
    assert np.all(film[..., :3] == 0) and np.all(film[..., 3] != 0)


def test_empty_scene(T, ctx):
    """No primitives at all: every ray misses, the film keeps only its filter weights."""
    scene = T.Scene(T.scenes.cornell_lights(), T.BVHAccel([], 1))
    cam = T.scenes.cornell_camera(16)
    integ = T.PathIntegrator(cam, T.SeededSampler(2, seed=1), 3)
    film = integ.render(scene, ctx)
    assert np.all(film[..., :3] == 0) and np.all(film[..., 3] != 0)
    assert integ.stats.closest_rays == 18 * 18 * 2 and integ.stats.shadow_rays == 0
    rays = T.scenes.incoherent_rays(1000, np.float32([-1, -1, -1]), np.float32([1, 1, 1]))
    flat = scene.flatten(ctx)
    assert np.all(flat.trace_closest(rays)["prim"] == -1) and not flat.trace_any(rays).any()


def test_cropped_film(T, ob, ctx):
    """Film crop window (film.jl:41-44): sample bounds, tiles and the gather all start away from (1, 1)."""
    flt = T.LanczosSincFilter([1.0, 1.0], 3.0)
    film = T.Film([40, 30], T.Bounds2([0.3, 0.2], [0.8, 0.9]), flt, 1.0, 1.0, "")
    base = T.scenes.cornell_camera(40)
    cam = T.PerspectiveCamera(base.camera_to_world, T.Bounds2([-1.0, -1.0], [1.0, 1.0]), 0.0, 1.0, 0.0, 1e6, 90.0, film)
    scene = T.scenes.cornell_scene()
    osc = ob.OracleScene.from_scene(scene, bvh=scene.flatten(ctx).bvh())
    sn = ob.make_sensor(cam, crop=(0.3, 0.2, 0.8, 0.9))
    ref, ref_L, _ = osc.render(cam, "path", 3, 4, seed=8, want_samples=True, sensor=sn)
    integ = T.PathIntegrator(cam, T.SeededSampler(3, seed=8), 4)
    got = integ.render(scene, ctx)
    assert got.shape == ref.shape == (21, 20, 4)
    assert_bits_equal(integ.sample_radiance(scene), ref_L, "per-sample radiance (cropped film)")
    assert_bits_equal(got, ref, "cropped film")
    for mode in (0, 1, 2, 3, 4, 9):  # 3 = splat descriptors; the default is 6 (4 x 4 pixels per thread from packed descriptors)
        ctx.set_option("film_block", mode)
        try:
            assert_bits_equal(T.PathIntegrator(cam, T.SeededSampler(3, seed=8), 4).render(scene, ctx), ref, f"cropped film, film_block {mode}")
        finally:
            ctx.set_option("film_block", 5)


def GP_sphere(T, centre, material, radius=0.03):
    return T.GeometricPrimitive(T.Sphere(T.ShapeCore(T.translate(centre), False), radius, 360.0), material)


def test_degenerate_triangles_and_single_leaf_threshold(T, ob, ctx):
    """Zero-area triangles are never hit (is_degenerate, triangle_mesh.jl:65-68; flagged at commit); 16 primitives make one
    leaf of the LIBRARY's tree (the accelerator of the default hybrid commit, th_trace3c.h), 17 a hierarchy; the canonical tree is the reference's either way."""
    white = T.MatteMaterial(T.ConstantTexture(T.RGBSpectrum(0.9)), T.ConstantTexture(0.0))
    core = T.ShapeCore(T.translate([0, 0, 0]), False)
    verts = np.float32([[0, 0, -2.5], [1, 0, -2.5], [1, 1, -2.5], [0, 1, -2.5], [0.5, 0.5, -2.2], [0.5, 0.5, -2.2], [0.2, 0.2, -2.3], [0.4, 0.4, -2.3], [0.8, 0.8, -2.3]])
    idx = np.uint32([1, 2, 3, 1, 3, 4, 5, 6, 7, 7, 8, 9])  # two real triangles, a point triangle, a collinear one
    for extra in (12, 13):  # 4 + 12 = 16 primitives -> one leaf; 17 -> a tree
        prims = [T.GeometricPrimitive(t, white) for t in T.create_triangle_mesh(core, 4, idx, 9, verts)]
        for k in range(6):  # (more than 32 spheres would leave the scene without an accelerator: th_trace3c.h kCertMaxSpheres)
            prims.append(GP_sphere(T, [0.1 + 0.14 * k, 0.5, -2.4], white))
        n_small = extra - 6
        v2 = np.float32([[0.05 + 0.12 * k + dx, 0.2 + dy, -2.35] for k in range(n_small) for dx, dy in ((0, 0), (0.1, 0), (0, 0.1))])
        prims += [T.GeometricPrimitive(t, white) for t in T.create_triangle_mesh(core, n_small, np.arange(1, 3 * n_small + 1, dtype=np.uint32), 3 * n_small, v2)]
        scene = T.Scene(T.scenes.cornell_lights(), T.BVHAccel(prims, 1))
        flat = scene.flatten(ctx)
        bounds, a, flags, order = flat.bvh()
        mode, acc_nodes, _ = flat.bvh_mode()
        assert mode == 2 and (acc_nodes == 1) == (extra == 12)
        osc = ob.OracleScene.from_scene(scene, bvh=(bounds, a, flags, order))
        cam = T.scenes.cornell_camera(32)
        rays = np.concatenate([ob.generate_rays(cam, T.scenes.camera_sample_grid(cam, 1, 3)), T.scenes.incoherent_rays(20000, np.float32([0, 0, -2.6]), np.float32([1, 1, -2.0]))])
        got = flat.trace_closest(rays)
        t_ref, prim_ref, _, _ = osc.trace_closest(rays)
        assert np.array_equal(got["prim"], prim_ref)
        assert_bits_equal(got["t"], t_ref, "t")
        hit_slots = set(prim_ref[prim_ref >= 0].tolist())
        degenerate_slots = {int(np.flatnonzero(order == k)[0]) for k in (2, 3)}
        assert not (hit_slots & degenerate_slots)
        ref, _, _ = osc.render(cam, "path", 2, 3, seed=6)
        assert_bits_equal(T.PathIntegrator(cam, T.SeededSampler(2, seed=6), 3).render(scene, ctx), ref, f"film ({a.size} nodes)")


def test_path_integrator_with_point_and_spot_lights_and_all_materials(T, ob, ctx):
    """uniform_sample_one_light over three lights (two kinds) on matte / Oren-Nayar / mirror / glass / rough glass / plastic."""
    from test_gpu_parity import MATERIALS
    from test_gpu_sppm import spot_light
    prims, _ = T.scenes.cornell_primitives(spheres=False)
    for k, name in enumerate(MATERIALS):
        prims.append(T.GeometricPrimitive(T.Sphere(T.ShapeCore(T.translate([0.15 + 0.14 * k, 0.13 + 0.05 * (k % 2), -2.3 - 0.08 * k]), False), 0.08, 360.0), MATERIALS[name](T)))
    lights = T.scenes.cornell_lights() + [spot_light(T), T.PointLight(T.translate([0.2, 0.5, -2.1]), T.RGBSpectrum(0.4, 0.6, 0.9))]
    scene = T.Scene(lights, T.BVHAccel(prims, 1))
    cam = T.scenes.cornell_camera(40)
    osc = ob.OracleScene.from_scene(scene, bvh=scene.flatten(ctx).bvh())
    ref, ref_L, _ = osc.render(cam, "path", 4, 6, seed=12, want_samples=True)
    integ = T.PathIntegrator(cam, T.SeededSampler(4, seed=12), 6)
    got = integ.render(scene, ctx)
    assert_bits_equal(integ.sample_radiance(scene), ref_L, "per-sample radiance")
    assert_bits_equal(got, ref, "film")


def test_material_less_primitive_is_rejected_by_render_but_traced(T, ctx):
    core = T.ShapeCore(T.translate([0, 0, 0]), False)
    tris = T.create_triangle_mesh(core, 1, np.uint32([1, 2, 3]), 3, np.float32([[0, 0, -2.5], [1, 0, -2.5], [1, 1, -2.5]]))
    scene = T.Scene(T.scenes.cornell_lights(), T.BVHAccel([T.GeometricPrimitive(tris[0], None)], 1))
    cam = T.scenes.cornell_camera(16)
    flat = scene.flatten(ctx)
    rays = T.scenes.incoherent_rays(2000, np.float32([0, 0, -2.6]), np.float32([1, 1, -2.0]))
    assert (flat.trace_closest(rays)["prim"] >= -1).all()
    with pytest.raises(T.TraceHipError, match="material-less"):
        T.PathIntegrator(cam, T.SeededSampler(1), 2).render(scene, ctx)


def test_bench_size_properties(T, ctx):
    """At the size of the bench (1024 x 1024, S-cornell, depth 8; all 256 spp) the oracle is too
    slow to compare against; size-independent properties instead: the render is reproducible bit for bit, does not depend on
    the traversal kernel, the film-gather variant or the batch size, and the two halves of the sample range add up to the whole."""
    scene, cam = T.scenes.cornell_scene(), T.scenes.cornell_camera(1024)

    def render(spp=256, offset=0, **opts):
        for k, v in opts.items():
            ctx.set_option(k, v)
        try:
            return T.PathIntegrator(cam, T.SeededSampler(spp, seed=0x5EED0001, sample_offset=offset), 8).render(scene, ctx).copy()
        finally:
            for k, v in {"traversal": 3, "film_block": 5, "batch_paths": 0, "overlap": 1}.items():
                ctx.set_option(k, v)

    a = render()
    assert np.isfinite(a).all() and a[..., 3].min() > 0
    assert_bits_equal(render(), a, "second run")
    assert_bits_equal(render(traversal=1), a, "literal traversal kernel")
    assert_bits_equal(render(film_block=0), a, "one film pixel per thread")
    assert_bits_equal(render(film_block=3), a, "film gather from splat descriptors")
    assert_bits_equal(render(film_block=2), a, "1 x 4 blocks recomputing the ranges (round 2's default)")
    assert_bits_equal(render(batch_paths=16 * 1026 * 1026, overlap=0), a, "four batches, one stream")
    assert_bits_equal(render(traversal=3), a, "binary children-in-parent walk (k_trace_leaf here: one-leaf scene)")
    h0, h1 = render(128, 0), render(128, 128)
    np.testing.assert_allclose(h0 + h1, a, rtol=3e-5, atol=1e-5)
    assert np.array_equal((h0 + h1)[..., 3] > 0, a[..., 3] > 0)


def test_device_detmath_equals_host_detmath(T, ctx):
    """include/trace_detmath.h is shared by the oracle (g++), the library's host side and the kernels (hipcc, device): the device
    copy must return the host copy's Float32 bit patterns — sin, cos, tan, atan2, acos, log and both parts of the fused sincos —
    over their working ranges, the special values and the range-reduction boundaries."""
    rng = np.random.default_rng(7)
    wide = np.concatenate([rng.uniform(-50, 50, 200000), rng.uniform(-1e4, 1e4, 50000), rng.normal(size=50000) * 1e-3,
                           np.arange(-64, 65) * (np.pi / 4), [0.0, -0.0, 1e-30, -1e-30, 1e-45, np.inf, -np.inf, np.nan, 1e10, -1e10]]).astype(np.float32)
    unit = np.concatenate([rng.uniform(-1, 1, 200000), [1.0, -1.0, 0.0, -0.0, 1.0 + 1e-7, -1.0 - 1e-7, np.nan]]).astype(np.float32)
    pos = np.concatenate([10.0 ** rng.uniform(-30, 30, 200000), [0.0, -0.0, 1.0, 1e-3, np.inf, -1.0, np.nan, 1e-45]]).astype(np.float32)
    cases = [(0, wide, None), (1, wide, None), (2, wide, None), (6, wide, None), (7, wide, None), (4, unit, None), (5, pos, None),
             (3, wide, np.roll(wide, 7919))]
    for fn, x, y in cases:
        host = T._ffi.detmath(fn, x, y)
        dev = ctx.detmath(fn, x, y)
        na, nb = np.isnan(host), np.isnan(dev)
        assert np.array_equal(na, nb), f"fn {fn}: NaN pattern"
        bad = (host.view(np.uint32) != dev.view(np.uint32)) & ~na
        assert not bad.any(), f"fn {fn}: {int(bad.sum())} of {x.size} values differ, e.g. x = {x[bad][0]!r}: host {host[bad][0]!r} device {dev[bad][0]!r}"


def test_set_bvh_rejects_trees_the_kernels_cannot_walk(T, ob, ctx):
    """trhip_scene_set_bvh: a node array that is not ONE tree in the reference's depth-first layout (a child index that closes a
    cycle would make the traversal kernels spin forever) or that is deeper than the 64-entry stack (the reference throws a
    BoundsError there, bvh.jl:222) is an error, not a hang or a silent miss; a valid tree whose boxes do not nest is walked by
    the literal kernels and still agrees with the oracle."""
    scene = T.scenes.mesh_scene(8)
    flat = scene.flatten(ctx)
    bounds, a, flags, order = [x.copy() for x in flat.bvh()]
    inner = np.flatnonzero((flags & 3) != 3)
    bad = a.copy()
    bad[inner[3]] = inner[3]  # second child = the node itself: a cycle
    with pytest.raises(T.TraceHipError):
        flat.set_bvh(bounds, bad, flags, order)
    bad = a.copy()
    bad[inner[2]] = inner[2] + 1  # second child = first child
    with pytest.raises(T.TraceHipError):
        flat.set_bvh(bounds, bad, flags, order)
    # a chain deeper than 64 levels: node i = interior {leaf, rest}
    n = 70
    prims = scene.aggregate.primitives
    tri = next(p for p in prims if isinstance(p, T.MeshPrimitives))
    verts, idx = tri.mesh.vertices, tri.mesh.indices.reshape(-1, 3)[:n]
    chain_scene = T.Scene(scene.lights, T.BVHAccel([T.create_mesh_primitives(T.ShapeCore(T.translate([0, 0, 0]), False), idx.reshape(-1), verts, tri.mesh.normals, tri.material)], 1))
    cflat = chain_scene.flatten(ctx)
    big = np.array([[-10, -10, -10, 10, 10, 10]], np.float32)
    cb, ca, cf = [], [], []
    for i in range(n - 1):
        cb += [big[0], big[0]]
        ca += [2 * i + 2, i]
        cf += [0, (1 << 2) | 3]
    cb.append(big[0])
    ca.append(n - 1)
    cf.append((1 << 2) | 3)
    with pytest.raises(T.TraceHipError) as e:
        cflat.set_bvh(np.array(cb, np.float32), np.array(ca, np.uint32), np.array(cf, np.uint32), np.arange(n, dtype=np.uint32))
    assert "64" in str(e.value)
    # loose boxes that do not nest (every node the whole scene, leaves too small for their triangles' neighbours): literal kernels, same hits as the oracle
    loose = bounds.copy()
    loose[inner] += np.float32([0.01, 0.01, 0.01, -0.01, -0.01, -0.01]) * 0  # interiors unchanged
    leaf = np.flatnonzero((flags & 3) == 3)
    loose[leaf[::2], :3] -= np.float32(0.05)  # every other leaf box grown beyond its parent
    loose[leaf[::2], 3:] += np.float32(0.05)
    flat.set_bvh(loose, a, flags, order)
    try:
        osc = ob.OracleScene.from_scene(scene, bvh=(loose, a, flags, order))
        wb = osc.world_bound()
        rays = T.scenes.incoherent_rays(20000, wb[:3], wb[3:])
        got = flat.trace_closest(rays)
        t_ref, prim_ref, _, _ = osc.trace_closest(rays)
        assert np.array_equal(got["prim"], prim_ref)
        assert_bits_equal(got["t"], t_ref, "t (foreign tree whose boxes do not nest)")
        assert np.array_equal(flat.trace_any(rays), osc.trace_any(rays)[0])
    finally:
        flat.set_bvh(bounds, a, flags, order)


def test_nested_bvh_as_primitive(T, ob, ctx):
    """test/test_intersection.jl:129-156: eight unit spheres at (i, i, 0), i = 0:3:21; BVHAccel(1:4) is a PRIMITIVE of the BVHAccel that
    also holds spheres 5:8.  Expected (Appendix B): the ray o = (-2, 0, 0), d = (1, 0, 0) hits at t = 1, the ray o = (0, 18, 0),
    d = (1, 0, 0) at t = 17.  The host splices nested aggregates (one BVH over the flat list)."""
    matte = T.MatteMaterial(T.ConstantTexture(T.RGBSpectrum(0.5)), T.ConstantTexture(0.0))
    spheres = [T.GeometricPrimitive(T.Sphere(T.ShapeCore(T.translate([float(i), float(i), 0.0]), False), 1.0, 360.0), matte) for i in range(0, 22, 3)]
    inner = T.BVHAccel(spheres[:4], 1)
    outer = T.BVHAccel([inner] + spheres[4:], 1)
    scene = T.Scene([], outer)
    ctx.set_option("tiny_scene_prims", 0)
    try:
        flat = scene.flatten(ctx)
    finally:
        ctx.set_option("tiny_scene_prims", 16)
    bounds = flat.bvh()[0]
    assert np.allclose(bounds[0], [-1, -1, -1, 22, 22, 1])
    rays = np.float32([[-2, 0, 0, np.inf, 1, 0, 0, 0], [0, 18, 0, np.inf, 1, 0, 0, 0], [0, 50, 0, np.inf, 1, 0, 0, 0]])
    hits = flat.trace_closest(rays)
    assert hits["t"][0] == 1.0 and hits["t"][1] == 17.0 and hits["prim"][2] == -1
    osc = ob.OracleScene.from_scene(scene, bvh=flat.bvh())
    t, prim, _, _ = osc.trace_closest(rays)
    assert np.array_equal(hits["prim"], prim) and np.array_equal(hits["t"].view(np.uint32), t.view(np.uint32))


@pytest.mark.parametrize("rows", [1, 2, 5])
def test_banded_frame_equals_whole_frame(T, ob, ctx, rows):
    """A frame whose per-sample buffers do not fit in HBM (4096^2 x 1024 spp: 412 GB) is rendered in bands of whole tile rows into one
    film (option band_tile_rows forces it here): every film pixel still receives its tiles in the reference's k order, so the film is
    the one-band film — and the oracle's — bit for bit, with every film-gather variant."""
    scene, cam = T.scenes.cornell_scene(), T.scenes.cornell_camera(100)
    whole = T.PathIntegrator(cam, T.SeededSampler(3, seed=12), 5).render(scene, ctx).copy()
    osc = ob.OracleScene.from_scene(scene, bvh=scene.flatten(ctx).bvh())
    ref, _, st_ref = osc.render(cam, "path", 3, 5, seed=12, threads=ob.lib().orc_num_threads())
    assert_bits_equal(whole, ref, "one band vs oracle")
    for film_block in (6, 2, 3, 0, 1, 5):
        ctx.set_option("band_tile_rows", rows)
        ctx.set_option("film_block", film_block)
        try:
            integ = T.PathIntegrator(cam, T.SeededSampler(3, seed=12), 5)
            banded = integ.render(scene, ctx).copy()
        finally:
            ctx.set_option("band_tile_rows", 0)
            ctx.set_option("film_block", 5)
        assert_bits_equal(banded, whole, f"bands of {rows} tile rows, film_block {film_block}")
        assert integ.stats.camera_samples == 102 * 102 * 3 and integ.stats.closest_rays == st_ref.closest_rays and integ.stats.launches_film == -(-7 // rows)
