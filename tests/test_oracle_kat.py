"""Pin the CPU oracle against every known-answer vector the reference's own tests hold for the hot path
(/root/reference/test/{runtests,test_intersection,test_materials}.jl).  The expectations below are
the literals of those tests; `≈` is Julia's isapprox (rtol = sqrt(eps(Float32)); exact comparison against 0).
"""
import ctypes as C

import numpy as np
import pytest

RTOL = 0.00034526698


def approx(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) <= RTOL * max(np.linalg.norm(a), np.linalg.norm(b))


def ray(o, d, t_max=np.inf, time=0.0):
    return np.array([o[0], o[1], o[2], t_max, d[0], d[1], d[2], time], np.float32)


def prim_intersect(ob, sc, prim, r):
    t = C.c_float()
    geom = np.empty(17, np.float32)
    hit = ob.lib().orc_prim_intersect(sc.h, prim, ob.fp(r), C.byref(t), ob.fp(geom))
    hit_p = ob.lib().orc_prim_intersect_p(sc.h, prim, ob.fp(r))
    return bool(hit), t.value, geom, bool(hit_p)


# ---- test_intersection.jl:1-20 ---------------------------------------------------------------------------------------------------
def test_ray_bounds_intersection(ob):
    b = np.array([1, 1, 1, 2, 2, 2], np.float32)
    b_neg = np.array([-2, -2, -2, -1, -1, -1], np.float32)
    r0, r1, ri = ray((0, 0, 0), (1, 0, 0)), ray((0, 0, 0), (1, 1, 1)), ray((1.5, 1.5, 1.5), (1, 1, 0))
    t = np.empty(2, np.float32)
    with np.errstate(all="ignore"):
        assert ob.lib().orc_bounds_intersect(ob.fp(b), ob.fp(r1), ob.fp(t)) == 1 and approx(t[0], 1) and approx(t[1], 2)
        assert ob.lib().orc_bounds_intersect(ob.fp(b), ob.fp(r0), ob.fp(t)) == 0 and t[0] == 0 and t[1] == 0
        assert ob.lib().orc_bounds_intersect(ob.fp(b), ob.fp(ri), ob.fp(t)) == 1 and t[0] == 0 and approx(t[1], 0.5)
    assert ob.lib().orc_bounds_intersect_p(ob.fp(b), ob.fp(r1)) == 1
    assert ob.lib().orc_bounds_intersect_p(ob.fp(b_neg), ob.fp(r1)) == 0


def test_bounds_intersect_p_keeps_the_larger_xy_exit(ob):
    """bounds.jl:190 reads `ty_max > tx_max && (tx_max = ty_max)`: the LARGER of the x and y exits survives, so the earlier one never
    bounds the z entry and a box wholly behind the ray in x (or y) still passes.  Derived by hand from the lines 180-200; the
    restatement must keep it (it decides which boxes a walk visits, and the GPU kernels' added clauses are defined against it)."""
    r = ray((0, 0, 0), (1, 1e-3, 1))                       # nearly in the x-z plane, heading +x +z
    behind_x = np.array([-3, -1, 1, -2, 1, 2], np.float32)  # x in [-3, -2]: entirely behind the origin; y range holds the ray, z ahead
    # tx in [-3, -2], ty in [-1000, 1000], tz in [1, 2]: x-y overlap passes, tx_max becomes 1000, z overlaps [max(-2,-1000), 1000] -> "hit"
    assert ob.lib().orc_bounds_intersect_p(ob.fp(behind_x), ob.fp(r)) == 1
    # the same box seen by a ray whose y interval ends early is rejected: there ty_max < tx_max keeps tx_max
    r2 = ray((0, 5, 0), (1, 1, 1))                          # ty in [-6, -4]
    assert ob.lib().orc_bounds_intersect_p(ob.fp(behind_x), ob.fp(r2)) == 0
    # a box behind the ray in z is rejected as it should be (the z exit is taken with `<`)
    behind_z = np.array([1, -1, -3, 2, 1, -2], np.float32)
    assert ob.lib().orc_bounds_intersect_p(ob.fp(behind_z), ob.fp(r)) == 0


# ---- test_intersection.jl:22-87 --------------------------------------------------------------------------------------------------
def test_ray_sphere_intersection(T, ob):
    sc = ob.OracleScene()
    sc.add_sphere(T.Transformation(), False, 1.0, -1.0, 1.0, 360.0)
    sc.add_sphere(T.translate([0, 2, 0]), False, 1.0, -1.0, 1.0, 360.0)

    hit, t, g, hp = prim_intersect(ob, sc, 0, ray((0, -2, 0), (0, 1, 0)))
    assert hit and hit == hp and approx(t, 1)
    assert approx(g[0:3], (0, -1, 0)) and approx(g[3:6], (0, -1, 0))
    assert approx(np.linalg.norm(g[3:6]), 1) and approx(np.linalg.norm(g[6:9]), 1)
    # spawn a new ray from the intersection: it must miss
    d = np.array([0, -1, 0], np.float32)
    o = g[0:3] + np.float32(1e-6) * d
    hit2, *_ = prim_intersect(ob, sc, 0, ray(o, d))
    assert not hit2

    hit, t, g, hp = prim_intersect(ob, sc, 0, ray((0, 0, -2), (0, 0, 1)))
    assert hit and hit == hp and approx(t, 1) and approx(g[0:3], (0, 0, -1)) and approx(g[3:6], (0, 0, -1))
    assert approx(np.linalg.norm(g[3:6]), 1) and approx(np.linalg.norm(g[6:9]), 1)

    # ray inside the sphere
    hit, t, g, _ = prim_intersect(ob, sc, 0, ray((0, 0, 0), (0, 1, 0)))
    assert hit and approx(t, 1) and approx(g[0:3], (0, 1, 0)) and approx(g[3:6], (0, 1, 0))
    # ray at the edge of the sphere
    hit, t, g, _ = prim_intersect(ob, sc, 0, ray((0, -1, 0), (0, -1, 0)))
    assert hit and abs(t) < 1e-6 and approx(g[0:3], (0, -1, 0)) and approx(g[3:6], (0, -1, 0))
    # translated sphere
    hit, t, g, hp = prim_intersect(ob, sc, 1, ray((0, 0, 0), (0, 1, 0)))
    assert hit and hit == hp and approx(t, 1) and approx(g[0:3], (0, 1, 0)) and approx(g[3:6], (0, -1, 0))


def test_sphere_hit_from_inside_ignores_t_max(T, ob):
    """sphere.jl:137-147: `t0 > t_max || t1 < 0` rejects, then `t0 < 0 && (t0 = t1)` — the exit point is returned without a second look at
    t_max.  A ray that starts inside the sphere with t_max = 0.5 is told it hit at t = 1: the caller then RAISES its t_max
    (primitive.jl:12-20).  Derived from the lines; the traversal kernels' stack handling depends on it."""
    sc = ob.OracleScene()
    sc.add_sphere(T.Transformation(), False, 1.0, -1.0, 1.0, 360.0)
    r = ray((0, 0, 0), (0, 0, 1))
    r[3] = 0.5
    hit, t, g, hp = prim_intersect(ob, sc, 0, r)
    assert hit and hp and approx(t, 1) and approx(g[0:3], (0, 0, 1))
    r_out = ray((0, 0, -3), (0, 0, 1))   # from outside the test is the usual one: nearer root beyond t_max -> no hit
    r_out[3] = 1.5
    hit, *_ = prim_intersect(ob, sc, 0, r_out)
    assert not hit


# ---- runtests.jl:34-41 ----------------------------------------------------------------------------------------------------------------
def test_sphere_bound(T, ob):
    sc = ob.OracleScene()
    sc.add_sphere(T.translate([0, 0, 0]), False, 1.0, -1.0, 1.0, 360.0)
    w, o = np.empty(6, np.float32), np.empty(6, np.float32)
    ob.lib().orc_prim_bounds(sc.h, 0, ob.fp(w), ob.fp(o))
    assert np.array_equal(o, [-1, -1, -1, 1, 1, 1])


# ---- test_intersection.jl:89-127 -----------------------------------------------------------------------------------------------------
def test_triangle(T, ob):
    sc = ob.OracleScene()
    sc.add_triangle_mesh(T.translate([0, 0, 2]), False, [[0, 0, 0], [1, 0, 0], [1, 1, 0]], [1, 2, 3], normals=[[0, 0, -1]] * 3)
    assert approx(ob.lib().orc_triangle_area(sc.h, 0), 0.5)
    w, o = np.empty(6, np.float32), np.empty(6, np.float32)
    ob.lib().orc_prim_bounds(sc.h, 0, ob.fp(w), ob.fp(o))
    assert approx(w, (0, 0, 2, 1, 1, 2)) and approx(o, (0, 0, 0, 1, 1, 0))
    r = ray((0, 0, -2), (0, 0, 1))
    hit, t, g, hp = prim_intersect(ob, sc, 0, r)
    assert hit and hp and approx(t, 4) and approx(g[0:3], (0, 0, 2))
    assert np.allclose(g[15:17], 0) and approx(g[3:6], (0, 0, -1)) and approx(g[9:12], (0, 0, -1))  # uv, n, wo = -d
    r = ray((1, 0.5, 0), (0, 0, 1))
    hit, t, g, hp = prim_intersect(ob, sc, 0, r)
    assert hit and hp and approx(t, 2) and approx(g[0:3], (1, 0.5, 2)) and approx(g[15:17], (1, 0.5)) and approx(g[3:6], (0, 0, -1))


# ---- test_intersection.jl:129-195 ------------------------------------------------------------------------------------------------------
def test_bvh_with_nested_bvh(T, ob):
    inner, outer = ob.OracleScene(), ob.OracleScene()
    for i in range(0, 12, 3):
        inner.add_sphere(T.translate([i, i, 0]), False, 1.0, -1.0, 1.0, 360.0)
    inner.commit_reference(1)
    for i in range(12, 24, 3):
        outer.add_sphere(T.translate([i, i, 0]), False, 1.0, -1.0, 1.0, 360.0)
    ob.lib().orc_scene_add_nested_bvh(outer.h, inner.h)
    outer.commit_reference(1)
    assert approx(inner.world_bound(), (-1, -1, -1, 10, 10, 1))
    assert approx(outer.world_bound(), (-1, -1, -1, 22, 22, 1))
    rays = np.stack([ray((-2, 0, 0), (1, 0, 0)), ray((0, 18, 0), (1, 0, 0))])
    t, prim, geom, _ = outer.trace_closest(rays, want_geom=True)
    assert prim[0] >= 0 and approx(t[0], 1) and approx(geom[0, 0:3], (-1, 0, 0))
    assert prim[1] >= 0 and approx(t[1], 17) and approx(geom[1, 0:3], (17, 18, 0))


def test_bvh_spheres_in_a_row(T, ob):
    sc = ob.OracleScene()
    sc.add_sphere(T.Transformation(), False, 1.0, -1.0, 1.0, 360.0)
    sc.add_sphere(T.translate([0, 0, 4]), False, 2.0, -2.0, 2.0, 360.0)
    sc.add_sphere(T.translate([0, 0, 11]), False, 4.0, -4.0, 4.0, 360.0)
    sc.commit_reference(1)
    assert approx(sc.world_bound(), (-4, -4, -1, 4, 4, 15))
    rays = np.stack([ray((0, 0, -2), (0, 0, 1)), ray((1.5, 0, -2), (0, 0, 1)), ray((3, 0, -2), (0, 0, 1))])
    t, prim, geom, _ = sc.trace_closest(rays, want_geom=True)
    assert (prim >= 0).all()
    assert approx(t[0], 1) and 2 < t[1] < 6 and 7 < t[2] < 15
    for k in range(3):
        assert approx(rays[k, 0:3] + rays[k, 4:7] * t[k], geom[k, 0:3])


# ---- test_materials.jl -------------------------------------------------------------------------------------------------------------
def test_fresnel_dielectric(ob):
    assert ob.lib().orc_fresnel_dielectric(1.0, 1.0, 1.0) == 0
    assert ob.lib().orc_fresnel_dielectric(0.5, 1.0, 1.0) == 0


def test_fresnel_conductor(ob):
    s = np.ones(3, np.float32)
    out = np.empty(3, np.float32)
    ob.lib().orc_fresnel_conductor(0.0, ob.fp(s), ob.fp(s), ob.fp(s), ob.fp(out))
    assert np.array_equal(out, s)
    for c in (float(np.cos(np.float32(np.pi) / np.float32(4))), 1.0):
        ob.lib().orc_fresnel_conductor(c, ob.fp(s), ob.fp(s), ob.fp(s), ob.fp(out))
        assert (out > 0).all()


REFL, TRAN, DIFF, GLOSSY, SPEC = 1, 2, 4, 8, 16


def bxdf_params(r=(1, 1, 1), t=(1, 1, 1), sigma=0, ax=1, ay=1, eta_a=1, eta_b=1, fresnel=0, fi=1, ft=1):
    return np.array([*r, *t, sigma, ax, ay, eta_a, eta_b, fresnel, fi, ft], np.float32)


def matches(bxdf_type, flags):  # Base.:&(b::BxDF, type)  reflection/bxdf.jl:9-11
    return (bxdf_type & flags) == bxdf_type


def test_bxdf_flags(ob):
    p = bxdf_params()
    assert matches(ob.lib().orc_bxdf_type(3, ob.fp(p)), SPEC | REFL)
    assert matches(ob.lib().orc_bxdf_type(4, ob.fp(p)), SPEC | TRAN)
    assert matches(ob.lib().orc_bxdf_type(5, ob.fp(p)), SPEC | REFL | TRAN)
    assert matches(ob.lib().orc_bxdf_type(6, ob.fp(p)), REFL | GLOSSY)
    assert matches(ob.lib().orc_bxdf_type(7, ob.fp(p)), TRAN | GLOSSY)
    assert not matches(ob.lib().orc_bxdf_type(4, ob.fp(p)), SPEC | REFL)


def sample_f(ob, kind, p, wo, u):
    wo, u = np.array(wo, np.float32), np.array(u, np.float32)
    out = np.empty(8, np.float32)
    ob.lib().orc_bxdf_sample_f(kind, ob.fp(p), ob.fp(wo), ob.fp(u), ob.fp(out))
    return out[0:3], out[3], out[4:7], int(out[7])


def test_fresnel_specular_sample_f(ob):
    wi, pdf, f, typ = sample_f(ob, 5, bxdf_params(), (0, 0, 1), (0, 0))
    assert approx(wi, (0, 0, -1)) and approx(pdf, 1) and typ == (SPEC | TRAN)


def test_microfacet_reflection_sample_f(ob):
    wi, pdf, f, typ = sample_f(ob, 6, bxdf_params(ax=1, ay=1), (0, 0, 1), (0, 0))
    assert approx(wi, (0, 0, 1))


def test_microfacet_transmission_sample_f(ob):
    wi, pdf, f, typ = sample_f(ob, 7, bxdf_params(ax=1, ay=1, eta_a=1, eta_b=2), (0, 0, 1), (0, 0))
    assert approx(wi, (0, 0, -1))


# ---- runtests.jl:43-58 ------------------------------------------------------------------------------------------------------------------
def test_lanczos_filter(ob):
    f = ob.lib().orc_filter_eval
    assert approx(f(4, 4, 3, 0, 0), 1)
    assert f(4, 4, 3, 4, 4) < 1e-6
    assert f(4, 4, 3, 5, 5) == 0


def film_1080p(T):
    flt = T.LanczosSincFilter([4.0, 4.0], 3.0)
    film = T.Film([1920.0, 1080.0], T.Bounds2([0.0, 0.0], [1.0, 1.0]), flt, 35.0, 1.0, "")
    cam = T.PerspectiveCamera(T.translate([0, 0, 0]), T.Bounds2([0, 0], [10, 10]), 0.0, 1.0, 0.0, 700.0, 45.0, film)
    return film, cam


def test_film(T, ob):
    film, cam = film_1080p(T)
    sn = ob.make_sensor(cam, screen_window=(0, 0, 10, 10), fov=45.0)
    i6, crop, table, r2c = ob.sensor_derived(cam, sn)
    assert (i6[1], i6[0]) == (1080, 1920)              # size(film.pixels)
    assert list(i6[2:]) == [-3, -3, 1924, 1084]        # get_sample_bounds(film)
    # the Python mirror derives the same film geometry, filter table and raster_to_camera, bit for bit
    assert film.size == (1080, 1920)
    sb = film.get_sample_bounds()
    assert [int(x) for x in (*sb.p_min, *sb.p_max)] == [-3, -3, 1924, 1084]
    assert np.array_equal(table.view(np.uint32), film.filter_table.view(np.uint32))
    assert np.array_equal(r2c.view(np.uint32), cam.raster_to_camera.m.view(np.uint32))


# ---- runtests.jl:60-133 ----------------------------------------------------------------------------------------------------------------
class Tile:
    def __init__(self, ob, sensor, bounds):
        self.ob = ob
        b = np.array(bounds, np.float32)
        self.bounds = np.empty(4, np.float32)
        size = np.empty(2, np.int32)
        self.h = ob.lib().orc_filmtile_new(C.byref(sensor), ob.fp(b), ob.fp(self.bounds), size.ctypes.data_as(C.POINTER(C.c_int32)))
        self.size = tuple(int(x) for x in size)

    def add(self, x, y, rgb=(1, 1, 1)):
        c = np.array(rgb, np.float32)
        self.ob.lib().orc_filmtile_add_sample(self.h, x, y, self.ob.fp(c), 1.0)

    def weights(self):
        out = np.empty((*self.size, 4), np.float32)
        self.ob.lib().orc_filmtile_read(self.h, self.ob.fp(out))
        return out[..., 3]

    def merged(self, film_size):
        out = np.empty((*film_size, 4), np.float32)
        self.ob.lib().orc_filmtile_merge(self.h, self.ob.fp(out))
        return out[..., 3]


def test_film_tile(T, ob):
    film, cam = film_1080p(T)
    sn = ob.make_sensor(cam, screen_window=(0, 0, 10, 10), fov=45.0)
    tile = Tile(ob, sn, (1, 1, 10, 10))
    assert tile.size == (14, 14) and list(tile.bounds) == [1, 1, 14, 14]
    w = tile.weights()
    assert all(w[i, i] == 0 for i in range(5))
    tile.add(1.0, 1.0)
    w = tile.weights()
    for i, j in zip(range(0, 4), range(1, 5)):  # 1-based (i, j) in zip(1:4, 2:5)
        assert w[i, i] > 0 and w[j, j] > 0 and w[i, i] > w[j, j]
    fw = tile.merged((1080, 1920))
    for i, j in zip(range(0, 4), range(1, 5)):
        assert fw[i, i] > 0 and fw[j, j] > 0 and fw[i, i] > fw[j, j]

    tile = Tile(ob, sn, (10, 10, 60, 60))
    assert tile.size == (59, 59) and list(tile.bounds) == [6, 6, 64, 64]
    tile.add(20.0, 20.0)
    w = tile.weights()
    J = lambda k: k - 1  # noqa: E731  (Julia 1-based -> 0-based)
    for i, j in zip(range(11, 15), range(18, 14, -1)):
        assert approx(w[J(i), J(i)], w[J(j), J(j)])
    for i, j in zip(range(11, 14), range(12, 15)):
        assert 0 < w[J(i), J(i)] < w[J(j), J(j)]
    for i, j in zip(range(16, 19), range(17, 20)):
        assert w[J(i), J(i)] > w[J(j), J(j)] > 0
    fw = tile.merged((1080, 1920))
    for i, j in zip(range(16, 20), range(23, 19, -1)):
        assert approx(fw[J(i), J(i)], fw[J(j), J(j)])
    for i, j in zip(range(16, 19), range(17, 20)):
        assert 0 < fw[J(i), J(i)] < fw[J(j), J(j)]
    for i, j in zip(range(20, 24), range(21, 25)):
        assert fw[J(i), J(i)] > fw[J(j), J(j)] > 0


# ---- runtests.jl:135-170 -------------------------------------------------------------------------------------------------------------
def test_perspective_camera(T, ob):
    film, cam = film_1080p(T)
    sn = ob.make_sensor(cam, screen_window=(0, 0, 10, 10), fov=45.0)
    samples = np.array([[1, 1, 1, 1, 0], [1920, 1080, 1920, 1080, 0], [2, 1, 1, 1, 0], [1, 2, 1, 1, 0]], np.float32)
    r = ob.generate_rays(cam, samples, sn)
    r1, r2, rx, ry = r
    assert np.array_equal(r1[0:3], (0, 0, 0)) and np.array_equal(r2[0:3], (0, 0, 0))
    assert r1[7] == r2[7] == cam.shutter_open
    assert r1[4] < r2[4] and r1[5] < r2[5]
    assert np.argmax(np.abs(r1[4:7])) == np.argmax(np.abs(r2[4:7])) == 2
    # differentials = rays through the pixels shifted by one in x / y (camera.jl:48-65)
    assert rx[4] > r1[4] and approx(rx[5], r1[5]) and approx(ry[4], r1[4]) and ry[5] > r1[5]


# ---- runtests.jl:11-32: Bounds2 iteration is x-fastest — the order the render driver and the film gather rely on ------------------
def test_sample_order_is_x_fastest(T, ob):
    scene = T.scenes.shadows_scene()
    cam = T.scenes.shadows_camera(8)
    osc = ob.OracleScene.from_scene(scene)
    _, L, st = osc.render(cam, "whitted", 1, 2, seed=1, want_samples=True)
    assert L.shape == (1, 10, 10, 3) and st.camera_samples == 100


# ---- golden image (docs/src/assets/shadows-sppm-1024x1024_mio.png): geometric pin of the camera + scene transcription ---------------
def test_golden_png_geometry(T, ob):
    """with the reference's load-bearing matrix bugs reproduced, the directly visible matte spheres of the
    shadows scene land where they are in the reference's own 1024² render (silhouette edges, +-2 px).  The fixture is
    measured from the PNG by tests/golden/make_golden_geometry.py; edges polluted by reflections/caustics are not used."""
    import json
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from make_golden_geometry import extents, masks
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "shadows_golden_geometry.json")))
    scene = T.scenes.shadows_scene()
    cam = T.scenes.shadows_camera(1024)
    osc = ob.OracleScene.from_scene(scene)
    xyzw, _, _ = osc.render(cam, "whitted", 1, 3, seed=1, threads=ob.lib().orc_num_threads())
    rgb = np.empty((1024, 1024, 3), np.float32)
    ob.lib().orc_film_to_rgb(ob.fp(xyzw), 1024, 1024, 1.0, ob.fp(rgb))
    mine = {k: extents(m) for k, m in masks(rgb[::-1]).items()}  # save() flips rows (film.jl:221)
    for obj, key in (("blue_sphere", "x_min"), ("blue_sphere", "y_min"), ("red_sphere", "x_min")):
        assert abs(mine[obj][key] - gold[obj][key]) <= 2, f"{obj}.{key}: {mine[obj][key]} vs golden {gold[obj][key]}"
