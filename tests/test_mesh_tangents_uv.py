"""Rows a8 / a11: the two optional TriangleMesh arrays, tangents and (u, v)s (shapes/triangle_mesh.jl:11-14, 76-83, 125-141, 160-185).
No scene or test of the reference sets them, so the oracle's restatement is pinned here by an independent float64 evaluation of
the same formulas (CPU), and the HIP path against the oracle bit for bit (`-m gpu`).
uv quirk kept: the reference reads `mesh.uv[t.i + j]` — by CORNER position 3k + j, not through the index list (:82)."""
import numpy as np
import pytest


def soup(rng, n):
    c = rng.random((n, 1, 3)) * 2.0 - 1.0
    v = (c + (rng.random((n, 3, 3)) - 0.5) * 0.8).astype(np.float32)
    return v


def make_scene(T, v, normals, tangents, uv, reverse=False, xform=None):
    n = v.shape[0]
    grey = T.MatteMaterial(T.ConstantTexture(T.RGBSpectrum(0.6)), T.ConstantTexture(0.0))
    core = T.ShapeCore(xform if xform is not None else T.translate([0, 0, 0]), reverse)
    mesh = T.create_mesh_primitives(core, np.arange(1, 3 * n + 1, dtype=np.uint32), v.reshape(-1, 3), normals, grey, tangents=tangents, uv=uv)
    light = T.PointLight(T.translate([0.0, 0.0, 3.0]), T.RGBSpectrum(20.0))
    return T.Scene([light], T.BVHAccel([mesh], 1))


def rays_at(v, rng, per_tri=2):
    """rays from outside aimed at random interior points of the triangles"""
    n = v.shape[0]
    b = rng.random((n * per_tri, 3)) + 0.05
    b /= b.sum(axis=1, keepdims=True)
    tri = np.repeat(np.arange(n), per_tri)
    p = (b[:, :, None] * v[tri]).sum(axis=1)
    d = rng.normal(size=p.shape)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o = p - 6.0 * d
    rays = np.zeros((p.shape[0], 8), np.float32)
    rays[:, 0:3] = o
    rays[:, 3] = np.inf
    rays[:, 4:7] = d
    return rays


def f64_frame(vs, bary, nrm, tg, uv, flip):
    """triangle_mesh.jl:125-141, 160-185, 219-240 + surface_interaction.jl:70-88 in float64: (core.n, shading.n, normalize(shading.∂p∂u))"""
    def nz(x):
        return x / np.linalg.norm(x)
    vs = vs.astype(np.float64)
    uv = np.array([[0, 0], [1, 0], [1, 1]], np.float64) if uv is None else uv.astype(np.float64)
    duv13, duv23 = uv[0] - uv[2], uv[1] - uv[2]
    dp13, dp23 = vs[0] - vs[2], vs[1] - vs[2]
    det = duv13[0] * duv23[1] - duv13[1] * duv23[0]
    assert det != 0
    dpdu = (duv23[1] * dp13 - duv13[1] * dp23) / det
    n = nz(np.cross(dp13, dp23))
    sh_n, ss_out = n, nz(dpdu)
    if nrm is not None or tg is not None:
        ns = n if nrm is None else nz((bary[:, None] * nrm.astype(np.float64)).sum(axis=0))
        ss = nz(dpdu) if tg is None else nz((bary[:, None] * tg.astype(np.float64)).sum(axis=0))
        ts = np.cross(ns, ss)
        assert ts @ ts > 1e-12
        ts = nz(ts)
        ss = np.cross(ts, ns)
        sh_n = nz(np.cross(ss, ts))
        if flip:
            sh_n = -sh_n
        if n @ sh_n < 0:
            n = -n
        ss_out = nz(ss)
    if nrm is not None:
        if n @ sh_n < 0:
            n = -n
    elif flip:
        n = -n
        sh_n = n
    return n, sh_n, ss_out


@pytest.mark.parametrize("case", ["uv", "tangents", "tangents+normals", "tangents+uv+normals", "tangents_flip", "uv_flip"])
def test_oracle_tangents_uv_against_float64(case):
    import oracle_bridge as ob
    import __graft_entry__ as g
    T = g.load_package()
    rng = np.random.default_rng(abs(hash(case)) % 1000)
    n = 40
    v = soup(rng, n)
    nrm = tg = uv = None
    if "normals" in case:
        face = np.cross(v[:, 0] - v[:, 2], v[:, 1] - v[:, 2])
        nrm = (np.repeat(face[:, None, :], 3, axis=1) + 0.3 * rng.normal(size=(n, 3, 3))).astype(np.float32)
        nrm /= np.linalg.norm(nrm, axis=2, keepdims=True)
        nrm = nrm.reshape(-1, 3)
    if "tangents" in case:
        tg = rng.normal(size=(3 * n, 3)).astype(np.float32)
    if "uv" in case:
        uv = rng.random((3 * n, 2)).astype(np.float32)
    flip = "flip" in case
    scene = make_scene(T, v, nrm, tg, uv, reverse=flip)
    osc = ob.OracleScene.from_scene(scene)
    rays = rays_at(v, rng)
    t, prim, geom, _ = osc.trace_closest(rays, want_geom=True)
    hit = np.flatnonzero(prim >= 0)
    assert hit.size > 40
    order = osc.get_bvh()[3]  # hits name the ordered slot (BVHAccel.primitives); order[slot] = the caller's triangle
    checked = 0
    for i in hit:
        k = int(order[prim[i]])
        vs = v[k]
        p = geom[i, 0:3].astype(np.float64)
        # barycentrics of the reported hit point
        A = np.stack([vs[0] - vs[2], vs[1] - vs[2]], axis=1).astype(np.float64)
        b01, *_ = np.linalg.lstsq(A, p - vs[2].astype(np.float64), rcond=None)
        bary = np.array([b01[0], b01[1], 1 - b01[0] - b01[1]])
        if bary.min() < 0.02:
            continue
        en, esn, ess = f64_frame(vs, bary, None if nrm is None else nrm[3 * k:3 * k + 3], None if tg is None else tg[3 * k:3 * k + 3], None if uv is None else uv[3 * k:3 * k + 3], flip)
        assert np.allclose(geom[i, 3:6], en, atol=2e-4), (case, i, geom[i, 3:6], en)
        assert np.allclose(geom[i, 6:9], esn, atol=2e-4), (case, i, geom[i, 6:9], esn)
        assert np.allclose(geom[i, 12:15], ess, atol=2e-3), (case, i, geom[i, 12:15], ess)
        checked += 1
    assert checked > 30


def test_oracle_uv_is_read_by_corner_position():
    """Two triangles sharing vertices through the index list: the uvs follow the corner slots 3k + j, not the vertex numbers (triangle_mesh.jl:82)."""
    import oracle_bridge as ob
    import __graft_entry__ as g
    T = g.load_package()
    verts = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], np.float32)
    idx = np.array([1, 2, 3, 1, 3, 4], np.uint32)
    uv = np.array([[0, 0], [1, 0], [1, 1], [0, 0], [1, 1], [1, 0]], np.float32)  # second triangle: its own three corners' (u, v)s, slots 4-6
    grey = T.MatteMaterial(T.ConstantTexture(T.RGBSpectrum(0.6)), T.ConstantTexture(0.0))
    mesh = T.create_mesh_primitives(T.ShapeCore(T.translate([0, 0, 0]), False), idx, verts, None, grey, uv=uv)
    osc = ob.OracleScene.from_scene(T.Scene([], T.BVHAccel([mesh], 1)))
    rays = np.zeros((2, 8), np.float32)
    rays[:, 0:3] = [[0.7, 0.2, 1.0], [0.2, 0.7, 1.0]]
    rays[:, 3] = np.inf
    rays[:, 4:7] = [0, 0, -1]
    _, prim, geom, _ = osc.trace_closest(rays, want_geom=True)
    assert osc.get_bvh()[3][prim].tolist() == [0, 1]
    assert np.allclose(geom[0, 12:15], [1, 0, 0], atol=1e-6)   # default-like uvs: ∂p∂u along +x
    assert np.allclose(geom[1, 12:15], [0, 1, 0], atol=1e-6)   # vertices (0,0,0) (1,1,0) (0,1,0) with (0,0) (1,1) (1,0): ∂p∂u = +y


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["uv", "tangents", "tangents+normals", "tangents+uv+normals", "tangents_flip", "uv_degenerate", "mixed_scene"])
def test_gpu_tangents_uv_equal_the_oracle(T, ob, ctx, case):
    from test_gpu_parity import assert_bits_equal
    rng = np.random.default_rng(11 + len(case))
    n = 300
    v = soup(rng, n)
    nrm = tg = uv = None
    if "normals" in case or case == "mixed_scene":
        nrm = rng.normal(size=(3 * n, 3)).astype(np.float32)
        nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    if "tangents" in case or case == "mixed_scene":
        tg = rng.normal(size=(3 * n, 3)).astype(np.float32)
        tg[::7] = 0.0  # a zero tangent at some vertices: the ts·ts > 0 branch and NaN frames must match too
    if "uv" in case or case == "mixed_scene":
        uv = rng.random((3 * n, 2)).astype(np.float32)
    if case == "uv_degenerate":
        uv[0:30] = 0.25  # det == 0: coordinate_system fallback (:132-136)
    scene = make_scene(T, v, nrm, tg, uv, reverse="flip" in case, xform=T.translate([0.1, -0.2, 0.05]))
    if case == "mixed_scene":  # a second mesh without the arrays and two spheres in the same BVH
        v2 = soup(rng, 100) + np.float32(0.3)
        grey = T.MatteMaterial(T.ConstantTexture(T.RGBSpectrum(0.4)), T.ConstantTexture(0.0))
        scene.aggregate.primitives.append(T.create_mesh_primitives(T.ShapeCore(T.translate([0, 0, 0]), False), np.arange(1, 301, dtype=np.uint32), v2.reshape(-1, 3), None, grey))
        scene.aggregate.primitives += T.scenes.shadows_scene().aggregate.primitives[-2:]
    flat = scene.flatten(ctx)
    osc = ob.OracleScene.from_scene(scene, bvh=flat.bvh())
    wb = osc.world_bound()
    rays = np.concatenate([rays_at(v + np.float32([0.1, -0.2, 0.05]), rng, 3), T.scenes.incoherent_rays(20000, wb[:3] - 0.3, wb[3:] + 0.3, seed=8)])
    geom = flat.hit_geometry(rays)
    _, prim_ref, geom_ref, _ = osc.trace_closest(rays, want_geom=True)
    assert (prim_ref >= 0).sum() > 500
    assert_bits_equal(geom, geom_ref, f"hit geometry ({case})")
    cam = T.scenes.cornell_camera(24)
    ref, ref_L, _ = osc.render(cam, "path", 2, 4, seed=21, want_samples=True)
    integ = T.PathIntegrator(cam, T.SeededSampler(2, seed=21), 4)
    film = integ.render(scene, ctx)
    assert_bits_equal(integ.sample_radiance(scene), ref_L, f"per-sample radiance ({case})")
    assert_bits_equal(film, ref, f"film ({case})")
    flat.free()
    scene._flat = None
