"""Hand-derived known answers for the shading rows the reference has no test for: light sampling, the
matte and mirror BSDFs through `compute_scattering!`, and the first vertex of the path integrator (direct light on a matte
floor).  Every expectation is computed here from the reference's formulas (cited), not from the oracle; they pin the
restatement to the text where the reference itself offers no vector.
"""
import numpy as np

f32 = np.float32


def one_triangle_scene(T, lights, material=None):
    mat = material or T.MatteMaterial(T.ConstantTexture(T.RGBSpectrum(0.5)), T.ConstantTexture(0.0))
    core = T.ShapeCore(T.translate([0, 0, 0]), False)
    tri = T.create_triangle_mesh(core, 1, np.array([1, 2, 3], np.uint32), 3, [[100, 100, 100], [101, 100, 100], [100, 101, 100]])
    return T.Scene(lights, T.BVHAccel([T.GeometricPrimitive(tri[0], mat)], 1))


def light_query(ob, osc, light, points):
    pts = np.ascontiguousarray(points, np.float32).reshape(-1, 3)
    out = np.empty((pts.shape[0], 8), np.float32)
    assert ob.lib().orc_light_query(osc.h, light, ob.fp(pts), pts.shape[0], ob.fp(out)) == 0
    return out


def test_point_light_sample_li(T, ob):
    """lights/point.jl:50-58: wi = normalize(position - p), radiance = I / distance², pdf = 1."""
    scene = one_triangle_scene(T, [T.PointLight(T.translate([0.5, 0.9, -2.5]), T.RGBSpectrum(2.5))])
    osc = ob.OracleScene.from_scene(scene)
    out = light_query(ob, osc, 0, [[0.5, 0.4, -2.5], [1.5, 0.9, -2.5], [0.5, 0.9, -4.5]])
    dy = f32(0.9) - f32(0.4)                                                  # Float32 like the reference: 0.49999997, not 0.5
    assert np.array_equal(out[0], f32([f32(2.5) / (dy * dy)] * 3 + [0, 1, 0, 1, 1])) and abs(out[0, 0] - 10) < 2e-6
    assert np.array_equal(out[1], f32([2.5, 2.5, 2.5, -1, 0, 0, 1, 1]))       # distance 1
    assert np.array_equal(out[2], f32([0.625, 0.625, 0.625, 0, 0, 1, 1, 1]))  # distance 2


def test_spot_light_falloff(T, ob):
    """lights/spot.jl:22-40: radiance = I · falloff(-wi) / distance²; falloff = 0 outside cos(total), 1 inside cos(falloff_start),
    δ⁴ in between with δ = (cosθ - cos_total) / (cos_start - cos_total).  The light looks down +z of its own frame."""
    scene = one_triangle_scene(T, [T.SpotLight(T.translate([0, 1, 0]), T.RGBSpectrum(8.0), 60.0, 30.0)])
    osc = ob.OracleScene.from_scene(scene)
    out = light_query(ob, osc, 0, [[0, 1, 2], [2, 1, 2], [5, 1, 1], [0, 1, -2]])
    assert np.array_equal(out[0, :3], f32([2, 2, 2])) and np.array_equal(out[0, 3:6], f32([0, 0, -1]))  # on the axis: 8 / 4
    cos_total, cos_start = np.cos(np.float64(f32(60.0) * (f32(np.pi) / f32(180.0)))), np.cos(np.float64(f32(30.0) * (f32(np.pi) / f32(180.0))))
    delta = (np.float64(f32(1.0) / np.sqrt(f32(2.0))) - cos_total) / (cos_start - cos_total)
    want = 8.0 * delta ** 4 / 8.0                                                                          # 45° off the axis, distance² 8
    assert np.allclose(out[1, :3], want, rtol=3e-6, atol=0) and abs(want - 0.10243) < 1e-4
    assert np.all(out[2, :3] == 0) and np.all(out[3, :3] == 0)                                             # outside the cone; behind the light
    assert np.all(out[:, 6] == 1)


def frame_z():
    return f32([0, 0, 1, 0, 0, 1, 1, 0, 0])  # ng, ns = +z; ss = +x


def test_matte_is_one_lambertian_lobe(T, ob):
    """materials/material.jl:16-31 with σ = 0 -> LambertianReflection(Kd): f = Kd / π in the same hemisphere, 0 across it (bsdf.jl:79-100
    selects reflection lobes by the geometric normal), pdf = |cosθ_i| / π (bxdf.jl:23-25); sample_f at u = (0.5, 0.5) is the pole of the
    concentric map: wi = n (bxdf.jl:34-42, Trace.jl:48-66)."""
    kd = (0.2, 0.4, 0.6)
    scene = one_triangle_scene(T, [], T.MatteMaterial(T.ConstantTexture(T.RGBSpectrum(*kd)), T.ConstantTexture(0.0)))
    osc = ob.OracleScene.from_scene(scene)
    wi = f32([0.3, 0.4, np.sqrt(0.75)])
    q = osc.bsdf_query(0, True, 0, 31, [frame_z(), frame_z()], [[0, 0, 1, *wi], [0, 0, 1, wi[0], wi[1], -wi[2]]])
    inv_pi = f32(1.0) / f32(np.pi)
    assert np.array_equal(q[0, :3], f32(kd) * inv_pi)
    assert abs(q[0, 3] - wi[2] * inv_pi) <= 1e-7
    assert np.all(q[1, :4] == 0)
    s = osc.bsdf_query(0, True, 1, 31, [frame_z()], [[0, 0, 1, 0.5, 0.5, 0]])[0]
    assert np.allclose(s[:3], [0, 0, 1], atol=1e-7) and np.array_equal(s[3:6], f32(kd) * inv_pi) and abs(s[6] - inv_pi) <= 1e-7
    assert int(s[7]) == (1 | 4)  # BSDF_REFLECTION | BSDF_DIFFUSE (bxdf.jl:1-7)


def test_mirror_is_a_perfect_specular_lobe(T, ob):
    """materials/material.jl:39-46 -> SpecularReflection(Kr, FresnelNoOp): f(wo, wi) = 0 for given directions, but their pdf is NOT 0 —
    specular.jl defines no `compute_pdf`, so the BxDF default applies (bxdf.jl:23-25: |cosθ_i| / π in the same hemisphere) and
    BSDF.compute_pdf (bsdf.jl:177-193) reports it; sample_f mirrors wo about the shading normal, pdf = 1, f = Kr / |cosθ_i| (specular.jl:1-39)."""
    scene = one_triangle_scene(T, [], T.MirrorMaterial(T.ConstantTexture(T.RGBSpectrum(0.9))))
    osc = ob.OracleScene.from_scene(scene)
    c = f32(1.0) / np.sqrt(f32(2.0))
    q = osc.bsdf_query(0, True, 0, 31, [frame_z()], [[c, 0, c, -c, 0, c]])[0]
    assert np.all(q[:3] == 0) and abs(q[3] - c * (f32(1.0) / f32(np.pi))) <= 1e-7
    s = osc.bsdf_query(0, True, 1, 31, [frame_z()], [[c, 0, c, 0.3, 0.7, 0]])[0]
    assert np.allclose(s[:3], [-c, 0, c], atol=1e-7) and s[6] == 1.0
    assert np.allclose(s[3:6], f32(0.9) / c, rtol=2e-7)
    assert int(s[7]) == (1 | 16)  # BSDF_REFLECTION | BSDF_SPECULAR


def test_path_first_vertex_is_kd_over_pi_cos_li(T, ob):
    """The first vertex of the path integrator on a matte floor under a point light, depth 1 (the SPPM camera-pass arithmetic,
    sppm.jl:208-230 with β = 1, estimate_direct :503-554): L = (Kd / π) · |wi · n| · I / d² for every camera sample whose ray hits the
    floor — the light is above an OPEN scene, so the shadow ray (t_max = Inf) meets nothing."""
    kd, inten = f32(0.5), f32(8.0)
    mat = T.MatteMaterial(T.ConstantTexture(T.RGBSpectrum(float(kd))), T.ConstantTexture(0.0))
    core = T.ShapeCore(T.translate([0, 0, 0]), False)
    floor = T.create_triangle_mesh(core, 2, np.array([1, 2, 3, 1, 3, 4], np.uint32), 4, [[-4, 0, -7], [5, 0, -7], [5, 0, 2], [-4, 0, 2]])
    light_p = f32([0.5, 2.0, -2.5])
    scene = T.Scene([T.PointLight(T.translate([float(x) for x in light_p]), T.RGBSpectrum(float(inten)))], T.BVHAccel([T.GeometricPrimitive(t, mat) for t in floor], 1))
    osc = ob.OracleScene.from_scene(scene)
    cam = T.scenes.cornell_camera(12)
    _, L, _ = osc.render(cam, "path", 2, 1, seed=5, want_samples=True)
    rays = ob.generate_rays(cam, T.scenes.camera_sample_grid(cam, 2, 5))
    t, prim, geom, _ = osc.trace_closest(rays, want_geom=True)
    L = L.reshape(-1, 3)
    assert L.shape[0] == rays.shape[0]
    hit = prim >= 0
    assert hit.sum() > 50 and (~hit).sum() > 10
    p = geom[hit, 0:3].astype(np.float64)
    d2 = ((light_p.astype(np.float64) - p) ** 2).sum(axis=1)
    cos = np.abs((light_p.astype(np.float64) - p)[:, 1]) / np.sqrt(d2)           # n = ±y
    want = np.float64(kd) / np.pi * cos * np.float64(inten) / d2
    assert np.allclose(L[hit], want[:, None], rtol=5e-6, atol=0)
    assert np.all(L[~hit] == 0)


def test_bsdf_sample_f_picks_a_lobe_and_divides_the_pdf(T, ob):
    """BSDF.sample_f (bsdf.jl:107-175) on a two-lobe BSDF: GlassMaterial without multiple lobes adds SpecularReflection(Kr,
    FresnelDielectric(1, η)) and SpecularTransmission(Kt, 1, η) (material.jl:75-116).  component = ceil(u₁ · 2): u₁ = 0.25 samples the
    reflection lobe, u₁ = 0.75 the transmission lobe; a specular lobe's pdf (1) is divided by the 2 matching components and f comes
    from the chosen lobe alone.  At normal incidence Fr = ((η - 1) / (η + 1))² = 0.04 for η = 1.5; f_r = Fr · Kr / |cosθ|,
    f_t = (1 - Fr) · Kt / |cosθ| (no η² factor: `T isa Radiance` is always false, A.11)."""
    glass = T.GlassMaterial(T.ConstantTexture(T.RGBSpectrum(0.9)), T.ConstantTexture(T.RGBSpectrum(0.8)), T.ConstantTexture(0.0), T.ConstantTexture(0.0),
                            T.ConstantTexture(1.5), True)
    scene = one_triangle_scene(T, [], glass)
    osc = ob.OracleScene.from_scene(scene)
    r = osc.bsdf_query(0, False, 1, 31, [frame_z(), frame_z()], [[0, 0, 1, 0.25, 0.5, 0], [0, 0, 1, 0.75, 0.5, 0]])
    assert np.allclose(r[0, :3], [0, 0, 1], atol=1e-7) and np.allclose(r[0, 3:6], 0.04 * 0.9, rtol=2e-6) and r[0, 6] == 0.5 and int(r[0, 7]) == (1 | 16)
    assert np.allclose(r[1, :3], [0, 0, -1], atol=1e-7) and np.allclose(r[1, 3:6], 0.96 * 0.8, rtol=2e-6) and r[1, 6] == 0.5 and int(r[1, 7]) == (2 | 16)
    # restricted to reflection there is one matching component: the pdf is not divided, whatever u₁
    q = osc.bsdf_query(0, False, 1, 1 | 16, [frame_z()], [[0, 0, 1, 0.75, 0.5, 0]])[0]
    assert np.allclose(q[:3], [0, 0, 1], atol=1e-7) and q[6] == 1.0 and np.allclose(q[3:6], 0.04 * 0.9, rtol=2e-6)
