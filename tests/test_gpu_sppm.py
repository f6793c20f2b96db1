"""GPU parity of the SPPM integrator (integrators/sppm.jl) against oracle/orc_sppm.h, through the C ABI.

Integer / index state (M, grid resolution, grid entries, photons that landed in the grid) and everything that does not
depend on the order of the photon atomics (Ld, visible points, radius, N) must match BIT FOR BIT.  ϕ and τ are Float32
sums of identical terms in a different order (the reference itself adds them with unordered atomics, sppm.jl:398-399):
they are compared with rtol 2e-5 relative to the sum of magnitudes, the bound for reordering a few hundred terms.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_same_bits(a, b, what):
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    na, nb = np.isnan(a), np.isnan(b)
    assert np.array_equal(na, nb), f"{what}: NaN pattern differs"
    bad = (bits(a) != bits(b)) & ~na
    assert not bad.any(), f"{what}: {int(bad.sum())} of {a.size} values differ, first at {np.argwhere(bad)[0]}: {a[bad][0]!r} vs {b[bad][0]!r}"


def spot_light(T):
    """A SpotLight under the Cornell ceiling, pointing down and slightly to the back (+z of the light frame is its axis)."""
    d = np.float32([0.1, -0.93, -0.36])  # built like docs/code/caustic_glass.jl:55-68: rows du, dv, dir map dir to +z
    d = d / np.float32(np.sqrt(np.float32(d @ d)))
    d, du, dv = T.coordinate_system(d)
    m = np.eye(4, dtype=np.float32)
    m[0, :3], m[1, :3], m[2, :3] = du, dv, d
    l2w = T.translate([0.45, 0.97, -2.45]) * T.inv(T.Transformation(m))
    return T.SpotLight(l2w, T.RGBSpectrum(3.0), 55.0, 35.0)


def run_pair(T, ob, ctx, scene, cam, radius, depth, iters, photons, seed):
    integ = T.SPPMIntegrator(cam, radius, depth, iters, photons, seed=seed)
    xyzw = integ.render(scene, ctx)
    got = integ.state()
    flat = scene.flatten(ctx)
    osc = ob.OracleScene.from_scene(scene, bvh=flat.bvh())
    ref = osc.sppm(cam, radius, depth, iters, photons, seed=seed)
    return integ, xyzw, got, ref


def check_pair(T, xyzw, got, ref, iters):
    gi, ri = got["info"], ref["info"]
    assert np.array_equal(gi["grid_res"], ri["grid_res"]), (gi, ri)
    assert gi["grid_entries"] == ri["grid_entries"]
    assert gi["photon_hits"] == ri["photon_hits"]
    assert gi["photons_per_iteration"] == ri["photons_per_iteration"]
    assert_same_bits(got["vp_p"][got["vp_beta"].any(-1)], ref["vp_p"][ref["vp_beta"].any(-1)], "visible point positions")
    assert_same_bits(got["vp_beta"], ref["vp_beta"], "visible point β")
    assert_same_bits(got["Ld"], ref["Ld"], "Ld")
    assert np.array_equal(got["M"], ref["M"]), f"M differs at {int((got['M'] != ref['M']).sum())} pixels"
    assert ref["M"].sum() > 0
    assert_same_bits(got["radius"], ref["radius"], "radius")
    assert np.array_equal(got["N"], ref["N"]), "N"
    scale = np.maximum(np.abs(ref["phi"]).max(), 1e-30)
    np.testing.assert_allclose(got["phi"], ref["phi"], rtol=2e-5, atol=2e-5 * scale)
    tscale = np.maximum(np.abs(ref["tau"]).max(), 1e-30)
    np.testing.assert_allclose(got["tau"], ref["tau"], rtol=5e-5, atol=5e-5 * tscale)
    # image -> film: xyz = to_XYZ(image), weight 1 (film.jl:195-202)
    img = ref["image"].astype(np.float32)
    m = np.array([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169], [0.019334, 0.119193, 0.950227]], np.float32)
    xyz = img @ m.T
    np.testing.assert_allclose(xyzw[..., :3], xyz, rtol=1e-4, atol=1e-4 * np.abs(xyz).max())
    assert np.all(xyzw[..., 3] == 1.0)


@pytest.mark.parametrize("iters,batch", [(1, 0), (3, 0), (3, 1), (3, 2), (5, 2)])
def test_sppm_cornell_point_light(T, ob, ctx, iters, batch):
    """Matte walls, mirror and glass spheres: specular chains before the visible point, photons through glass.
    ``sppm_batch``: iterations sharing the traversal launches (0 = all that fit); the result may not depend on it."""
    scene = T.scenes.cornell_scene()
    cam = T.scenes.cornell_camera(48)
    ctx.set_option("sppm_batch", batch)
    try:
        _, xyzw, got, ref = run_pair(T, ob, ctx, scene, cam, 0.08, 5, iters, 20000, seed=11)
    finally:
        ctx.set_option("sppm_batch", 0)
    check_pair(T, xyzw, got, ref, iters)
    assert (got["radius"] < np.float32(0.08)).any()


def test_sppm_default_photon_count_and_depth_limit(T, ob, ctx):
    """photons_per_iteration = area(crop_bounds) (sppm.jl:121-124); max_depth 2 cuts specular chains short."""
    scene = T.scenes.cornell_scene()
    cam = T.scenes.cornell_camera(40)
    integ, xyzw, got, ref = run_pair(T, ob, ctx, scene, cam, 0.1, 2, 2, -1, seed=5)
    assert integ.photons_per_iteration == 39 * 39 == got["info"]["photons_per_iteration"]
    check_pair(T, xyzw, got, ref, 2)


def test_sppm_spot_light_and_mesh(T, ob, ctx):
    """SpotLight emission (uniform_sample_cone + falloff, spot.jl:46-55) and a BVH with real depth."""
    scene = T.scenes.mesh_scene(24)
    scene = T.Scene([spot_light(T)] + scene.lights, scene.aggregate)  # two lights: sample_discrete over their power
    cam = T.scenes.cornell_camera(40)
    _, xyzw, got, ref = run_pair(T, ob, ctx, scene, cam, 0.07, 4, 2, 30000, seed=3)
    check_pair(T, xyzw, got, ref, 2)


def test_sppm_rejects_offset_crop(T, ctx):
    scene = T.scenes.cornell_scene()
    cam = T.scenes.cornell_camera(32)
    cam.film.crop_bounds = T.Bounds2(np.float32([2, 1]), cam.film.crop_bounds.p_max)
    with pytest.raises(T.TraceHipError):
        T.SPPMIntegrator(cam, 0.05, 3, 1).render(scene, ctx)


def test_sppm_periodic_image_is_the_image_of_the_first_k_iterations(T, ctx):
    """integrators/sppm.jl:166-171 through trhip_render_sppm_ex: the callback fires after every iteration below the last that write_frequency divides, with
    _sppm_to_image(i, pixels, k) — which is what a k-iteration call returns (M, radius, N, Ld bit for bit; τ up to the order of the photon atomics).  The batches of
    iterations end at those iterations, and the call's own result is unchanged by the callback."""
    scene, cam = T.scenes.cornell_scene(), T.scenes.cornell_camera(48)
    seen = {}
    integ = T.SPPMIntegrator(cam, 0.06, 5, 5, 3000, write_frequency=2, seed=11)
    final = integ.render(scene, ctx, on_write=lambda k, img: seen.__setitem__(k, img.copy())).copy()
    assert sorted(seen) == [2, 4], sorted(seen)
    plain = T.SPPMIntegrator(cam, 0.06, 5, 5, 3000, seed=11).render(scene, ctx)
    assert np.allclose(final, plain, rtol=2e-4, atol=1e-7), "the callback changed the call's result"
    for k, img in seen.items():
        want = T.SPPMIntegrator(cam, 0.06, 5, k, 3000, seed=11).render(scene, ctx)
        assert np.allclose(img, want, rtol=2e-4, atol=1e-7), f"image after {k} iterations: max difference {np.abs(img - want).max()}"
        assert not np.allclose(img, final, rtol=1e-3, atol=1e-6), "an intermediate image equal to the final one proves nothing"
    # a callback that fails stops the call with an error, nothing propagates through the C frames
    def boom(k, img):
        raise RuntimeError("disk full")
    with pytest.raises(RuntimeError, match="disk full"):
        T.SPPMIntegrator(cam, 0.06, 5, 5, 3000, write_frequency=2, seed=11).render(scene, ctx, on_write=boom)
