"""GPU parity of the WhittedIntegrator (integrators/sampler.jl:58-199): the level-by-level ray tree +
bottom-up fold must equal the oracle's recursion bit for bit, per sample and in the film."""
import numpy as np
import pytest

from test_gpu_parity import assert_bits_equal, scene_pair

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("res,spp,depth", [(48, 2, 5), (40, 1, 8), (32, 3, 1)])
def test_whitted_shadows_bit_exact(T, ob, ctx, res, spp, depth):
    """Config C1's scene: glass sphere (two-way branching), mirror sphere + mirror triangles (chains), matte (leaves)."""
    scene = T.scenes.shadows_scene()
    flat, osc = scene_pair(T, ob, scene)
    cam = T.scenes.shadows_camera(res)
    integ = T.WhittedIntegrator(cam, T.SeededSampler(spp, seed=0x5EED0001), depth)
    xyzw = integ.render(scene)
    L = integ.sample_radiance(scene)
    ref_xyzw, ref_L, st = osc.render(cam, "whitted", spp, depth, seed=0x5EED0001, want_samples=True)
    assert ref_L.max() > 0
    assert_bits_equal(L, ref_L, "per-sample radiance")
    assert_bits_equal(xyzw, ref_xyzw, "film")
    assert integ.stats.closest_rays == st.closest_rays and integ.stats.shadow_rays == st.shadow_rays


def test_whitted_two_lights_glass_and_plastic(T, ob, ctx):
    """Light-order accumulation (two lights), rough glass / plastic (no specular lobes: no children), nested glass."""
    mat = T.MatteMaterial(T.ConstantTexture(T.RGBSpectrum(0.7, 0.6, 0.5)), T.ConstantTexture(20.0))
    glass = T.GlassMaterial(T.ConstantTexture(T.RGBSpectrum(1.0)), T.ConstantTexture(T.RGBSpectrum(0.9)), T.ConstantTexture(0.0), T.ConstantTexture(0.0), T.ConstantTexture(1.5), True)
    rough = T.GlassMaterial(T.ConstantTexture(T.RGBSpectrum(1.0)), T.ConstantTexture(T.RGBSpectrum(1.0)), T.ConstantTexture(0.2), T.ConstantTexture(0.2), T.ConstantTexture(1.3), True)
    plastic = T.PlasticMaterial(T.ConstantTexture(T.RGBSpectrum(0.5)), T.ConstantTexture(T.RGBSpectrum(0.3)), T.ConstantTexture(0.1), True)
    mirror = T.MirrorMaterial(T.ConstantTexture(T.RGBSpectrum(0.9)))
    prims = [
        T.GeometricPrimitive(T.Sphere(T.ShapeCore(T.translate([0.3, 0.3, -2.5]), False), 0.25, 360.0), glass),
        T.GeometricPrimitive(T.Sphere(T.ShapeCore(T.translate([0.3, 0.3, -2.5]), False), 0.12, 360.0), glass),
        T.GeometricPrimitive(T.Sphere(T.ShapeCore(T.translate([0.75, 0.2, -2.4]), False), 0.18, 360.0), rough),
        T.GeometricPrimitive(T.Sphere(T.ShapeCore(T.translate([0.7, 0.65, -2.7]), False), 0.2, 360.0), mirror),
    ]
    floor = T.create_triangle_mesh(T.ShapeCore(T.translate([0, 0, -3]), False), 4, np.array([1, 2, 3, 1, 3, 4, 1, 4, 5, 5, 4, 6], np.uint32), 6,
                                   [[0, 0, 0], [1, 0, 0], [1, 0, 1], [0, 0, 1], [0, 1, 0], [0, 1, 1]], [[0, 1, 0]] * 4 + [[1, 0, 0]] * 2)
    prims += [T.GeometricPrimitive(t, m) for t, m in zip(floor, (mat, mat, plastic, plastic))]
    scene = T.Scene([T.PointLight(T.translate([0.5, 0.95, -2.1]), T.RGBSpectrum(3.0)), T.SpotLight(T.translate([0.9, 0.9, -2.0]), T.RGBSpectrum(5.0), 60.0, 40.0)], T.BVHAccel(prims, 1))
    flat, osc = scene_pair(T, ob, scene)
    cam = T.scenes.shadows_camera(56)
    integ = T.WhittedIntegrator(cam, T.SeededSampler(2, seed=5), 7)
    xyzw = integ.render(scene)
    ref_xyzw, ref_L, st = osc.render(cam, "whitted", 2, 7, seed=5, want_samples=True)
    assert_bits_equal(integ.sample_radiance(scene), ref_L, "per-sample radiance")
    assert_bits_equal(xyzw, ref_xyzw, "film")
    assert integ.stats.closest_rays == st.closest_rays and integ.stats.shadow_rays == st.shadow_rays


def test_whitted_batches(T, ob, ctx):
    scene = T.scenes.shadows_scene()
    cam = T.scenes.shadows_camera(32)
    a = T.WhittedIntegrator(cam, T.SeededSampler(3, seed=2), 5).render(scene).copy()
    ctx.set_option("batch_paths", 34 * 34)
    try:
        b = T.WhittedIntegrator(cam, T.SeededSampler(3, seed=2), 5).render(scene).copy()
    finally:
        ctx.set_option("batch_paths", 0)
    assert_bits_equal(a, b, "batched Whitted film")
