"""BASELINE.json's configurations at their LITERAL sizes (run with -m gpu).

* C1 (configs[0], the reference's own CPU-runnable case): docs/src/shadows.md — 4 spheres, 4 triangles, one PointLight — at 256 x 256, 8 spp, max depth 5, through the
  PathIntegrator and the WhittedIntegrator, bit for bit against the oracle: once in the default configuration (the canonical tree is the reference's own construction,
  accel/bvh.jl:55-206; the rays walk the accelerator under the certificate of csrc/th_trace3c.h), once on the library's SAH tree alone (bvh_builder 0; the oracle
  walks that same tree, handed over through trhip_scene_get_bvh).
* C4 (configs[3]): docs/code/caustic_glass.jl at 1024 x 1024, 100 SPPM iterations, max depth 8.  The oracle needs minutes per iteration at this size, so what is
  checked are the properties that do not depend on it: the integer / order-independent state (M, N, radius, Ld) is reproducible from run to run, identical with a
  1-rank communicator and without one, and identical under traversal 1 (the literal accel/bvh.jl loop) and 3; ϕ / τ / the image within the reordering tolerance of
  tests/test_gpu_sppm.py; the last iteration's photon-hit count and grid resolution are the same in every run.
"""
import os

import numpy as np
import pytest

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


@pytest.mark.parametrize("tree", ["default (reference tree + accelerator)", "library SAH tree alone"])
def test_c1_shadows_256_8spp_depth5_path_and_whitted(T, ob, ctx, tree):
    scene, cam = T.scenes.shadows_scene(), T.scenes.shadows_camera(256)
    ctx.set_option("traversal", 3)
    ctx.set_option("hybrid", 1)
    ctx.set_option("bvh_builder", 0 if tree.startswith("library") else -1)
    try:
        flat = scene.flatten(ctx)
        mode = flat.bvh_mode()[0]
        if tree.startswith("library"):
            assert mode == 0
            osc = ob.OracleScene.from_scene(scene, bvh=flat.bvh())  # the oracle walks the tree the library built
        else:
            assert mode == 2
            osc = ob.OracleScene.from_scene(scene)  # the oracle builds the reference's tree itself (oracle/orc_build.h)
            assert np.array_equal(flat.bvh()[1], osc.get_bvh()[1]) and np.array_equal(bits(flat.bvh()[0]), bits(osc.get_bvh()[0]))
        threads = ob.lib().orc_num_threads()
        for kind, Integ in (("path", T.PathIntegrator), ("whitted", T.WhittedIntegrator)):
            ref_film, ref_L, _ = osc.render(cam, kind, 8, 5, seed=0x5EED0001, threads=threads, want_samples=True)
            integ = Integ(cam, T.SeededSampler(8, seed=0x5EED0001), 5)
            film = integ.render(scene, ctx)
            assert film.shape == (256, 256, 4)
            assert_bits_equal(integ.sample_radiance(scene), ref_L, f"C1 {kind}, {tree}: per-sample radiance")
            assert_bits_equal(film, ref_film, f"C1 {kind}, {tree}: film")
            assert ref_film[..., :3].max() > 0 and int(integ.stats.camera_samples) == 258 * 258 * 8
    finally:
        ctx.set_option("bvh_builder", -1)
        if scene._flat is not None:
            scene._flat.free()
            scene._flat = None


def test_c4_caustic_glass_1024_100_iterations_depth8_properties(T, ctx):
    ply = os.path.join(GOLDEN, "caustic-glass.ply")
    scene = T.scenes.caustic_scene(ply if os.path.exists(ply) else "")
    cam = T.scenes.caustic_camera(1024)
    ctx.set_option("bvh_builder", -1)
    ctx.set_option("hybrid", 1)

    def run(context, traversal):
        context.set_option("traversal", traversal)
        integ = T.SPPMIntegrator(cam, 0.075, 8, 100, -1, seed=0x5EED0004)
        img = integ.render(scene, context).copy()
        st = integ.state()
        scene._flat.free()
        scene._flat = None
        return img, {k: np.array(v, copy=True) for k, v in st.items() if isinstance(v, np.ndarray)}, st["info"]

    img_a, a, info_a = run(ctx, 3)
    assert a["M"].shape == (1024, 1024) and a["M"].sum() > 0 and (a["Ld"] > 0).any() and np.isfinite(img_a).all()
    assert info_a["photons_per_iteration"] == 1023 * 1023 and info_a["photon_hits"] > 0  # sppm.jl:24-27: the default photon count is the film's pixel count as the script computes it
    runs = [("again", ctx, 3), ("traversal 1", ctx, 1)]
    comm = T.Context(0)
    try:
        comm.comm_init(T._ffi.comm_unique_id(), 0, 1)
        runs.append(("1-rank communicator", comm, 3))
        for what, context, trav in runs:
            img_b, b, info_b = run(context, trav)
            assert info_b["photon_hits"] == info_a["photon_hits"] and np.array_equal(info_b["grid_res"], info_a["grid_res"]), f"C4 {what}: photon hits / grid"
            for k in ("M", "N"):
                assert np.array_equal(a[k], b[k]), f"C4 {what}: {k} differs"
            for k in ("radius", "Ld"):
                assert np.array_equal(bits(a[k]), bits(b[k])), f"C4 {what}: {k} differs"
            # ϕ / τ / image: Float32 sums whose order is not fixed from run to run (photon hits of a grid cell are binned with atomics, like the reference's own
            # Threads.Atomic adds, sppm.jl:398-399): the tolerance of tests/test_gpu_sppm.py
            np.testing.assert_allclose(b["tau"], a["tau"], rtol=5e-5, atol=5e-5 * np.abs(a["tau"]).max(), err_msg=f"C4 {what}: tau")
            np.testing.assert_allclose(img_b, img_a, rtol=1e-4, atol=1e-4 * np.abs(img_a).max(), err_msg=f"C4 {what}: image")
    finally:
        ctx.set_option("traversal", 3)
        comm.comm_destroy()
        comm.close()
