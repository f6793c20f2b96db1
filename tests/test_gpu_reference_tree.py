"""Row a18 on the GPU (run with -m gpu): the HIP path on the REFERENCE's own BVH topology.

Every other GPU test hands the oracle the library's tree (binned SAH, th_bvh.h), so what they prove is "same tree => same bits".  A Trace.jl user
holds the tree of accel/bvh.jl:87-206; where two primitives are accepted at (nearly) the same t the later tested one wins (bvh.jl:229-237,
triangle_mesh.jl:211-214), so agreement with Trace.jl on those rays needs THAT tree.  Here:

* option "bvh_builder" = 2 makes trhip_scene_commit build it (th_bvh_ref.h): its arrays must equal, bit for bit, the tree the oracle's restatement
  of the reference builder makes from the same scene (oracle/orc_build.h — compared on the CPU in tests/test_reference_bvh.py);
* the oracle's reference tree handed over through trhip_scene_set_bvh (what TraceHIP.jl does with `bvh.nodes`) gives the same device scene;
* closest hits, occlusion and whole frames on that tree equal the oracle walking ITS OWN reference tree, with every traversal kernel — including the
  0-primitive leaves with invalid bounds the construction emits.
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


@pytest.fixture
def ref_ctx(ctx):
    ctx.set_option("bvh_builder", 2)
    yield ctx
    ctx.set_option("bvh_builder", -1)
    ctx.set_option("traversal", 3)


def cases(T):
    yield "shadows", T.scenes.shadows_scene(), T.scenes.shadows_camera(48), ([-1.2, -0.3, -3.2], [1.3, 1.2, 1.0])
    yield "cornell", T.scenes.cornell_scene(), T.scenes.cornell_camera(40), ([0, 0, -3], [1, 1, -2])
    yield "mesh64", T.scenes.mesh_scene(64), T.scenes.cornell_camera(48), ([0, 0, -3], [1, 1, -2])
    ply = os.path.join(GOLDEN, "caustic-glass.ply")
    if os.path.exists(ply):
        yield "caustic-glass.ply", T.scenes.caustic_scene(ply), T.scenes.caustic_camera(32), None


def test_commit_builds_the_reference_tree_and_every_kernel_agrees_with_the_oracle_on_it(T, ob, ref_ctx):
    ctx = ref_ctx
    for name, scene, cam, box in cases(T):
        osc = ob.OracleScene.from_scene(scene)  # bvh=None: the oracle builds the reference's tree itself (orc_build.h)
        rb, ra, rf, ro = osc.get_bvh()
        flat = scene.flatten(ctx)
        b, a, f, o = flat.bvh()
        assert a.size == ra.size and np.array_equal(o, ro) and np.array_equal(a, ra) and np.array_equal(f, rf), f"{name}: topology differs from the oracle's reference tree"
        assert np.array_equal(bits(b), bits(rb)), f"{name}: node bounds differ"
        n_empty = int((((f & 3) == 3) & ((f >> 2) == 0)).sum())
        wb = osc.world_bound()
        lo, hi = (wb[:3], wb[3:]) if box is None else (np.float32(box[0]), np.float32(box[1]))
        rays = np.concatenate([ob.generate_rays(cam, T.scenes.camera_sample_grid(cam, 1, seed=3)), T.scenes.incoherent_rays(30000, lo, hi, seed=17)])
        t_ref, prim_ref, bary_ref, _ = osc.trace_closest(rays)
        occ_ref, _ = osc.trace_any(rays)
        from conftest import supported
        for trav in supported(ctx, "traversal", (3, 7, 2, 6, 1)):
            ctx.set_option("traversal", trav)
            hits = flat.trace_closest(rays)
            assert np.array_equal(hits["prim"], prim_ref), f"{name}, traversal {trav}: primitives differ ({n_empty} empty leaves in the tree)"
            assert np.array_equal(bits(hits["t"]), bits(t_ref)), f"{name}, traversal {trav}: t differs"
            assert np.array_equal(flat.trace_any(rays), occ_ref), f"{name}, traversal {trav}: occlusion differs"
        ctx.set_option("traversal", 3)
        ref_film, ref_L, _ = osc.render(cam, "path", 2, 6, seed=31, want_samples=True)
        integ = T.PathIntegrator(cam, T.SeededSampler(2, seed=31), 6)
        film = integ.render(scene, ctx)
        assert_bits_equal(integ.sample_radiance(scene), ref_L, f"{name}: per-sample radiance on the reference tree")
        assert_bits_equal(film, ref_film, f"{name}: film on the reference tree")
        scene._flat = None
        flat.free()


def test_reference_tree_through_set_bvh_is_the_same_device_scene(T, ob, ctx):
    """The Julia shim's route: the host already holds BVHAccel.nodes / .primitives and hands them over (trhip_scene_set_bvh)."""
    scene, cam = T.scenes.mesh_scene(40), T.scenes.cornell_camera(40)
    osc = ob.OracleScene.from_scene(scene)
    tree = osc.get_bvh()
    ctx.set_option("bvh_builder", 0)
    flat = scene.flatten(ctx)  # the library's own tree first
    assert flat.bvh()[1].size != tree[1].size or not np.array_equal(flat.bvh()[3], tree[3]), "the SAH tree happens to equal the reference's: the test proves nothing"
    ctx.set_option("bvh_builder", 2)  # … replaced by the host's tree, alone (the default would add the library's tree as an accelerator: tests/test_gpu_hybrid.py)
    flat.set_bvh(*tree)
    ctx.set_option("bvh_builder", -1)
    assert flat.bvh_mode()[0] == 1
    got = flat.bvh()
    for x, y in zip(got, tree):
        assert np.array_equal(np.ascontiguousarray(x).view(np.uint32), np.ascontiguousarray(y).view(np.uint32))
    rays = np.concatenate([ob.generate_rays(cam, T.scenes.camera_sample_grid(cam, 1, seed=5)), T.scenes.incoherent_rays(20000, np.float32([0, 0, -3]), np.float32([1, 1, -2]), seed=9)])
    t_ref, prim_ref, _, _ = osc.trace_closest(rays)
    hits = flat.trace_closest(rays)
    assert np.array_equal(hits["prim"], prim_ref) and np.array_equal(bits(hits["t"]), bits(t_ref))
    ref_film, _, _ = osc.render(cam, "path", 2, 5, seed=8)
    film = T.PathIntegrator(cam, T.SeededSampler(2, seed=8), 5).render(scene, ctx)
    assert_bits_equal(film, ref_film, "film through set_bvh(reference tree)")
    scene._flat = None
    flat.free()


def test_sppm_on_the_reference_tree(T, ob, ref_ctx):
    """C4's integrator on the reference's tree of the reference's own mesh: the integer / order-independent state equals the oracle's."""
    ply = os.path.join(GOLDEN, "caustic-glass.ply")
    scene = T.scenes.caustic_scene(ply if os.path.exists(ply) else "")
    cam = T.scenes.caustic_camera(24)
    osc = ob.OracleScene.from_scene(scene)
    flat = scene.flatten(ref_ctx)
    assert np.array_equal(flat.bvh()[1], osc.get_bvh()[1])
    integ = T.SPPMIntegrator(cam, 0.075, 5, 2, 20000, seed=11)
    integ.render(scene, ref_ctx)
    got = integ.state()
    ref = osc.sppm(cam, 0.075, 5, 2, 20000, seed=11)
    assert np.array_equal(got["M"], ref["M"]) and np.array_equal(bits(got["radius"]), bits(ref["radius"])) and np.array_equal(got["N"], ref["N"])
    assert_bits_equal(got["Ld"], ref["Ld"], "Ld")
    scene._flat = None
    flat.free()
