"""The multi-GPU entry points of the C ABI on ONE GPU (the box the tests run on has one): a 1-rank RCCL communicator created
from a unique id exactly as an N-rank job creates it; the collectives must leave single-rank results untouched, and
trhip_render_sppm with a communicator (its photon pass sharded 1 way, ϕ / M all-reduced every iteration) must reproduce the
render without one.  The N = 2 semantics are covered on CPU by tests/test_sharding_gloo.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture()
def comm_ctx(T):
    ctx = T.Context(0)
    uid = T._ffi.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    ctx.comm_init(uid, 0, 1)
    assert ctx.comm_rank() == (0, 1)
    yield ctx
    ctx.comm_destroy()
    assert ctx.comm_rank() == (0, 1)
    ctx.close()


def test_film_reduce_single_rank_is_identity(T, comm_ctx):
    import torch
    scene, cam = T.scenes.shadows_scene(), T.scenes.shadows_camera(48)
    h, w = cam.film.size
    film = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    integ = T.PathIntegrator(cam, T.SeededSampler(4, seed=9), 4)
    integ.render(scene, comm_ctx, device_out=film.data_ptr())
    before = film.cpu().numpy().copy()
    comm_ctx.film_reduce(film.data_ptr(), h * w, 0)
    assert np.array_equal(film.cpu().numpy().view(np.uint32), before.view(np.uint32))
    comm_ctx.film_allreduce(film.data_ptr(), h * w)
    assert np.array_equal(film.cpu().numpy().view(np.uint32), before.view(np.uint32))
    with pytest.raises(T.TraceHipError):
        comm_ctx.film_reduce(film.data_ptr(), h * w, 3)  # root outside the job
    scene._flat.free()
    scene._flat = None


def test_comm_errors(T, ctx):
    with pytest.raises(T.TraceHipError):
        ctx.comm_init(bytes(128), 2, 2)  # rank outside the job
    assert ctx.comm_rank() == (0, 1)
    ctx.film_reduce(0x1000, 16, 0)  # no communicator, single-process job: nothing to do (the pointer is not touched)


def test_sppm_with_communicator_equals_sppm_without(T, ctx, comm_ctx):
    scene, cam = T.scenes.cornell_scene(), T.scenes.cornell_camera(40)
    a = T.SPPMIntegrator(cam, 0.08, 5, 3, 20000, seed=11)
    img_a = a.render(scene, ctx).copy()
    st_a = a.state()
    scene._flat.free()
    scene._flat = None
    b = T.SPPMIntegrator(cam, 0.08, 5, 3, 20000, seed=11)
    img_b = b.render(scene, comm_ctx).copy()
    st_b = b.state()
    scene._flat.free()
    scene._flat = None
    assert np.array_equal(st_a["M"], st_b["M"]) and st_a["M"].sum() > 0
    for k in ("radius", "Ld"):
        assert np.array_equal(st_a[k].view(np.uint32), st_b[k].view(np.uint32)), k
    assert np.array_equal(st_a["N"], st_b["N"])
    # ϕ / τ / image: Float32 sums whose order is not fixed from run to run (the photon hits of a grid cell are binned with atomics,
    # like the reference's own Threads.Atomic adds, sppm.jl:398-399): the tolerance of tests/test_gpu_sppm.py
    np.testing.assert_allclose(st_b["phi"], st_a["phi"], rtol=2e-5, atol=2e-5 * np.abs(st_a["phi"]).max())
    np.testing.assert_allclose(st_b["tau"], st_a["tau"], rtol=5e-5, atol=5e-5 * np.abs(st_a["tau"]).max())
    np.testing.assert_allclose(img_b, img_a, rtol=1e-4, atol=1e-4 * np.abs(img_a).max())


def test_torch_nccl_group_and_library_communicator_coexist(T):
    """What `bench.py --gpus N` does in every rank, on the one GPU there is: torch.distributed holds an `nccl` process group (torch's own copy of RCCL) while the library
    dlopens librccl.so.1 for its communicator (th_comm.h) — two RCCL instances in one process.  The frame is rendered with the second (low-priority) stream live
    (option overlap = 1: shadow rays beside the next depth's closest-hit rays), then summed by trhip_film_reduce inside the library, then all-reduced by torch's group:
    neither may disturb the other, the film must come through bit for bit."""
    import os
    import subprocess
    import sys
    import textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import os, sys
        import numpy as np
        sys.path.insert(0, %r)
        import torch
        import torch.distributed as dist
        import __graft_entry__ as g
        T = g.load_package()
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29731", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1)
        warm = torch.ones(1024, device="cuda")
        dist.all_reduce(warm)                      # torch's RCCL instance is up and has built its communicator
        torch.cuda.synchronize()
        ctx = T.Context(0)
        ctx.comm_init(T._ffi.comm_unique_id(), 0, 1)  # the library's communicator (its own dlopen'ed RCCL), made exactly as an N-rank job makes it
        assert ctx.comm_rank() == (0, 1)
        ctx.set_option("overlap", 1)
        scene, cam = T.scenes.mesh_scene(96), T.scenes.cornell_camera(128)
        h, w = cam.film.size
        film = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
        integ = T.PathIntegrator(cam, T.SeededSampler(8, seed=21), 6)
        ref = integ.render(scene, ctx).copy()       # host copy of the same frame
        for _ in range(3):
            integ.render(scene, ctx, device_out=film.data_ptr())
            ctx.film_reduce(film.data_ptr(), h * w, 0)      # ncclReduce inside the library, on the library's stream
            dist.all_reduce(film)                           # … and torch's group right behind it
            torch.cuda.synchronize()
            assert np.array_equal(film.cpu().numpy().view(np.uint32), ref.view(np.uint32)), "film changed under the two collectives"
        assert integ.stats.shadow_rays > 0 and int(integ.stats.traversal) == 9
        ctx.comm_destroy()
        dist.barrier()
        dist.destroy_process_group()
        print("COEXIST OK", integ.stats.closest_rays, integ.stats.shadow_rays)
    """) % root
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0 and "COEXIST OK" in res.stdout, (res.stdout[-1500:], res.stderr[-3000:])
