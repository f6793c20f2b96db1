"""The N > 1 path on CPU: two processes (gloo), each producing the film of its sample-index shard with the CPU oracle,
one sum-reduce to rank 0 (trace_jl_amd.parallel.reduce_film — the function bench.py calls over RCCL).  The reduced film
must equal the oracle's single-process render of all samples up to Float32 summation order."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import __graft_entry__ as graft
T = graft.load_package()
import oracle_bridge as ob
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo")
spp = 3
scene, cam = T.scenes.shadows_scene(), T.scenes.shadows_camera(24)
osc = ob.OracleScene.from_scene(scene)
xyzw, _, _ = osc.render(cam, "path", spp, 4, seed=11, sample_offset=T.parallel.shard_sample_offset(rank, spp))
film = torch.from_numpy(xyzw.copy())
T.parallel.reduce_film(film, dst=0)
if rank == 0:
    np.save(sys.argv[2], film.numpy())
dist.barrier()
dist.destroy_process_group()
'''


def test_two_rank_film_reduce(T, ob, tmp_path):
    out = tmp_path / "film.npy"
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29571", OMP_NUM_THREADS="2")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29571",
                           str(script), ROOT, str(out)], env=env, timeout=600)
    reduced = np.load(out)
    scene, cam = T.scenes.shadows_scene(), T.scenes.shadows_camera(24)
    full, _, _ = ob.OracleScene.from_scene(scene).render(cam, "path", 6, 4, seed=11)
    assert reduced.shape == full.shape and full[..., :3].max() > 0
    np.testing.assert_allclose(reduced, full, rtol=3e-5, atol=1e-6)
    # weights are sums of table values: exact up to order as well
    assert np.abs(reduced[..., 3] - full[..., 3]).max() <= 1e-4


STRONG_WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import __graft_entry__ as graft
T = graft.load_package()
import oracle_bridge as ob
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo")
total_spp = 7  # ONE frame of 7 samples per pixel split over the ranks (bench.py --scaling strong): 4 + 3
spp, off = T.parallel.shard_samples(total_spp, rank, world)
assert (spp, off) == ((4, 0) if rank == 0 else (3, 4))
scene, cam = T.scenes.shadows_scene(), T.scenes.shadows_camera(24)
osc = ob.OracleScene.from_scene(scene)
xyzw, _, _ = osc.render(cam, "path", spp, 4, seed=11, sample_offset=off)
film = torch.from_numpy(xyzw.copy())
T.parallel.reduce_film(film, dst=0)
if rank == 0:
    np.save(sys.argv[2], film.numpy())
dist.barrier()
dist.destroy_process_group()
'''

SPPM_WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import __graft_entry__ as graft
T = graft.load_package()
import oracle_bridge as ob
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo")
scene, cam = T.scenes.cornell_scene(), T.scenes.cornell_camera(32)
osc = ob.OracleScene.from_scene(scene)
P = 9000
lo, hi = T.parallel.photon_slice(P, rank, world)      # the slice trhip_render_sppm gives rank r of a communicator (tu_sppm.hip)
calls = []
def exchange(phi, M):                                  # one all-reduce of phi (3 floats) and M per pixel per iteration (SURVEY.md 8e)
    t_phi, t_M = torch.from_numpy(phi), torch.from_numpy(M)
    dist.all_reduce(t_phi, op=dist.ReduceOp.SUM)
    dist.all_reduce(t_M, op=dist.ReduceOp.SUM)
    calls.append(int(M.sum()))
r = osc.sppm(cam, 0.08, 5, 3, P, seed=11, photon_range=(lo, hi), exchange=exchange)
assert len(calls) == 3
np.savez(sys.argv[2] + f".{rank}.npz", image=r["image"], M=r["M"], radius=r["radius"], N=r["N"], tau=r["tau"], Ld=r["Ld"], phi=r["phi"])
dist.barrier()
dist.destroy_process_group()
'''


def _run_ranks(tmp_path, body, out, port, n=2, check=True):
    script = tmp_path / "worker.py"
    script.write_text(body)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2" if n <= 2 else "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), ROOT, str(out)]
    if check:
        subprocess.check_call(cmd, env=env, timeout=900)
        return None
    return subprocess.run(cmd, env=env, timeout=900, capture_output=True, text=True)


def _run_two_ranks(tmp_path, body, out, port):
    _run_ranks(tmp_path, body, out, port, 2)


def test_two_rank_strong_scaling_frame(T, ob, tmp_path):
    """bench.py --scaling strong: ONE frame's samples split over the ranks (uneven: 4 + 3), films sum-reduced: equals the
    single-process frame of all 7 samples up to Float32 summation order."""
    out = tmp_path / "film.npy"
    _run_two_ranks(tmp_path, STRONG_WORKER, out, 29573)
    reduced = np.load(out)
    scene, cam = T.scenes.shadows_scene(), T.scenes.shadows_camera(24)
    full, _, _ = ob.OracleScene.from_scene(scene).render(cam, "path", 7, 4, seed=11)
    assert full[..., :3].max() > 0
    np.testing.assert_allclose(reduced, full, rtol=3e-5, atol=1e-6)


def test_two_rank_sppm_photon_sharding(T, ob, tmp_path):
    """SPPM over two processes as trhip_render_sppm does it with a communicator: the camera pass replicated, each rank tracing its
    slice of every iteration's photons, ONE all-reduce of ϕ and M per iteration before _update_pixels!.  Both ranks end with the same
    pixels; M, radius and N equal the single-process run exactly (integers / functions of M), ϕ, τ and the image up to the
    order of the Float32 adds (the reference's own atomics are unordered, sppm.jl:398-399)."""
    out = tmp_path / "sppm"
    _run_two_ranks(tmp_path, SPPM_WORKER, out, 29575)
    r0, r1 = np.load(str(out) + ".0.npz"), np.load(str(out) + ".1.npz")
    for k in r0.files:
        assert np.array_equal(r0[k], r1[k]), f"ranks disagree on {k}"
    scene, cam = T.scenes.cornell_scene(), T.scenes.cornell_camera(32)
    ref = ob.OracleScene.from_scene(scene).sppm(cam, 0.08, 5, 3, 9000, seed=11)
    assert np.array_equal(r0["M"], ref["M"]) and ref["M"].sum() > 0
    assert np.array_equal(r0["radius"], ref["radius"]) and np.array_equal(r0["N"], ref["N"])
    assert np.array_equal(r0["Ld"].view(np.uint32), ref["Ld"].view(np.uint32))
    scale = np.abs(ref["phi"]).max()
    np.testing.assert_allclose(r0["phi"], ref["phi"], rtol=2e-5, atol=2e-5 * scale)
    np.testing.assert_allclose(r0["tau"], ref["tau"], rtol=5e-5, atol=5e-5 * np.abs(ref["tau"]).max())
    np.testing.assert_allclose(r0["image"], ref["image"], rtol=1e-4, atol=1e-4 * np.abs(ref["image"]).max())


def test_c5_sample_sharding_arithmetic(T):
    """BASELINE configs[4] (4096 x 4096, 1024 spp, 8 GPUs) is sharded by GLOBAL SAMPLE INDEX: rank r renders samples [offset_r, offset_r + spp_r) of every
    pixel with the frame's seed, so that the union over the ranks is exactly the single-process frame's sample set — whatever the split."""
    P = T.parallel
    shards = [P.shard_samples(1024, r, 8) for r in range(8)]
    assert shards == [(128, 128 * r) for r in range(8)]
    for total, world in ((1000, 8), (1001, 8), (1023, 8), (7, 8), (1024, 3), (1, 2), (256, 1)):
        shards = [P.shard_samples(total, r, world) for r in range(world)]
        sizes = [n for n, _ in shards]
        assert sum(sizes) == total and max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True), (total, world, sizes)
        covered = []
        for n, off in shards:
            covered += list(range(off, off + n))
        assert covered == list(range(total)), (total, world)  # contiguous, disjoint, in rank order
    assert P.shard_samples(1000, 0, 8) == (125, 0) and P.shard_samples(1001, 0, 8) == (126, 0) and P.shard_samples(1001, 1, 8) == (125, 126)
    # weak scaling (bench.py --scaling weak): every rank its own `spp` samples, offsets r * spp
    assert [P.shard_sample_offset(r, 128) for r in range(8)] == [128 * r for r in range(8)]
    # SPPM photons (trhip_render_sppm with a communicator): contiguous slices of every iteration's photon indices
    for photons, world in ((1023 * 1023, 8), (9000, 2), (5, 8)):
        sl = [P.photon_slice(photons, r, world) for r in range(world)]
        assert sl[0][0] == 0 and sl[-1][1] == photons and all(sl[r][1] == sl[r + 1][0] for r in range(world - 1)) and all(b >= a for a, b in sl)


EIGHT_WORKER = STRONG_WORKER.replace("total_spp = 7  # ONE frame of 7 samples per pixel split over the ranks (bench.py --scaling strong): 4 + 3", "total_spp = 11") \
    .replace("assert (spp, off) == ((4, 0) if rank == 0 else (3, 4))", "assert world == 8 and spp == (2 if rank < 3 else 1)") \
    .replace("T.scenes.shadows_camera(24)", "T.scenes.shadows_camera(16)")


def test_eight_rank_strong_scaling_frame(T, ob, tmp_path):
    """C5's decomposition at its rank count: ONE frame's 11 samples per pixel over 8 processes (2, 2, 2, 1, 1, 1, 1, 1), films sum-reduced onto rank 0: the
    single-process frame of all 11 samples up to Float32 summation order."""
    out = tmp_path / "film8.npy"
    _run_ranks(tmp_path, EIGHT_WORKER, out, 29577, n=8)
    reduced = np.load(out)
    scene, cam = T.scenes.shadows_scene(), T.scenes.shadows_camera(16)
    full, _, _ = ob.OracleScene.from_scene(scene).render(cam, "path", 11, 4, seed=11)
    assert full[..., :3].max() > 0
    np.testing.assert_allclose(reduced, full, rtol=5e-5, atol=2e-6)


BENCH_GUARD_WORKER = r'''
import os, sys
import torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import __graft_entry__ as graft
import bench
T = graft.load_package()
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo")


class OneRankContext:
    """What bench.py sees when the library's communicator did not come up with the job's size: trhip_comm_init was called, trhip_comm_rank says (0, 1)."""
    def __init__(self):
        self.inits = []

    def comm_init(self, uid, rank, world):
        assert len(uid) == T._ffi.UNIQUE_ID_BYTES
        self.inits.append((rank, world))

    def comm_rank(self):
        return (0, 1)


ctx = OneRankContext()
orig = T._ffi.comm_unique_id
T._ffi.comm_unique_id = lambda: bytes(range(128))  # (the real one needs librccl and a GPU; the id's content is irrelevant here)
try:
    bench.require_communicator(T, ctx, rank, world, "cpu")
finally:
    T._ffi.comm_unique_id = orig
print("UNREACHABLE: a bench line would follow")
'''


def test_bench_refuses_a_line_without_an_n_rank_library_communicator(T, tmp_path):
    """bench.py --gpus N prints its line only when trhip_comm_rank reports N ranks on every rank; otherwise every rank leaves with exit code 3 (the driver then
    records no SCALE entry instead of a number that came from a torch.distributed stand-in)."""
    r = _run_ranks(tmp_path, BENCH_GUARD_WORKER, tmp_path / "unused", 29579, n=2, check=False)
    assert r.returncode != 0, r.stdout + r.stderr
    assert "UNREACHABLE" not in r.stdout and "no bench line" in r.stderr, r.stdout + r.stderr
    assert "exitcode  : 3" in r.stderr or "exitcode: 3" in r.stderr or "exit code 3" in r.stderr.lower() or "(exitcode: 3)" in r.stderr, r.stderr[-2000:]


def test_c5_band_boundaries(T):
    """BASELINE configs[4] on one rank of eight: 4096 x 4096, 128 of the 1024 spp.  The per-sample buffers (17 B per camera sample with the default film pass) are larger than what
    trhip_render_path gives them (half of the free HBM), so the frame renders in bands of whole 16-pixel tile rows (trhip_plan_bands: the library's own arithmetic, host only).
    The bands must tile the sample rows exactly — contiguous, disjoint, tile-aligned, every row once — and each must fit 32-bit sample indices."""
    import ctypes as C
    cam = T.scenes.cornell_camera(4096)
    sn = cam.sensor()
    sb = cam.film.get_sample_bounds()
    y_min, y_max = int(sb.p_min[1]), int(sb.p_max[1])
    sb_w = int(sb.p_max[0]) - int(sb.p_min[0]) + 1
    lib = T._ffi.lib()
    for spp, budget in ((128, 120 << 30), (128, 40 << 30), (1024, 120 << 30), (128, 1 << 40), (8, 1 << 20)):
        n = C.c_uint32(0)
        first, rows = (C.c_int32 * 512)(), (C.c_int32 * 512)()
        assert lib.trhip_plan_bands(C.byref(sn), spp, budget, 17, 512, C.byref(n), first, rows) == 0
        nb = n.value
        assert 1 <= nb <= 512
        bands = [(first[k], rows[k]) for k in range(nb)]
        assert bands[0][0] == y_min and bands[-1][0] + bands[-1][1] - 1 == y_max, (spp, budget, bands[:3])
        for (y0, r0), (y1, _) in zip(bands, bands[1:]):
            assert y1 == y0 + r0 and r0 % 16 == 0 and (y0 - y_min) % 16 == 0  # contiguous, whole tile rows
        assert all(r > 0 and sb_w * r * spp < (1 << 32) for _, r in bands)
        if budget >= (1 << 40):
            assert nb == 1 or sb_w * (y_max - y_min + 1) * spp >= (1 << 32)   # everything fits: one band unless 32-bit indices forbid it
        per_band_bytes = max(r for _, r in bands) * sb_w * spp * 17
        assert nb == 1 or per_band_bytes <= budget or max(r for _, r in bands) == 16  # (a single tile row is the smallest band)
