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
