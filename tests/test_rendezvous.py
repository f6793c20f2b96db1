"""The file rendezvous of the RCCL unique id (trace.jl_amd/parallel.py file_rendezvous; the same protocol in julia/TraceHIP.jl init_job!) on CPU: real processes, a
shared temporary directory, and the debris of a crashed earlier job in it — a stale id file, stale hello / ack files.  Every rank must come back with the id rank 0
made in THIS run (two ranks calling ncclCommInitRank with different ids hang), and a rank whose peers never arrive must raise instead of waiting forever."""
import multiprocessing as mp
import os

import pytest


def _rank(path, rank, world, delay, q):
    import time
    import __graft_entry__ as graft
    T = graft.load_package()
    time.sleep(delay)
    try:
        uid = T.parallel.file_rendezvous(path, rank, world, lambda: bytes([0xA0 + world]) * 128, 128, timeout_s=20.0, poll_s=0.01)
        q.put((rank, uid))
    except Exception as e:  # noqa: BLE001
        q.put((rank, repr(e)))


@pytest.mark.parametrize("order", ["rank 0 first", "rank 0 last"])
def test_file_rendezvous_ignores_the_debris_of_an_earlier_job(tmp_path, order):
    path = str(tmp_path / "rccl_id")
    world = 3
    # a crashed job of the same name: its id (another one), the tokens its readers wrote, their acks
    open(path, "wb").write(bytes([0x11]) * 128 + b"1:deadbeef\n2:deadbeef\n")
    for r in (1, 2):
        open(f"{path}.hello{r}", "wb").write(b"deadbeef")
        open(f"{path}.ack{r}", "wb").write(b"deadbeef")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    delays = {0: 0.0, 1: 0.6, 2: 1.2} if order == "rank 0 first" else {0: 1.2, 1: 0.0, 2: 0.4}
    procs = [ctx.Process(target=_rank, args=(path, r, world, delays[r], q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=60) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
    assert all(got[r] == bytes([0xA0 + world]) * 128 for r in range(world)), got


def test_file_rendezvous_times_out(tmp_path, T):
    path = str(tmp_path / "rccl_id")
    open(path, "wb").write(bytes([0x11]) * 128)  # a stale id nobody vouches for
    with pytest.raises(T.TraceHipError):
        T.parallel.file_rendezvous(path, 1, 2, None, 128, timeout_s=0.5, poll_s=0.01)
    with pytest.raises(T.TraceHipError):
        T.parallel.file_rendezvous(path, 0, 2, lambda: bytes(128), 128, timeout_s=0.5, poll_s=0.01)


def test_job_suffix_is_the_same_on_every_rank(T, monkeypatch):
    for name in ("TRACEHIP_JOB_ID", "SLURM_JOB_ID", "PBS_JOBID", "LSB_JOBID", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        monkeypatch.delenv(name, raising=False)
    assert T.parallel.Job.job_suffix() == ""  # no per-process ingredient (pid, ppid): ranks under srun / mpirun / wrapper shells agree
    monkeypatch.setenv("SLURM_JOB_ID", "4711")
    assert T.parallel.Job.job_suffix() == ".4711"
    monkeypatch.setenv("TRACEHIP_JOB_ID", "mine")
    assert T.parallel.Job.job_suffix() == ".mine"
