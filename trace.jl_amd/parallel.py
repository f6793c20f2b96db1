"""Multi-GPU jobs: one process per GPU, the scene replicated, samples sharded by global sample index, ONE
collective per frame — the sum of the per-rank film accumulators (Film pixels are additive: film.jl:161-162, 190-191, what
merge_film_tile! relies on, film.jl:182-193) — and for SPPM one all-reduce of ϕ / M per iteration inside trhip_render_sppm.

The collectives live behind the C ABI (trhip_comm_init / trhip_film_reduce: RCCL over xGMI inside libtracehip.so), so a Julia
host gets the same path (julia/TraceHIP.jl).  What a host has to supply is the rendezvous: rank 0 makes the 128-byte RCCL id
(`comm_unique_id`) and hands it to the other processes.  `Job` does that over an already initialised torch.distributed group
(bench.py), or through a file (`id_file`) when there is no such group.  CPU tests (tests/test_sharding_gloo.py) run the same
sharding arithmetic with the oracle and gloo.
"""
from __future__ import annotations

import os
import time

from . import _ffi


def shard_samples(total_spp: int, rank: int, world: int):
    """Strong scaling: the `total_spp` samples of ONE frame split over the ranks -> (spp of this rank, first global sample index)."""
    base, rem = divmod(int(total_spp), int(world))
    return base + (1 if rank < rem else 0), rank * base + min(rank, rem)


def shard_sample_offset(rank: int, spp_per_rank: int) -> int:
    """Weak scaling: every rank renders `spp_per_rank` samples; rank r owns global sample indices [r * spp, (r + 1) * spp)."""
    return int(rank) * int(spp_per_rank)


def photon_slice(photons_per_iteration: int, rank: int, world: int):
    """The photon indices [lo, hi) of every SPPM iteration that rank r traces (what trhip_render_sppm does with a communicator)."""
    p = int(photons_per_iteration)
    return p * rank // world, p * (rank + 1) // world


def reduce_film(film, dst: int = 0):
    """Sum-reduce (H, W, 4) film accumulators held in torch tensors over torch.distributed (gloo in the CPU tests; the fallback of
    bench.py when the library's own communicator cannot be created)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(film, dst=dst, op=dist.ReduceOp.SUM)
    return film


class Job:
    """This process's membership in an N-GPU job: creates the library-side RCCL communicator on `ctx`.

    rank / world default to RANK / WORLD_SIZE.  Rendezvous of the unique id: `id_file` (rank 0 writes it, the others poll for
    it) or, when torch.distributed is initialised, a broadcast over that group."""

    def __init__(self, ctx: _ffi.Context, rank: int | None = None, world: int | None = None, id_file: str | None = None, timeout_s: float = 120.0):
        self.ctx = ctx
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
        self.ok = False
        self._id_path = None
        if self.world <= 1:
            return
        uid = self._exchange_id(id_file, timeout_s)
        ctx.comm_init(uid, self.rank, self.world)
        self.ok = True

    @staticmethod
    def job_nonce() -> str:
        """What tells this job's id file from an earlier job's at the same path: TRACEHIP_JOB_ID, else the launcher's rendezvous
        (TORCHELASTIC_RUN_ID + MASTER_PORT), else the launcher's pid (the ranks of one node share their parent)."""
        env = os.environ
        if env.get("TRACEHIP_JOB_ID"):
            return env["TRACEHIP_JOB_ID"]
        if env.get("MASTER_PORT"):
            return f"{env.get('TORCHELASTIC_RUN_ID', 'run')}-{env['MASTER_PORT']}"
        return f"ppid{os.getppid()}"

    def _exchange_id(self, id_file, timeout_s) -> bytes:
        if id_file:
            # the file is per job (nonce in its name), written atomically, and removed by rank 0 in close(): a second job at the same
            # path never reads the first one's id (ncclCommInitRank with two different ids hangs)
            path = self._id_path = f"{id_file}.{self.job_nonce()}"
            if self.rank == 0:
                if os.path.exists(path):
                    os.unlink(path)  # a stale file of a crashed job with the same nonce
                uid = _ffi.comm_unique_id()
                tmp = path + f".tmp{os.getpid()}"
                with open(tmp, "wb") as f:
                    f.write(uid)
                os.replace(tmp, path)
                return uid
            t0 = time.time()
            while True:
                try:
                    with open(path, "rb") as f:
                        uid = f.read()
                    if len(uid) == _ffi.UNIQUE_ID_BYTES:
                        return uid
                except FileNotFoundError:
                    pass
                if time.time() - t0 > timeout_s:
                    raise _ffi.TraceHipError(f"no RCCL id at {path} after {timeout_s} s")
                time.sleep(0.05)
        import torch
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            raise _ffi.TraceHipError("Job needs an id_file or an initialised torch.distributed group to distribute the RCCL id")
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        buf = torch.zeros(_ffi.UNIQUE_ID_BYTES, dtype=torch.uint8, device=dev)
        if self.rank == 0:
            buf.copy_(torch.frombuffer(bytearray(_ffi.comm_unique_id()), dtype=torch.uint8))
        dist.broadcast(buf, src=0)
        return bytes(buf.cpu().numpy().tobytes())

    def reduce_film(self, device_ptr: int, n_pixels: int, root: int = 0):
        """In-place sum of the ranks' film accumulators onto `root` (trhip_film_reduce).  A job of several processes whose communicator
        is missing fails here: trhip_film_reduce on a context without one is a no-op (the film of a single process IS the sum)."""
        if self.world > 1:
            _, n = self.ctx.comm_rank()
            if not self.ok or n != self.world:
                raise _ffi.TraceHipError(f"film reduce of a {self.world}-process job, but the library's communicator has {n} rank(s)")
            self.ctx.film_reduce(device_ptr, n_pixels, root)

    def close(self):
        if self.ok:
            self.ctx.comm_destroy()  # every rank has read the id by now (ncclCommInitRank returned everywhere)
            self.ok = False
        if self.rank == 0 and self._id_path and os.path.exists(self._id_path):
            os.unlink(self._id_path)
            self._id_path = None
