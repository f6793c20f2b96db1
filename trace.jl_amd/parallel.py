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
    """Sum-reduce (H, W, 4) film accumulators held in torch tensors over torch.distributed: what the CPU tests (gloo, tests/test_sharding_gloo.py) use to check
    the sharding arithmetic with the oracle.  bench.py does NOT use it: an N-GPU line comes from the library's RCCL communicator (trhip_film_reduce) or not at all."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(film, dst=dst, op=dist.ReduceOp.SUM)
    return film


def _write_atomically(path: str, data: bytes):
    tmp = f"{path}.tmp{os.getpid()}"
    with open(tmp, "wb") as f:
        f.write(data)
    os.replace(tmp, path)


def _read(path: str) -> bytes:
    try:
        with open(path, "rb") as f:
            return f.read()
    except (FileNotFoundError, PermissionError):
        return b""


def file_rendezvous(path: str, rank: int, world: int, make_id, id_bytes: int, timeout_s: float = 120.0, poll_s: float = 0.05) -> bytes:
    """The RCCL unique id from rank 0 to the others through a shared directory, safe against files a crashed job left behind (two ranks that call ncclCommInitRank
    with different ids hang).  No file is trusted for being there: every reader proves to rank 0 that it is alive NOW, and accepts an id only from a rank 0 that has
    seen that proof.

        reader r:  writes  <path>.hello<r> = a fresh random token;  polls <path> until it holds the line "r:<its token>";  then writes <path>.ack<r> =
                   "<its token>:<rank 0's token>" (the line "0:..." of the same file)
        rank 0:    makes the id and a fresh random token once;  repeats { read the hello files that exist; (re)write <path> = id + "0:<token>" + one line "r:token"
                   per reader seen; } until every reader's ack file holds "<the token of its current hello file>:<rank 0's token>".

    A stale <path> does not carry the new tokens; a stale hello is echoed, ignored by the live reader, and replaced as soon as that reader writes its own; a stale
    ack does not carry rank 0's new token.  Raises TraceHipError after `timeout_s` (both sides).  The same protocol, byte for byte, is in julia/TraceHIP.jl (init_job!)."""
    import secrets
    t0 = time.time()

    def expired():
        return time.time() - t0 > timeout_s

    if rank == 0:
        uid = bytes(make_id())
        assert len(uid) == id_bytes
        mine = secrets.token_hex(16).encode()
        written = None
        while True:
            tokens = {}
            for r in range(1, world):
                tok = _read(f"{path}.hello{r}").strip()
                if tok:
                    tokens[r] = tok
            if tokens != written:
                _write_atomically(path, uid + b"0:%s\n" % mine + b"".join(b"%d:%s\n" % (r, tok) for r, tok in sorted(tokens.items())))
                written = dict(tokens)
            if len(tokens) == world - 1 and all(_read(f"{path}.ack{r}").strip() == tokens[r] + b":" + mine for r in range(1, world)):
                return uid
            if expired():
                raise _ffi.TraceHipError(f"RCCL id rendezvous at {path}: {world - 1 - len(tokens)} rank(s) never showed up, or never acknowledged, within {timeout_s} s")
            time.sleep(poll_s)
    token = secrets.token_hex(16).encode()
    _write_atomically(f"{path}.hello{rank}", token)
    want = b"%d:%s" % (rank, token)
    while True:
        data = _read(path)
        lines = data[id_bytes:].split(b"\n") if len(data) > id_bytes else []
        if want in lines and lines[0].startswith(b"0:"):
            _write_atomically(f"{path}.ack{rank}", token + b":" + lines[0][2:])
            return data[:id_bytes]
        if expired():
            raise _ffi.TraceHipError(f"no RCCL id for rank {rank} at {path} after {timeout_s} s (is rank 0 running, and is the directory shared?)")
        time.sleep(poll_s)


class Job:
    """This process's membership in an N-GPU job: creates the library-side RCCL communicator on `ctx`.

    rank / world default to RANK / WORLD_SIZE.  Rendezvous of the unique id: `id_file` (a path in a directory all ranks share: file_rendezvous above) or, when
    torch.distributed is initialised, a broadcast over that group."""

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
    def job_suffix() -> str:
        """Optional: TRACEHIP_JOB_ID (or the batch system's job id) keeps two jobs that share `id_file` apart in the file system.  It is NOT what makes a stale file
        harmless — the token handshake of file_rendezvous is; without any of these variables every rank uses the bare path (all ranks compute the same name under any
        launcher: srun, mpirun, wrapper shells)."""
        env = os.environ
        for name in ("TRACEHIP_JOB_ID", "SLURM_JOB_ID", "PBS_JOBID", "LSB_JOBID"):
            if env.get(name):
                return "." + env[name]
        return ""

    def _exchange_id(self, id_file, timeout_s) -> bytes:
        if id_file:
            self._id_path = id_file + self.job_suffix()
            return file_rendezvous(self._id_path, self.rank, self.world, _ffi.comm_unique_id, _ffi.UNIQUE_ID_BYTES, timeout_s)
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
        if self.rank == 0 and self._id_path:
            for f in [self._id_path] + [f"{self._id_path}.{kind}{r}" for kind in ("hello", "ack") for r in range(1, self.world)]:
                if os.path.exists(f):
                    os.unlink(f)
            self._id_path = None
