"""Process-per-GPU plumbing for the benchmark and for multi-process callers.

The hashing path has no exchange step: every object id depends on that object
alone, so ranks only (1) agree on who hashes what and (2) synchronise for
timing.  torch.distributed carries the barrier and the small reductions
("nccl" = RCCL when the ranks own GPUs, "gloo" on CPU-only ranks); no object
bytes and no digests travel between ranks.
"""
import ctypes as C
import os

import numpy as np


def env():
    """(rank, local_rank, world_size) from the launcher's environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend=None):
    """Join the process group when WORLD_SIZE > 1.  Returns the torch.distributed
    module, or None for a single process."""
    rank, local_rank, world = env()
    # VIT_FORCE_DIST=1 forms the group even for one rank: a way to exercise the RCCL
    # path (group set-up, device reductions, barrier) on a single-GPU machine.
    if world <= 1 and not os.environ.get("VIT_FORCE_DIST"):
        return None
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    os.environ.setdefault("RANK", str(rank))
    os.environ.setdefault("WORLD_SIZE", str(world))
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend=backend)
    return dist


def _device(dist):
    import torch
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def max_over_ranks(dist, value):
    """The slowest rank's value (a timing), identical on every rank."""
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=_device(dist))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(dist, value):
    if dist is None:
        return int(value)
    import torch
    t = torch.tensor([int(value)], dtype=torch.int64, device=_device(dist))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


_DONE_KEY = "vit_rank0_side_legs_done"


def release_ranks(dist):
    """Rank 0: tell the waiting ranks that its single-process measurements are over."""
    if dist is None:
        return
    from torch.distributed.distributed_c10d import _get_default_store
    _get_default_store().set(_DONE_KEY, "1")


def wait_for_rank0(dist, timeout_s=3600):
    """Other ranks: block on the rendezvous store, on the CPU, until rank 0 calls release_ranks().
    Not a collective: a barrier on RCCL would park a spinning kernel on every waiting rank's GPU,
    and rank 0 is meanwhile measuring ONE hasher that drives all of those GPUs."""
    if dist is None:
        return
    import datetime
    from torch.distributed.distributed_c10d import _get_default_store
    _get_default_store().wait([_DONE_KEY], datetime.timedelta(seconds=timeout_s))


def weak_first_index(rank, n_per_rank):
    """Weak scaling: rank r hashes synthetic objects [r*n, (r+1)*n)."""
    return rank * n_per_rank


def strong_shard(rank, world, nblocks):
    """Strong scaling over one object list: indices this rank hashes, balanced by
    block count with the library's own partitioner (vit_shard_plan)."""
    from . import _abi
    nblocks = np.ascontiguousarray(nblocks, dtype=np.uint32)
    plan = np.empty(max(len(nblocks), 1), dtype=np.uint32)
    rc = _abi.load().vit_shard_plan(nblocks.ctypes.data, len(nblocks), world, plan.ctypes.data)
    if rc != 0:
        raise RuntimeError("vit_shard_plan failed: %d" % rc)
    return np.nonzero(plan[:len(nblocks)] == rank)[0]


def xor_summary(digests):
    """Order-independent 20-byte summary of a set of digests."""
    d = np.ascontiguousarray(digests, dtype=np.uint8).reshape(-1, 20)
    return np.bitwise_xor.reduce(d, axis=0) if len(d) else np.zeros(20, dtype=np.uint8)


def combine_summaries(dist, summary):
    """XOR of every rank's summary (a checksum of checksums over the whole job)."""
    if dist is None:
        return np.asarray(summary, dtype=np.uint8)
    import torch
    t = torch.from_numpy(np.asarray(summary, dtype=np.uint8).astype(np.int64)).to(_device(dist))
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    acc = np.zeros(20, dtype=np.uint8)
    for o in out:
        acc ^= o.cpu().numpy().astype(np.uint8)
    return acc
