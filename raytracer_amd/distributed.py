"""Photon-range sharding across the ranks of a one-process-per-GPU job.

Photons are independent and their variates are keyed by photon id, so the
only exchange step is one sum of the tally block at the end of a run:
`torch.distributed.all_reduce(SUM)` (RCCL over xGMI with backend "nccl",
gloo on CPU).  All tallies are unsigned integers, so the sum is exact and
independent of rank count and order.
"""
from __future__ import annotations

import os

import numpy as np

from . import _abi


def _dist():
    try:
        import torch.distributed as dist
    except Exception:       # torch absent: single process
        return None
    if dist.is_available() and dist.is_initialized():
        return dist
    return None


def rank_world():
    d = _dist()
    if d is None:
        return 0, 1
    return d.get_rank(), d.get_world_size()


def local_device():
    return int(os.environ.get("LOCAL_RANK", "0"))


def shard_range(n, rank, world):
    """Contiguous block [first, first+count) of `n` photon ids for `rank`;
    the first n % world ranks get one photon more."""
    base, extra = divmod(int(n), int(world))
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


def shared_entropy():
    """A fresh 63-bit seed, identical on every rank."""
    seed = int.from_bytes(os.urandom(8), "little") >> 1
    d = _dist()
    if d is not None:
        import torch
        dev = _tensor_device(d)
        t = torch.tensor([seed], dtype=torch.int64, device=dev)
        d.broadcast(t, src=0)
        seed = int(t.item())
    return seed


def _tensor_device(d, device_index=None):
    import torch
    if d.get_backend() == "nccl":
        idx = local_device() if device_index is None else device_index
        return torch.device("cuda", idx)
    return torch.device("cpu")


def tally_payload_bytes(block):
    """Size of the one message a rank contributes to the sum of the tallies."""
    _, arrays = block
    return 8 * (len(_abi.TALLY_SCALARS) + sum(arrays[k].size for k in list(_abi.TALLY_ARRAYS) + ["spectrum"]))


def collective_info(block=None):
    """What the exchange step of a run really saw: backend and number of ranks of the process
    group (from torch.distributed itself, not from the command line), bytes per rank."""
    d = _dist()
    info = {"op": "all_reduce(sum) of the integer tallies, once per run",
            "backend": None if d is None else str(d.get_backend()),
            "world_size": 1 if d is None else int(d.get_world_size())}
    if block is not None:
        info["payload_bytes"] = int(tally_payload_bytes(block))
    return info


def allreduce_tallies(block, device_index=None):
    """Sum a (Tallies struct, arrays) block over all ranks, in place.

    The counters travel as one int64 tensor (bit pattern of the uint64
    values; two's-complement addition is the same operation)."""
    d = _dist()
    if d is None:
        return block
    import torch
    t, arrays = block
    scalars = np.array([getattr(t, k) for k in _abi.TALLY_SCALARS], dtype=np.uint64)
    keys = list(_abi.TALLY_ARRAYS) + ["spectrum"]
    flat = np.concatenate([scalars] + [arrays[k] for k in keys]).view(np.int64)
    ten = torch.from_numpy(flat.copy()).to(_tensor_device(d, device_index))
    d.all_reduce(ten, op=d.ReduceOp.SUM)
    summed = ten.cpu().numpy().view(np.uint64)
    for i, k in enumerate(_abi.TALLY_SCALARS):
        setattr(t, k, int(summed[i]))
    pos = len(_abi.TALLY_SCALARS)
    for k in keys:
        n = arrays[k].size
        arrays[k][:] = summed[pos:pos + n]
        pos += n
    return block
