"""
Multi-GPU sharding of photon batches: photon paths are
independent and the RNG stream of a photon is its global index, so a batch of N
photons is cut into contiguous index ranges -- one per rank, one rank per GPU --
with no collective on the data path. The only exchange is one sum-reduce of the
per-GPU time histograms (RCCL over xGMI; `torch.distributed` backend "nccl" is
RCCL on ROCm). The message is n_bins x 4 B per batch, i.e. latency bound.
"""
from __future__ import annotations

import os

__all__ = ["init_from_env", "shard_range", "reduce_histograms", "gather_reduce_histograms", "reduce_counters", "world", "rank"]


def world() -> int:
    return int(os.environ.get("WORLD_SIZE", "1"))


def rank() -> int:
    return int(os.environ.get("RANK", "0"))


def shard_range(n_total: int, r: int, w: int) -> tuple[int, int]:
    """Photon index range [begin, end) of rank r out of w: contiguous, sizes differ
    by at most one, union is [0, n_total)."""
    base, rem = divmod(int(n_total), int(w))
    begin = r * base + min(r, rem)
    return begin, begin + base + (1 if r < rem else 0)


def init_from_env(backend: str | None = None):
    """Initialises torch.distributed from RANK/WORLD_SIZE/MASTER_* when WORLD_SIZE > 1.
    Returns the torch.distributed module, or None for a single process."""
    if world() <= 1:
        return None
    import torch
    import torch.distributed as dist

    if not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group(backend=backend, rank=rank(), world_size=world())
    return dist


def reduce_histograms(hist, dist=None):
    """In-place sum of the per-rank histograms (tensor of any shape) over all ranks:
    the single collective of the sharded path. No-op for one process."""
    if dist is None:
        return hist
    if getattr(hist, "is_cuda", False):
        import torch

        # the tracer / reducer kernels run on the library's own stream, which is not ordered
        # with torch's: finish them before RCCL reads the histograms
        torch.cuda.synchronize(hist.device)
    dist.all_reduce(hist, op=dist.ReduceOp.SUM)
    return hist


def gather_reduce_histograms(hist, dist=None):
    """The deterministic form of `reduce_histograms` ("gather + fixed-order Kahan"): every rank gathers
    all ranks' histograms in rank order and sums them with the slab reducer (`HistogramReducer`: Kahan partials, fixed
    order) -- the same bits on every rank and in every run, whatever order a ring all-reduce would have added them in.
    The torch.distributed counterpart of the C ABI's `thx_hist_gather_reduce`. In place; no-op for one process."""
    if dist is None:
        return hist
    import torch

    from . import _backend

    if getattr(hist, "is_cuda", False):
        torch.cuda.synchronize(hist.device)
    flat = hist.reshape(-1)
    parts = [torch.empty_like(flat) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, flat)
    stacked = torch.stack(parts).cpu().numpy()
    flat.copy_(torch.from_numpy(_backend.get().reduce_histograms(stacked, 1.0)).to(flat.device))
    return hist


def reduce_counters(counters, dist=None):
    """Sum of the u64 event / traversal counters (`EventStatisticCallback`, 12 + 8 of them) over all ranks: numpy uint64
    array in, numpy uint64 array out. The torch.distributed counterpart of `thx_stats_allreduce`."""
    import numpy as np

    c = np.asarray(counters, dtype=np.uint64)
    if dist is None:
        return c
    import torch

    t = torch.from_numpy(c.astype(np.int64))  # (counts stay far below 2^63)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy().astype(np.uint64)
