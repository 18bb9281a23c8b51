"""N > 1 path on CPU: world_size-2 `gloo` processes, each tracing its photon index
range (on the oracle stand-in), one all-reduce of the histograms; the sum equals
the single-process light curve of the whole batch."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, out_dir):
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch

    import scenarios as S
    from oracle_backend import OracleBackend
    from theia_amd import _backend, distributed

    torch.set_num_threads(1)
    dist = distributed.init_from_env("gloo")
    _backend.install(OracleBackend(threads=2))
    begin, end = distributed.shard_range(n_total, rank, world)
    sc = S.config2_sphere(end - begin, stats=True)
    sc.response.normalization = 1.0 / n_total
    sc.tracer.photonOffset = begin
    mine = S.run_batches(sc, 1)[0].copy()
    hist = torch.from_numpy(mine.copy())
    distributed.reduce_histograms(hist, dist)
    # the deterministic form (all-gather + fixed-order Kahan) and the u64 counter sum of SURVEY.md 8(e)
    exact = torch.from_numpy(mine.copy())
    distributed.gather_reduce_histograms(exact, dist)
    counters = distributed.reduce_counters(sc.tracer.program.read_stats(), dist)
    np.save(os.path.join(out_dir, f"exact{rank}.npy"), exact.numpy())
    if rank == 0:
        np.save(os.path.join(out_dir, "reduced.npy"), hist.numpy())
        np.save(os.path.join(out_dir, "counters.npy"), counters)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_ranges_partition():
    from theia_amd.distributed import shard_range

    for n in (0, 1, 7, 8, 1000, 10**8):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gloo_reduce_equals_single_process(tmp_path, oracle_backend):
    import torch.multiprocessing as mp

    import scenarios as S

    n = 30001  # odd on purpose: ragged shards
    mp.spawn(_worker, args=(2, _free_port(), n, str(tmp_path)), nprocs=2, join=True)
    reduced = np.load(tmp_path / "reduced.npy")
    full = S.run_batches(S.config2_sphere(n), 1)[0]
    assert full.sum() > 0
    assert np.abs(reduced - full).max() <= 1e-5 * full.sum()  # same photons, different fp32 summation order
    # gather + fixed-order Kahan: the same bits on both ranks, and the same curve
    e0, e1 = np.load(tmp_path / "exact0.npy"), np.load(tmp_path / "exact1.npy")
    assert np.array_equal(e0, e1) and np.abs(e0 - full).max() <= 1e-5 * full.sum()
    # the event counters of the two shards add up to those of the whole batch
    sc = S.config2_sphere(n, stats=True)
    S.run_batches(sc, 1)
    whole = sc.tracer.program.read_stats()
    counters = np.load(tmp_path / "counters.npy")
    assert counters[0] == n and np.array_equal(counters[:12], whole[:12]) and counters[19] == whole[19]
