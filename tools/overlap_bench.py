#!/usr/bin/env python3
"""
Batches side by side: wall time of N batches of B photons of one workload launched through the C ABI with
`thx_tracer_set_concurrency(n)` for n = 1, 2, 4, 8 -- every launch on 1 / n of the chip's workgroups and a
stream of its own, so that the fixed tail of a launch (the life of its last photons, profiles/r03_launch_tail.txt)
is covered by the other batches in flight. Also checks that the curves do not depend on n beyond the order of
the fp32 slab sums.

    python tools/overlap_bench.py [--workload config2] [--batch 1000000] [--batches 20]
"""
from __future__ import annotations

import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from bench import make_scenario  # noqa: E402
from theia_amd import _backend  # noqa: E402
from theia_amd import workloads as S  # noqa: E402
from theia_amd.hp.pipeline import Pipeline  # noqa: E402


def run(workload: str, B: int, N: int, conc: int, div: int, slots: int, reps: int = 3):
    hip = _backend.get()
    sc = make_scenario(S, workload, B, -1)
    sc.response.normalization = 1.0 / B
    pipe = Pipeline(sc.stages())
    tr, prog = sc.tracer, sc.tracer.program
    n_bins = sc.response.nBins
    prog.set_concurrency(conc, div)
    out = np.zeros((N, n_bins), dtype=np.float32)
    best = float("inf")
    for rep in range(reps + 1):
        sc.rng.setParams(offset=0)  # every repetition traces the same photons
        hip.sync()
        t0 = time.perf_counter()
        pending = []
        for k in range(N):
            slot = k % slots
            if len(pending) == slots:  # the slot's previous batch has to be read before it is used again
                j, s = pending.pop(0)
                prog.collect_hist(s, out[j])
            pipe.update(k % 2)
            hip.tracer_run(prog.handle, slot, tr._assemble(k % 2))
            hip.tracer_reduce(prog.handle, slot, 1.0 / B)
            hip.tracer_fetch_hist(prog.handle, slot)
            pending.append((k, slot))
        for j, s in pending:
            prog.collect_hist(s, out[j])
        dt = time.perf_counter() - t0
        if rep > 0:
            best = min(best, dt)
    hip.tracer_check(prog.handle)
    return best, out.copy()


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="config2")
    ap.add_argument("--batch", type=int, default=1_000_000)
    ap.add_argument("--batches", type=int, default=20)
    ap.add_argument("--conc", default="1/1,2/2,4/4,8/4,8/8,16/8,16/16", help="lanes/grid_divisor pairs")
    args = ap.parse_args()
    ref = None
    for pair in args.conc.split(","):
        conc, div = (int(x) for x in pair.split("/"))
        for slots in (max(conc, 2),):
            dt, curves = run(args.workload, args.batch, args.batches, conc, div, slots)
            if ref is None:
                ref = curves
            err = float(np.abs(curves - ref).max() / max(ref.sum(axis=1).max(), 1e-30))
            print(f"{args.workload} B={args.batch} N={args.batches} lanes={conc} grid 1/{div} slots={slots}: {dt * 1e3:8.3f} ms total, "
                  f"{dt / args.batches * 1e3:7.4f} ms per batch, curves vs concurrency 1: max |d| / total = {err:.2e}", flush=True)


if __name__ == "__main__":
    main()
