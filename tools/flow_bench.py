#!/usr/bin/env python3
"""
The path a theia user calls, end to end: N tasks of B photons through `PipelineScheduler` with a `processFn` that
copies the light curve (the flow of /root/reference/examples/02_scene_tracing.ipynb cells 20-24: `tasks = [{}] * 20`,
batches of 512 * 1024 photons), timed by wall clock from `schedule()` to the end of `wait()`.

    python tools/flow_bench.py [--workload config2] [--batch 524288,1000000] [--batches 20] [--modes 1/1,2/2,8/4]
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
from theia_amd import workloads as S  # noqa: E402
from theia_amd.hp import pipeline as pl  # noqa: E402


def flow(workload: str, B: int, N: int, lanes: int, divisor: int, reps: int = 3, inFlight=None):
    """best wall time of `reps` runs of N tasks, the curves of the last run"""
    sc = make_scenario(S, workload, B, -1)
    curves: list[np.ndarray] = []
    sched = pl.PipelineScheduler(pl.Pipeline(sc.stages()), processFn=lambda c, b, a: curves.append(sc.response.result(c).copy()),
                                 lanes=lanes, gridDivisor=divisor, inFlight=inFlight)
    best = float("inf")
    for rep in range(reps + 1):  # the first run warms up (kernel load, clocks, slot allocation)
        curves.clear()
        sc.rng.setParams(offset=0)  # every repetition traces the same photons
        t0 = time.perf_counter()
        sched.schedule([{}] * N)
        sched.wait()
        dt = time.perf_counter() - t0
        if rep:
            best = min(best, dt)
    depth = sched.batchesInFlight
    sched.destroy()
    return best, np.array(curves), depth


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="config2")
    ap.add_argument("--batch", default="524288,1000000")
    ap.add_argument("--batches", type=int, default=20)
    ap.add_argument("--modes", default="1/1,2/2,4/4,8/4", help="lanes/gridDivisor pairs")
    args = ap.parse_args()
    for B in (int(b) for b in args.batch.split(",")):
        ref = None
        for mode in args.modes.split(","):
            lanes, div, *rest = (int(x) for x in mode.split("/"))
            dt, curves, depth = flow(args.workload, B, args.batches, lanes, div, inFlight=rest[0] if rest else None)
            if ref is None:
                ref = curves
            err = float(np.abs(curves - ref).max() / max(ref.sum(axis=1).max(), 1e-30))
            print(f"{args.workload} B={B} N={args.batches} lanes={lanes} grid 1/{div} in flight {depth}: {dt * 1e3:8.3f} ms total, "
                  f"{dt / args.batches * 1e3:7.4f} ms per batch; curves vs first mode: max |d| / total = {err:.2e}", flush=True)


if __name__ == "__main__":
    main()
