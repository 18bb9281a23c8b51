#!/usr/bin/env python3
"""A recording tracer end to end: N tasks of B photons of config 2 with a `HitRecorder` (every hit copied to the host per batch),
one after the other (`Pipeline.run`) and through `PipelineScheduler` (one hit queue per pipeline config: a batch runs while the
previous one's queue is read). Wall clock per batch; processFn counts the hits.   tools/record_flow_bench.py [photons] [batches]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from theia_amd import workloads as S
from theia_amd.response import HitRecorder
from theia_amd.hp import pipeline as pl

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for mode in ("serial loop", "scheduler"):
    rec = HitRecorder()
    sc = S.config2_sphere(batch=B, response=rec)
    pipe = pl.Pipeline(sc.stages())
    hits = []
    best = float("inf")
    if mode == "scheduler":
        sched = pl.PipelineScheduler(pipe, processFn=lambda c, b, a: hits.append(rec.queue.view(c).count))
    for rep in range(4):
        hits.clear()
        t0 = time.perf_counter()
        if mode == "scheduler":
            sched.schedule([{}] * N)
            sched.wait()
        else:
            for k in range(N):
                pipe.run(k % 2)
                hits.append(rec.queue.view(k % 2).count)
        dt = time.perf_counter() - t0
        if rep:
            best = min(best, dt)
    if mode == "scheduler":
        sched.destroy()
    print(f"{mode:12s} B={B} N={N}: {best / N * 1e3:8.3f} ms per batch, {int(np.mean(hits))} hits per batch "
          f"({int(np.mean(hits)) * 52 / 1e6:.1f} MB copied per batch)", flush=True)
