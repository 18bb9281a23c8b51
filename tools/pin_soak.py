#!/usr/bin/env python3
"""Life cycle of page-locked IOQueue buffers (`thx_host_register`): recording tracers of changing size are built, run through
`PipelineScheduler`, and dropped -- some destroyed in order, some left to the garbage collector, some alive at interpreter
exit -- while a second thread keeps another recorder busy. Run under `python -X faulthandler`.   tools/pin_soak.py [rounds]"""
import gc, os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from theia_amd import workloads as S
from theia_amd.response import HitRecorder
from theia_amd.hp import pipeline as pl

R = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(7)
keep = []
stop = threading.Event()


def one(B, N, how):
    rec = HitRecorder()
    sc = S.config2_sphere(batch=B, response=rec)
    counts = []
    sched = pl.PipelineScheduler(pl.Pipeline(sc.stages()), processFn=lambda c, b, a: counts.append(rec.queue.view(c).count))
    sched.schedule([{}] * N)
    sched.wait()
    assert len(counts) == N and min(counts) > 0, counts
    if how == 0:
        sched.destroy()
    elif how == 1:
        keep.append((rec, sc, sched))  # alive at exit (scheduler threads included)
    return counts  # how == 2: left to the collector, scheduler not destroyed


def background():
    while not stop.is_set():
        one(20_000, 4, 0)


t = threading.Thread(target=background)
t.start()
total = 0
for r in range(R):
    B = int(rng.choice([1_000, 4_096, 50_000, 300_000]))
    total += sum(one(B, int(rng.integers(2, 7)), r % 3 if r % 30 == 1 or r % 3 != 1 else 0))
    if r % 10 == 9:
        gc.collect()
        print(f"round {r + 1}: {total} hits read, {len(keep)} recorders kept", flush=True)
stop.set()
t.join()
print("OK", flush=True)
