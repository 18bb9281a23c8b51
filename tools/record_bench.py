#!/usr/bin/env python3
"""Kernel time of config 2 with a HitRecorder instead of the histogram response (development aid): tools/record_bench.py [photons]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from theia_amd import workloads as S
from theia_amd.response import HitRecorder
from theia_amd.hp.pipeline import Pipeline

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
for name, resp in (("histogram", None), ("recorder", HitRecorder())):
    sc = S.config2_sphere(batch=n, response=resp)
    pipe = Pipeline(sc.stages())
    ms = []
    for k in range(12):
        pipe.run(k % 2)
        ms.append(sc.tracer.program.last_kernel_ms(k % 2))
    info = sc.tracer.program.launch_info(n)
    extra = f" hits of the last batch: {sc.response.queue.view(1).count}" if resp is not None else ""
    print(f"{name:10s} kernel ms = {np.median(ms[-5:]):8.3f}  block={info['block']}{extra}", flush=True)
