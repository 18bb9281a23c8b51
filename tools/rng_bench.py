#!/usr/bin/env python3
"""config 2 / config 3 kernel time with PhiloxRNG vs SobolQRNG."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from theia_amd import workloads as S
import theia.random as TRN
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
for name, rng in (("philox", None), ("sobol", TRN.SobolQRNG(seed=0xC0FFEE, advanceSeed=7))):
    sc = S.config2_sphere(batch=n, rng=rng)
    S.run_batches(sc, 2)
    h = S.run_batches(sc, 3)
    ms = np.mean(sc.tracer.program.kernel_times(3))
    print(f"config2 {name:7s} kernel {ms:7.3f} ms  light curve sum {np.mean([x.sum() for x in h]):.6g}", flush=True)
