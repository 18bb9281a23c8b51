#!/usr/bin/env python3
"""Odd batch sizes and all scene workloads through the staged kernel: event counters must add up
(every photon created exactly once, no photon lost by the scheduler)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from theia_amd import workloads as S
for make, name in ((S.config2_sphere, "config2"), (S.config3_cube_torus, "config3"), (S.config5_suzanne_grid, "config5")):
    for n in (1, 63, 64, 65, 1000, 4097, 100_003, 1_000_001):
        if name == "config5" and n > 200_000:
            continue
        sc = make(batch=n, stats=True)
        h = S.run_batches(sc, 2)
        cb = sc.callback
        ok = cb.created == 2 * n and np.isfinite(h[0]).all()
        print(f"{name} n={n:8d} created={cb.created} steps={cb.traversal['steps']} finite={np.isfinite(h[0]).all()} {'ok' if ok else 'FAIL'}", flush=True)
        assert ok
print("all ok")
