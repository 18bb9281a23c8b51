#!/usr/bin/env python3
"""One launch of 10^8 photons (config 2) and a photon offset close to 2^32: counters add up, the
light curve matches the mean of ten 10^7 launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from theia_amd import workloads as S
n = 100_000_000
sc = S.config2_sphere(batch=n, stats=True)
h = S.run_batches(sc, 1)[0]
cb = sc.callback
print("created", cb.created, "steps", cb.traversal["steps"], "sum", h.sum(), "ms", sc.tracer.program.last_kernel_ms(0))
assert cb.created == n and np.isfinite(h).all()
sc2 = S.config2_sphere(batch=10_000_000)
h2 = np.mean(S.run_batches(sc2, 4), 0)
print("rel diff of integrals", abs(h.sum() / h2.sum() - 1))
assert abs(h.sum() / h2.sum() - 1) < 5e-3
sc3 = S.config2_sphere(batch=1_000_000, stats=True)
sc3.tracer.photonOffset = 2**32 - 500_000  # wraps around
h3 = S.run_batches(sc3, 1)[0]
print("wrap: created", sc3.callback.created, "sum", h3.sum())
assert sc3.callback.created == 1_000_000 and abs(h3.sum() / h2.sum() - 1) < 3e-2
print("ok")
