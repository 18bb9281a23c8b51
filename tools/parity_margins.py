#!/usr/bin/env python3
"""Observed oracle-vs-HIP deviations of the parity cases (development aid: how much margin do the tolerances of
tests/test_gpu_parity.py leave?)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np
import test_gpu_parity as T
from theia_amd import _backend, workloads as S

hip = _backend.HipBackend()
_backend._backend = hip
cases = [("volume mis pol=%d" % p, S.config1_volume, dict(batch=20000, polarized=bool(p), stats=True, track=True), False) for p in (0, 1)]
cases += [("volume no-mis pol=%d" % p, S.config1_volume, dict(batch=20000, polarized=bool(p), stats=True, track=True, disableTargetSampling=True), True) for p in (0, 1)]
cases += [("sphere guide=%d pol=%d" % (g, p), S.config2_sphere, dict(batch=20000, polarized=bool(p), stats=True, track=True, guide=bool(g)), False) for g in (1, 0) for p in (0, 1)]
cases += [("cube_torus", S.config3_cube_torus, dict(batch=20000, polarized=True, stats=True, track=True), False),
          ("suzanne 10^3", S.config5_suzanne_grid, dict(batch=6000, n_side=10, polarized=True, stats=True, track=True), False)]
for name, make, kw, chaotic in cases:
    (_, ho, so, to), (_, hg, sg, tg) = T._both(make, **kw)
    if chaotic:
        keep = T._mask_after_first_detection(to) & T._mask_after_first_detection(tg)
        to = np.where(keep[:, :, None], to, 0.0); tg = np.where(keep[:, :, None], tg, 0.0)
    same = np.all(to[:, :, 0] == tg[:, :, 0], axis=1)
    print(f"{name:24s} max |dh| / total = {np.abs(hg - ho).max() / ho.sum():.2e}   photons with identical event codes = {same.mean():.4f}   "
          f"max counter difference = {np.abs(so[:12].astype(np.int64) - sg[:12].astype(np.int64)).max()}")
