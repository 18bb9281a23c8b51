#!/usr/bin/env python3
"""Is the PRODUCTION kernel variant reproducible from launch to launch? (profiles/r02_park.txt: built with hipcc's SDWA operand
folding, one form of the walk made it trace ~1 % of the two-shell scene's photons differently from launch to launch.) Six
launches of the same photons per scene; totals and curves are compared (the LDS / slab sums differ in their last bits from launch
to launch by construction, a photon going elsewhere is >= 1e-5 of a 10^5-photon curve).

    THEIA_AMD_LIB=theia_amd/lib_pp/libtheia_hip.so python tools/sdwa_check.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import determinism_scenes as D  # noqa: E402
from theia_amd import workloads as S  # noqa: E402

worst = 0.0
for name, make, kw in D.PRODUCTION_CASES:
    curves = []
    for rep in range(6):
        sc = make(**kw)
        curves.append(S.run_batches(sc, 1)[0].astype(np.float64))
        del sc
    c = np.array(curves)
    tot = c.sum(axis=1)
    dev = float(np.abs(c - c[0]).max() / tot[0])
    worst = max(worst, dev)
    print(f"{name:10s} totals {tot.min():.9e} .. {tot.max():.9e}  spread {float(tot.max() / tot.min() - 1):.2e}  max |curve - first| / total {dev:.2e}", flush=True)
print("library:", os.environ.get("THEIA_AMD_LIB", "default"), "-> worst", f"{worst:.2e}", "REPRODUCIBLE" if worst < 2e-6 else "NOT REPRODUCIBLE")
sys.exit(0 if worst < 2e-6 else 1)
