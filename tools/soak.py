#!/usr/bin/env python3
"""Soak test of the staged scene kernel's lock-free scheduler: random workloads, batch sizes and
flags for a given number of seconds; every launch must account for all of its photons (event
counters) and return a finite light curve (a tripped watchdog makes the read fail)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from theia_amd import workloads as S

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
gen = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
makers = [("config2", S.config2_sphere, 2_000_000), ("config3", S.config3_cube_torus, 400_000), ("config5", S.config5_suzanne_grid, 60_000)]
t0, launches, photons = time.time(), 0, 0
while time.time() - t0 < seconds:
    name, make, top = makers[gen.integers(len(makers))]
    n = int(np.exp(gen.uniform(0.0, np.log(top))))
    kw = {"polarized": bool(gen.integers(2))}
    if name == "config2":
        kw["guide"] = bool(gen.integers(2))
        kw["maxPathLength"] = int(gen.integers(1, 30))
    # scheduler knobs: how eagerly walk batches set their last rays aside, how much of the traversal stack lives in LDS
    knobs = {}
    if gen.integers(2):
        knobs = {"THX_PARK_BELOW": str(int(gen.integers(0, 66))), "THX_PARK_ROUNDS": str(int(gen.integers(1, 5))),
                 "THX_STACK_LDS": str(int(gen.integers(2, 33)))}
    # ... and which compiled form of the walk runs: pair records in LDS or not, 16- or 32-bit stack entries
    if gen.integers(3) == 0:
        knobs["THX_LDS_NODES"] = str(int(gen.integers(2)))
    if gen.integers(4) == 0:
        knobs["THX_STACK32"] = "1"
    if gen.integers(3) == 0:  # ... 4-wide records (what large scenes get), forced onto these small ones
        knobs["THX_WIDE"] = "1"
    for k in ("THX_PARK_BELOW", "THX_PARK_ROUNDS", "THX_STACK_LDS", "THX_LDS_NODES", "THX_STACK32", "THX_WIDE"):
        os.environ.pop(k, None)
    os.environ.update(knobs)
    sc = make(batch=n, stats=True, key=int(gen.integers(1 << 40)), **kw)
    reps = int(gen.integers(1, 4))
    hists = S.run_batches(sc, reps)
    cb = sc.callback
    ok = cb.created == reps * n and all(np.isfinite(h).all() for h in hists)
    launches += reps; photons += reps * n
    if not ok:
        print(f"FAIL {name} n={n} {kw} {knobs} created={cb.created} expected={reps * n}", flush=True)
        sys.exit(1)
    if launches % 50 < reps:
        print(f"{time.time() - t0:6.0f} s  {launches} launches  {photons / 1e6:.0f} M photons  last: {name} n={n} {kw} {knobs}", flush=True)
print(f"soak ok: {launches} launches, {photons / 1e6:.0f} M photons in {time.time() - t0:.0f} s")
