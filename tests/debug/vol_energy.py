#!/usr/bin/env python3
"""Deviations of the reference's volume energy test (tests/test_reference_energy.py::test_VolumeForwardTracer)
for every parameter row, on the HIP backend or (argument `oracle`, reduced size) on the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import hephaistos.pipeline as pl
import scenarios as S
import theia, theia.light, theia.material, theia.random, theia.response, theia.trace, theia.units as u
from theia.target import InnerSphereTarget
from theia_amd import _backend
oracle = len(sys.argv) > 1 and sys.argv[1] == "oracle"
if oracle:
    from oracle_backend import OracleBackend
    _backend.install(OracleBackend(threads=8))
rows = [(0.0, 0.005, 0.05, 0.0, False, True, False, 4.5e-4), (0.0, 0.005, 0.05, 0.0, False, False, False, 0.011),
        (0.0, 0.005, 0.05, 0.0, True, True, False, 0.011), (0.05, 0.01, 0.05, 0.0, False, True, False, 1.8e-3),
        (0.05, 0.01, 0.05, 0.0, True, True, False, 2.6e-3), (0.05, 0.01, 0.05, -0.9, False, True, False, 0.012),
        (0.05, 0.01, 0.05, 0.9, True, True, False, 2.8e-4), (0.05, 0.01, 0.05, 0.9, True, False, False, 2.9e-3),
        (0.05, 0.01, np.nan, 0.9, False, True, True, 2.9e-3), (0.0, 0.005, 0.05, 0.0, False, True, True, 5e-4),
        (0.0, 0.005, 0.05, 0.0, True, True, True, 1.2e-3), (0.05, 0.01, 0.05, -0.9, False, True, True, 0.012)]
B, NB = (512 * 1024, 10) if not oracle else (256 * 1024, 2)
nscat = int(os.environ.get("NSCAT", "10"))
for (mu_a, mu_s, ms, g, dd, st, pol, err) in rows:
    model = S.EnergyTestModel(mu_a, mu_s, g)
    store = theia.material.MaterialStore([], media=[model.createMedium()])
    rng = theia.random.PhiloxRNG(key=0xC0FFEE)
    photons = theia.light.UniformWavelengthSource(lambdaRange=(400.0, 400.0))
    pos = (12.0, 15.0, 0.2)
    light = theia.light.SphericalLightSource(position=pos, timeRange=(10.0, 10.0), budget=1e9)
    target = InnerSphereTarget(position=pos, radius=100.0)
    rec = theia.response.HitRecorder(polarized=pol)
    tr = theia.trace.VolumeForwardTracer(B, light, target, photons, rec, rng, medium=store.media["homogenous"], maxTime=float("inf"),
                                         nScattering=nscat, scatterCoefficient=ms, polarized=pol, disableDirectLighting=dd,
                                         disableTargetSampling=not st)
    rng.autoAdvance = tr.nRNGSamples
    pipe = pl.Pipeline(tr.collectStages())
    tot = 0.0
    for k in range(NB):
        pipe.run(k % 2)
        v = rec.queue.view(k % 2); r = v[: v.count]
        d = (1 / 1.33 * u.c) * (r["time"] - 10.0)
        tot += (r["contrib"] * np.exp(mu_a * d)).sum(dtype=np.float64)
    contrib = 1e9 - 1e9 * np.exp(-mu_s * 100.0) if dd else 1e9
    dev = tot / (B * NB) / contrib - 1
    print(f"mu_a={mu_a} mu_s={mu_s} ms={ms} g={g} noDirect={dd} MIS={st} pol={pol}: dev={dev:+.2e} tol={err:.1e} {'ok' if abs(dev) < err else 'FAIL'}", flush=True)
