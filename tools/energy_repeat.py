"""development aid: the multi-media energy scenario of tests/test_reference_energy.py, repeated (is the result reproducible?)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_reference_energy as E
from theia_amd import _backend
class _Fix:  # what the `backend` fixture does for "hip"
    pass
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    budget, batch, tasks = 1e9, int(os.environ.get("BATCH", 2 * E.Mi)), int(os.environ.get("TASKS", 10))
    inner = E.medium_model(0.0, 0.01, 0.9, n=1.8).createMedium(name="inner")
    outer = E.medium_model(0.0, 0.04, 0.9, n=1.2).createMedium(name="outer")
    store = E.TM.MaterialStore([E.TM.Material("det", outer, None, flags="DB"), E.TM.Material("inner", inner, outer, flags="TR")])
    meshes = E.TS.MeshStore({"sphere": E.S.bundled_mesh("sphere")})
    shells = [meshes.createInstance("sphere", name, E.TS.Transform.TRS(scale=r, translate=E.OFF_CENTRE)) for name, r in (("inner", 65.0 * E.u.m), ("det", 100.0 * E.u.m))]
    scene = E.TS.Scene(shells, store.material)
    rng = E.PhiloxRNG(key=0xC0FFEE)
    response = E.TR.HistogramHitResponse(E.TR.UniformValueResponse(), nBins=400, binSize=100.0 * E.u.ns)
    tracer = E.TT.SceneForwardTracer(batch, E.point_source((2.0, 5.0, 0.0) * E.u.m, budget), E.TL.ConstWavelengthSource(E.LAMBDA), response, rng,
                                     scene, maxPathLength=50, scatterCoefficient=0.03, sourceMedium=store.media["inner"], maxTime=float("inf"), polarized=False)
    rng.autoAdvance = tracer.nRNGSamples
    curve = E.mean_curve(tracer, response, tasks)
    print(f"{curve.sum() / budget - 1.0:+.9f}", flush=True)
