"""development aid: two production-kernel launches of the shells scenario, bin by bin"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_reference_energy as E
from theia_amd import workloads as S
def one(batch, stats=False):
    inner = E.medium_model(0.0, 0.01, 0.9, n=1.8).createMedium(name="inner")
    outer = E.medium_model(0.0, 0.04, 0.9, n=1.2).createMedium(name="outer")
    store = E.TM.MaterialStore([E.TM.Material("det", outer, None, flags="DB"), E.TM.Material("inner", inner, outer, flags="TR")])
    meshes = E.TS.MeshStore({"sphere": E.S.bundled_mesh("sphere")})
    inst = [meshes.createInstance("sphere", name, E.TS.Transform.TRS(scale=r, translate=E.OFF_CENTRE)) for name, r in (("inner", 65.0 * E.u.m), ("det", 100.0 * E.u.m))]
    scene = E.TS.Scene(inst, store.material)
    rng = E.PhiloxRNG(key=0xC0FFEE)
    response = E.TR.HistogramHitResponse(E.TR.UniformValueResponse(), nBins=400, binSize=100.0 * E.u.ns)
    kw = dict(callback=E.TT.EventStatisticCallback()) if stats else {}
    tracer = E.TT.SceneForwardTracer(batch, E.point_source((2.0, 5.0, 0.0) * E.u.m, 1e9), E.TL.ConstWavelengthSource(E.LAMBDA), response, rng, scene,
                                     maxPathLength=50, scatterCoefficient=0.03, sourceMedium=store.media["inner"], maxTime=float("inf"), polarized=False, **kw)
    sc = S.Scenario(name="shells", tracer=tracer, response=response, rng=rng, scene=scene, store=store, meshes=meshes)
    return S.run_batches(sc, 1)[0].astype(np.float64)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ref = one(n, stats=True)
print("instrumented total", ref.sum())
for k in range(3):
    h = one(n)
    d = h - ref
    idx = np.argsort(-np.abs(d))[:5]
    print(f"production run {k}: total {h.sum():.6e} diff {d.sum():+.4e}; bins differing by > 1e-5 of the total: {(np.abs(d) > 1e-5 * ref.sum()).sum()}; largest:",
          [(int(i), float(d[i] / (1e9 / n))) for i in idx], flush=True)
