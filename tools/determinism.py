"""development aid: is a launch reproducible bit for bit? Runs scenarios three times and compares every photon's events."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from theia_amd import workloads as S
import test_reference_energy as E
from theia_amd.trace import EventStatisticCallback


def shells(batch, polarized=False):
    inner = E.medium_model(0.0, 0.01, 0.9, n=1.8).createMedium(name="inner")
    outer = E.medium_model(0.0, 0.04, 0.9, n=1.2).createMedium(name="outer")
    store = E.TM.MaterialStore([E.TM.Material("det", outer, None, flags="DB"), E.TM.Material("inner", inner, outer, flags="TR")])
    meshes = E.TS.MeshStore({"sphere": E.S.bundled_mesh("sphere")})
    inst = [meshes.createInstance("sphere", name, E.TS.Transform.TRS(scale=r, translate=E.OFF_CENTRE)) for name, r in (("inner", 65.0 * E.u.m), ("det", 100.0 * E.u.m))]
    scene = E.TS.Scene(inst, store.material)
    rng = E.PhiloxRNG(key=0xC0FFEE)
    response = E.TR.HistogramHitResponse(E.TR.UniformValueResponse(), nBins=400, binSize=100.0 * E.u.ns)
    tracer = E.TT.SceneForwardTracer(batch, E.point_source((2.0, 5.0, 0.0) * E.u.m, 1e9), E.TL.ConstWavelengthSource(E.LAMBDA), response, rng, scene,
                                     maxPathLength=50, scatterCoefficient=0.03, sourceMedium=store.media["inner"], maxTime=float("inf"),
                                     polarized=polarized, callback=EventStatisticCallback(), trackEvents=True)
    return S.Scenario(name="shells", tracer=tracer, response=response, rng=rng, scene=scene, callback=tracer.callback, store=store, meshes=meshes)


def run(make, **kw):
    sc = make(**kw)
    h = S.run_batches(sc, 1)[0]
    return h, sc.tracer.program.read_stats(), sc.tracer.program.read_track(sc.tracer.batchSize)


cases = [("config2", S.config2_sphere, dict(batch=200000, stats=True, track=True)),
         ("config3", S.config3_cube_torus, dict(batch=100000, polarized=True, stats=True, track=True)),
         ("shells", shells, dict(batch=int(os.environ.get("BATCH", 200000))))]
for name, make, kw in cases:
    a = run(make, **kw)
    for rep in range(2):
        x = run(make, **kw)
        ta, tx = a[2].view(np.uint32), x[2].view(np.uint32)
        bad = np.nonzero((ta != tx).any(axis=(1, 2)))[0]
        print(name, "tracks identical:", bad.size == 0, "photons differing:", bad.size, "stats equal:", np.array_equal(a[1][:20], x[1][:20]), flush=True)
        if bad.size:
            k = int(bad[0])
            ev = int(np.nonzero((ta[k] != tx[k]).any(axis=1))[0][0])
            print("  first differing photon", k, "event", ev)
            print("   run 0:", a[2][k, max(ev - 1, 0):ev + 2, :12])
            print("   run n:", x[2][k, max(ev - 1, 0):ev + 2, :12])
