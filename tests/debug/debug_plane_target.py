"""dev aid: volume tracer with a disk target, oracle vs HIP, hit by hit"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import hephaistos.pipeline as pl
import theia.light, theia.material, theia.response, theia.trace, theia.target as TT, theia.units as u
from theia.random import PhiloxRNG
from theia.testing import WaterTestModel
from oracle_backend import OracleBackend
from theia_amd import _backend

pol = len(sys.argv) > 1 and sys.argv[1] == "pol"
n = 16 * 1024
out = {}
for name in ("oracle", "hip"):
    be = OracleBackend(threads=8) if name == "oracle" else _backend.HipBackend()
    _backend._backend = be
    water = WaterTestModel().createMedium()
    store = theia.material.MaterialStore([], media=[water])
    rng = PhiloxRNG(key=0xC01DC0FFEE)
    rec = theia.response.HitRecorder(polarized=pol)
    target = TT.DiskTarget(position=(5.0, 2.0, -8.0), radius=4.0, normal=(-0.6, -0.5, 0.62), up=(0.0, 0.0, 1.0))
    tracer = theia.trace.VolumeForwardTracer(
        n, theia.light.SphericalLightSource(position=(-1.0, -7.0, 0.0), budget=1000.0, timeRange=(10.0, 20.0)),
        target, theia.light.UniformWavelengthSource(), rec, rng, medium=store.media["water"], nScattering=6,
        polarized=pol, maxTime=1000.0)
    pl.runPipeline(tracer.collectStages())
    q = rec.queue.view(0)
    o = np.argsort(q["time"][: q.count])
    out[name] = {f: np.array(q[f][: q.count])[o] for f in q.fields}
    print(name, q.count, out[name]["contrib"].sum())
a, b = out["oracle"], out["hip"]
idx = np.clip(np.searchsorted(b["time"], a["time"]), 1, len(b["time"]) - 1)
idx = np.where(np.abs(b["time"][idx - 1] - a["time"]) < np.abs(b["time"][idx] - a["time"]), idx - 1, idx)
bad = ~np.isclose(a["contrib"], b["contrib"][idx], rtol=1e-3) | (np.abs(b["time"][idx] - a["time"]) > 1e-3)
print("bad", bad.sum(), "of", len(bad))
for k in np.nonzero(bad)[0][:12]:
    print("O t=%.4f c=%.5g pos=%s nrm=%s | H t=%.4f c=%.5g pos=%s nrm=%s" % (a["time"][k], a["contrib"][k], a["position"][k], a["normal"][k],
          b["time"][idx[k]], b["contrib"][idx[k]], b["position"][idx[k]], b["normal"][idx[k]]))
