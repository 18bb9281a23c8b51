import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import scenarios as S
from oracle_backend import OracleBackend
from theia_amd import _backend
np.set_printoptions(linewidth=220, precision=7, suppress=False)

def both(make, **kw):
    out = []
    for be in (OracleBackend(threads=8), None):
        prev = _backend._backend
        if be is not None: _backend._backend = be
        try:
            sc = make(**kw); hist = S.run_batches(sc, 1)[0]
            out.append((hist, sc.tracer.program.read_stats(), sc.tracer.program.read_track(sc.tracer.batchSize)))
        finally:
            _backend._backend = prev
    return out

which = sys.argv[1] if len(sys.argv) > 1 else "vol"
if which == "vol":
    (ho, so, to), (hg, sg, tg) = both(S.config1_volume, batch=20000, stats=True, track=True, disableTargetSampling=True)
else:
    (ho, so, to), (hg, sg, tg) = both(S.config2_sphere, batch=20000, stats=True, track=True)
same = np.all(to[:, :, 0] == tg[:, :, 0], axis=1)
print("code agreement", same.mean(), "hist rel err", np.abs(hg-ho).max()/ho.sum())
print("stats o", so); print("stats g", sg)
err = np.abs(to[:, :, 1:4] - tg[:, :, 1:4]).max(axis=(1, 2))
err[~same] = 0
worst = np.argsort(err)[::-1][:3]
for w in worst:
    print("photon", w, "err", err[w])
    print("oracle\n", to[w]); print("gpu\n", tg[w])
bad = np.where(~same)[0][:3]
for w in bad:
    print("DIFF photon", w); print("oracle codes", to[w,:,0]); print("gpu codes", tg[w,:,0])
    print("oracle\n", to[w]); print("gpu\n", tg[w])
