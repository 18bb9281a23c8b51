#!/usr/bin/env python3
"""Micro-benchmark of the BVH walk alone: n rays aimed at the config-2 sphere."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from theia_amd import workloads as S
from theia_amd import _backend, _capi
from theia_amd.scene import Transform
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
be = _backend.get()
m = S.bundled_mesh("sphere")
h = be.scene_create([(m.vertices, m.indices)], [(0, Transform.Scale(5.0).numpy(), 0, 0)])
print(be.scene_info(h))
rng = np.random.default_rng(1)
for name, dist, coherent in [("from 6 m (primary-like)", 6.0, False), ("from 100 m into the cone (shadow-like)", 100.0, False), ("from 100 m, sorted by target point", 100.0, True), ("from 100 m, identical rays per wave", 100.0, "same")][: (1 if "one" in sys.argv else 4)]:
    o = rng.normal(size=(n, 3)); o = o / np.linalg.norm(o, axis=1, keepdims=True) * dist
    tgt = rng.normal(size=(n, 3)); tgt = tgt / np.linalg.norm(tgt, axis=1, keepdims=True) * 5.0 * rng.uniform(0, 1, (n, 1)) ** (1 / 3)
    if coherent == 'same':
        idx = (np.arange(n) // 64) * 64
        o = o[idx]; tgt = tgt[idx]
    elif coherent:
        o[:] = o[0]
        order = np.lexsort((tgt[:, 2], tgt[:, 1], tgt[:, 0])); tgt = tgt[order]
    d = tgt - o; d /= np.linalg.norm(d, axis=1, keepdims=True)
    out = be.scene_query(h, o.astype(np.float32), d.astype(np.float32), 1e4)
    ms = C.c_float(); _capi.check(be._lib.thx_scene_query_ms(C.c_void_p(h), C.byref(ms)))
    import zlib
    print(f"{name:45s} crc={zlib.crc32(out.tobytes()):08x} rays={n} hit={out[:,0].mean():.3f} kernel={ms.value:.3f} ms  -> {n/ms.value/1e6:.2f} Grays/s")
