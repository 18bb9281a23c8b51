#!/usr/bin/env python3
"""Config 2 with the detector sphere subdivided to 1280 * 4^k triangles: scene build time, kernel
time and throughput as the BVH outgrows L1 / L2 (k = 5: 1.3 M triangles, beyond the staged kernel's
20-bit primitive ids, so the lane kernel runs)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from theia_amd import workloads as S

prod = "--prod" in sys.argv  # production kernel variant (no counters): kernel time only
argv = [a for a in sys.argv if a != "--prod"]
n = int(argv[1]) if len(argv) > 1 else 1_000_000
levels = [int(a) for a in argv[2:]] or [0, 2, 3, 4, 5]
for k in levels:
    if prod:
        sc = S.config2_sphere(batch=n, subdivide=k)
        ms = []
        for i in range(12):
            S.run_batches(sc, 1)
            ms.append(sc.tracer.program.last_kernel_ms(0))
        li = sc.tracer.program.launch_info(n)
        print(f"k={k} tris={1280 * 4**k:8d} block={li['block']:4d} pool-lds={li['lds_bytes']} kernel={np.median(ms[-5:]):8.3f} ms (production variant)", flush=True)
        continue
    t0 = time.time()
    sc = S.config2_sphere(batch=n, stats=True, subdivide=k)
    info = sc.scene.info if hasattr(sc.scene, "info") else {}
    t1 = time.time()
    S.run_batches(sc, 1)
    sc.callback.reset()
    S.run_batches(sc, 2)
    ms = sc.tracer.program.last_kernel_ms(0)
    steps = sc.callback.traversal["steps"] / 2
    tr = sc.callback.traversal
    li = sc.tracer.program.launch_info(n)
    print(f"k={k} tris={1280 * 4**k:8d} setup={t1 - t0:6.2f}s block={li['block']:4d} kernel={ms:8.3f} ms "
          f"{steps / ms / 1e3:8.1f} Mstep/s nodes/query={tr['nodes'] / tr['queries']:.1f} tris/query={tr['triangles'] / tr['queries']:.2f}", flush=True)
