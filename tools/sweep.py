#!/usr/bin/env python3
"""Kernel time of the scene workloads under environment overrides (development aid):
   tools/sweep.py <photons> <workload,...> [VAR=v1,v2,... ...]   -- full cross product of the VAR lists.
   (THEIA_AMD_LIB cannot be one of them: the library is loaded once per process -- A/B two builds with two invocations.)"""
import itertools, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from theia_amd import workloads as S
from theia_amd.hp.pipeline import Pipeline

MAKE = {"config2": S.config2_sphere, "config3": S.config3_cube_torus, "config5": S.config5_suzanne_grid,
        "mid": lambda batch: S.config2_sphere(batch, subdivide=3), "bigmesh": lambda batch: S.config2_sphere(batch, subdivide=6)}
n = int(sys.argv[1])
WARM = int(os.environ.get("SWEEP_WARMUP", "10"))
names = sys.argv[2].split(",")
axes = [(a.split("=")[0], a.split("=")[1].split(",")) for a in sys.argv[3:]]
if any(k == "THEIA_AMD_LIB" for k, _ in axes):
    sys.exit("THEIA_AMD_LIB is read once, when the library is loaded: set it in the environment of one invocation per build")
for name in names:
    for combo in itertools.product(*[v for _, v in axes]) if axes else [()]:
        for (k, _), v in zip(axes, combo):
            if v == "-":
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        tag = " ".join(f"{k}={v}" for (k, _), v in zip(axes, combo))
        try:
            sc = MAKE[name](batch=n)
            pipe = Pipeline(sc.stages())
            ms = []
            for k in range(WARM + 5):  # the clocks of a fresh GPU take a second of work to settle
                pipe.run(k % 2)
                ms.append(sc.tracer.program.last_kernel_ms(k % 2))
            info = sc.tracer.program.launch_info(n)
        except Exception as exc:  # a combination the library refuses (e.g. a pool that does not fit)
            print(f"{name:8s} {tag:40s} refused: {exc}", flush=True)
            continue
        print(f"{name:8s} {tag:40s} kernel ms = {np.median(ms[-5:]):8.3f}   block={info['block']} lds={info['lds_bytes']}", flush=True)
        del sc, pipe
