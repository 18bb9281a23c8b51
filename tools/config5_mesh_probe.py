#!/usr/bin/env python3
"""Where does config 5's time go -- the TLAS and the instance entries, or the walks inside the instances? The same 10 x 10 x 10 grid
with the suzanne (968 triangles) replaced by a cube (12) and by the icosphere (1280): kernel ms of the production variant and, from an
instrumented run, photon-steps and traversal units per step -> ns per photon-step.    tools/config5_mesh_probe.py [photons]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from theia_amd import workloads as S
from theia_amd.hp.pipeline import Pipeline

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
real = S.bundled_mesh
for name in ("suzanne", "cube", "sphere"):
    S.bundled_mesh = lambda which, _name=name: real(_name if which == "suzanne" else which)
    sc = S.config5_suzanne_grid(batch=B)
    pipe = Pipeline(sc.stages())
    ms = []
    for k in range(8):
        pipe.run(k % 2)
        ms.append(sc.tracer.program.last_kernel_ms(k % 2))
    ms = float(np.median(ms[-5:]))
    st = S.config5_suzanne_grid(batch=200_000, stats=True)
    S.run_batches(st, 1)
    s = st.tracer.program.read_stats().astype(np.float64)
    steps = s[19] / 200_000
    print(f"{name:8s} kernel {ms:8.3f} ms for {B} photons; {steps:5.2f} steps per photon -> {ms * 1e6 / (B * steps):6.3f} ns per photon-step; per step: "
          f"queries {s[12] / s[19]:.2f} box tests {s[13] / s[19]:.1f} triangles {s[14] / s[19]:.2f} instance entries {s[15] / s[19]:.2f}", flush=True)
S.bundled_mesh = real
