#!/usr/bin/env python3
"""Quick kernel timing of the BASELINE scenarios on the GPU (development aid)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from theia_amd import workloads as S
from theia_amd import _backend


def run(name, make, batch, reps=5, **kw):
    # steps per photon from an instrumented run
    sc = make(batch=min(batch, 200_000), stats=True, **kw)
    S.run_batches(sc, 1)
    st = sc.callback.traversal
    per = {k: v / sc.tracer.batchSize for k, v in st.items()}
    sc = make(batch=batch, **kw)
    hists = S.run_batches(sc, 2)  # warm-up
    be = _backend.get()
    from theia_amd.hp.pipeline import Pipeline

    pipe = Pipeline(sc.stages())
    ms = []
    t0 = time.time()
    for k in range(reps):
        pipe.run(k % 2)
        ms.append(sc.tracer.program.last_kernel_ms(k % 2))
    wall = (time.time() - t0) / reps * 1e3
    steps = per["steps"] * batch
    info = sc.tracer.program.launch_info(batch)
    bytes_step = 32 * per["nodes"] / per["steps"] + 48 * per["triangles"] / per["steps"] + 112 * per["instances"] / per["steps"]
    m = float(np.median(ms))
    print(f"{name:28s} batch={batch:>9d} kernel={m:8.3f} ms wall={wall:8.3f} ms steps/photon={per['steps']:.2f} "
          f"Msteps/s={steps / m / 1e3:9.1f} bvhB/step={bytes_step:7.1f} algGB/s={bytes_step * steps / m / 1e6:8.1f} grid={info}")
    return m


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    only = sys.argv[2] if len(sys.argv) > 2 else ""
    if only == "config2":
        run("config2_sphere", S.config2_sphere, n)
        sys.exit(0)
    if only == "config3":
        run("config3_cube_torus_pol", S.config3_cube_torus, n)
        sys.exit(0)
    if only == "config5":
        run("config5_suzanne_grid_pol", S.config5_suzanne_grid, n)
        sys.exit(0)
    print(_backend.get().device_name(), _backend.get().device_info())
    run("config1_volume", S.config1_volume, n)
    run("config1_volume_pol", S.config1_volume, n, polarized=True)
    run("config2_sphere", S.config2_sphere, n)
    run("config2_sphere_pol", S.config2_sphere, n, polarized=True)
    run("config3_cube_torus_pol", S.config3_cube_torus, n)
    run("config3_cube_torus_unpol", S.config3_cube_torus, n, polarized=False)
