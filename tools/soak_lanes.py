#!/usr/bin/env python3
"""Soak test of batches side by side: random scenes, task lists with a batch size of their own per task, random lanes / share of
the chip / batches in flight, through `PipelineScheduler`. Every task list must account for all of its photons (event counters of
the instrumented kernels, summed over the batches in flight), return finite curves, and -- for one list in eight -- equal the
curves of a plain `Pipeline.run` loop over the same tasks.

    python tools/soak_lanes.py [seconds] [seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from theia_amd import workloads as S  # noqa: E402
from theia_amd.hp import pipeline as pl  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
gen = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
import theia_amd.light as TL  # noqa: E402
import theia_amd.units as u  # noqa: E402


def cherenkov(batch, **kw):
    """config 2's detector lit by one of the Cherenkov-family sources (sampled by the sampler kernel in front of the tracing kernel:
    a pair of sample queues per lane)"""
    kind = kw.pop("family")
    if kind == 0:
        light = TL.MuonTrackLightSource((-6.0, -20.0, 2.5), 0.0, (-6.0, 20.0, 2.5), 133.4, 2.0 * u.TeV)
    elif kind == 1:
        light = TL.ParticleCascadeLightSource(startPosition=(-6.0, -3.0, 2.5), direction=(0.0, 1.0, 0.0), effectiveLength=500.0, a_angular=0.54,
                                              b_angular=3.3, a_long=4.2, b_long=0.57)
    elif kind == 2:
        light = TL.CherenkovLightSource(trackStart=(-6.0, -20.0, 2.5), trackEnd=(-6.0, 20.0, 2.5), startTime=0.0, endTime=133.4)
    else:
        track = TL.ParticleTrack(4)
        track.setVertices([[-6.0, -20.0, 2.5, 0.0], [-6.0, 0.0, 2.5, 66.7], [-6.5, 20.0, 2.5, 133.4]])
        light = TL.CherenkovTrackLightSource(track)
    return S.config2_sphere(batch=batch, light=light, photons=TL.UniformWavelengthSource(lambdaRange=(350.0, 550.0) * u.nm), **kw)


makers = [("config2", S.config2_sphere, 600_000), ("config3", S.config3_cube_torus, 150_000), ("config5", S.config5_suzanne_grid, 40_000),
          ("config1", S.config1_volume, 400_000), ("cherenkov", cherenkov, 400_000)]
t0, lists, batches, photons, compared = time.time(), 0, 0, 0, 0
while time.time() - t0 < seconds:
    name, make, top = makers[gen.integers(len(makers))]
    cap = int(np.exp(gen.uniform(np.log(64), np.log(top))))
    stats = bool(gen.integers(2))
    kw = {"stats": stats, "key": int(gen.integers(1 << 40))}
    if name != "config1":
        kw["polarized"] = bool(gen.integers(2)) if name in ("config2", "cherenkov") else True
    if name == "cherenkov":
        kw["family"] = int(gen.integers(4))
    lanes = int(gen.integers(1, 17))
    div = int(gen.integers(1, lanes + 1))
    depth = int(gen.integers(1, 17))
    n_tasks = int(gen.integers(1, 41))
    sizes = [int(gen.integers(1, cap + 1)) for _ in range(n_tasks)]
    tasks = [{"tracer__batchSize": s} for s in sizes]
    sc = make(batch=cap, **kw)
    got = []
    sched = pl.PipelineScheduler(pl.Pipeline(sc.stages()), processFn=lambda c, b, a: got.append(sc.response.result(c).copy()),
                                 lanes=lanes, gridDivisor=div, inFlight=depth)
    sched.schedule(tasks)
    sched.wait()
    sched.destroy()
    ok = len(got) == n_tasks and all(np.isfinite(h).all() for h in got)
    if stats:
        ok = ok and sc.callback.created == sum(sizes)
    what = f"{name} cap={cap} {kw} lanes={lanes}/{div} inFlight={depth} tasks={n_tasks}"
    if ok and lists % 8 == 0:  # the same tasks one after the other
        ref = make(batch=cap, **kw)
        pipe = pl.Pipeline(ref.stages())
        for k, t in enumerate(tasks):
            pipe.setParams(**t)
            pipe.run(k % 2)
            h = ref.response.result(k % 2)
            tot = max(float(np.abs(h).sum()), 1e-30)
            if np.abs(h - got[k]).max() > 5e-6 * tot:
                print(f"MISMATCH batch {k} (size {sizes[k]}): {float(np.abs(h - got[k]).max() / tot):.2e}  {what}", flush=True)
                ok = False
                break
        compared += 1
    lists += 1
    batches += n_tasks
    photons += sum(sizes)
    if not ok:
        print(f"FAIL {what} created={sc.callback.created if stats else '-'} expected={sum(sizes)}", flush=True)
        sys.exit(1)
    if lists % 20 == 0:
        print(f"{time.time() - t0:6.0f} s  {lists} task lists  {batches} batches  {photons / 1e6:.0f} M photons  ({compared} compared with a serial loop)  last: {what}", flush=True)
print(f"soak ok: {lists} task lists, {batches} batches, {photons / 1e6:.0f} M photons in {time.time() - t0:.0f} s; {compared} lists compared with a serial loop")
