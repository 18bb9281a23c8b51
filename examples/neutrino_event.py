#!/usr/bin/env python3
"""
Light curve a detector sphere sees from one neutrino-like event: a muon of a few TeV passing by and the cascades it leaves along its
way -- every particle turned into its light source by `theia.cascades.createParamsFromParticle` and traced as one task of a
`PipelineScheduler` over two pipelines (tracks, cascades), the way theia's examples/03_multiple_lightsources.ipynb switches between
light sources. Written against the alias packages of this repository (`import theia`, `import hephaistos` resolve to `theia_amd`);
needs an MI355X and the built library.

    python examples/neutrino_event.py [photons per particle] [cascades]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np

import hephaistos.pipeline as pl
import theia.cascades as tc
import theia.light
import theia.material
import theia.random
import theia.response
import theia.scene
import theia.trace
import theia.units as u
from theia.scene import Transform
from theia.target import SphereTargetGuide
from theia.testing import WaterTestModel


def build(batch: int, key: int = 0xC0FFEE):
    """the detector, and one scene tracer per kind of light source sharing everything else -> (pipelines by name, response, owners).
    `owners` must stay alive as long as the pipelines are used: material store and mesh store own the device memory whose addresses
    (plain integers, as in theia) the scene and the tracers carry."""
    water = WaterTestModel().createMedium()
    store = theia.material.MaterialStore([theia.material.Material("det", None, water, flags="DB")])
    mesh = np.load(os.path.join(ROOT, "theia_amd", "data", "meshes.npz"))
    sphere = theia.scene.Mesh()
    sphere.vertices, sphere.indices = mesh["sphere/vertices"], mesh["sphere/faces"]
    meshes = theia.scene.MeshStore({"sphere": sphere})
    radius, centre = 0.5 * u.m, (0.0, 0.0, 0.0) * u.m
    module = meshes.createInstance("sphere", "det", Transform.TRS(scale=radius, translate=centre), detectorId=1)
    scene = theia.scene.Scene([module], store.material, medium=store.media["water"])
    rng = theia.random.PhiloxRNG(key=key)
    photons = theia.light.UniformWavelengthSource(lambdaRange=(350.0, 600.0) * u.nm)
    response = theia.response.HistogramHitResponse(theia.response.UniformValueResponse(), nBins=100, binSize=4.0 * u.ns,
                                                   normalization=1.0 / batch)
    guide = SphereTargetGuide(position=centre, radius=radius)
    common = dict(maxPathLength=12, targetId=1, targetGuide=guide, scatterCoefficient=0.02 / u.m)
    tracers = {"track": theia.trace.SceneForwardTracer(batch, theia.light.MuonTrackLightSource(), photons, response, rng, scene, **common),
               "cascade": theia.trace.SceneForwardTracer(batch, theia.light.ParticleCascadeLightSource(), photons, response, rng, scene, **common)}
    rng.autoAdvance = max(t.nRNGSamples for t in tracers.values())
    return {name: pl.Pipeline(t.collectStages()) for name, t in tracers.items()}, response, (store, meshes, scene, tracers)


def event(n_cascades: int, seed: int = 7):
    """a 3 TeV muon passing 8 m from the detector and the stochastic losses along its 120 m -> [(pipeline name, parameters)]"""
    gen = np.random.default_rng(seed)
    start, along, length = np.array([-60.0, 8.0, 3.0]) * u.m, np.array([1.0, 0.0, 0.0]), 120.0 * u.m
    particles = [tc.Particle(tc.ParticleType.MU_MINUS, tuple(start), tuple(along), 3.0 * u.TeV, 0.0 * u.ns, length)]
    for _ in range(n_cascades):
        s = gen.random() * length
        kind = tc.ParticleType.E_MINUS if gen.random() < 0.7 else tc.ParticleType.PI_PLUS
        particles.append(tc.Particle(kind, tuple(start + s * along), tuple(along), float(10.0 ** gen.uniform(0.5, 2.5)) * u.GeV, s / u.c))
    tasks = []
    for p in particles:
        source, params, _yield = tc.createParamsFromParticle(p)
        tasks.append(("track" if source is theia.light.MuonTrackLightSource else "cascade", params))
    return tasks


def trace(pipelines, response, tasks):
    """one batch per particle; returns the event's light curve = the sum over its particles (photons per bin)"""
    curves = []
    scheduler = pl.PipelineScheduler(pipelines, processFn=lambda config, batch, args: curves.append(response.result(config).copy()))
    scheduler.schedule(tasks)
    scheduler.wait()
    scheduler.destroy()
    return np.sum(curves, axis=0), curves


if __name__ == "__main__":
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024 * 1024
    n_cascades = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    pipelines, response, owners = build(batch)
    tasks = event(n_cascades)
    trace(pipelines, response, tasks[:2])  # (warm-up: kernels loaded, buffers allocated)
    t0 = time.perf_counter()
    total, curves = trace(pipelines, response, tasks)
    dt = time.perf_counter() - t0
    print(f"{len(tasks)} particles x {batch} photons in {dt * 1e3:.1f} ms ({len(tasks) * batch / dt / 1e9:.2f} G photons/s)")
    t = (np.arange(response.nBins) + 0.5) * response.binSize
    peak = int(np.argmax(total))
    print(f"expected photons on the module: {total.sum():.3g}; peak {total[peak]:.3g} per bin at t = {t[peak]:.0f} ns")
    print(f"the muon alone: {curves[0].sum():.3g}; its cascades: {np.sum(curves[1:]):.3g}")
