#!/usr/bin/env python3
"""
Light curve of a point source next to a detector sphere in sea water -- the flow of theia's
examples/02_scene_tracing.ipynb, written against the alias packages of this repository
(`import theia`, `import hephaistos` resolve to `theia_amd`). Needs an MI355X and the built
library (`python -c "import __graft_entry__ as g; g.build()"`).

    python examples/quickstart_scene.py [photons per batch] [batches]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np

import hephaistos.pipeline as pl
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

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4 * 1024 * 1024
n_batches = int(sys.argv[2]) if len(sys.argv) > 2 else 8

# media and materials: one blob on the device, addressed like in the reference
water = WaterTestModel().createMedium()
detector = theia.material.Material("det", None, water, flags="DB")  # detect + black body
store = theia.material.MaterialStore([detector])

# geometry: the icosphere this repository bundles for its tests (any STL / PLY works: theia_amd.meshio)
mesh = np.load(os.path.join(ROOT, "theia_amd", "data", "meshes.npz"))
sphere = theia.scene.Mesh()
sphere.vertices, sphere.indices = mesh["sphere/vertices"], mesh["sphere/faces"]
meshes = theia.scene.MeshStore({"sphere": sphere})
radius, centre = 0.5 * u.m, (0.0, 0.0, 0.0) * u.m
instance = meshes.createInstance("sphere", "det", Transform.TRS(scale=radius, translate=centre), detectorId=1)
scene = theia.scene.Scene([instance], store.material, medium=store.media["water"])

# components of the tracer
rng = theia.random.PhiloxRNG(key=0xC0FFEE)
photons = theia.light.UniformWavelengthSource(lambdaRange=(400.0, 700.0) * u.nm)
light = theia.light.SphericalLightSource(position=(5.0, 0.0, 0.0) * u.m, timeRange=(0.0, 1.0) * u.ns, budget=1.0e9)
response = theia.response.HistogramHitResponse(theia.response.UniformValueResponse(), nBins=100, binSize=2.0 * u.ns,
                                               normalization=1.0 / batch)
guide = SphereTargetGuide(position=centre, radius=radius)  # next event estimation towards the detector
tracer = theia.trace.SceneForwardTracer(batch, light, photons, response, rng, scene, maxPathLength=20, targetId=1,
                                        targetGuide=guide, scatterCoefficient=0.05 / u.m)
rng.autoAdvance = tracer.nRNGSamples

curves = []
pipeline = pl.Pipeline(tracer.collectStages())
scheduler = pl.PipelineScheduler(pipeline, processFn=lambda cfg, k, args: curves.append(response.result(cfg).copy()))
t0 = time.time()
scheduler.schedule([{} for _ in range(n_batches)])
scheduler.wait()
scheduler.destroy()
dt = time.time() - t0

curve = np.mean(curves, 0)
print(f"{n_batches} batches of {batch} photons in {dt:.2f} s ({n_batches * batch / dt / 1e6:.0f} M photons/s incl. host)")
print(f"detected share of the budget: {curve.sum() / 1.0e9:.3e}")
peak = int(np.argmax(curve))
print(f"light curve peaks in bin {peak} ({peak * 2.0:.0f}-{peak * 2.0 + 2.0:.0f} ns); first bins:", np.array2string(curve[:12], precision=3))
