#!/usr/bin/env python3
"""Event counters of the beam-splitter scene of tests/test_reference_tracers.py::test_tracer_reflection."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import hephaistos.pipeline as pl
import scenarios as S
import theia, theia.light, theia.material, theia.random, theia.response, theia.scene, theia.trace, theia.units as u
from theia.scene import Transform
from theia_amd import _backend
if len(sys.argv) > 1 and sys.argv[1] == "oracle":
    from oracle_backend import OracleBackend
    _backend.install(OracleBackend(threads=8))
N = 32 * 256
flag = "TR"
glass = theia.material.BK7Model().createMedium()
mat = theia.material.Material("mat", glass, None, flags=(flag, "T"))
det = theia.material.Material("det", glass, None, flags="DB")
matStore = theia.material.MaterialStore([mat, det])
store = theia.scene.MeshStore({"cube": S.bundled_mesh("cube")})
splitter = store.createInstance("cube", "mat", Transform.TRS(scale=(0.1, 5.0, 5.0), rotate=(0.0, 0.0, 1.0, 45.0), translate=(0.141, 0.0, 0.0)))
ref_det = store.createInstance("cube", "det", Transform.TRS(scale=(50.0, 0.5, 50.0), translate=(0.0, -50.0, 0.0)), detectorId=1)
trans_det = store.createInstance("cube", "det", Transform.TRS(scale=(0.5, 50.0, 50.0), translate=(50.0, 0.0, 0.0)), detectorId=2)
scene = theia.scene.Scene([splitter, trans_det, ref_det], matStore.material)
rng = theia.random.PhiloxRNG(key=0xC01DC0FFEE)
photons = theia.light.UniformWavelengthSource(lambdaRange=(500.0, 500.0) * u.nm)
source = theia.light.PencilLightSource(position=(-20.0, 0.0, 0.0) * u.m, direction=(1.0, 0.0, 0.0), timeRange=(0.0, 0.0), budget=1.0)
recorder = theia.response.HitRecorder()
stats = theia.trace.EventStatisticCallback()
tracer = theia.trace.SceneForwardTracer(N, source, photons, recorder, rng, scene=scene, targetId=1, maxPathLength=4, callback=stats)
pipeline = pl.Pipeline(tracer.collectStages())
pipeline.run(0)
print("hits", recorder.queue.view(0).count)
print(stats)
print(stats.traversal)
