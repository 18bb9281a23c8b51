#!/usr/bin/env python3
"""How fast would BASELINE config 5 be if its 1000 instanced suzannes were ONE world-space mesh (968 000 triangles, no TLAS, no instance
entries)? An upper bound for what de-instancing inside the library could win -- the merged mesh is intersected in world space, which is
not the arithmetic of the instanced scene (hits differ in the last bits). Kernel ms, median of 5 launches after warm-up; light curves compared.
    tools/flatten_probe.py [photons]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from theia_amd import workloads as S
from theia_amd import units as u
from theia_amd.hp import Mesh
from theia_amd.hp.pipeline import Pipeline
from theia_amd.light import SphericalLightSource, UniformWavelengthSource
from theia_amd.material import BK7Model, Material, MaterialStore
from theia_amd.random import PhiloxRNG
from theia_amd.response import HistogramHitResponse, UniformValueResponse
from theia_amd.scene import MeshStore, Scene, Transform
from theia_amd.target import SphereTargetGuide
from theia_amd.testing import WaterTestModel
from theia_amd.trace import SceneForwardTracer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000


def flat(batch, n_side=10, key=42):
    water = WaterTestModel().createMedium(400.0 * u.nm, 800.0 * u.nm, name="water")
    glass = BK7Model().createMedium(400.0 * u.nm, 800.0 * u.nm, name="glass")
    store = MaterialStore([Material("glass", glass, water, flags="TR"), Material("det", None, water, flags="BD")])
    monkey = S.bundled_mesh("suzanne")
    gen = np.random.default_rng(0xC0FFEE)
    pitch, verts, faces = 4.0, [], []
    for ix in range(n_side):
        for iy in range(n_side):
            for iz in range(n_side):
                t = Transform.TRS(scale=1.0, rotate=(0.0, 0.0, 1.0, gen.uniform(0.0, 360.0)), translate=(ix * pitch, iy * pitch, iz * pitch)).numpy()
                v = np.array(monkey.vertices, dtype=np.float64)
                w = np.empty_like(v)
                w[:, :3] = v[:, :3] @ t[:, :3].T + t[:, 3]
                w[:, 3:] = v[:, 3:] @ t[:, :3].T  # (rotation + unit scale: normals transform like directions)
                faces.append(np.asarray(monkey.indices) + len(verts) * len(v))
                verts.append(w)
    merged = Mesh()
    merged.vertices, merged.indices = np.concatenate(verts).astype(np.float32), np.concatenate(faces).astype(np.uint32)
    meshes = MeshStore({"monkeys": merged, "sphere": S.bundled_mesh("sphere")})
    c = 0.5 * (n_side - 1) * pitch
    inst = [meshes.createInstance("monkeys", "glass", Transform.TRS(scale=1.0)),
            meshes.createInstance("sphere", "det", Transform.TRS(scale=1.0, translate=(c, c, c)), detectorId=1)]
    scene = Scene(inst, store.material, medium=store.media["water"])
    response = HistogramHitResponse(UniformValueResponse(), nBins=100, binSize=5.0 * u.ns)
    rng = PhiloxRNG(key=key)
    tracer = SceneForwardTracer(batch, SphericalLightSource(position=(c - pitch,) * 3, timeRange=(0.0, 10.0) * u.ns, budget=1.0e5),
                                UniformWavelengthSource(lambdaRange=(400.0 * u.nm, 800.0 * u.nm)), response, rng, scene, maxPathLength=20, targetId=1,
                                targetGuide=SphereTargetGuide(position=(c, c, c), radius=1.0 * u.m), scatterCoefficient=0.05 / u.m, polarized=True)
    rng.autoAdvance = tracer.nRNGSamples
    return S.Scenario(name="config5 flattened", store=store, meshes=meshes, scene=scene, response=response, rng=rng, tracer=tracer)


curves = {}
for name, make in (("instanced (BASELINE config 5)", lambda: S.config5_suzanne_grid(batch=B)), ("one world-space mesh", lambda: flat(B))):
    sc = make()
    pipe = Pipeline(sc.stages())
    ms = []
    for k in range(9):
        pipe.run(k % 2)
        ms.append(sc.tracer.program.last_kernel_ms(k % 2))
    curves[name] = np.array(sc.response.result(0), dtype=np.float64)
    print(f"{name:32s} kernel ms = {np.median(ms[-5:]):8.3f}   scene {sc.scene.info}", flush=True)
a, b = curves.values()
print(f"light curves: totals {a.sum():.6g} / {b.sum():.6g}, max |difference| / total = {np.abs(a - b).max() / a.sum():.2e}")
