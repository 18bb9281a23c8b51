"""
Workload builders: the BASELINE.json configurations expressed through the public
(theia-compatible) API of theia_amd. bench.py, the smoke test and the test suite build their
scenes here (`tests/scenarios.py` re-exports this module). Whatever backend is installed in
`theia_amd._backend` executes them -- the HIP library in production, the oracle stand-in in
CPU tests.
"""
from __future__ import annotations

import os

import numpy as np

from . import units as u
from .light import ConstWavelengthSource, SphericalLightSource, UniformWavelengthSource
from .material import (BK7Model, DispersionFreeMedium, HenyeyGreensteinPhaseFunction,
                                KokhanovskyOceanWaterPhaseMatrix, Material, MaterialStore, MediumModel)
from .random import PhiloxRNG
from .response import HistogramHitResponse, UniformValueResponse
from .scene import MeshStore, Scene, Transform
from .target import SphereTarget, SphereTargetGuide
from .testing import WaterTestModel
from .trace import EmptyEventCallback, EventStatisticCallback, SceneForwardTracer, VolumeForwardTracer
from .hp import Mesh

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def bundled_mesh(name: str) -> Mesh:
    """Mesh arrays converted from the reference's assets (theia_amd/data/meshes.npz, made by tools/gen_golden.py)."""
    data = np.load(os.path.join(_DATA, "meshes.npz"))
    m = Mesh()
    m.vertices = data[f"{name}/vertices"]
    m.indices = data[f"{name}/faces"]
    return m


def subdivided_sphere(levels: int) -> Mesh:
    """The bundled icosphere (1280 triangles) with every triangle split in four `levels` times and the
    new vertices pushed onto the unit sphere: 1280 * 4^levels triangles (stress meshes)."""
    base = bundled_mesh("sphere")
    v = np.asarray(base.vertices, dtype=np.float64)[:, :3]
    f = np.asarray(base.indices, dtype=np.int64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    for _ in range(levels):
        edges = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
        edges.sort(axis=1)
        uniq, inv = np.unique(edges, axis=0, return_inverse=True)
        mid = v[uniq[:, 0]] + v[uniq[:, 1]]
        mid /= np.linalg.norm(mid, axis=1, keepdims=True)
        m = len(v) + inv.reshape(3, -1)  # midpoints of edges 01, 12, 20 per face
        v = np.concatenate([v, mid])
        f = np.concatenate([np.stack([f[:, 0], m[0], m[2]], 1), np.stack([f[:, 1], m[1], m[0]], 1),
                            np.stack([f[:, 2], m[2], m[1]], 1), np.stack([m[0], m[1], m[2]], 1)])
    out = Mesh()
    out.vertices = np.concatenate([v, v], axis=1).astype(np.float32)  # position + normal
    out.indices = f.astype(np.uint32)
    return out


class HomogeneousModel(DispersionFreeMedium, HenyeyGreensteinPhaseFunction, MediumModel):
    """examples/01_volume_tracing.ipynb cell 5"""

    ModelName = "homogenous"

    def __init__(self, *, n, mu_a, mu_s, g) -> None:
        DispersionFreeMedium.__init__(self, n=n, ng=n, mu_a=mu_a, mu_s=mu_s)
        HenyeyGreensteinPhaseFunction.__init__(self, g)


class EnergyTestModel(DispersionFreeMedium, HenyeyGreensteinPhaseFunction, KokhanovskyOceanWaterPhaseMatrix, MediumModel):
    """tests/test_trace_energy.py:31-45 of the reference"""

    ModelName = "homogenous"

    def __init__(self, a, s, g, *, n=1.33, ng=1.33) -> None:
        DispersionFreeMedium.__init__(self, n=n, ng=ng, mu_a=a, mu_s=s)
        HenyeyGreensteinPhaseFunction.__init__(self, g)
        KokhanovskyOceanWaterPhaseMatrix.__init__(self, p90=0.66, theta0=0.25, alpha=4.0, xi=25.6)


class Scenario:
    """Holds every object of one configuration (keeps device memory alive)."""

    def __init__(self, **kw) -> None:
        self.__dict__.update(kw)

    def stages(self):
        return self.tracer.collectStages()


def config1_volume(batch: int = 100_000, *, polarized: bool = False, stats: bool = False, track: bool = False,
                   disableTargetSampling: bool = False, disableDirectLighting: bool = False, key: int = 42, rng=None) -> Scenario:
    """BASELINE config 1 = example 01: VolumeForwardTracer in a homogeneous HG medium."""
    model = HomogeneousModel(n=1.33, mu_a=0.05 / u.m, mu_s=0.02 / u.m, g=0.2)
    medium = model.createMedium()
    store = MaterialStore([], media=[medium])
    wavelength = ConstWavelengthSource(500.0 * u.nm)
    light = SphericalLightSource(position=(1.0, 0.0, 0.0) * u.m, timeRange=(0.0, 10.0) * u.ns, budget=1.0e5)
    target = SphereTarget(position=(-1.0, 0.0, 0.0) * u.m, radius=50.0 * u.cm)
    response = HistogramHitResponse(UniformValueResponse(), nBins=100, binSize=5.0 * u.ns)
    rng = PhiloxRNG(key=key) if rng is None else rng
    callback = EventStatisticCallback() if stats else EmptyEventCallback()
    tracer = VolumeForwardTracer(batch, light, target, wavelength, response, rng, scatterCoefficient=0.05 / u.m,
                                 nScattering=10, medium=store.media["homogenous"], polarized=polarized, callback=callback,
                                 disableTargetSampling=disableTargetSampling, disableDirectLighting=disableDirectLighting,
                                 trackEvents=track)
    if isinstance(rng, PhiloxRNG):
        rng.autoAdvance = tracer.nRNGSamples
    return Scenario(name="config1_volume", store=store, light=light, target=target, wavelength=wavelength,
                    response=response, rng=rng, tracer=tracer, callback=callback, medium=medium)


def config2_sphere(batch: int = 1 << 20, *, polarized: bool = False, stats: bool = False, track: bool = False,
                   guide: bool = True, mu_a: float = 0.05, mu_s: float = 0.005, g: float = 0.5,
                   maxPathLength: int = 10, key: int = 0xC0FFEE, subdivide: int = 0, custom_guide=None, rng=None, response=None,
                   light=None, photons=None) -> Scenario:
    """BASELINE config 2 = sphere-in-medium (reference tests/test_trace_energy.py:187-337,
    Crosscheck geometry): detector sphere r=5 m at the origin, light 6 m away."""
    model = EnergyTestModel(mu_a, mu_s, g)
    medium = model.createMedium()
    material = Material("det", None, medium, flags="DB")
    store = MaterialStore([material])
    meshes = MeshStore({"sphere": subdivided_sphere(subdivide) if subdivide else bundled_mesh("sphere")})
    radius = 5.0 * u.m
    trafo = Transform.TRS(scale=radius, translate=(0.0, 0.0, 0.0))
    target = meshes.createInstance("sphere", "det", trafo, detectorId=0)
    scene = Scene([target], store.material, medium=store.media["homogenous"])
    lam = 400.0 * u.nm
    if photons is None:
        photons = UniformWavelengthSource(lambdaRange=(lam, lam))
    if light is None:
        light = SphericalLightSource(position=(-6.0, 0.0, 0.0) * u.m, timeRange=(30.0, 30.0) * u.ns, budget=1e9)
    if response is None:
        response = HistogramHitResponse(UniformValueResponse(), nBins=30, binSize=20.0, t0=0.0, normalization=1 / batch)
    rng = PhiloxRNG(key=key) if rng is None else rng
    tg = SphereTargetGuide(position=(0.0, 0.0, 0.0), radius=radius) if guide else custom_guide
    callback = EventStatisticCallback() if stats else EmptyEventCallback()
    tracer = SceneForwardTracer(batch, light, photons, response, rng, scene, maxPathLength=maxPathLength, targetGuide=tg,
                                scatterCoefficient=0.01, maxTime=600.0, polarized=polarized, callback=callback,
                                trackEvents=track)
    if isinstance(rng, PhiloxRNG):
        rng.autoAdvance = tracer.nRNGSamples
    return Scenario(name="config2_sphere", store=store, meshes=meshes, scene=scene, light=light, photons=photons,
                    response=response, rng=rng, guide=tg, tracer=tracer, callback=callback, medium=medium)


def config3_cube_torus(batch: int = 1_000_000, *, polarized: bool = True, stats: bool = False, track: bool = False,
                       maxPathLength: int = 10, key: int = 42) -> Scenario:
    """BASELINE config 3: BK7 cube + torus detector in sea water, polarized."""
    water = WaterTestModel().createMedium(400.0 * u.nm, 800.0 * u.nm, name="water")
    glass = BK7Model().createMedium(400.0 * u.nm, 800.0 * u.nm, name="glass")
    mats = [Material("glass", glass, water, flags="TR"), Material("det", None, water, flags="BD")]
    store = MaterialStore(mats)
    meshes = MeshStore({"cube": bundled_mesh("cube"), "torus": bundled_mesh("torus")})
    cube = meshes.createInstance("cube", "glass", Transform.TRS(scale=0.5, translate=(2.0, 0.0, 0.0)))
    torus = meshes.createInstance("torus", "det", Transform.TRS(scale=1.0, translate=(0.0, 3.0, 0.0)), detectorId=1)
    scene = Scene([cube, torus], store.material, medium=store.media["water"])
    photons = UniformWavelengthSource(lambdaRange=(400.0 * u.nm, 800.0 * u.nm))
    light = SphericalLightSource(position=(0.0, 0.0, 0.0), timeRange=(0.0, 10.0) * u.ns, budget=1.0e5)
    guide = SphereTargetGuide(position=(0.0, 3.0, 0.0), radius=1.25 * u.m)
    response = HistogramHitResponse(UniformValueResponse(), nBins=100, binSize=5.0 * u.ns)
    rng = PhiloxRNG(key=key)
    callback = EventStatisticCallback() if stats else EmptyEventCallback()
    tracer = SceneForwardTracer(batch, light, photons, response, rng, scene, maxPathLength=maxPathLength, targetId=1,
                                targetGuide=guide, scatterCoefficient=0.05 / u.m, polarized=polarized, callback=callback,
                                trackEvents=track)
    rng.autoAdvance = tracer.nRNGSamples
    return Scenario(name="config3_cube_torus", store=store, meshes=meshes, scene=scene, light=light, photons=photons,
                    response=response, rng=rng, guide=guide, tracer=tracer, callback=callback)


def config5_suzanne_grid(batch: int = 1_000_000, *, polarized: bool = True, stats: bool = False, track: bool = False, n_side: int = 10,
                         maxPathLength: int = 20, key: int = 42) -> Scenario:
    """BASELINE config 5: suzanne x n_side^3 instances (BK7) + sphere detector."""
    water = WaterTestModel().createMedium(400.0 * u.nm, 800.0 * u.nm, name="water")
    glass = BK7Model().createMedium(400.0 * u.nm, 800.0 * u.nm, name="glass")
    mats = [Material("glass", glass, water, flags="TR"), Material("det", None, water, flags="BD")]
    store = MaterialStore(mats)
    meshes = MeshStore({"suzanne": bundled_mesh("suzanne"), "sphere": bundled_mesh("sphere")})
    gen = np.random.default_rng(0xC0FFEE)
    pitch = 4.0
    instances = []
    for ix in range(n_side):
        for iy in range(n_side):
            for iz in range(n_side):
                angle = gen.uniform(0.0, 360.0)
                t = Transform.TRS(scale=1.0, rotate=(0.0, 0.0, 1.0, angle), translate=(ix * pitch, iy * pitch, iz * pitch))
                instances.append(meshes.createInstance("suzanne", "glass", t))
    # detector and light sit in the voids between the monkeys (cell centres of the grid):
    # the detector in the central cell, the light one cell down the diagonal, so that the
    # line of sight between them passes through the monkey at the shared grid point
    c = 0.5 * (n_side - 1) * pitch
    det_pos = (c, c, c)
    instances.append(meshes.createInstance("sphere", "det", Transform.TRS(scale=1.0, translate=det_pos), detectorId=1))
    scene = Scene(instances, store.material, medium=store.media["water"])
    photons = UniformWavelengthSource(lambdaRange=(400.0 * u.nm, 800.0 * u.nm))
    light = SphericalLightSource(position=(c - pitch, c - pitch, c - pitch), timeRange=(0.0, 10.0) * u.ns, budget=1.0e5)
    guide = SphereTargetGuide(position=det_pos, radius=1.0 * u.m)
    response = HistogramHitResponse(UniformValueResponse(), nBins=100, binSize=5.0 * u.ns)
    rng = PhiloxRNG(key=key)
    callback = EventStatisticCallback() if stats else EmptyEventCallback()
    tracer = SceneForwardTracer(batch, light, photons, response, rng, scene, maxPathLength=maxPathLength, targetId=1,
                                targetGuide=guide, scatterCoefficient=0.05 / u.m, polarized=polarized, callback=callback,
                                trackEvents=track)
    rng.autoAdvance = tracer.nRNGSamples
    return Scenario(name="config5_suzanne_grid", store=store, meshes=meshes, scene=scene, light=light, photons=photons,
                    response=response, rng=rng, guide=guide, tracer=tracer, callback=callback)


def many_cubes(batch: int = 10_000, *, n_side: int = 17, stats: bool = False, track: bool = False, key: int = 7) -> Scenario:
    """n_side^3 small glass cubes (>= 4096 instances: beyond the staged kernel's packed hit ids, so the
    lane kernel runs) around a detector sphere; unpolarized, no guide."""
    water = WaterTestModel().createMedium(400.0 * u.nm, 800.0 * u.nm, name="water")
    glass = BK7Model().createMedium(400.0 * u.nm, 800.0 * u.nm, name="glass")
    store = MaterialStore([Material("glass", glass, water, flags="TR"), Material("det", None, water, flags="BD")])
    meshes = MeshStore({"cube": bundled_mesh("cube"), "sphere": bundled_mesh("sphere")})
    pitch = 2.0
    instances = [meshes.createInstance("cube", "glass", Transform.TRS(scale=0.3, translate=(ix * pitch, iy * pitch, iz * pitch)))
                 for ix in range(n_side) for iy in range(n_side) for iz in range(n_side)]
    c = 0.5 * (n_side - 1) * pitch + 1.0
    instances.append(meshes.createInstance("sphere", "det", Transform.TRS(scale=0.6, translate=(c, c, c)), detectorId=1))
    scene = Scene(instances, store.material, medium=store.media["water"])
    photons = UniformWavelengthSource(lambdaRange=(400.0 * u.nm, 800.0 * u.nm))
    light = SphericalLightSource(position=(c - pitch, c - pitch, c - pitch), timeRange=(0.0, 10.0) * u.ns, budget=1.0e5)
    response = HistogramHitResponse(UniformValueResponse(), nBins=100, binSize=5.0 * u.ns)
    rng = PhiloxRNG(key=key)
    callback = EventStatisticCallback() if stats else EmptyEventCallback()
    tracer = SceneForwardTracer(batch, light, photons, response, rng, scene, maxPathLength=12, targetId=1,
                                scatterCoefficient=0.05 / u.m, callback=callback, trackEvents=track)
    rng.autoAdvance = tracer.nRNGSamples
    return Scenario(name="many_cubes", store=store, meshes=meshes, scene=scene, light=light, photons=photons,
                    response=response, rng=rng, guide=None, tracer=tracer, callback=callback)


def run_batches(sc: Scenario, n_batches: int = 1) -> list[np.ndarray]:
    """Runs the scenario's pipeline synchronously and returns one histogram per batch."""
    from .hp.pipeline import Pipeline

    pipe = Pipeline(sc.stages())
    hists = []
    for k in range(n_batches):
        cfg = k % 2
        pipe.run(cfg)
        hists.append(sc.response.result(cfg).copy())
    return hists
