"""
The reference's analytic end-to-end pins of the forward and photon tracers, at the REFERENCE'S OWN
sample sizes, on the HIP kernels (`-m gpu` only: 84 M / 200 M / 21 M photons per case are seconds
on an MI355X and hours on the CPU oracle, whose reduced-size versions are in
tests/test_oracle_tracers.py). Same geometry, media, keys, batch counts and tolerances as
/root/reference/tests/test_trace_energy.py, restated in this suite's own form:
`ground truth` (:58-176: light at the centre of a detector sphere; all energy arrives),
`crosscheck` (:187-337: target-guide MIS against plain path tracing of twice the length),
`multi media` (:479-568: nested media with refraction and reflection, 0.61 %),
`volume` (:815-938: light inside an InnerSphereTarget, tolerances 2.8e-4 ... 1.2e-2),
`photon tracers` (:1342-1590: detected photon count against the forward tracer's light curve).
Only the mesh source differs (bundled sphere.stl fixture instead of assets/sphere.stl).
"""
import numpy as np
import pytest

import hephaistos.pipeline as pl
import scenarios as S
import theia.light as TL
import theia.material as TM
import theia.response as TR
import theia.scene as TS
import theia.trace as TT
import theia.units as u
from hephaistos.queue import dumpQueue
from theia.random import PhiloxRNG
from theia.target import InnerSphereTarget, SphereTargetGuide

pytestmark = pytest.mark.gpu


def CASES(table):
    """(argument names, rows) of one case table of tests/golden/energy_cases.json: the reference's own cases and tolerances"""
    import json
    import os

    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "energy_cases.json"), encoding="utf-8") as fh:
        t = json.load(fh)[table]
    rows = [tuple(float("nan") if v is None else v for v in row) for row in t["rows"]]
    return ",".join(t["columns"]), rows

LAMBDA, T_EMIT = 400.0 * u.nm, 10.0 * u.ns
OFF_CENTRE = (12.0, 15.0, 0.2) * u.m
Mi = 1024 * 1024


@pytest.fixture(autouse=True)
def _hip_backend():
    from theia_amd import _backend

    prev = _backend._backend
    _backend._backend = _backend.HipBackend()
    yield
    _backend._backend = prev


def medium_model(a, s, g, **kw):
    """dispersion free, Henyey-Greenstein, Kokhanovsky phase matrix (test_trace_energy.py:31-45)"""
    return S.EnergyTestModel(a, s, g, **kw)


def detector_sphere(model, radius, centre, *, inside: bool):
    """one detector sphere (flags DB): `inside` = the medium fills it (light inside), else it
    surrounds it. Returns (scene, material store)."""
    medium = model.createMedium()
    material = TM.Material("det", medium, None, flags="DB") if inside else TM.Material("det", None, medium, flags="DB")
    store = TM.MaterialStore([material])
    meshes = TS.MeshStore({"sphere": S.bundled_mesh("sphere")})
    instance = meshes.createInstance("sphere", "det", TS.Transform.TRS(scale=radius, translate=centre), detectorId=0)
    return TS.Scene([instance], store.material, medium=store.media["homogenous"]), store


def point_source(where, budget, t=T_EMIT):
    return TL.SphericalLightSource(position=where, timeRange=(t, t), budget=budget)


def run_tasks(tracer, count, collect):
    """`count` batches through a scheduler; `collect(config)` after each (the generator advances by
    itself, no parameters change between tasks)"""
    scheduler = pl.PipelineScheduler(pl.Pipeline(tracer.collectStages()), processFn=lambda cfg, k, args: collect(cfg))
    scheduler.schedule([{} for _ in range(count)])
    scheduler.wait()
    scheduler.destroy()


def mean_curve(tracer, response, count):
    curves = []
    run_tasks(tracer, count, lambda cfg: curves.append(response.result(cfg).copy()))
    return np.mean(curves, 0)


def unattenuated(model, mu_a, time, contrib, t0=T_EMIT):
    """contribution with the absorption along the flight path undone"""
    return contrib * np.exp(mu_a * (1.0 / model.ng * u.c) * (time - t0))


def assert_stokes_valid(stokes):
    assert np.abs(stokes[:, 0] - 1.0).max() < 1e-5
    assert -1.0 <= stokes[:, 1:].min() and stokes[:, 1:].max() <= 1.0
    assert np.all(np.square(stokes[:, 1:]).sum(-1) <= 1.0 + 1e-6)


@pytest.mark.parametrize(*CASES("scene_ground_truth"))
def test_scene_ground_truth(mu_a, mu_s, g, polarized):
    """light at the centre of a 100 m detector sphere full of scattering medium: with the absorption
    undone all of the budget arrives (5 %, from below: longer paths are cut), uniformly over the sphere"""
    budget, batch, tasks = 1e9, 2 * Mi, 40
    model = medium_model(mu_a, mu_s, g)
    scene, _store = detector_sphere(model, 100.0 * u.m, OFF_CENTRE, inside=True)
    rng, recorder = PhiloxRNG(key=0xC0FFEE), TR.HitRecorder(polarized=polarized)
    tracer = TT.SceneForwardTracer(batch, point_source(OFF_CENTRE, budget), TL.UniformWavelengthSource(lambdaRange=(LAMBDA, LAMBDA)),
                                   recorder, rng, scene=scene, maxPathLength=10, scatterCoefficient=0.05, maxTime=float("inf"),
                                   polarized=polarized)
    rng.autoAdvance = tracer.nRNGSamples
    dumps = []
    run_tasks(tracer, tasks, lambda cfg: dumps.append(dumpQueue(recorder.queue.view(cfg))))
    column = lambda f: np.concatenate([d[f] for d in dumps])
    arrived = unattenuated(model, mu_a, column("time"), column("contrib")).sum(dtype=np.float64) / (batch * tasks)
    assert abs(arrived / budget - 1.0) < 0.05 and arrived < budget
    assert np.abs(column("position").mean(0)).max() < 5e-3
    assert np.abs(np.vstack([d["position"].var(0) for d in dumps]) - 1 / 3).max() < 0.01
    if polarized:
        assert_stokes_valid(column("stokes"))


@pytest.mark.parametrize(*CASES("scene_crosscheck"))
def test_scene_crosscheck(mu_a, mu_s, g, polarized):
    """5 m detector sphere at the origin, light 6 m off: paths of length 10 with target-guide MIS
    against plain paths of length 20 (2 % in total, 1 % in the mean relative error of the log curves;
    600 ns cut: the plain estimate is too noisy later)"""
    budget, batch, tasks, radius = 1e9, 1 * Mi, 100, 5.0 * u.m
    scene, _store = detector_sphere(medium_model(mu_a, mu_s, g), radius, (0.0, 0.0, 0.0) * u.m, inside=False)
    light, photons = point_source((-6.0, 0.0, 0.0) * u.m, budget, 30.0 * u.ns), TL.UniformWavelengthSource(lambdaRange=(LAMBDA, LAMBDA))
    rng = PhiloxRNG(key=0xC0FFEE)

    def curve(length, guide):
        response = TR.HistogramHitResponse(TR.UniformValueResponse(), nBins=30, binSize=20.0, t0=0.0, normalization=1 / batch)
        tracer = TT.SceneForwardTracer(batch, light, photons, response, rng, scene=scene, maxPathLength=length, targetGuide=guide,
                                       scatterCoefficient=0.01, maxTime=600.0, polarized=polarized)
        rng.autoAdvance = tracer.nRNGSamples
        return mean_curve(tracer, response, tasks)

    plain = curve(20, None)
    guided = curve(10, SphereTargetGuide(position=(0.0, 0.0, 0.0) * u.m, radius=radius))
    assert plain.sum() > 0.0 and abs(guided.sum() / plain.sum() - 1.0) < 0.02
    with np.errstate(divide="ignore", invalid="ignore"):
        rel = np.nan_to_num((np.log10(guided) - np.log10(plain)) / np.log10(guided), nan=0.0)
    assert np.abs(rel).mean() < 0.01


@pytest.mark.parametrize("polarized", [True, False])
def test_scene_multi_media(polarized):
    """off-centre light inside a 65 m glass-like shell (n 1.8) inside the 100 m detector sphere
    (n 1.2), no absorption: reflection and transmission together return all the energy (0.61 %)"""
    budget, batch, tasks = 1e9, 2 * Mi, 10
    inner = medium_model(0.0, 0.01, 0.9, n=1.8).createMedium(name="inner")
    outer = medium_model(0.0, 0.04, 0.9, n=1.2).createMedium(name="outer")
    store = TM.MaterialStore([TM.Material("det", outer, None, flags="DB"), TM.Material("inner", inner, outer, flags="TR")])
    meshes = TS.MeshStore({"sphere": S.bundled_mesh("sphere")})
    shells = [meshes.createInstance("sphere", name, TS.Transform.TRS(scale=r, translate=OFF_CENTRE))
              for name, r in (("inner", 65.0 * u.m), ("det", 100.0 * u.m))]
    scene = TS.Scene(shells, store.material)
    rng = PhiloxRNG(key=0xC0FFEE)
    response = TR.HistogramHitResponse(TR.UniformValueResponse(), nBins=400, binSize=100.0 * u.ns)
    tracer = TT.SceneForwardTracer(batch, point_source((2.0, 5.0, 0.0) * u.m, budget), TL.ConstWavelengthSource(LAMBDA), response, rng,
                                   scene, maxPathLength=50, scatterCoefficient=0.03, sourceMedium=store.media["inner"],
                                   maxTime=float("inf"), polarized=polarized)
    rng.autoAdvance = tracer.nRNGSamples
    assert abs(mean_curve(tracer, response, tasks).sum() / budget - 1.0) < 0.0061


@pytest.mark.parametrize(*CASES("volume"))
def test_volume_energy(mu_a, mu_s, mu_sample, g, no_direct, sample_target, polarized, tol):
    """light at the centre of a 100 m InnerSphereTarget: the budget (minus the unscattered share when
    direct light is off) arrives once absorption is undone"""
    budget, batch, tasks, radius = 1e9, 512 * 1024, 10, 100.0 * u.m
    model = medium_model(mu_a, mu_s, g)
    store = TM.MaterialStore([], media=[model.createMedium()])
    rng, recorder = PhiloxRNG(key=0xC0FFEE), TR.HitRecorder(polarized=polarized)
    tracer = TT.VolumeForwardTracer(batch, point_source(OFF_CENTRE, budget), InnerSphereTarget(position=OFF_CENTRE, radius=radius),
                                    TL.UniformWavelengthSource(lambdaRange=(LAMBDA, LAMBDA)), recorder, rng,
                                    medium=store.media["homogenous"], maxTime=float("inf"), nScattering=10,
                                    scatterCoefficient=mu_sample, polarized=polarized, disableDirectLighting=no_direct,
                                    disableTargetSampling=not sample_target)
    rng.autoAdvance = tracer.nRNGSamples
    acc = {"sum": 0.0, "stokes": []}

    def collect(cfg):
        q = recorder.queue.view(cfg)
        q = q[: q.count]
        acc["sum"] += unattenuated(model, mu_a, q["time"], q["contrib"]).sum(dtype=np.float64)
        if polarized:
            acc["stokes"].append(q["stokes"].copy())

    run_tasks(tracer, tasks, collect)
    expected = budget - budget * np.exp(-mu_s * radius) if no_direct else budget
    assert abs(acc["sum"] / (batch * tasks) / expected - 1.0) < tol
    if polarized:
        assert_stokes_valid(np.concatenate(acc["stokes"]))


def assert_photon_count(count, expected, budget, mu_a):
    if mu_a != 0.0:  # a count of random survivors: 5 sigma
        assert abs(count - expected) < 5.0 * np.sqrt(expected)
    else:  # nothing absorbs: every photon arrives
        assert count == int(budget)


@pytest.mark.parametrize(*CASES("photon"))
def test_scene_photon_tracer(mu_a, mu_s, g, polarized):
    """photons a ScenePhotonTracer detects (one batch of 10^6, 20 runs x 5 steps) against the light
    curve integral of the SceneForwardTracer (20 Mi paths of length 80) in a 60 m detector sphere"""
    budget, batch, tasks = 1e6, 1 * Mi, 20
    scene, _store = detector_sphere(medium_model(mu_a, mu_s, g), 60.0 * u.m, OFF_CENTRE, inside=True)
    rng, light = PhiloxRNG(key=0xC01DC0FFEE), point_source(OFF_CENTRE, budget)
    photons = TL.UniformWavelengthSource(lambdaRange=(LAMBDA, LAMBDA))
    curve = TR.HistogramHitResponse(TR.UniformValueResponse(), binSize=50.0 * u.ns, nBins=400)
    forward = TT.SceneForwardTracer(batch, light, photons, curve, rng, scene, maxPathLength=80, scatterCoefficient=0.05,
                                    maxTime=float("inf"), polarized=polarized)
    rng.autoAdvance = forward.nRNGSamples
    expected = mean_curve(forward, curve, tasks).sum()
    stored = TR.StoreTimeHitResponse(TR.UniformValueResponse())
    tracer = TT.ScenePhotonTracer(int(budget), light, photons, stored, rng, scene, nScatteringPerRun=5, nRuns=20,
                                  maxTime=float("inf"), polarized=polarized, callback=TT.EventStatisticCallback())
    pl.runPipeline(tracer.collectStages())
    assert stored.result(0) is not None
    assert_photon_count(stored.result(0).count, expected, budget, mu_a)


@pytest.mark.parametrize(*CASES("photon"))
def test_volume_photon_tracer(mu_a, mu_s, g, polarized):
    """the same for the VolumePhotonTracer (40 runs x 5 steps) inside a 60 m InnerSphereTarget against
    the VolumeForwardTracer with target sampling"""
    budget, batch, tasks = 1e6, 1 * Mi, 20
    store = TM.MaterialStore([], media=[medium_model(mu_a, mu_s, g).createMedium()])
    rng, light = PhiloxRNG(key=0xC0FFEE), point_source(OFF_CENTRE, budget)
    photons, target = TL.UniformWavelengthSource(lambdaRange=(LAMBDA, LAMBDA)), InnerSphereTarget(position=OFF_CENTRE, radius=60.0 * u.m)
    curve = TR.HistogramHitResponse(TR.UniformValueResponse(), binSize=50.0 * u.ns, nBins=400)
    forward = TT.VolumeForwardTracer(batch, light, target, photons, curve, rng, medium=store.media["homogenous"],
                                     maxTime=float("inf"), nScattering=80, scatterCoefficient=0.05, polarized=polarized)
    rng.autoAdvance = forward.nRNGSamples
    expected = mean_curve(forward, curve, tasks).sum()
    stored = TR.StoreTimeHitResponse(TR.UniformValueResponse())
    tracer = TT.VolumePhotonTracer(int(budget), light, target, photons, stored, rng, medium=store.media["homogenous"],
                                   nScatteringPerRun=5, nRuns=40, maxTime=float("inf"), polarized=polarized,
                                   callback=TT.EventStatisticCallback())
    pl.runPipeline(tracer.collectStages())
    assert stored.result(0) is not None
    assert_photon_count(stored.result(0).count, expected, budget, mu_a)
