"""
What the reference's tests check about the light and wavelength sources a forward tracer can be
fed from, restated in this suite's own form with the same sizes, seeds, expectations and
tolerances (/root/reference/tests/test_light.py: :23-71 host sources, :74-141 streaming sources,
:144-151 sampler / host queue mismatch, :154-169 const, :172-194 uniform and :197-230 function
wavelengths, :233-278 cone, :339-372 pencil, :375-411 spherical source, forward mode). The sources
are observed through `LightSampler`; every case runs on the CPU oracle (`-m "not gpu"`) and,
through the C ABI, on the HIP sampler kernel (`-m gpu`). The last tests feed a tracer from host
queues.

Not restated: the backward-mode cases (backward tracers are out of scope). The Cherenkov / muon / cascade
sources: tests/test_reference_cherenkov.py.
"""
import numpy as np
import pytest

import hephaistos.pipeline as pl
import theia.light as TL
import theia.material
import theia.response
import theia.trace
import theia.units as u
from hephaistos.queue import dumpQueue
from theia.random import PhiloxRNG
from theia.target import SphereTarget
from theia.testing import WaterTestModel

BACKENDS = [pytest.param("oracle", id="oracle"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]
SEED = 0xC0FFEE  # tests/conftest.py:73-76 of the reference: a fresh default_rng per test
KEY = 0xC0110FFC0FFEE
N = 32 * 256


@pytest.fixture(params=BACKENDS)
def backend(request):
    from theia_amd import _backend

    prev = _backend._backend
    if request.param == "oracle":
        from oracle_backend import OracleBackend

        be = OracleBackend(threads=8)
    else:
        be = _backend.HipBackend()
    _backend._backend = be
    yield request.param
    _backend._backend = prev
    if request.param == "oracle":
        be.close()


def unit_sq(v):
    return np.square(v).sum(-1)


def sampled(light, photons, n=N, *, rng=None, polarized=False):
    """runs a LightSampler once; returns (light samples, wavelength samples)"""
    sampler = TL.LightSampler(light, photons, n, rng=rng, polarized=polarized)
    pl.runPipeline(sampler.collectStages())
    return sampler.lightQueue.view(0), sampler.wavelengthQueue.view(0)


def random_rays(gen, n, polarized, *, swapped=False):
    """positions in a 10 m cube and directions from two angles, drawn in the order of the reference's
    tests (`swapped`: the direction components as test_streaming builds them)"""
    pos = np.stack([(gen.random(n) * 10.0 - 5.0) * u.m for _ in range(3)], axis=-1)
    cos_t = 2.0 * gen.random(n) - 1.0
    sin_t = np.sqrt(1.0 - cos_t**2)
    phi = gen.random(n) * 2.0 * np.pi
    if swapped:
        d = np.stack([sin_t * np.cos(phi), cos_t * np.cos(phi), np.sin(phi)], axis=-1)
    else:
        d = np.stack([sin_t * np.cos(phi), sin_t * np.sin(phi), cos_t], axis=-1)
    return pos, d


@pytest.mark.parametrize("polarized", [True, False])
def test_host_sources_pass_their_queues_through(backend, polarized):
    gen = np.random.default_rng(SEED)
    lam_src, ray_src = TL.HostWavelengthSource(N), TL.HostLightSource(N, polarized=polarized)
    rays, lams = ray_src.queue.view(0), lam_src.queue.view(0)
    rays["position"], rays["direction"] = random_rays(gen, N, polarized)
    lams["wavelength"] = (gen.random(N) * 600.0 + 200.0) * u.nm
    lams["contrib"] = gen.random(N) + 5.0
    rays["startTime"] = gen.random(N) * 50.0 * u.ns
    rays["contrib"] = gen.random(N) + 1.0
    if polarized:  # any values will do
        rays["stokes"], rays["polarizationRef"] = gen.random((N, 4)), gen.random((N, 3))
    got, got_lam = sampled(ray_src, lam_src, polarized=polarized)
    assert got.count == N
    for f in rays.fields:
        assert np.allclose(got[f], rays[f]), f
    for f in lams.fields:
        assert np.allclose(got_lam[f], lams[f]), f


@pytest.mark.parametrize("polarized", [False, True])
def test_streaming_sources_walk_through_their_data(backend, polarized):
    """two data sets of 2560 rays in batches of 1024 / 1024 / 512; the second set arrives with the
    fourth task and rewinds the stream"""
    total, batch = 5 * 512, 1024
    gen = np.random.default_rng(SEED)

    def make():
        d = {"contrib": 10.0 * gen.random(total), "wavelength": 200.0 + 500.0 * gen.random(total)}
        d["position"], d["direction"] = random_rays(gen, total, polarized, swapped=True)
        d["startTime"] = 50.0 * gen.random(total)
        if polarized:
            d["stokes"], d["polarizationRef"] = gen.random((total, 4)), gen.random((total, 3))
        return d

    first, second = make(), make()
    lam_src = TL.StreamingHostWavelengthSource(batch, data=first, ignoreContrib=True)
    ray_src = TL.StreamingHostLightSource(batch, data=first, polarized=polarized)
    sampler = TL.LightSampler(ray_src, lam_src, batch, polarized=polarized)
    seen = []
    scheduler = pl.PipelineScheduler(pl.Pipeline(sampler.collectStages()),
                                     processFn=lambda cfg, k, args: seen.append((dumpQueue(sampler.lightQueue.view(cfg)),
                                                                                 dumpQueue(sampler.wavelengthQueue.view(cfg)))))
    scheduler.schedule([{}, {}, {"count": 512}, {"data": second, "count": 1024}, {}, {"count": 512}])
    scheduler.wait()
    assert len(seen) == 6
    for data, part in ((first, seen[:3]), (second, seen[3:])):
        for f in sampler.lightQueue.view(0).fields:
            assert np.allclose(np.concatenate([r[f] for r, _ in part]), data[f]), f
        assert np.allclose(np.concatenate([w["wavelength"] for _, w in part]), data["wavelength"])
    assert all(np.all(w["contrib"] == 1.0) for _, w in seen)  # ignoreContrib


def test_sampler_refuses_a_host_queue_of_other_polarization():
    for a, b in ((True, False), (False, True)):
        with pytest.raises(RuntimeError):
            TL.LightSampler(TL.HostLightSource(128, polarized=a), TL.ConstWavelengthSource(), 128, polarized=b)


def test_const_wavelength(backend):
    lam = 450.0 * u.nm
    _, got = sampled(TL.PencilLightSource(timeRange=(0.0, 0.0)), TL.ConstWavelengthSource(lam), rng=PhiloxRNG(key=KEY))
    assert np.all(got["contrib"] == 1.0) and np.all(got["wavelength"] == lam)


@pytest.mark.parametrize("normalize", [True, False])
def test_uniform_wavelength(backend, normalize):
    lo, hi = (350.0, 750.0) * u.nm
    photons = TL.UniformWavelengthSource(lambdaRange=(lo, hi), normalize=normalize)
    _, got = sampled(TL.PencilLightSource(timeRange=(0.0, 0.0) * u.ns), photons, rng=PhiloxRNG(key=KEY))
    assert np.all(got["contrib"] == (1.0 if normalize else hi - lo))  # L / p(lambda)
    assert abs(got["wavelength"].min() - lo) < 1.0 and abs(got["wavelength"].max() - hi) < 1.0


def test_function_wavelength(backend):
    """importance sampling of -x ln x (x = lambda / 1000): constant contribution = its integral,
    the histogram of the samples follows the antiderivative"""
    n, (lo, hi) = 256 * 1024, (300.0, 700.0) * u.nm
    pdf = lambda lam: -(lam / 1000) * np.log(lam / 1000)
    cdf = lambda lam: 250.0 * (lam / 1000) ** 2 * (1.0 - 2.0 * np.log(lam / 1000))
    photons = TL.FunctionWavelengthSource(pdf, lambdaRange=(lo, hi))
    _, got = sampled(TL.PencilLightSource(timeRange=(0.0, 0.0)), photons, n, rng=PhiloxRNG(key=SEED))
    integral = cdf(hi) - cdf(lo)
    assert np.allclose(got["contrib"], integral)
    assert abs(got["wavelength"].min() - lo) < 1.0 and abs(got["wavelength"].max() - hi) < 1.0
    counts, edges = np.histogram(got["wavelength"], bins=40)
    assert np.abs(counts / n - np.diff(cdf(edges)) / integral).max() < 9e-4


SOURCE = dict(position=(14.0, -2.0, 3.0) * u.m, timeRange=(10.0, 10.0) * u.ns, budget=12.0)
AXIS, STOKES, REF = (0.8, 0.36, 0.48), (1.0, 0.9, 0.1, -0.5), (0.0, 0.48, -0.36)


@pytest.mark.parametrize("polarized", [True, False])
def test_cone_source(backend, polarized):
    cos_open = 0.33
    light = TL.ConeLightSource(direction=AXIS, cosOpeningAngle=cos_open, stokes=STOKES, polarizationReference=REF,
                               polarized=polarized, **SOURCE)
    got, _ = sampled(light, TL.ConstWavelengthSource(wavelength=100.0 * u.nm), rng=PhiloxRNG(key=KEY), polarized=polarized)
    assert got.count == N and np.allclose(got["position"], SOURCE["position"])
    d = got["direction"]
    assert (d * AXIS).sum(-1).min() >= cos_open and np.abs(unit_sq(d) - 1.0).max() < 1e-5
    assert np.all(got["contrib"] == SOURCE["budget"])
    if polarized:
        ref = got["polarizationRef"]
        assert np.allclose(got["stokes"], STOKES) and np.abs(unit_sq(ref) - 1.0).max() < 1e-5
        # the reference is only tilted into the plane perpendicular to the ray, not rotated about
        # it: ray, given reference and sampled reference stay coplanar
        assert np.abs((np.cross(d, REF) * ref).sum(-1)).max() < 1e-5


@pytest.mark.parametrize("polarized", [True, False])
def test_pencil_source(backend, polarized):
    light = TL.PencilLightSource(direction=AXIS, stokes=STOKES, polarizationRef=REF, **SOURCE)
    got, _ = sampled(light, TL.ConstWavelengthSource(wavelength=100.0 * u.nm), rng=PhiloxRNG(key=KEY), polarized=polarized)
    assert got.count == N
    assert np.allclose(got["position"], SOURCE["position"]) and np.allclose(got["direction"], AXIS)
    assert np.all(got["contrib"] == SOURCE["budget"])
    if polarized:
        assert np.allclose(got["stokes"], STOKES) and np.allclose(got["polarizationRef"], REF)


@pytest.mark.parametrize("polarized", [True, False])
def test_spherical_source(backend, polarized):
    light = TL.SphericalLightSource(**SOURCE)
    got, _ = sampled(light, TL.ConstWavelengthSource(wavelength=100.0 * u.nm), rng=PhiloxRNG(key=KEY), polarized=polarized)
    d = got["direction"]
    assert got.count == N and np.all(got["position"] == np.float32(SOURCE["position"]))
    assert np.allclose(np.sqrt(unit_sq(d)), 1.0)
    assert np.abs(d.mean(0)).max() < 0.01  # isotropic within the statistics
    assert np.allclose(got["contrib"], SOURCE["budget"])
    if polarized:  # unpolarized light: any reference perpendicular to the ray
        ref = got["polarizationRef"]
        assert np.allclose(got["stokes"], (1.0, 0.0, 0.0, 0.0))
        assert np.abs((ref * d).sum(-1)).max() < 1e-5 and (unit_sq(ref) - 1.0).max() < 1e-5


# ---- the host sources feeding a tracer ---------------------------------------------------------


def _volume_tracer(n, source, photons, polarized, key=0xC01DC0FFEE):
    water = WaterTestModel().createMedium()
    store = theia.material.MaterialStore([], media=[water])
    rng = PhiloxRNG(key=key)
    recorder = theia.response.HitRecorder(polarized=polarized)
    tracer = theia.trace.VolumeForwardTracer(
        n, source, SphereTarget(position=(5.0, 2.0, -8.0) * u.m, radius=4.0 * u.m), photons, recorder, rng,
        medium=store.media["water"], nScattering=5, polarized=polarized, maxTime=1.0 * u.us,
    )
    rng.autoAdvance = tracer.nRNGSamples
    return tracer, recorder, store


@pytest.mark.parametrize("polarized", [False, True])
def test_tracer_fed_from_host_queues_matches_the_sampled_source(backend, polarized):
    """Samples a cone source + uniform wavelengths with the LightSampler, feeds the samples back
    through HostLightSource / HostWavelengthSource and compares the hits with the tracer that
    samples the same sources itself. The host sources draw no random numbers, so the two tracers
    use different random streams from the first scatter on: the hits are compared statistically
    (same number to 3 sigma, same mean arrival time to 3 sigma, same summed contribution to 10 %)."""
    light = TL.ConeLightSource(position=(-1.0, -7.0, 0.0) * u.m, direction=(0.6, 0.8, 0.0), cosOpeningAngle=0.2,
                               timeRange=(10.0, 20.0) * u.ns, budget=1000.0, polarized=polarized,
                               stokes=(1.0, 0.3, 0.2, 0.1), polarizationReference=(0.0, 0.0, 1.0))
    photons = TL.UniformWavelengthSource(lambdaRange=(400.0, 600.0) * u.nm)
    tracer, recorder, _store = _volume_tracer(N, light, photons, polarized)
    pl.runPipeline(tracer.collectStages())
    q = recorder.queue.view(0)
    direct = {f: np.array(q[f][: q.count]) for f in ("time", "contrib")}
    sampler = TL.LightSampler(light, photons, N, rng=PhiloxRNG(key=0xABCDEF), polarized=polarized)
    pl.runPipeline(sampler.collectStages())
    host_light, host_lam = TL.HostLightSource(N, polarized=polarized), TL.HostWavelengthSource(N)
    src, dst = sampler.lightQueue.view(0), host_light.queue.view(0)
    for f in dst.fields:
        dst[f] = src[f]
    for f in ("wavelength", "contrib"):
        host_lam.queue.view(0)[f] = sampler.wavelengthQueue.view(0)[f]
    tracer2, recorder2, _store2 = _volume_tracer(N, host_light, host_lam, polarized)
    assert tracer2.nRNGSamples == tracer.nRNGSamples - 4  # 1 wavelength + 3 light draws fewer
    pl.runPipeline(tracer2.collectStages())
    q = recorder2.queue.view(0)
    fed = {f: np.array(q[f][: q.count]) for f in ("time", "contrib")}
    n1, n2 = len(direct["time"]), len(fed["time"])
    assert n1 > 1000 and abs(n1 - n2) < 3.0 * np.sqrt(n1 + n2)
    sem = np.hypot(direct["time"].std() / np.sqrt(n1), fed["time"].std() / np.sqrt(n2))
    assert abs(direct["time"].mean() - fed["time"].mean()) < 3.0 * sem
    assert abs(direct["contrib"].sum() / fed["contrib"].sum() - 1.0) < 0.1


@pytest.mark.gpu
@pytest.mark.parametrize("polarized", [False, True])
def test_host_fed_tracer_hip_matches_oracle(polarized):
    """same host queues through the oracle and the HIP volume tracer: the recorded hits agree
    hit by hit (matched by time)"""
    from oracle_backend import OracleBackend
    from theia_amd import _backend

    n, gen = 20_000, np.random.default_rng(SEED)
    cos_t = 2.0 * gen.random(n) - 1.0
    sin_t, phi = np.sqrt(1.0 - cos_t**2), gen.random(n) * 2.0 * np.pi
    data = {
        "position": np.tile(np.array([-1.0, -7.0, 0.0]), (n, 1)) + gen.random((n, 3)),
        "direction": np.stack([sin_t * np.cos(phi), sin_t * np.sin(phi), cos_t], axis=-1),
        "startTime": gen.random(n) * 10.0,
        "contrib": gen.random(n) + 0.5,
    }
    if polarized:
        data["stokes"] = np.tile(np.array([1.0, 0.2, -0.1, 0.3]), (n, 1))
        ref = np.cross(data["direction"], np.array([0.0, 0.0, 1.0]))
        data["polarizationRef"] = ref / np.linalg.norm(ref, axis=-1, keepdims=True)
    lam = 400.0 + 200.0 * gen.random(n)
    out, prev = {}, _backend._backend
    for name in ("oracle", "hip"):
        be = OracleBackend(threads=8) if name == "oracle" else _backend.HipBackend()
        _backend._backend = be
        try:
            host_light, host_lam = TL.HostLightSource(n, polarized=polarized), TL.HostWavelengthSource(n)
            for f, v in data.items():
                host_light.queue.view(0)[f] = v
            host_lam.queue.view(0)["wavelength"], host_lam.queue.view(0)["contrib"] = lam, 1.0
            tracer, recorder, _store = _volume_tracer(n, host_light, host_lam, polarized)
            pl.runPipeline(tracer.collectStages())
            q = recorder.queue.view(0)
            order = np.argsort(q["time"][: q.count])
            out[name] = {f: np.array(q[f][: q.count])[order] for f in q.fields}
        finally:
            _backend._backend = prev
            if name == "oracle":
                be.close()
    a, b = out["oracle"], out["hip"]
    assert len(a["time"]) > 500
    assert abs(len(a["time"]) - len(b["time"])) <= max(2, len(a["time"]) // 500)
    idx = np.clip(np.searchsorted(b["time"], a["time"]), 1, len(b["time"]) - 1)
    idx = np.where(np.abs(b["time"][idx - 1] - a["time"]) < np.abs(b["time"][idx] - a["time"]), idx - 1, idx)
    close = np.abs(b["time"][idx] - a["time"]) < 1e-3
    assert close.mean() > 0.995
    # a few pairs are mismatched (two hits within the matching window) or differ by a rounding-level
    # change of a scattering angle that long paths amplify
    assert np.isclose(a["contrib"][close], b["contrib"][idx][close], rtol=2e-3, atol=1e-7).mean() > 0.995
    assert np.isclose(a["position"][close], b["position"][idx][close], atol=2e-3).all(-1).mean() > 0.995


def test_sampler_count_cannot_exceed_its_capacity(backend):
    sampler = TL.LightSampler(TL.PencilLightSource(), TL.ConstWavelengthSource(), 256, rng=PhiloxRNG(key=1))
    sampler.setParams(count=257)
    with pytest.raises(ValueError):
        pl.runPipeline(sampler.collectStages())


@pytest.mark.gpu
def test_batch_larger_than_the_host_queue_fails_loudly():
    """a tracer may be larger than the host source feeding it, a batch may not"""
    from theia_amd import _backend, _capi

    prev, _backend._backend = _backend._backend, _backend.HipBackend()
    try:
        tracer, _recorder, _store = _volume_tracer(512, TL.HostLightSource(256), TL.HostWavelengthSource(512), False)
        with pytest.raises(_capi.ThxError):
            pl.runPipeline(tracer.collectStages())
    finally:
        _backend._backend = prev


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["cone+function", "pencil+uniform", "spherical+sobol"])
def test_tracer_with_every_source_hip_matches_oracle(kind):
    """light curves of a volume tracer driven by each light / wavelength source (and by the Sobol
    generator): oracle and HIP agree bin by bin"""
    from oracle_backend import OracleBackend
    from theia_amd import _backend
    import theia.random as TRN

    n, out, prev = 32 * 1024, {}, _backend._backend
    for name in ("oracle", "hip"):
        be = OracleBackend(threads=8) if name == "oracle" else _backend.HipBackend()
        _backend._backend = be
        try:
            water = WaterTestModel().createMedium()
            store = theia.material.MaterialStore([], media=[water])
            where = (-1.0, -7.0, 0.0) * u.m
            if kind == "cone+function":
                light = TL.ConeLightSource(position=where, direction=(0.6, 0.8, 0.0), cosOpeningAngle=0.3, budget=100.0,
                                           timeRange=(0.0, 5.0) * u.ns, polarized=True, stokes=(1.0, 0.4, 0.0, 0.1),
                                           polarizationReference=(0.0, 0.0, 1.0))
                photons = TL.FunctionWavelengthSource(lambda lam: 1.0 + (lam - 400.0) / 300.0, lambdaRange=(400.0, 700.0) * u.nm)
            elif kind == "pencil+uniform":
                light = TL.PencilLightSource(position=where, direction=(0.6, 0.8, 0.0), budget=100.0, timeRange=(0.0, 5.0) * u.ns,
                                             stokes=(1.0, 0.0, 0.3, 0.0), polarizationRef=(0.0, 0.0, 1.0))
                photons = TL.UniformWavelengthSource(lambdaRange=(400.0, 700.0) * u.nm)
            else:
                light = TL.SphericalLightSource(position=where, budget=100.0, timeRange=(0.0, 5.0) * u.ns)
                photons = TL.ConstWavelengthSource(500.0 * u.nm)
            rng = TRN.SobolQRNG(seed=0xBEEF) if kind.endswith("sobol") else PhiloxRNG(key=0xFEED)
            response = theia.response.HistogramHitResponse(theia.response.UniformValueResponse(), nBins=64, binSize=10.0 * u.ns)
            tracer = theia.trace.VolumeForwardTracer(n, light, SphereTarget(position=(5.0, 2.0, -8.0) * u.m, radius=4.0 * u.m), photons,
                                                     response, rng, medium=store.media["water"], nScattering=6, polarized=True)
            pl.runPipeline(tracer.collectStages())
            out[name] = np.array(response.result(0))
        finally:
            _backend._backend = prev
            if name == "oracle":
                be.close()
    assert out["oracle"].sum() > 0
    assert np.allclose(out["oracle"], out["hip"], rtol=2e-3, atol=1e-5 * out["oracle"].max())
