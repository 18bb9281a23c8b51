"""
What the reference's response tests check through `HitReplay`, restated in this suite's own
form (same sizes, seeds, expectations and tolerances; /root/reference/tests/test_response.py:17-60
record + polarization mismatch, :62-99 histogram, :133-180 kernel histogram, :253-285 time
store). Every case runs on the CPU oracle (`-m "not gpu"`) and, through the C ABI, on the HIP
replay kernel (`-m gpu`); expectations come from numpy / scipy. A last test compares the two
backends on one queue.

Not restated: value queues and the estimators on them (`test_histogramEstimator`,
`test_uniformResponse`; out of scope) and the camera response sampler.
"""
import numpy as np
import pytest
from scipy.stats import norm

import theia.response as R
import theia.units as u
from hephaistos.pipeline import runPipeline
from theia.random import PhiloxRNG

BACKENDS = [pytest.param("oracle", id="oracle"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]
SEED = 0xC0FFEE  # tests/conftest.py:73-76 of the reference: a fresh default_rng per test


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


def fill_geometry(q, gen, n, polarized=False):
    """position, direction, normal, wavelength [, stokes, reference] in the order the reference's
    tests draw them, so that the generator is in the same state afterwards"""
    q["position"] = (10.0 * gen.random((n, 3)) - 5.0) * u.m
    for name in ("direction", "normal"):
        q[name] = gen.random((n, 3))
    q["wavelength"] = (gen.random(n) * 100.0 + 400.0) * u.nm
    if polarized:
        q["stokes"], q["polarizationRef"] = gen.random((n, 4)), gen.random((n, 3))


@pytest.mark.parametrize("polarized", [True, False])
def test_record(backend, polarized):
    """every column of every hit comes back bit-identical; the contribution carries the index
    since the device appends in any order"""
    n, gen = 8192, np.random.default_rng(SEED)
    recorder = R.HitRecorder(polarized=polarized)
    replay = R.HitReplay(n, recorder, polarized=polarized)
    src = replay.queue.view(0)
    fill_geometry(src, gen, n)
    src["time"] = gen.random(n) * 100.0 * u.ns
    src["contrib"] = np.arange(n, dtype=np.float32)
    if polarized:
        src["stokes"], src["polarizationRef"] = gen.random((n, 4)), gen.random((n, 3))
    runPipeline(replay.collectStages())
    got = recorder.queue.view(0)
    assert got.count == n
    got = got[np.argsort(got["contrib"])]
    for field in src.fields:
        if field != "objectId":
            assert np.array_equal(src[field], got[field]), field


def test_record_refuses_mismatched_polarization():
    for a, b in ((True, False), (False, True)):
        with pytest.raises(RuntimeError):
            R.HitReplay(128, R.HitRecorder(polarized=a), polarized=b)


@pytest.mark.parametrize("useSharedMemory", [True, False])
def test_histogram(backend, useSharedMemory):
    n, bins, width, t0, scale = 256 * 1024, 50, 4.0 * u.ns, 20.0 * u.ns, 1e-5
    t1 = t0 + bins * width
    gen = np.random.default_rng(SEED)
    response = R.HistogramHitResponse(R.UniformValueResponse(), nBins=bins, t0=t0, binSize=width, normalization=scale,
                                      useSharedMemory=useSharedMemory)
    replay = R.HitReplay(n, response, blockSize=128)
    q = replay.queue.view(0)
    fill_geometry(q, gen, n)
    q["time"] = gen.random(n) * t1
    q["contrib"] = gen.random(n) * 10.0
    runPipeline(replay.collectStages())
    want, _ = np.histogram(q["time"], np.arange(bins + 1) * width + t0, weights=q["contrib"])
    assert np.allclose(response.result(0), want * scale, rtol=1e-4)


@pytest.mark.parametrize("useSharedMemory", [False, True])
def test_kernel_histogram(backend, useSharedMemory):
    n, bins, width, t0, scale = 64 * 1024, 200, 1.2 * u.ns, 20.0 * u.ns, 5e-4
    sigma, support = 10.0 * u.ns, 50.0 * u.ns
    peak = t0 + 0.3 * bins * width
    gen = np.random.default_rng(SEED)
    response = R.KernelHistogramHitResponse(R.UniformValueResponse(), nBins=bins, t0=t0, binSize=width, kernelBandwidth=sigma,
                                            kernelSupport=support, normalization=scale, useSharedMemory=useSharedMemory)
    replay = R.HitReplay(n, response)
    q = replay.queue.view(0)
    fill_geometry(q, gen, n)
    q["time"] = gen.normal(peak, 10.0 * width, n)
    q["contrib"] = gen.random(n) + np.sqrt(q["time"])
    runPipeline(replay.collectStages())
    # every hit is a Gaussian of width sigma: its share of a bin is the difference of the normal
    # CDF at the bin's edges (the support cut-off of 5 sigma is below the tolerance)
    edges = np.arange(bins + 1) * width + t0
    t, w = np.float64(q["time"]), np.float64(q["contrib"])
    cdf = np.array([(w * norm.cdf(e, loc=t, scale=sigma)).sum() for e in edges])
    assert np.abs(response.result(0) - np.diff(cdf) * scale).max() < 6e-4


@pytest.mark.parametrize("storeObjectId", [True, False])
def test_store_time(backend, storeObjectId):
    """contribution = acceptance probability: about half of the hits survive; time stamps double
    as indices so that every stored item can be traced back"""
    n, gen = 32 * 256, np.random.default_rng(SEED)
    response = R.StoreTimeHitResponse(R.UniformValueResponse(), storeObjectId=storeObjectId)
    replay = R.HitReplay(n, response, rng=PhiloxRNG(key=0xC0FFEE))
    q = replay.queue.view(0)
    fill_geometry(q, gen, n)
    q["time"] = np.arange(n)
    q["contrib"] = gen.random(n)
    q["objectId"] = np.arange(n, dtype=np.int32)
    runPipeline(replay.collectStages())
    out = response.result(0)
    assert out is not None and abs(out.count / n - 0.5) < 0.005
    t = out["time"]
    assert 0 <= t.min() and t.max() < n
    assert len(np.unique(t)) == out.count
    if storeObjectId:
        assert np.array_equal(t, out["objectId"])
    else:
        assert "objectId" not in out.fields


def test_replay_requires_rng():
    with pytest.raises(ValueError):  # response.py:316-318
        R.HitReplay(128, R.StoreTimeHitResponse(R.UniformValueResponse()))


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["histogram", "kernel", "store"])
def test_replay_hip_matches_oracle(kind):
    """same queue through the oracle's and the HIP replay: store-time keeps exactly the same items
    (same Philox draws), the histograms agree to float summation order"""
    from oracle_backend import OracleBackend
    from theia_amd import _backend

    n, gen = 50_000, np.random.default_rng(SEED)
    times, weights = gen.random(n) * 300.0, gen.random(n)
    out, prev = {}, _backend._backend
    for name in ("oracle", "hip"):
        be = OracleBackend(threads=8) if name == "oracle" else _backend.HipBackend()
        _backend._backend = be
        try:
            value, hist = R.UniformValueResponse(), dict(nBins=128, t0=10.0, binSize=2.0, normalization=1e-3)
            if kind == "store":
                response = R.StoreTimeHitResponse(value)
                replay = R.HitReplay(n, response, rng=PhiloxRNG(key=0xABCD))
            elif kind == "kernel":
                response = R.KernelHistogramHitResponse(value, kernelBandwidth=3.0, kernelSupport=9.0, **hist)
                replay = R.HitReplay(n, response)
            else:
                response = R.HistogramHitResponse(value, **hist)
                replay = R.HitReplay(n, response)
            q = replay.queue.view(0)
            q["time"], q["contrib"], q["objectId"] = times, weights, np.arange(n, dtype=np.int32)
            q.count = n - 17  # the last items are not replayed
            runPipeline(replay.collectStages())
            if kind == "store":
                r = response.result(0)
                out[name] = np.sort(np.array(r["objectId"][: r.count]))
            else:
                out[name] = np.array(response.result(0))
        finally:
            _backend._backend = prev
            if name == "oracle":
                be.close()
    if kind == "store":
        assert out["hip"].max() < n - 17 and np.array_equal(out["oracle"], out["hip"])
    else:
        assert out["oracle"].sum() > 0 and np.allclose(out["oracle"], out["hip"], rtol=2e-5, atol=1e-7)
