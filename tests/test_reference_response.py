"""
The reference's own response tests that run through `HitReplay`
(/root/reference/tests/test_response.py:17-60 `test_record`, `test_hitPolarizationMismatch`;
:62-99 `test_histogramResponse`; :133-180 `test_kernelHistogramEstimator`; :253-285
`test_StoreTimeHitResponse`), restated with their original bodies against the `theia` /
`hephaistos` alias packages. Every case runs on the CPU oracle (`-m "not gpu"`) and, through
the C ABI, on the HIP replay kernel (`-m gpu`); the expectations are the reference's (numpy
histograms, scipy's normal CDF). A last test compares both backends item by item.

Not restated: `test_histogramEstimator`, `test_uniformResponse` (value queues / estimators --
out of scope), `test_CameraHitResponseSamples` (cameras).
"""
import numpy as np
import pytest
import scipy.stats as stats

import theia.response
import theia.units as u
from hephaistos.pipeline import runPipeline
from theia.random import PhiloxRNG

BACKENDS = [pytest.param("oracle", id="oracle"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


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


@pytest.fixture
def rng():
    # tests/conftest.py:73-76 of the reference
    return np.random.default_rng(0xC0FFEE)


@pytest.mark.parametrize("polarized", [True, False])
def test_record(backend, rng, polarized: bool):
    N = 8192

    record = theia.response.HitRecorder(polarized=polarized)
    replay = theia.response.HitReplay(N, record, polarized=polarized)

    samples = replay.queue.view(0)
    samples["position"] = (10.0 * rng.random((N, 3)) - 5.0) * u.m
    samples["direction"] = rng.random((N, 3))
    samples["normal"] = rng.random((N, 3))
    samples["wavelength"] = (rng.random((N,)) * 100.0 + 400.0) * u.nm
    samples["time"] = (rng.random((N,)) * 100.0) * u.ns
    # use contrib to encode ordering (gpu will scramble items)
    samples["contrib"] = np.arange(N).astype(np.float32)
    if polarized:
        samples["stokes"] = rng.random((N, 4))
        samples["polarizationRef"] = rng.random((N, 3))

    runPipeline(replay.collectStages())

    results = record.queue.view(0)
    assert results.count == N
    # restore ordering via contrib
    results = results[np.argsort(results["contrib"])]
    assert np.all(samples["position"] == results["position"])
    assert np.all(samples["direction"] == results["direction"])
    assert np.all(samples["normal"] == results["normal"])
    assert np.all(samples["wavelength"] == results["wavelength"])
    assert np.all(samples["time"] == results["time"])
    assert np.all(samples["contrib"] == results["contrib"])
    if polarized:
        assert np.all(samples["stokes"] == results["stokes"])
        assert np.all(samples["polarizationRef"] == results["polarizationRef"])


def test_hitPolarizationMismatch():
    with pytest.raises(RuntimeError):
        record = theia.response.HitRecorder(polarized=True)
        theia.response.HitReplay(128, record, polarized=False)
    with pytest.raises(RuntimeError):
        record = theia.response.HitRecorder(polarized=False)
        theia.response.HitReplay(128, record, polarized=True)


@pytest.mark.parametrize("useSharedMemory", [True, False])
def test_histogramResponse(backend, rng, useSharedMemory: bool):
    N = 256 * 1024
    N_BINS = 50
    BIN_SIZE = 4.0 * u.ns
    T0 = 20.0 * u.ns
    T1 = T0 + N_BINS * BIN_SIZE
    NORM = 1e-5

    value = theia.response.UniformValueResponse()
    response = theia.response.HistogramHitResponse(
        value,
        nBins=N_BINS,
        t0=T0,
        binSize=BIN_SIZE,
        normalization=NORM,
        useSharedMemory=useSharedMemory,
    )
    replay = theia.response.HitReplay(N, response, blockSize=128)

    samples = replay.queue.view(0)
    samples["position"] = (10.0 * rng.random((N, 3)) - 5.0) * u.m
    samples["direction"] = rng.random((N, 3))
    samples["normal"] = rng.random((N, 3))
    samples["wavelength"] = (rng.random((N,)) * 100.0 + 400.0) * u.nm
    samples["time"] = rng.random((N,)) * T1
    samples["contrib"] = rng.random((N,)) * 10.0

    runPipeline(replay.collectStages())

    result = response.result(0)
    # calculate expected results
    bin_edge = np.arange(N_BINS + 1) * BIN_SIZE + T0
    exp_hist, _ = np.histogram(samples["time"], bin_edge, weights=samples["contrib"])
    exp_hist *= NORM
    # check result
    assert np.allclose(result, exp_hist, rtol=1e-4)


@pytest.mark.parametrize("useSharedMemory", [False, True])
def test_kernelHistogramEstimator(backend, rng, useSharedMemory: bool):
    N = 64 * 1024
    N_BINS = 200
    BIN_SIZE = 1.2 * u.ns
    T0 = 20.0 * u.ns
    T1 = T0 + N_BINS * BIN_SIZE
    T_PEAK = T0 + 0.3 * (T1 - T0)
    BANDWIDTH = 10.0 * u.ns
    SUPPORT = 50.0 * u.ns
    NORM = 5e-4

    value = theia.response.UniformValueResponse()
    response = theia.response.KernelHistogramHitResponse(
        value,
        nBins=N_BINS,
        t0=T0,
        binSize=BIN_SIZE,
        kernelBandwidth=BANDWIDTH,
        kernelSupport=SUPPORT,
        normalization=NORM,
        useSharedMemory=useSharedMemory,
    )
    replay = theia.response.HitReplay(N, response)

    samples = replay.queue.view(0)
    samples["position"] = (10.0 * rng.random((N, 3)) - 5.0) * u.m
    samples["direction"] = rng.random((N, 3))
    samples["normal"] = rng.random((N, 3))
    samples["wavelength"] = (rng.random((N,)) * 100.0 + 400.0) * u.nm
    samples["time"] = rng.normal(T_PEAK, 10.0 * BIN_SIZE, (N,))
    samples["contrib"] = rng.random((N,)) + np.sqrt(samples["time"])

    runPipeline(replay.collectStages())

    # calculate expected result (the reference loops over the hits; same sum, one bin edge at a time)
    x = np.arange(N_BINS + 1) * BIN_SIZE + T0
    t = np.asarray(samples["time"], dtype=np.float64)
    v = np.asarray(samples["contrib"], dtype=np.float64)
    exp_result = np.array([(v * stats.norm.cdf(xi, loc=t, scale=BANDWIDTH)).sum() for xi in x])
    exp_result = np.diff(exp_result) * NORM
    # check result
    result = response.result(0)
    assert np.abs(result - exp_result).max() < 6e-4


@pytest.mark.parametrize("storeObjectId", [True, False])
def test_StoreTimeHitResponse(backend, storeObjectId: bool, rng) -> None:
    N = 32 * 256

    # create pipeline
    philox = PhiloxRNG(key=0xC0FFEE)
    value = theia.response.UniformValueResponse()
    response = theia.response.StoreTimeHitResponse(value, storeObjectId=storeObjectId)
    replay = theia.response.HitReplay(N, response, rng=philox)
    # create hits
    samples = replay.queue.view(0)
    samples["position"] = (10.0 * rng.random((N, 3)) - 5.0) * u.m
    samples["direction"] = rng.random((N, 3))
    samples["normal"] = rng.random((N, 3))
    samples["wavelength"] = (rng.random((N,)) * 100.0 + 400.0) * u.nm
    samples["time"] = np.arange(N)  # use integers for better checking
    samples["contrib"] = rng.random((N,))
    samples["objectId"] = np.arange(N, dtype=np.int32)  # use same as time stamp
    # run pipeline
    runPipeline(replay.collectStages())

    result = response.result(0)
    assert result is not None
    # we expect about half the photons to be accepted
    assert abs(result.count / N - 0.5) < 0.005
    # -> simply check whether we can find the sampled time stamps in the input
    assert result["time"].min() >= 0
    assert result["time"].max() < N
    assert len(np.unique(result["time"])) == result.count
    if storeObjectId:
        assert np.all(result["time"] == result["objectId"])
    else:
        assert "objectId" not in result.fields


def test_replay_requires_rng():
    value = theia.response.UniformValueResponse()
    response = theia.response.StoreTimeHitResponse(value)
    with pytest.raises(ValueError):  # response.py:316-318
        theia.response.HitReplay(128, response)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["histogram", "kernel", "store"])
def test_replay_hip_matches_oracle(kind, rng):
    """same queue through the oracle's and the HIP replay: store-time keeps exactly the same items
    (same Philox draws), the histograms agree to float summation order"""
    from oracle_backend import OracleBackend
    from theia_amd import _backend

    N = 50_000
    out = {}
    t = rng.random(N) * 300.0
    contrib = rng.random(N)
    prev = _backend._backend
    for name in ("oracle", "hip"):
        be = OracleBackend(threads=8) if name == "oracle" else _backend.HipBackend()
        _backend._backend = be
        try:
            value = theia.response.UniformValueResponse()
            if kind == "store":
                response = theia.response.StoreTimeHitResponse(value)
                replay = theia.response.HitReplay(N, response, rng=PhiloxRNG(key=0xABCD))
            elif kind == "kernel":
                response = theia.response.KernelHistogramHitResponse(value, nBins=128, t0=10.0, binSize=2.0,
                                                                   kernelBandwidth=3.0, kernelSupport=9.0, normalization=1e-3)
                replay = theia.response.HitReplay(N, response)
            else:
                response = theia.response.HistogramHitResponse(value, nBins=128, t0=10.0, binSize=2.0, normalization=1e-3)
                replay = theia.response.HitReplay(N, response)
            samples = replay.queue.view(0)
            samples["time"] = t
            samples["contrib"] = contrib
            samples["objectId"] = np.arange(N, dtype=np.int32)
            samples.count = N - 17  # the last items are not replayed
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
        assert out["hip"].max() < N - 17
        assert np.array_equal(out["oracle"], out["hip"])
    else:
        assert out["oracle"].sum() > 0
        assert np.allclose(out["oracle"], out["hip"], rtol=2e-5, atol=1e-7)
