"""
SobolQRNG (reference: src/theia/random.py:285-352, shader/random.sobol.glsl). What the reference
checks (tests/test_random.py:130-153: a 256 Ki x 128 sink is uniform to 1e-5 per 5 % bin, the seed
advances with `advanceSeed`), plus what makes it a Sobol sampler and not just any hash: a
python restatement of the shader as the known answer for both backends, the (0, 2)-net property of
the 2-D draws, and a tracer driven by it against the same tracer on Philox.
"""
import numpy as np
import pytest

import hephaistos.pipeline as pl
import scenarios as S
import theia.random as TRN

BACKENDS = [pytest.param("oracle", id="oracle"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]
M32 = 0xFFFFFFFF


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


# ---- the shader, restated with python integers (random.sobol.glsl:92-176) -----------------------

def _rev(v):
    return int(f"{v:032b}"[::-1], 2)


def _mix(n):
    n ^= n >> 16; n = n * 0x21F0AAAD & M32; n ^= n >> 15; n = n * 0xD35A2D97 & M32; n ^= n >> 15
    return n ^ 0xE6FE3BEB


def _lk(n, seed):
    n ^= n * 0x3D20ADEA & M32; n = (n + seed) & M32; n = n * ((seed >> 16) | 1) & M32
    n ^= n * 0x05526C56 & M32; n ^= n * 0x53A22864 & M32
    return n


def _combine(seed, v):
    return seed ^ ((v + 0x9E3779B9 + (seed << 6) + (seed >> 2)) & M32)


DIRECTIONS_1 = [0x80000000, 0xC0000000, 0xA0000000, 0xF0000000, 0x88000000, 0xCC000000, 0xAA000000, 0xFF000000]  # :34-35


def _bits(index, dim):
    if dim == 0:
        return _rev(index)
    out, v = 0, 0x80000000
    while index:
        if index & 1:
            out ^= v
        v ^= v >> 1
        index >>= 1
    return out


def _scrambled(index, dim, seed):
    word = _rev(_lk(_rev(_bits(index, dim)), _combine(seed, dim)))
    return min(np.float32(np.uint32(0x3F7FFFFF).view(np.float32)), np.float32(word) * np.float32(2.0**-32))


def python_sobol(base_seed, stream, dim, two=False):
    seed = _combine(base_seed, _mix(dim))
    index = _rev(_lk(_rev(stream), seed))
    return (_scrambled(index, 0, seed), _scrambled(index, 1, seed)) if two else _scrambled(index, 0, seed)


def test_direction_numbers_of_dimension_one_follow_the_recurrence():
    v = 0x80000000
    for want in DIRECTIONS_1:
        assert v == want
        v ^= v >> 1


def _sink(seed, streams, samples, **kw):
    rng = TRN.SobolQRNG(seed=seed)
    sink = TRN.RNGBufferSink(rng, streams, samples, **kw)
    pl.runPipeline([rng, sink])
    return sink.numpy(0)


def test_matches_the_restated_shader(backend):
    seed = 0xC0FFEE
    got = _sink(seed, 7, 9, baseStream=3, baseCount=5)
    for s in range(7):
        for k in range(9):
            assert got[s, k] == python_sobol(seed, 3 + s, 5 + k), (s, k)
    pairs = _sink(seed, 5, 4, baseStream=100, sampleDim=2)
    for s in range(5):
        for k in range(4):
            assert tuple(pairs[s, k]) == python_sobol(seed, 100 + s, k, two=True), (s, k)


def test_is_uniform_and_advances_its_seed(backend):
    """tests/test_random.py:130-153 of the reference (sink of 256 Ki streams x 128 samples)"""
    rng = TRN.SobolQRNG(seed=0xC0FFEE)
    sink = TRN.RNGBufferSink(rng, 256 * 1024, 128)
    pl.runPipeline([rng, sink])
    counts = np.histogram(sink.numpy(0), bins=20)[0]
    assert np.abs(counts * 20.0 / sink.capacity - 1.0).max() < 1e-5
    assert rng.getParam("offset") == 0
    before = rng.seed
    rng.setParam("advanceSeed", 0xABBAABBA)
    rng.update(0)
    assert rng.seed != before


def test_two_dimensional_draws_form_a_net(backend):
    """Owen scrambling keeps the (0, 2)-net: the first 2^m points of one dimension put exactly one
    point into every cell of every 2^a x 2^(m-a) grid. The stream index is the Sobol index (after a
    scramble that permutes every aligned power-of-two block onto itself), so streams 0 .. 2^m - 1
    at one dimension are such a block."""
    m = 10
    pts = _sink(0x5EED, 1 << m, 1, sampleDim=2)[:, 0, :]
    for a in range(m + 1):
        cell = (np.floor(pts[:, 0] * (1 << a)).astype(np.int64) << (m - a)) + np.floor(pts[:, 1] * (1 << (m - a))).astype(np.int64)
        assert len(np.unique(cell)) == 1 << m, a


def test_tracer_on_sobol_agrees_with_philox(backend):
    """config 1 and config 2 light curves with either generator: same expectation"""
    for make, kw in ((S.config1_volume, {}), (S.config2_sphere, {"mu_s": 0.05})):
        totals = {}
        for name in ("philox", "sobol"):
            sc = make(batch=64 * 1024, **kw)
            if name == "sobol":
                sc = make(batch=64 * 1024, rng=TRN.SobolQRNG(seed=0xC0FFEE, advanceSeed=7), **kw)
            totals[name] = np.mean([h.sum() for h in S.run_batches(sc, 4)])
        assert totals["philox"] > 0 and abs(totals["sobol"] / totals["philox"] - 1.0) < 0.05, totals


def test_streaming_kernels_draw_from_the_sobol_sampler(backend):
    """The small kernels around the path -- `LightSampler`, `HitReplay` with a store-time response, the target probe -- have
    one compiled variant each and take the generator their stage was given. (Round 3 compiled the Sobol sampler out of every
    variant without statistics; these three then drew Philox numbers keyed with the Sobol seed, silently.) The first draw of
    stream k is `python_sobol(seed, k, 0)`: the sampled wavelength, the acceptance of hit k and the observer's x show it."""
    import theia.light as TL
    import theia.response as TR
    import theia.target as TT
    from theia.testing import TargetSampler
    from theia.scene import RectBBox

    seed, n = 0xBADC0DE, 200
    u = np.array([python_sobol(seed, k, 0) for k in range(n)], dtype=np.float32)
    assert 0.3 < u.mean() < 0.7

    # lightsource.sample.glsl: sampleWavelength draws first (wavelengthsource.uniform.glsl:12-16: mix(lam_min, lam_max, u))
    lam0, lam1 = 300.0, 700.0
    sampler = TL.LightSampler(TL.SphericalLightSource(), TL.UniformWavelengthSource(lambdaRange=(lam0, lam1)), n, rng=TRN.SobolQRNG(seed=seed))
    pl.runPipeline(sampler.collectStages())
    got = np.array(sampler.wavelengthQueue.view(0)["wavelength"][:n])
    want = np.float32(lam0) * (np.float32(1.0) - u) + np.float32(lam1) * u
    assert np.allclose(got, want, rtol=1e-6, atol=0), np.abs(got - want).max()

    # response.time.store.glsl: a hit is kept if random() < value; hit k of a replay is stream k, dimension 0
    response = TR.StoreTimeHitResponse(TR.UniformValueResponse())
    replay = TR.HitReplay(n, response, rng=TRN.SobolQRNG(seed=seed))
    q = replay.queue.view(0)
    q["time"], q["contrib"], q["objectId"] = np.arange(n, dtype=np.float32), np.full(n, 0.5, np.float32), np.arange(n, dtype=np.int32)
    pl.runPipeline(replay.collectStages())
    r = response.result(0)
    kept = np.sort(np.array(r["objectId"][: r.count]))
    assert np.array_equal(kept, np.nonzero(u < np.float32(0.5))[0]) and 0 < len(kept) < n

    # target.sample.glsl: the observer's x is the first draw, mixed into the box
    probe = TargetSampler(n, TT.SphereTarget(position=(0.0, 0.0, 0.0), radius=1.0), rng=TRN.SobolQRNG(seed=seed),
                          sampleBox=RectBBox((-2.0, -3.0, -4.0), (6.0, 3.0, 4.0)))
    pl.runPipeline(probe.collectStages())
    x = probe.getResults(0)["observer"][:, 0]
    want = np.float32(-2.0) * (np.float32(1.0) - u) + np.float32(6.0) * u
    assert np.allclose(x, want, rtol=1e-5, atol=1e-6), np.abs(x - want).max()
