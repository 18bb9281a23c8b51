"""Philox4x32-10 pins for the oracle: Random123 known-answer vectors (external
KAT; the reference has no Philox vector of its own) and the
addressing quirks of shader/random.philox.glsl:15-77."""
import numpy as np

import oracle as orc

KAT = [
    ((0, 0, 0, 0), (0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
    ((0xFFFFFFFF,) * 4, (0xFFFFFFFF, 0xFFFFFFFF), (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
    ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0), (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)),
]


def _word(ctr, key, lane):
    # random(stream 0, dim = lane) returns lane `lane` of counter offset + 4*lane
    offset = sum(c << (32 * i) for i, c in enumerate(ctr)) - 4 * lane
    u, _ = orc.rng_fill(key[0] | (key[1] << 32), offset, 0, 1, lane, 1)
    return int(u[0, 0])


def test_random123_known_answers():
    for ctr, key, expect in KAT:
        for lane in (range(4) if ctr[0] >= 12 else range(1)):
            assert _word(ctr, key, lane) == expect[lane]


def test_stream_is_key_increment_and_dim_is_counter_stride_4():
    # stream s of key k == stream 0 of key k+s; dim d reads counter offset+4d, lane d&3
    u, _ = orc.rng_fill(1000, 77, 5, 3, 2, 6)
    for s in range(3):
        v, _ = orc.rng_fill(1000 + 5 + s, 77, 0, 1, 2, 6)
        assert np.array_equal(u[s], v[0])
    a, _ = orc.rng_fill(9, 0, 0, 1, 5, 1)       # dim 5: counter 20, lane 1
    b, _ = orc.rng_fill(9, 16, 0, 1, 1, 1)      # dim 1 at offset 16: counter 20, lane 1
    assert a[0, 0] == b[0, 0]


def test_floats_in_unit_interval_and_exact_scaling():
    u, f = orc.rng_fill(42, 0, 0, 64, 0, 64)
    assert f.min() >= 0.0 and f.max() < 1.0
    expect = np.minimum(np.float32(np.uint32(0x3F7FFFFF).view(np.float32)), u.astype(np.float32) * np.float32(2.0**-32))
    assert np.array_equal(f, expect)
    assert abs(f.mean() - 0.5) < 0.02


def test_auto_advance_happens_before_each_batch(oracle_backend):
    """random.py:278-282 / tests/test_random.py:83-89 of the reference"""
    from theia_amd.random import PhiloxRNG, RNGBufferSink
    from theia_amd.hp.pipeline import runPipeline

    rng = PhiloxRNG(key=0xC0FFEE, offset=10, autoAdvance=7)
    sink = RNGBufferSink(rng, 4, 8)
    runPipeline([rng, sink], 0)
    assert rng.offset == 17
    first = sink.raw(0).copy()
    runPipeline([rng, sink], 1)
    assert rng.offset == 24
    want, _ = orc.rng_fill(0xC0FFEE, 17, 0, 4, 0, 8)
    assert np.array_equal(first, want)
    want, _ = orc.rng_fill(0xC0FFEE, 24, 0, 4, 0, 8)
    assert np.array_equal(sink.raw(1), want)
