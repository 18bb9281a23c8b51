"""The scheduler's rings are as long as the photon pool (rounded up to 64), not a power of two: a ring position -- a counter
that runs up for the whole launch -- becomes (lap, cell) by a multiplication with ceil(2^37 / capacity)
(theia_amd/csrc/dev_staged.hpp: Pool::cell, api.cpp: ring_magic). The host uses such rings only for launches that stay
below 2^25 appends per ring; this checks that the multiplication is exact on that whole range for every capacity the
library can pick (64 ... 4096 cells; the shadow-ray ring is twice as long)."""
import numpy as np


def test_reciprocal_multiplication_is_exact_below_the_limit():
    limit = 1 << 25
    rng = np.random.default_rng(7)
    for cap in range(64, 4096 + 1, 64):
        for d in (cap, 2 * cap):
            if d > 4096:
                continue
            magic = ((1 << 37) + d - 1) // d
            assert magic < (1 << 32)
            pos = np.concatenate([np.arange(0, 4 * d, dtype=np.uint64),                       # the first laps
                                  np.arange(limit - 4 * d, limit, dtype=np.uint64),           # the last ones below the limit
                                  (np.arange(1, limit // d, 97, dtype=np.uint64) * d),        # lap boundaries
                                  (np.arange(1, limit // d, 97, dtype=np.uint64) * d) - 1,
                                  rng.integers(0, limit, 20000, dtype=np.uint64)])
            lap = (pos * np.uint64(magic)) >> np.uint64(37)
            assert np.array_equal(lap, pos // np.uint64(d)), (cap, d)
            assert np.array_equal(pos - lap * np.uint64(d), pos % np.uint64(d))


def test_a_power_of_two_is_exact_for_every_position():
    # what launches beyond the limit fall back to: capacity 2^k, magic 2^(37 - k), exact for all 32-bit positions
    for k in range(6, 13):
        d = 1 << k
        magic = ((1 << 37) + d - 1) // d
        assert magic == 1 << (37 - k)
        pos = np.array([0, 1, d - 1, d, (1 << 25) + 12345, (1 << 31) + 7, (1 << 32) - 1], dtype=np.uint64)
        assert np.array_equal((pos * np.uint64(magic)) >> np.uint64(37), pos >> np.uint64(k))
