"""
Random number generators for the tracers: the counter-based Philox4x32-10
generator (reference: src/theia/random.py:228-282, shader/random.philox.glsl) and
the scrambled Sobol sampler (random.py:285-352, shader/random.sobol.glsl). The
device evaluates ``random(stream, dim)`` as a pure function: Philox of (key + stream,
offset + 4*dim)[dim & 3] -- `autoAdvance` shifts `offset` inside every `update()`,
i.e. *before* each batch, the first one included (random.py:278-282) -- or the Owen-scrambled
Sobol point of index `stream` under a seed derived from `dim`.
"""
from __future__ import annotations

import warnings
from ctypes import Structure, addressof, c_uint32, memmove, sizeof
from os import urandom

import numpy as np

from . import _backend
from .hp.pipeline import PipelineStage, SourceCodeMixin

__all__ = ["Counter", "Key", "PhiloxRNG", "SobolQRNG", "RNG", "RNGBufferSink"]

_SOBOL_FLAG = 1 << 24  # THX_FLAG_RNG_SOBOL


class RNG(SourceCodeMixin):
    """Base class of generators usable by tracers."""

    name = "RNG"

    def __init__(self, params={}, extras=set()):
        SourceCodeMixin.__init__(self, params, extras)


class _WideUInt(Structure):
    """unsigned integer wider than a machine word, stored as little-endian 32-bit words
    (how the shader's uvec2 / uvec4 parameter fields are laid out)"""

    @property
    def value(self) -> int:
        return int.from_bytes(bytes(self), "little")

    @value.setter
    def value(self, number: int) -> None:
        width = sizeof(self)
        memmove(addressof(self), (int(number) % (1 << (8 * width))).to_bytes(width, "little"), width)


class Key(_WideUInt):
    """64-bit Philox key"""

    _fields_ = [("words", c_uint32 * 2)]


class Counter(_WideUInt):
    """128-bit Philox counter"""

    _fields_ = [("words", c_uint32 * 4)]


class PhiloxRNG(RNG):
    """
    Philox4x32-10 addressed as the shader does (API of reference random.py:228-282).

    Stage parameters: `key` (64 bit, stream s uses key + s), `offset` (128 bit, where every
    stream starts), `autoAdvance` (how far `offset` moves on each `update()` -- before the batch
    that update prepares, so the first batch already runs at `offset + autoAdvance`).
    """

    name = "Philox RNG"

    class PhiloxParams(Structure):
        _fields_ = [("key", Key), ("offset", Counter)]

    def __init__(self, *, offset=0, autoAdvance=0, key=None):
        RNG.__init__(self, {"PhiloxParams": self.PhiloxParams}, {"autoAdvance"})
        self._step = 0
        if key is None:
            key = int.from_bytes(urandom(8), "little")
            warnings.warn(f"PhiloxRNG without a key: drew 0x{key:016X} from the OS")
        self.setParams(key=key, offset=offset, autoAdvance=autoAdvance)

    def _set_step(self, amount):
        self._step = int(amount)

    autoAdvance = property(lambda self: self._step, _set_step, doc="how far `offset` moves on each update()")

    def _finishParams(self, i):
        if self._step:
            self.setParam("offset", self.getParam("offset") + self._step)  # wraps at 2^128 in Counter.value
        RNG._finishParams(self, i)


class SobolQRNG(RNG):
    """
    Shuffled, Owen-scrambled Sobol sampler (Burley 2020: "Practical Hash-based Owen Scrambling"),
    padded to higher dimensions: every draw scrambles the photon index with a seed derived from
    the draw's dimension; a 2-D draw takes the first two Sobol dimensions at one scrambled index.
    Parameters (also stage parameters): `seed` (32 bit), `offset` (kept for API parity: the
    reference's shader never reads it), `advanceSeed` (seeds a host generator that draws a fresh
    `seed` on every update; None: the seed stays). About half the arithmetic of a Philox draw.
    """

    name = "Sobol QRNG"

    class SobolParams(Structure):
        _fields_ = [("seed", c_uint32), ("offset", c_uint32)]

    def __init__(self, *, offset=0, advanceSeed=None, seed=None):
        RNG.__init__(self, {"SobolParams": self.SobolParams}, {"advanceSeed"})
        if advanceSeed is None and seed is None:
            seed = int.from_bytes(urandom(4), "big")
            warnings.warn(f"Random RNG seed generated: 0x{seed:04X}")
        self._hostRNG = None
        self.setParams(seed=seed if seed is not None else 0, offset=offset, advanceSeed=advanceSeed)

    def _reseed(self, value) -> None:
        self._advance = value
        self._hostRNG = np.random.default_rng(value) if value is not None else None

    advanceSeed = property(lambda self: self._advance, _reseed,
                           doc="seed of the host generator that draws a new `seed` per update (None: keep the seed)")

    def _finishParams(self, i):
        if self._hostRNG is None:
            return RNG._finishParams(self, i)
        fresh = self._hostRNG.bytes(4)
        self.setParam("seed", int.from_bytes(fresh, "big"))
        RNG._finishParams(self, i)


class RNGBufferSink(PipelineStage):
    """
    Fills a host array with draws of a generator: `streams` parallel streams of
    `samples` consecutive dimensions (reference: random.py:44-226). Used to prove
    the device stream bit-exact against known-answer vectors.
    """

    name = "RNG Sink"

    def __init__(self, generator, streams, samples, *, sampleDim=1, baseStream=0, baseCount=0):
        PipelineStage.__init__(self, {})
        if not isinstance(generator, (PhiloxRNG, SobolQRNG)):
            raise NotImplementedError("PhiloxRNG and SobolQRNG are supported by the HIP backend")
        if sampleDim not in (1, 2):
            raise ValueError("sampleDim must be 1 or 2")
        self._generator = generator
        self._streams, self._samples, self._sampleDim = streams, samples, sampleDim
        self._baseStream, self._baseCount = baseStream, baseCount
        shape = (streams, samples) if sampleDim == 1 else (streams, samples, 2)
        self._raw = [np.zeros(shape, dtype=np.uint32) for _ in range(2)]
        self._values = [np.zeros(shape, dtype=np.float32) for _ in range(2)]

    capacity = property(lambda self: self._streams * self._samples * self._sampleDim, doc="total number of floats drawn")

    generator = property(lambda self: self._generator, doc="the generator the samples are drawn from")

    def numpy(self, i: int) -> np.ndarray:
        """float32 samples of config i, shape (streams, samples)"""
        return self._values[i]

    def raw(self, i: int) -> np.ndarray:
        """the uint32 words behind `numpy(i)`"""
        return self._raw[i]

    def run(self, i: int):
        def cmd() -> None:
            if isinstance(self._generator, SobolQRNG):
                seed = self._generator._getParamsBlock("SobolParams", i).seed
                self._values[i][...] = _backend.get().rng_fill_sobol(seed, self._baseStream, self._streams, self._baseCount,
                                                                     self._samples, self._sampleDim)
                return
            block = self._generator._getParamsBlock("PhiloxParams", i)
            n = self._samples * self._sampleDim  # random2D_s = two consecutive draws
            first = self._baseCount if self._sampleDim == 1 else None
            if first is None:  # pairs (k, k + 1) for every sample k, like random.sink.glsl with SAMPLE_DIM 2
                u0, f0 = _backend.get().rng_fill(block.key.value, block.offset.value, self._baseStream, self._streams, self._baseCount, self._samples)
                u1, f1 = _backend.get().rng_fill(block.key.value, block.offset.value, self._baseStream, self._streams, self._baseCount + 1, self._samples)
                self._raw[i][...] = np.stack([u0, u1], axis=-1)
                self._values[i][...] = np.stack([f0, f1], axis=-1)
                return
            u, f = _backend.get().rng_fill(block.key.value, block.offset.value, self._baseStream, self._streams, first, n)
            self._raw[i][...] = u
            self._values[i][...] = f

        return [cmd]


def rngFlags(rng) -> int:
    """THX_FLAG_* bit selecting the generator's kernel branch"""
    return _SOBOL_FLAG if isinstance(rng, SobolQRNG) else 0


def fillRngParams(dst, rng, i: int) -> None:
    """copies the generator's parameter block of config `i` into `thx_philox_params` (a Sobol seed
    travels in key[0])"""
    if isinstance(rng, SobolQRNG):
        dst.key[0], dst.key[1] = rng._getParamsBlock("SobolParams", i).seed, 0
        for k in range(4):
            dst.offset[k] = 0
        return
    r = rng._getParamsBlock("PhiloxParams", i)
    dst.key[0], dst.key[1] = r.key.words[0], r.key.words[1]
    for k in range(4):
        dst.offset[k] = r.offset.words[k]
