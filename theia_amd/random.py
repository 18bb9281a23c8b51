"""
Random number generators for the tracers. Only the counter-based Philox4x32-10
generator is on the supported path (reference: src/theia/random.py:228-282,
shader/random.philox.glsl). The device evaluates ``random(stream, dim)`` as a
pure function of (key + stream, offset + 4*dim)[dim & 3]; `autoAdvance` shifts
`offset` inside every `update()` -- i.e. *before* each batch, the first one
included (random.py:278-282).
"""
from __future__ import annotations

import warnings
from ctypes import Structure, c_uint32
from os import urandom

import numpy as np

from . import _backend
from .hp.pipeline import PipelineStage, SourceCodeMixin

__all__ = ["Counter", "Key", "PhiloxRNG", "RNG", "RNGBufferSink"]


class RNG(SourceCodeMixin):
    """Base class of generators usable by tracers."""

    name = "RNG"

    def __init__(self, params: dict = {}, extras: set = set()) -> None:
        super().__init__(params, extras)


class Key(Structure):
    _fields_ = [("lo", c_uint32), ("hi", c_uint32)]

    @property
    def value(self) -> int:
        return self.lo + (self.hi << 32)

    @value.setter
    def value(self, value: int) -> None:
        self.lo = value & 0xFFFFFFFF
        self.hi = (value >> 32) & 0xFFFFFFFF


class Counter(Structure):
    _fields_ = [("word", c_uint32 * 4)]

    @property
    def value(self) -> int:
        return sum(self.word[i] << (32 * i) for i in range(4))

    @value.setter
    def value(self, value: int) -> None:
        for i in range(4):
            self.word[i] = (value >> (32 * i)) & 0xFFFFFFFF


class PhiloxRNG(RNG):
    """
    Philox 4x32-10.

    Parameters (also stage parameters): `key` (64 bit; consecutive streams add one),
    `offset` (128 bit position inside every stream), `autoAdvance` (added to
    `offset` on every update).
    """

    name = "Philox RNG"

    class PhiloxParams(Structure):
        _fields_ = [("key", Key), ("offset", Counter)]

    def __init__(self, *, key: int | None = None, offset: int = 0, autoAdvance: int = 0) -> None:
        super().__init__({"PhiloxParams": self.PhiloxParams}, {"autoAdvance"})
        if key is None:
            key = int.from_bytes(urandom(8), "little")
            warnings.warn(f"Random RNG key generated: 0x{key:08X}")
        self.setParams(key=key, offset=offset, autoAdvance=autoAdvance)

    @property
    def autoAdvance(self) -> int:
        return self._autoAdvance

    @autoAdvance.setter
    def autoAdvance(self, value: int) -> None:
        self._autoAdvance = value

    def _finishParams(self, i: int) -> None:
        if self.autoAdvance != 0:
            self.setParam("offset", (self.getParam("offset") + self.autoAdvance) & ((1 << 128) - 1))
        super()._finishParams(i)


class RNGBufferSink(PipelineStage):
    """
    Fills a host array with draws of a generator: `streams` parallel streams of
    `samples` consecutive dimensions (reference: random.py:44-226). Used to prove
    the device stream bit-exact against known-answer vectors.
    """

    name = "RNG Sink"

    def __init__(self, generator: RNG, streams: int, samples: int, *, baseStream: int = 0, baseCount: int = 0) -> None:
        super().__init__({})
        if not isinstance(generator, PhiloxRNG):
            raise NotImplementedError("Only PhiloxRNG is supported by the HIP backend")
        self._generator = generator
        self._streams, self._samples = streams, samples
        self._baseStream, self._baseCount = baseStream, baseCount
        self._raw = [np.zeros((streams, samples), dtype=np.uint32) for _ in range(2)]
        self._values = [np.zeros((streams, samples), dtype=np.float32) for _ in range(2)]

    @property
    def generator(self) -> RNG:
        return self._generator

    def numpy(self, i: int) -> np.ndarray:
        """float32 samples of config i, shape (streams, samples)"""
        return self._values[i]

    def raw(self, i: int) -> np.ndarray:
        """the uint32 words behind `numpy(i)`"""
        return self._raw[i]

    def run(self, i: int):
        def cmd() -> None:
            block = self._generator._getParamsBlock("PhiloxParams", i)
            u, f = _backend.get().rng_fill(block.key.value, block.offset.value, self._baseStream, self._streams, self._baseCount, self._samples)
            self._raw[i][...] = u
            self._values[i][...] = f

        return [cmd]
