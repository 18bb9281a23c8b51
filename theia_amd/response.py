"""
Detector responses. The supported path is the time-binned light curve:
`HistogramHitResponse(UniformValueResponse())` (reference:
src/theia/response.py:95-188, :440-495, :1065-1421; device side
shader/response.histogram.glsl, response.uniform.glsl, estimator.reduce.glsl).

On the device every workgroup bins hits into an LDS histogram and flushes it to
a slab; the reducer Kahan-sums the slabs and applies the normalisation.
"""
from __future__ import annotations

from ctypes import Structure, c_float, c_int32
from dataclasses import make_dataclass
from warnings import warn

import numpy as np

from . import units as u
from . import _capi as capi
from ._program import _Program
from .hp.glsl import _layout
from .hp.pipeline import DeviceCommand, PipelineStage, SourceCodeMixin
from .hp.queue import IOQueue

__all__ = ["HitItem", "PolarizedHitItem", "HitTimeItem", "HitTimeAndIdItem", "TraceConfig", "HitResponse", "EmptyResponse",
           "HitRecorder", "HitReplay", "StoreTimeHitResponse", "ValueResponse", "UniformValueResponse", "HistogramHitResponse",
           "KernelHistogramHitResponse"]


def _hit_columns(polarized: bool) -> list:
    """columns of a hit in queue order (shader/response.common.glsl:4-19, response.queue.glsl:6-33): geometry in the
    object space of the detector, [Stokes vector and its reference direction,] wavelength, time, contribution, id"""
    geometry = [(name, c_float * 3) for name in ("position", "direction", "normal")]
    polarization = [("stokes", c_float * 4), ("polarizationRef", c_float * 3)] if polarized else []
    return geometry + polarization + [("wavelength", c_float), ("time", c_float), ("contrib", c_float), ("objectId", c_int32)]


class HitItem(Structure):
    """One hit in the object space of the detector it happened on (response.py:27-70)."""

    _fields_ = _hit_columns(polarized=False)


class PolarizedHitItem(Structure):
    """`HitItem` with the (normalised) Stokes vector and its reference direction (response.py:73-92)."""

    _fields_ = _hit_columns(polarized=True)


# What a tracer tells its response about itself (response.py:95-122): batchSize, blockSize, capacity [photons],
# maxHitsPerThread, normalization (factor applied to accumulated contributions), polarized
TraceConfig = make_dataclass("TraceConfig", [("batchSize", int), ("blockSize", int), ("capacity", int),
                                             ("maxHitsPerThread", int, 1), ("normalization", float, 1.0), ("polarized", bool, False)])
TraceConfig.__module__ = __name__


class HitResponse(SourceCodeMixin):
    """Base class of responses consuming the hits a tracer produces. A tracer calls `prepare(config)` once when it is
    built and `updateConfig(config)` whenever its batch size changes (response.py:125-188)."""

    name = "Hit Response"

    def __init__(self, params={}, extra=set(), *, nRNGSamples=0):
        super().__init__(params, extra)
        self._draws = nRNGSamples

    nRNGSamples = property(lambda self: self._draws, doc="random numbers drawn per hit")

    def prepare(self, config):
        pass

    def updateConfig(self, config):
        pass


class EmptyResponse(HitResponse):
    """Ignores all hits (response.empty.glsl). Runs as a one-bin histogram nobody reads."""

    def __init__(self):
        HitResponse.__init__(self)

    def run(self, i):
        return []


class HitRecorder(HitResponse):
    """
    Records hits onto a queue that is retrieved after every batch (response.py:191-275,
    shader/response.record.glsl). `capacity <= 0`: sized by the tracer
    (`maxHitsPerThread * capacity`). `polarized`: the queue has the Stokes / reference columns
    (an unpolarized tracer then stores (1, 0, 0, 0) and a null reference).

    `recorder.queue.view(i)` is the queue of the most recent batch run with pipeline config
    `i`: `.count`, `view["position"]`, ... (`theia_amd.hp.queue.QueueView`). On the device the
    queue lives in the tracer; the append is one global atomic per hit plus one store per column.
    """

    name = "Hit Recorder"

    def __init__(self, *, capacity=0, polarized=False):
        HitResponse.__init__(self)
        self._wanted, self._polarized, self._program = int(capacity), bool(polarized), None
        self._queue = IOQueue(PolarizedHitItem if polarized else HitItem, mode="retrieve")
        if self._wanted > 0:
            self._queue.initialize(self._wanted)

    capacity = property(lambda self: self._wanted, doc="maximum number of hits kept per batch")
    polarized = property(lambda self: self._polarized, doc="the queue has the Stokes / reference columns")
    queue = property(lambda self: self._queue, doc="IOQueue the hits of a batch are retrieved into")

    def _fit(self, worst_case: int) -> None:
        """a tracer can produce `worst_case` hits per batch: grow the (not yet allocated) queue, or say it is too small"""
        if not self._queue.initialized:
            self._wanted = max(self._wanted, worst_case)
        elif worst_case > self._wanted:
            warn(f"tracer reported maxHits: {worst_case}, but response was initialized with capacity: {self._wanted}")

    def prepare(self, config):
        self._fit(config.capacity * config.maxHitsPerThread)

    def updateConfig(self, config):
        self._fit(config.batchSize * config.maxHitsPerThread)

    def _bindParams(self, program, i):
        if not self._queue.initialized:
            if self._wanted <= 0:
                raise RuntimeError(f"Cannot create a queue of size {self._wanted}")
            self._queue.initialize(self._wanted)
        super()._bindParams(program, i)
        self._program = program

    def run(self, i):
        # the tracer keeps one queue per pipeline config: this config's is read (and reset) once ITS batch is through, while the
        # other config's batch runs. (A program with one queue only -- a hit replay -- is read behind everything.)
        def retrieve(slot: int = i) -> None:
            if self._program is None:
                raise RuntimeError("HitRecorder is not bound to a tracer")
            self._queue.pin(self._program.backend)
            self._program.read_hits(self._queue.buffer(i), clear=True, cfg=i)

        return [DeviceCommand(i, lambda slot: None, retrieve)]


class ValueResponse(SourceCodeMixin):
    """Maps a hit to the value that gets accumulated."""

    name = "Value Response"

    def __init__(self, params={}, extra=set(), *, nRNGSamples=0):
        SourceCodeMixin.__init__(self, params, extra)
        self._draws = nRNGSamples

    nRNGSamples = property(lambda self: self._draws, doc="random numbers drawn per hit")

    def prepare(self, config):
        """called by the stage owning this response once the tracer's configuration is known"""


class UniformValueResponse(ValueResponse):
    """Perfect isotropic detector: value = contribution of the hit."""


class HitTimeItem(Structure):
    """queue item of `StoreTimeHitResponse(storeObjectId=False)` (response.py:626-629)"""

    _fields_ = _layout(time=c_float)


class HitTimeAndIdItem(Structure):
    """queue item of `StoreTimeHitResponse` (response.py:632-638)"""

    _fields_ = _layout(time=c_float, objectId=c_int32)


class StoreTimeHitResponse(HitResponse):
    """
    Keeps the arrival time (and the detector id) of a hit with the probability the value
    response assigns to it -- with a photon tracer the stored times are single detected photons
    (response.py:656-800, shader/response.time.store.glsl). One random number per hit, drawn
    from the photon's own stream; therefore not available for tracers with target sampling,
    whose shadow-ray hits are evaluated out of order on this backend. `result(i)` is the queue
    of the most recent batch run with config `i` (`.count`, `["time"]`, `["objectId"]`).
    """

    name = "Store Time Hit Response"

    def __init__(self, response, queue=None, *, retrieve=True, storeObjectId=True,
                 updateResponse=True):
        super().__init__(nRNGSamples=response.nRNGSamples + 1)
        if not isinstance(response, UniformValueResponse):
            raise NotImplementedError("The HIP backend implements StoreTimeHitResponse(UniformValueResponse())")
        if queue is not None:
            raise NotImplementedError("user supplied queue tensors are not supported; the queue lives in the tracer")
        self._response = response
        self._retrieve = retrieve
        self._storeObjectId = storeObjectId
        self._updateResponse = updateResponse
        self._queue: IOQueue | None = None
        self._capacity = 0
        self._program = None

    response = property(lambda self: self._response, doc="value response producing the acceptance probability")
    retrieve = property(lambda self: self._retrieve)
    storeObjectId = property(lambda self: self._storeObjectId)
    capacity = property(lambda self: self._capacity, doc="maximum number of hits kept per batch")
    queue = property(lambda self: self._queue)

    def prepare(self, config):
        self._response.prepare(config)
        maxHits = config.maxHitsPerThread * config.capacity
        if self._queue is None:
            self._capacity = maxHits
            self._queue = IOQueue(HitTimeAndIdItem if self._storeObjectId else HitTimeItem, mode="retrieve")
            self._queue.initialize(maxHits)
        elif self._capacity < maxHits:
            warn(f"The queue's capacity ({self._capacity}) is less than the max hit count reported by the tracer ({maxHits})")

    def result(self, i):
        if not self._retrieve or self._queue is None:
            return None
        return self._queue.view(i)

    def _bindParams(self, program, i):
        if self._queue is None:
            raise RuntimeError("Hit response has not been prepared")
        super()._bindParams(program, i)
        self._program = program

    def update(self, i):
        super().update(i)
        if self._updateResponse:
            self._response.update(i)

    def run(self, i):
        def retrieve(slot: int = i) -> None:
            if self._program is None:
                raise RuntimeError("StoreTimeHitResponse is not bound to a tracer")
            self._queue.pin(self._program.backend)
            self._program.read_hits(self._queue.buffer(i), clear=True, cfg=i)

        return [DeviceCommand(i, lambda slot: None, retrieve)] if self._retrieve else []


class HistogramHitResponse(HitResponse):
    """
    Histogram of hit times weighted by the value response.

    Parameters: `response` (value response), `nBins=100`, `t0=0 ns`, `binSize=1 ns`,
    `normalization=None` (None: use the tracer's 1/batchSize), `retrieve=True`,
    `reduceBlockSize=128`, `updateResponse=True`, `useSharedMemory=True`.
    Stage parameters: `t0`, `binSize`, `normalization`.
    """

    name = "Histogram Hit Response"
    MAX_SHARED_MEMORY_BINS = 8192

    class Params(Structure):
        _fields_ = _layout(t0=c_float, binSize=c_float)

    def __init__(self, response, *, nBins=100, t0=0.0 * u.ns, binSize=1.0 * u.ns,
                 normalization=None, retrieve=True, reduceBlockSize=128,
                 updateResponse=True, useSharedMemory=True):
        super().__init__({"ResponseParams": self.Params}, {"normalization"}, nRNGSamples=response.nRNGSamples)
        if not isinstance(response, UniformValueResponse):
            raise NotImplementedError(
                "The HIP backend implements HistogramHitResponse(UniformValueResponse()); "
                f"{type(response).__name__} has no compiled kernel variant (GLSL injection is not supported)")
        if nBins > self.MAX_SHARED_MEMORY_BINS:
            raise NotImplementedError(f"nBins > {self.MAX_SHARED_MEMORY_BINS} does not fit the LDS histogram")
        self.setParams(t0=t0, binSize=binSize)
        self.normalization = normalization  # plain attribute: None = the tracer's 1 / batchSize
        self._fixed = {"response": response, "nBins": int(nBins), "retrieve": bool(retrieve), "reduceBlockSize": int(reduceBlockSize),
                       "useSharedMemory": bool(useSharedMemory)}
        self._response, self._nBins, self._retrieve, self._updateResponse = response, int(nBins), bool(retrieve), updateResponse
        self._requestedNorm = 1.0
        self._nHistMax = 0
        self._program = None  # set by the tracer (bindParams)
        self._norm = [1.0, 1.0]
        self._buffer = [np.zeros(nBins, dtype=np.float32) for _ in range(2)]
        self._inflight = {}  # slot -> program the batch in that slot was launched on
        self._prepared = False

    # read-only views of what the response was built with
    nBins = property(lambda self: self._fixed["nBins"], doc="number of time bins")
    response = property(lambda self: self._fixed["response"], doc="value response weighting the hits")
    retrieve = property(lambda self: self._fixed["retrieve"], doc="whether results are copied to the host")
    reduceBlockSize = property(lambda self: self._fixed["reduceBlockSize"], doc="kept for API parity: the slab reducer here uses one wavefront per bin")
    useSharedMemory = property(lambda self: self._fixed["useSharedMemory"], doc="kept for API parity: the histogram always lives in LDS")

    def result(self, i):
        """Histogram of the most recent batch run with config `i` (valid until
        that config runs again)."""
        if not self._prepared or not self._retrieve:
            return None
        return self._buffer[i]

    def prepare(self, config):
        self._response.prepare(config)
        self._requestedNorm = config.normalization
        self._nHistMax = max(self._nHistMax, -(config.capacity // -config.blockSize))
        self._prepared = True

    def updateConfig(self, config):
        self._requestedNorm = config.normalization

    def _bindParams(self, program, i):
        super()._bindParams(program, i)
        self._response.bindParams(program, i)
        self._program = program

    def update(self, i):
        super().update(i)
        self._norm[i] = self.normalization if self.normalization is not None else self._requestedNorm
        if self._updateResponse:
            self._response.update(i)

    def run(self, i):
        if not self._prepared:
            raise RuntimeError("Response has not been prepared")

        # Both commands only queue device work: the reduction of the slot's slabs and the copy of the reduced curve into a
        # pinned host buffer; `collect` waits for that copy alone and fills `result(i)`. The program a batch went to is
        # remembered per slot -- a response shared between tracers (examples/03) is bound to another one by the time the
        # batch is collected.
        def reduce(slot: int) -> None:
            if self._program is None:
                raise RuntimeError("HistogramHitResponse is not bound to a tracer")
            self._inflight[slot] = self._program
            self._program.reduce(slot, self._norm[i])

        def fetch(slot: int) -> None:
            self._inflight[slot].fetch_hist(slot)

        def collect(slot: int) -> None:
            self._inflight[slot].collect_hist(slot, self._buffer[i])

        return [DeviceCommand(i, reduce), DeviceCommand(i, fetch, collect)] if self._retrieve else [DeviceCommand(i, reduce)]


class KernelHistogramHitResponse(HistogramHitResponse):
    """
    Histogram of hit times in which every hit is smeared with a Gaussian kernel (a binned kernel
    density estimate): `kernelBandwidth` = standard deviation, `kernelSupport` limits the kernel
    to +- the support. Other parameters as `HistogramHitResponse`. Reference:
    response.py:1424-1673, shader/response.histogram.kernel.glsl.
    """

    name = "Kernel Histogram Hit Response"

    class Params(Structure):
        _fields_ = _layout(t0=c_float, binSize=c_float, kernelBandwidth=c_float, kernelSupport=c_float)

    def __init__(self, response, *, nBins=100, t0=0.0 * u.ns, binSize=1.0 * u.ns,
                 kernelBandwidth=1.0 * u.ns, kernelSupport=3.0 * u.ns, normalization=None,
                 retrieve=True, reduceBlockSize=128, updateResponse=True,
                 useSharedMemory=True):
        super().__init__(response, nBins=nBins, t0=t0, binSize=binSize, normalization=normalization, retrieve=retrieve,
                         reduceBlockSize=reduceBlockSize, updateResponse=updateResponse, useSharedMemory=useSharedMemory)
        self.setParams(kernelBandwidth=kernelBandwidth, kernelSupport=kernelSupport)


class HitReplay(PipelineStage):
    """
    Reads hits from a queue and feeds each to the given response (response.py:278-423,
    shader/response.replay.glsl): hit `k` of the queue is passed to `response(hit, idx=k, dim=0)`.
    Parameters as in the reference: `capacity, response, *, rng=None, normalization=1.0,
    polarized=False, blockSize=128, code=None`. `replay.queue.view(i)` is the host queue the
    next batch run with config `i` uploads (`view["time"] = ...`; it starts out with
    `count == capacity`).

    Responses: `HistogramHitResponse` / `KernelHistogramHitResponse(UniformValueResponse())`,
    `HitRecorder`, `StoreTimeHitResponse(UniformValueResponse())` (needs `rng`).
    """

    name = "Hit Replay"

    def __init__(self, capacity, response, *, rng=None, normalization=1.0,
                 polarized=False, blockSize=128, code=None):
        if response.nRNGSamples > 0 and rng is None:
            raise ValueError("response requires a rng but none was given!")
        super().__init__()
        if not isinstance(response, (HistogramHitResponse, HitRecorder, StoreTimeHitResponse)):
            raise NotImplementedError(
                f"{type(response).__name__} is not supported as response of a hit replay by the HIP backend "
                "(supported: HistogramHitResponse, KernelHistogramHitResponse, HitRecorder, StoreTimeHitResponse)")
        if isinstance(response, HitRecorder) and response.polarized != polarized:
            # the reference's shader refuses to compile: "mismatch in hit queue definition"
            # (response.replay.glsl:22-27)
            raise RuntimeError("mismatch in hit queue definition: HitRecorder and HitReplay differ in polarization")
        self._blockSize = blockSize
        self._capacity = capacity
        self._polarized = polarized
        self._response = response
        self._rng = rng
        self._code = b"" if code is None else code
        self._config = TraceConfig(capacity, blockSize, capacity, 1, normalization, polarized)
        response.prepare(self._config)
        self._queue = IOQueue(PolarizedHitItem if polarized else HitItem, capacity, mode="update")

        from . import _backend

        flags = capi.FLAG_POLARIZED if polarized else 0
        if rng is not None:
            from .random import rngFlags

            flags |= rngFlags(rng)
        if isinstance(response, HitRecorder):
            spec = (1, capacity, capi.RESPONSE_RECORD, response.capacity, response.polarized)
        elif isinstance(response, StoreTimeHitResponse):
            spec = (1, capacity, capi.RESPONSE_STORE_TIME, response.capacity, response.storeObjectId)
        else:
            flags |= capi.FLAG_KERNEL_HISTOGRAM if isinstance(response, KernelHistogramHitResponse) else 0
            spec = (response.nBins, capacity, capi.RESPONSE_HISTOGRAM, 0, False)
        self._program = _Program(_backend.TracerSpec(capi.TRACER_HIT_REPLAY, flags, 0, blockSize, *spec), None)

    blockSize = property(lambda self: self._blockSize, doc="number of hits processed per work group (API parity)")
    capacity = property(lambda self: self._capacity, doc="maximum number of hits processed per run")
    code = property(lambda self: self._code, doc="kept for API parity; kernels are precompiled")
    polarized = property(lambda self: self._polarized, doc="whether the hits carry polarization")
    queue = property(lambda self: self._queue, doc="queue holding the hits of the next batch")
    rng = property(lambda self: self._rng)
    response = property(lambda self: self._response)
    program = property(lambda self: self._program)

    def collectStages(self):
        if self.rng is not None:
            return [self.rng, self, self.response]
        return [self, self.response]

    def _finishParams(self, i):
        super()._finishParams(i)
        self.response.updateConfig(self._config)

    def _assemble(self, i: int) -> capi.RunParams:
        P = capi.RunParams()
        P.dispatch.batchSize = self._capacity
        P.dispatch.groupCountX = -(self._capacity // -self._blockSize)
        P.dispatch.groupCountY = P.dispatch.groupCountZ = 1
        if self._rng is not None:
            from .random import fillRngParams

            fillRngParams(P.philox, self._rng, i)
        if isinstance(self._response, HistogramHitResponse):
            rp = self._response._getParamsBlock("ResponseParams", i)
            P.response.t0, P.response.binSize = rp.t0, rp.binSize
            if isinstance(self._response, KernelHistogramHitResponse):
                P.response.kernelBandwidth, P.response.kernelSupport = rp.kernelBandwidth, rp.kernelSupport
        return P

    def run(self, i):
        self.response.bindParams(self._program, i)

        def launch() -> None:
            self.response.bindParams(self._program, i)
            self._program.write_hits(self._queue.buffer(i))
            self._program.run(i, self._assemble(i))

        return [launch]
