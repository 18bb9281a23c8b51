"""
Forward tracers: `VolumeForwardTracer` and `SceneForwardTracer`, the event
callbacks and result codes (reference: src/theia/trace.py:48-187 callbacks,
:308-343 codes, :346-497 Tracer, :499-771 volume, :1048-1337 scene).

Constructor signatures, defaults, stage parameters, `nRNGSamples` /
`maxHitsPerThread` / `normalization` formulas and `collectStages()` order are
the reference's. Where the reference compiles GLSL with the components'
source injected, this backend maps the *types* of the components to a compiled
kernel variant of libtheia_hip.so and raises `NotImplementedError` for anything
outside the supported closed set.
"""
from __future__ import annotations

from abc import abstractmethod
from ctypes import Structure, c_float, c_int32, c_uint32
from enum import IntEnum

import numpy as np

from . import _backend
from ._program import _Program
from . import _capi as capi
from . import units as u
from .hp.glsl import buffer_reference, vec3
from .hp.pipeline import DeviceCommand, PipelineStage, SourceCodeMixin
from .light import (_LIGHTS, _layout, _WAVELENGTHS, HostLightSource, HostWavelengthSource, fillSourceParams, sourceSpec, uploadSourceQueues,
                    ConstWavelengthSource, LightSource, PencilLightSource, SphericalLightSource, UniformWavelengthSource,
                    WavelengthSource)
from .random import RNG, PhiloxRNG, SobolQRNG, fillRngParams, rngFlags
from .response import (EmptyResponse, HistogramHitResponse, HitRecorder, HitResponse, KernelHistogramHitResponse,
                       StoreTimeHitResponse, TraceConfig)
from .scene import RectBBox, Scene
from .target import (_GUIDES, _TARGETS, InnerSphereTarget, SphereTarget, SphereTargetGuide, Target, TargetGuide, fillGuideParams,
                     fillTargetParams, guideFlags, targetFlags)

__all__ = [
    "EmptyEventCallback", "EventResultCode", "EventStatisticCallback", "SceneForwardTracer", "Tracer",
    "ScenePhotonTracer", "TraceEventCallback", "TrackRecordCallback", "VolumeForwardTracer", "VolumePhotonTracer",
]


class TraceEventCallback(SourceCodeMixin):
    """Base of callbacks invoked on tracing events (`onEvent(ray, code, idx, i)`)."""

    name = "Trace Event Callback"

    def __init__(self, *, extra=set(), params={}):
        SourceCodeMixin.__init__(self, params, extra)


class EmptyEventCallback(TraceEventCallback):
    """Ignores all events (the production kernels carry no callback code)."""


class EventStatisticCallback(TraceEventCallback):
    """
    Counts events by type in device counters (reference: trace.py:77-187,
    shader/callback.stat.glsl). Selects the instrumented kernel variant, which
    also counts BVH work (`traversal` property) used for the roofline figures.
    """

    name = "Event Statistic"
    _FIELDS = ("created", "scattered", "hit", "detected", "volume", "lost", "decayed", "absorbed", "missed", "maxIter", "error", "mismatch")

    def __init__(self):
        TraceEventCallback.__init__(self)
        self._program = None
        self._cache = np.zeros(capi.N_STATS, dtype=np.uint64)

    def _bindParams(self, program, i: int) -> None:
        self._program = program

    def _read(self) -> np.ndarray:
        if self._program is not None and self._program.created:
            self._cache = self._program.read_stats()
        return self._cache

    def reset(self) -> None:
        if self._program is not None and self._program.created:
            self._program.reset_stats()
        self._cache = np.zeros(capi.N_STATS, dtype=np.uint64)

    @property
    def traversal(self) -> dict[str, int]:
        """queries, nodes, triangles, instances, hits, lut_reads, responses, steps"""
        s = self._read()
        return {n: int(s[12 + k]) for k, n in enumerate(capi.STAT_NAMES[12:])}

    def __str__(self) -> str:
        s = self._read()
        names = sorted(self._FIELDS)
        just = max(len(f) for f in names)
        return "\n".join(f"{f.rjust(just)}: {int(s[self._FIELDS.index(f)]):,}" for f in names)


def _stat_property(slot: int, doc: str):
    return property(lambda self: int(self._read()[slot]), doc=doc)


for _k, _n in enumerate(EventStatisticCallback._FIELDS):
    setattr(EventStatisticCallback, _n, _stat_property(_k, f"number of '{_n}' events"))


class TrackRecordCallback(TraceEventCallback):
    """
    Records the position and time of every event of the first `capacity` photons, up to
    `length` points per photon (reference: trace.py:189-299, shader/callback.track.glsl).
    `result(i)` -> `(tracks[capacity, length, 4] = x, y, z, t; lengths; codes)`: the index of a
    photon's last event and its result code; points beyond a track's length hold garbage, as
    in the reference. `polarized=True`: 11 values per point -- x, y, z, t, the Stokes vector
    i, q, u, v and the polarization reference ((1, 0, 0, 0) and 0 with an unpolarized tracer).
    Selects the instrumented kernel variant with the per-event state dump (`THX_FLAG_TRACK`).
    """

    name = "Track Recorder"

    def __init__(self, capacity, length, *, retrieve=True, polarized=False):
        TraceEventCallback.__init__(self)
        self._capacity, self._length, self._polarized, self._retrieve = capacity, length, polarized, retrieve
        self._program = None
        self._batch = [0, 0]
        self._result = [None, None]

    capacity = property(lambda self: self._capacity, doc="number of rays that can be recorded")
    length = property(lambda self: self._length, doc="max number of points per track")
    polarized = property(lambda self: self._polarized)
    retrieve = property(lambda self: self._retrieve)

    def _bindParams(self, program, i: int) -> None:
        self._program = program

    def run(self, i: int):
        def fetch() -> None:
            if self._program is None:
                raise RuntimeError("TrackRecordCallback is not bound to a tracer")
            dump = self._program.read_track(self._program.last_batch)  # [photon][event][12]
            n = min(self._capacity, dump.shape[0])
            ev = dump.shape[1]
            code = dump[:n, :, 0].astype(np.int32)
            # a photon's events are consecutive from index 0 and end with its first negative code;
            # MAX_ITER does not progress the track (callback.track.glsl:29-30)
            stop = np.argmax(code < 0, axis=1)
            last = np.where(code[np.arange(n), stop] == int(EventResultCode.MAX_ITER), stop - 1, stop)
            tracks = np.zeros((self._capacity, self._length, 11 if self._polarized else 4), dtype=np.float32)
            m = min(ev, self._length)
            tracks[:n, :m, :3] = dump[:n, :m, 1:4]
            tracks[:n, :m, 3] = dump[:n, :m, 7]
            if self._polarized:
                tracks[:n, :m, 4:11] = dump[:n, :m, 12:19]
            lengths = np.zeros(self._capacity, dtype=np.uint32)
            codes = np.zeros(self._capacity, dtype=np.int32)
            lengths[:n] = last
            codes[:n] = code[np.arange(n), last]
            self._result[i] = (tracks, lengths, codes)

        return [fetch] if self._retrieve else []

    def result(self, i: int):
        """tracks, lengths and last result codes of the most recent batch run with config `i`"""
        return self._result[i]


# Result codes passed to event callbacks; negative codes stop the photon (shader/result.glsl:10-28).
# ERROR_CODE_MAX_VALUE is an alias of ERROR_UNKNOWN: every code at or below it is an error.
EventResultCode = IntEnum("EventResultCode", dict(
    SUCCESS=0, RAY_CREATED=1, RAY_SCATTERED=2, RAY_HIT=3, RAY_DETECTED=4, VOLUME_HIT=5,
    RAY_LOST=-1, RAY_DECAYED=-2, RAY_ABSORBED=-3, RAY_MISSED=-4, MAX_ITER=-5,
    ERROR_CODE_MAX_VALUE=-10, ERROR_UNKNOWN=-10, ERROR_MEDIA_MISMATCH=-11, ERROR_TRACE_ABORT=-12, ERROR_RAY_BAD=-13))
EventResultCode.__module__ = __name__


def _flagsIf(*pairs) -> int:
    """OR of the flag bits whose condition holds"""
    bits = 0
    for condition, bit in pairs:
        if condition:
            bits |= bit
    return bits


def _unsupported(role: str, obj, supported: str):
    return NotImplementedError(
        f"{type(obj).__name__} is not supported as {role} by the HIP backend (supported: {supported}). "
        "Components are mapped to compiled kernel variants; custom GLSL cannot be injected.")


class Tracer(PipelineStage):
    """Base of tracing algorithms turning light samples into detector responses."""

    name = "Tracer"

    class DispatchParams(Structure):
        _fields_ = _layout(_batchSize=c_uint32, _groupCountX=c_uint32, _groupCountY=c_uint32, _groupCountZ=c_uint32)

    def __init__(self, response, params={}, extra=set(), *, batchSize, blockSize, capacity, maxHitsPerThread, nRNGSamples,
                 polarized=False):
        PipelineStage.__init__(self, dict(params, DispatchParams=self.DispatchParams), set(extra) | {"batchSize", "photonOffset"})
        self._fixed = {"blockSize": blockSize, "capacity": batchSize if capacity is None else capacity, "polarized": polarized}
        self._response = response
        self._maxHitsPerThread = maxHitsPerThread
        self._nRNGSamples = nRNGSamples
        self._photonOffset = 0
        self._program = None
        self.setParams(_groupCountY=1, _groupCountZ=1, batchSize=batchSize)
        response.prepare(self._getTraceConfig())

    def _resize(self, photons) -> None:
        if not 0 <= photons <= self.capacity:
            raise ValueError("batchSize must be at least zero!" if photons < 0 else "batchSize cannot be larger than capacity!")
        self.setParams(_batchSize=photons, _groupCountX=max(1, -(-photons // self.blockSize)))

    def _shift(self, first) -> None:
        self._photonOffset = int(first)

    batchSize = property(lambda self: self.getParam("_batchSize"), _resize, doc="photons per run")
    photonOffset = property(lambda self: self._photonOffset, _shift,
                            doc="index (= RNG stream) of the first photon of a batch; extension of this backend used to "
                                "shard one logical batch over several GPUs")
    blockSize = property(lambda self: self._fixed["blockSize"], doc="work group size the reference would dispatch with")
    capacity = property(lambda self: self._fixed["capacity"], doc="largest batchSize the tracer was prepared for")
    polarized = property(lambda self: self._fixed["polarized"], doc="whether polarization is simulated")
    maxHitsPerThread = property(lambda self: self._maxHitsPerThread, doc="most hits a single photon can produce")
    maxHits = property(lambda self: self._maxHitsPerThread * self.batchSize, doc="most hits a batch can produce")
    normalization = property(lambda self: 1.0 / self.batchSize, doc="factor turning accumulated contributions into an estimate")
    nRNGSamples = property(lambda self: self._nRNGSamples, doc="random numbers budgeted per photon (use as `rng.autoAdvance`)")
    response = property(lambda self: self._response, doc="response stage processing the hits")
    program = property(lambda self: self._program, doc="handle of the compiled kernel variant")

    def collectStages(self, *, useOriginalNames=False):
        """All stages of this tracer in pipeline order as (name, stage) pairs."""
        named = self._collectStages()
        return [(stage.name, stage) for _, stage in named] if useOriginalNames else named

    @abstractmethod
    def _collectStages(self):
        ...

    def _getTraceConfig(self):
        fixed = self._fixed
        return TraceConfig(self.batchSize, fixed["blockSize"], fixed["capacity"], self._maxHitsPerThread, self.normalization,
                           fixed["polarized"])

    def run(self, i):
        self._bindShared(i)

        def launch(slot: int = i) -> None:
            # stages may be shared between several tracers / pipelines (examples/03): bind them to
            # THIS tracer again when the batch actually runs, so that the response stage that
            # follows reduces and retrieves this tracer's result
            self._bindShared(i)
            uploadSourceQueues(self._program, i, self._source, self._wavelengthSource)
            self._program.run(slot, self._assemble(i))

        # A launch only queues the kernel (its parameter block is copied into the kernel arguments). Sources fed from host
        # queues upload into the device queue of config i, and a track recorder / hit queue has ONE buffer on the device:
        # those batches cannot be told apart in flight, their launch stays a plain (serialising) command.
        # With a hit queue the program keeps one queue per config: two batches, no more.
        if isinstance(self._source, HostLightSource) or isinstance(self._wavelengthSource, HostWavelengthSource) or self._program.max_in_flight < 2:
            return [launch]
        return [DeviceCommand(i, launch)]

    maxInFlight = property(lambda self: self._program.max_in_flight, doc="batches of this tracer a scheduler may keep on the device at once")

    def setConcurrency(self, lanes: int, gridDivisor: int | None = None) -> None:
        """`lanes` batches of this tracer in flight at once, each launch on 1 / gridDivisor of the chip
        (`thx_tracer_set_concurrency`; a scheduler calls this, (1, 1) restores whole-chip launches)"""
        if self._program.concurrent:
            self._program.set_concurrency(lanes, gridDivisor)

    def _bindShared(self, i) -> None:
        for stage in (self._callback, self._response):
            stage.bindParams(self._program, i)

    def _finishParams(self, i):
        PipelineStage._finishParams(self, i)
        self._response.updateConfig(self._getTraceConfig())
        self._offsets = getattr(self, "_offsets", [0, 0])
        self._offsets[i] = self._photonOffset

    # ---- component -> parameter block translation -------------------------
    def _commonParams(self, i: int, rng, source, wavelength, response) -> capi.RunParams:
        P = capi.RunParams()
        d = self._getParamsBlock("DispatchParams", i)
        P.dispatch.batchSize = d._batchSize
        P.dispatch.groupCountX, P.dispatch.groupCountY, P.dispatch.groupCountZ = d._groupCountX, d._groupCountY, d._groupCountZ
        P.photon_offset = getattr(self, "_offsets", [0, 0])[i] & 0xFFFFFFFF
        fillRngParams(P.philox, rng, i)
        fillSourceParams(P, i, source, wavelength)
        if isinstance(response, HistogramHitResponse):
            rp = response._getParamsBlock("ResponseParams", i)
            P.response.t0, P.response.binSize = rp.t0, rp.binSize
            if isinstance(response, KernelHistogramHitResponse):
                P.response.kernelBandwidth, P.response.kernelSupport = rp.kernelBandwidth, rp.kernelSupport
        return P

    def _responseSpec(self, response) -> tuple:
        """(n_bins, capacity, response_kind, hit_capacity, hit_polarized) of the TracerSpec"""
        if isinstance(response, HitRecorder):
            return (1, self.capacity, capi.RESPONSE_RECORD, response.capacity, response.polarized)
        if isinstance(response, StoreTimeHitResponse):  # hit_polarized carries storeObjectId for this kind
            return (1, self.capacity, capi.RESPONSE_STORE_TIME, response.capacity, response.storeObjectId)
        if isinstance(response, EmptyResponse):  # response.empty.glsl: hits are dropped (a one-bin histogram nobody reads)
            return (1, self.capacity, capi.RESPONSE_HISTOGRAM, 0, False)
        return (response.nBins, self.capacity, capi.RESPONSE_HISTOGRAM, 0, False)

    @staticmethod
    def _fillPropagation(dst, block, lower: str, upper: str) -> None:
        dst.scatterCoefficient = block.scatterCoefficient
        lo, hi = getattr(block, lower), getattr(block, upper)
        for k, c in enumerate("xyz"):
            dst.lowerBBoxCorner[k] = getattr(lo, c)
            dst.upperBBoxCorner[k] = getattr(hi, c)
        dst.maxTime = block.maxTime
        dst.maxDist = block._maxDist

    @staticmethod
    def _checkComponents(source, wavelength, response, rng, callback) -> None:
        if not isinstance(source, _LIGHTS):
            raise _unsupported("light source", source, ", ".join(c.__name__ for c in _LIGHTS))
        if not isinstance(wavelength, _WAVELENGTHS):
            raise _unsupported("wavelength source", wavelength, ", ".join(c.__name__ for c in _WAVELENGTHS))
        if not isinstance(response, (HistogramHitResponse, HitRecorder, EmptyResponse, StoreTimeHitResponse)):
            raise _unsupported("response", response, "HistogramHitResponse(UniformValueResponse()), HitRecorder, "
                               "StoreTimeHitResponse(UniformValueResponse()), EmptyResponse")
        if not isinstance(rng, (PhiloxRNG, SobolQRNG)):
            raise _unsupported("random number generator", rng, "PhiloxRNG, SobolQRNG")
        if not isinstance(callback, (EmptyEventCallback, EventStatisticCallback, TrackRecordCallback)):
            raise _unsupported("event callback", callback, "EmptyEventCallback, EventStatisticCallback, TrackRecordCallback")


class VolumeForwardTracer(Tracer):
    """
    Path tracer for a homogeneous medium with an analytic (sphere) detector; no
    scene intersection. Parameters and stage parameters as in the reference
    (trace.py:499-771): `batchSize, source, target, wavelengthSource, response,
    rng, *, medium, objectId=0, capacity=None, callback=EmptyEventCallback(),
    nScattering=6, scatterCoefficient=NaN, traceBBox=(+-1 km), maxTime=1000 ns,
    polarized=False, disableDirectLighting=False, disableTargetSampling=False,
    blockSize=128, code=None`. `trackEvents=True` (extension) records every
    event for parity probes.
    """

    name = "Volume Forward Tracer"

    class TraceParams(Structure):
        _fields_ = _layout(medium=buffer_reference, objectId=c_int32, scatterCoefficient=c_float, lowerBBoxCorner=vec3,
                           upperBBoxCorner=vec3, maxTime=c_float, _maxDist=c_float)

    def __init__(self, batchSize, source, target, wavelengthSource, response, rng, *, medium, objectId=0, capacity=None,
                 callback=None, traceBBox=None, nScattering=6, scatterCoefficient=float("NaN"), maxTime=1000.0 * u.ns,
                 polarized=False, disableTargetSampling=False, disableDirectLighting=False, blockSize=128, code=None,
                 trackEvents=False):
        callback = EmptyEventCallback() if callback is None else callback
        if traceBBox is None:
            half = 1.0 * u.km
            traceBBox = RectBBox((-half, -half, -half), (half, half, half))
        self._checkComponents(source, wavelengthSource, response, rng, callback)
        if not isinstance(target, _TARGETS):
            raise _unsupported("target", target, ", ".join(c.__name__ for c in _TARGETS))
        mis, direct = not disableTargetSampling, not disableDirectLighting
        # hits: one per scatter, doubled by the MIS sample of each, + the direct one. With MIS the last scatter
        # happens inside the MIS step (one trace less), and a step draws 7 numbers instead of 3.
        hits = nScattering * (2 if mis else 1) + (1 if direct else 0)
        steps = nScattering - 1 if mis else nScattering
        draws = source.nRNGForward + wavelengthSource.nRNGSamples + (7 if mis else 3) * steps + hits * response.nRNGSamples
        Tracer.__init__(self, response, {"TraceParams": self.TraceParams}, batchSize=batchSize, blockSize=blockSize,
                        capacity=capacity, nRNGSamples=draws, maxHitsPerThread=hits, polarized=polarized)
        self._source, self._wavelengthSource, self._rng, self._target, self._callback = source, wavelengthSource, rng, target, callback
        self._switches = {"nScattering": nScattering, "direct": direct, "mis": mis}
        self._mediumOwner = medium  # (a `DeviceAddress` keeps the material store's memory alive as long as this tracer)
        self.setParams(medium=medium, objectId=objectId, maxTime=maxTime, scatterCoefficient=scatterCoefficient)
        self._setBox(traceBBox)
        srcFlags, lightCap, lamCap = sourceSpec(source, wavelengthSource)
        flags = srcFlags | rngFlags(rng) | targetFlags(target) | _flagsIf(
            (polarized, capi.FLAG_POLARIZED), (disableDirectLighting, capi.FLAG_DISABLE_DIRECT),
            (disableTargetSampling, capi.FLAG_DISABLE_MIS), (isinstance(callback, EventStatisticCallback), capi.FLAG_STATISTICS),
            (trackEvents or isinstance(callback, TrackRecordCallback), capi.FLAG_TRACK),
            (isinstance(response, KernelHistogramHitResponse), capi.FLAG_KERNEL_HISTOGRAM))
        self._code = code or b""
        self._program = _Program(_backend.TracerSpec(capi.TRACER_VOLUME_FORWARD, flags, steps, blockSize, *self._responseSpec(response),
                                                     light_queue_capacity=lightCap, wavelength_queue_capacity=lamCap), None)

    callback = property(lambda self: self._callback, doc="callback called for each tracing event")
    code = property(lambda self: self._code, doc="kept for API parity; kernels are precompiled")
    directLightingDisabled = property(lambda self: not self._switches["direct"])
    nScattering = property(lambda self: self._switches["nScattering"])
    rng = property(lambda self: self._rng)
    source = property(lambda self: self._source)
    target = property(lambda self: self._target)
    targetSamplingDisabled = property(lambda self: not self._switches["mis"])
    wavelengthSource = property(lambda self: self._wavelengthSource)

    def _setBox(self, box) -> None:
        self.setParams(_maxDist=box.diagonal, upperBBoxCorner=box.upperCorner, lowerBBoxCorner=box.lowerCorner)

    traceBBox = property(lambda self: RectBBox(*(self.getParam(side + "BBoxCorner") for side in ("lower", "upper"))), _setBox,
                         doc="box photons are traced in; leaving it ends a path (RAY_LOST)")

    def _collectStages(self):
        order = ("rng", "photons", "lightSource", "target", "tracer", "callback", "response")
        parts = (self._rng, self._wavelengthSource, self._source, self._target, self, self._callback, self._response)
        return list(zip(order, parts))

    def _finishParams(self, i):
        Tracer._finishParams(self, i)
        self.setParam("_maxDist", self.traceBBox.diagonal)

    def _assemble(self, i: int) -> capi.RunParams:
        P = self._commonParams(i, self._rng, self._source, self._wavelengthSource, self._response)
        t = self._getParamsBlock("TraceParams", i)
        P.volume.medium[0], P.volume.medium[1] = t.medium.lo, t.medium.hi
        P.volume.objectId = t.objectId
        self._fillPropagation(P.volume.propagation, t, "lowerBBoxCorner", "upperBBoxCorner")
        fillTargetParams(P.target, self._target, i)
        return P



class SceneForwardTracer(Tracer):
    """
    Path tracer against a triangle-mesh scene with surface reflection /
    transmission and optional next-event estimation through a target guide.
    Parameters and stage parameters as in the reference (trace.py:1048-1337):
    `batchSize, source, wavelengthSource, response, rng, scene, *, capacity=None,
    callback=EmptyEventCallback(), maxPathLength=6, targetId=-1, targetGuide=None,
    scatterCoefficient=NaN, sourceMedium=None, maxTime=1000 ns, polarized=False,
    disableDirectLighting=False, disableTransmission=False,
    disableVolumeBorder=False, useRefractedHitDir=False, blockSize=128, code=None`.
    """

    name = "Scene Forward Tracer"

    class TraceParams(Structure):
        _fields_ = _layout(targetId=c_int32, sourceMedium=buffer_reference, scatterCoefficient=c_float, _lowerBBoxCorner=vec3,
                           _upperBBoxCorner=vec3, maxTime=c_float, _maxDist=c_float)

    def __init__(self, batchSize, source, wavelengthSource, response, rng, scene, *, capacity=None, callback=None,
                 targetGuide=None, targetId=-1, sourceMedium=None, maxPathLength=6, scatterCoefficient=float("NaN"),
                 maxTime=1000.0 * u.ns, polarized=False, disableDirectLighting=False, disableTransmission=False,
                 disableVolumeBorder=False, useRefractedHitDir=False, blockSize=128, code=None, trackEvents=False):
        callback = EmptyEventCallback() if callback is None else callback
        self._checkComponents(source, wavelengthSource, response, rng, callback)
        guided = targetGuide is not None
        if guided and not isinstance(targetGuide, _GUIDES):
            raise _unsupported("target guide", targetGuide, ", ".join(c.__name__ for c in _GUIDES) + ", None")
        if guided and response.nRNGSamples > 0:
            raise NotImplementedError("responses that draw random numbers (StoreTimeHitResponse) are not available together "
                                      "with a target guide on the HIP backend")
        # hits: one per step but the first, doubled by the guide's shadow rays, + the direct one;
        # a step draws 4 numbers + what the guide needs
        hits = (maxPathLength - 1) * (2 if guided else 1) + (0 if disableDirectLighting else 1)
        perStep = 4 + (targetGuide.nRNGSamples if guided else 0)
        draws = source.nRNGForward + wavelengthSource.nRNGSamples + perStep * maxPathLength + hits * response.nRNGSamples
        Tracer.__init__(self, response, {"TraceParams": self.TraceParams}, batchSize=batchSize, blockSize=blockSize,
                        capacity=capacity, maxHitsPerThread=hits, nRNGSamples=draws, polarized=polarized)
        self._source, self._wavelengthSource, self._callback, self._rng = source, wavelengthSource, callback, rng
        self._scene, self._targetGuide = scene, targetGuide
        self._switches = {"maxPathLength": maxPathLength, "noDirect": bool(disableDirectLighting),
                          "noTransmission": bool(disableTransmission), "noBorder": bool(disableVolumeBorder),
                          "refractedDir": bool(useRefractedHitDir)}
        box = scene.bbox
        self._mediumOwner = (scene.medium, sourceMedium)
        self.setParams(sourceMedium=scene.medium if sourceMedium is None else sourceMedium, targetId=targetId, maxTime=maxTime,
                       scatterCoefficient=scatterCoefficient, _maxDist=box.diagonal, _lowerBBoxCorner=box.lowerCorner,
                       _upperBBoxCorner=box.upperCorner)
        srcFlags, lightCap, lamCap = sourceSpec(source, wavelengthSource)
        flags = srcFlags | rngFlags(rng) | (guideFlags(targetGuide) if guided else capi.FLAG_DISABLE_MIS) | _flagsIf(
            (polarized, capi.FLAG_POLARIZED), (disableDirectLighting, capi.FLAG_DISABLE_DIRECT),
            (disableTransmission, capi.FLAG_DISABLE_TRANSMISSION), (disableVolumeBorder, capi.FLAG_DISABLE_VOLUME_BORDER),
            (useRefractedHitDir, capi.FLAG_HIT_USE_TRANSMITTED), (isinstance(callback, EventStatisticCallback), capi.FLAG_STATISTICS),
            (trackEvents or isinstance(callback, TrackRecordCallback), capi.FLAG_TRACK),
            (isinstance(response, KernelHistogramHitResponse), capi.FLAG_KERNEL_HISTOGRAM))
        self._code = code or b""
        self._program = _Program(_backend.TracerSpec(capi.TRACER_SCENE_FORWARD, flags, maxPathLength, blockSize, *self._responseSpec(response),
                                                     light_queue_capacity=lightCap, wavelength_queue_capacity=lamCap), scene)

    callback = property(lambda self: self._callback)
    code = property(lambda self: self._code)
    directLightingDisabled = property(lambda self: self._switches["noDirect"])
    hitsUseRefractedDir = property(lambda self: self._switches["refractedDir"])
    maxPathLength = property(lambda self: self._switches["maxPathLength"])
    rng = property(lambda self: self._rng)
    scene = property(lambda self: self._scene)
    source = property(lambda self: self._source)
    targetGuide = property(lambda self: self._targetGuide)
    transmissionDisabled = property(lambda self: self._switches["noTransmission"])
    volumeBorderDisabled = property(lambda self: self._switches["noBorder"])
    wavelengthSource = property(lambda self: self._wavelengthSource)

    def _collectStages(self):
        guide = [] if self._targetGuide is None else [("guide", self._targetGuide)]
        front = [("rng", self._rng), ("photons", self._wavelengthSource), ("lightSource", self._source)]
        return front + guide + [("tracer", self), ("callback", self._callback), ("response", self._response)]

    def _assemble(self, i: int) -> capi.RunParams:
        P = self._commonParams(i, self._rng, self._source, self._wavelengthSource, self._response)
        t = self._getParamsBlock("TraceParams", i)
        P.scene.targetId = t.targetId
        P.scene.sceneMedium[0], P.scene.sceneMedium[1] = t.sourceMedium.lo, t.sourceMedium.hi
        self._fillPropagation(P.scene.propagation, t, "_lowerBBoxCorner", "_upperBBoxCorner")
        if self._targetGuide is not None:
            fillGuideParams(P.guide, self._targetGuide, i)
        return P



class ScenePhotonTracer(SceneForwardTracer):
    """
    Path tracer simulating individual photons: a path carries one photon, produces at most one
    hit and is absorbed at random according to the media's absorption (survival roulette after
    every step); the hit's contribution is its survival chance. Parameters and stage parameters
    as in the reference (trace.py:2370-2665): `batchSize, source, wavelengthSource, response, rng,
    scene, *, capacity=None, callback, targetId=-1, sourceMedium=None, maxTime=1 ms,
    nScatteringPerRun=10, nRuns=10, polarized, disableTransmission, disableVolumeBorder,
    useRefractedHitDir, blockSize=128`.

    The reference runs `nRuns` dispatches of `nScatteringPerRun` steps with a photon queue in
    between to re-pack the surviving photons; the staged kernel of this backend re-packs photons
    after every stage anyway, so one launch traces all `nRuns * nScatteringPerRun` steps. What the
    runs determine is kept: run r draws its random numbers from dimension `r * nRNGPerRun` on.
    """

    name = "Scene Photon Tracer"

    def __init__(self, batchSize, source, wavelengthSource, response, rng, scene, *, nRuns=10, nScatteringPerRun=10,
                 capacity=None, callback=None, targetId=-1, sourceMedium=None, maxTime=1.0 * u.ms, polarized=False,
                 disableTransmission=False, disableVolumeBorder=False, useRefractedHitDir=False, blockSize=128):
        self._nRuns, self._nScatteringPerRun = nRuns, nScatteringPerRun
        self._nRNGPerRun = response.nRNGSamples + 5 * nScatteringPerRun
        SceneForwardTracer.__init__(self, batchSize, source, wavelengthSource, response, rng, scene, capacity=capacity, callback=callback,
                         maxPathLength=nRuns * nScatteringPerRun, targetId=targetId, targetGuide=None,
                         scatterCoefficient=float("NaN"), sourceMedium=sourceMedium, maxTime=maxTime, polarized=polarized,
                         disableDirectLighting=False, disableTransmission=disableTransmission,
                         disableVolumeBorder=disableVolumeBorder, useRefractedHitDir=useRefractedHitDir, blockSize=blockSize)
        # trace.py:2487-2501: one hit per photon, 5 draws per step
        self._maxHitsPerThread = 1
        self._nRNGSamples = source.nRNGForward + wavelengthSource.nRNGSamples + 5 * nRuns * nScatteringPerRun + response.nRNGSamples
        response.prepare(self._getTraceConfig())
        spec = self._program._spec
        spec.flags |= capi.FLAG_PHOTON
        self._program = _Program(spec, scene)

    nRuns = property(lambda self: self._nRuns, doc="number of runs; nRuns * nScatteringPerRun steps are simulated")
    nScatteringPerRun = property(lambda self: self._nScatteringPerRun, doc="scatter events per run")

    def _assemble(self, i: int) -> capi.RunParams:
        P = super()._assemble(i)
        P.photon_steps_per_run = self._nScatteringPerRun
        P.photon_rng_per_run = self._nRNGPerRun
        return P


class VolumePhotonTracer(VolumeForwardTracer):
    """
    Photon tracer in a homogeneous medium with an analytic target: a path carries one photon,
    is absorbed at random (survival roulette after every step) and produces at most one hit whose
    contribution is its survival chance. Parameters as in the reference (trace.py:2668-2959):
    `batchSize, source, target, wavelengthSource, response, rng, *, medium, objectId=0, capacity=None,
    callback, traceBBox, maxTime=1000 ns, nScatteringPerRun=10, nRuns=10, polarized, blockSize=128`.
    One launch traces all `nRuns * nScatteringPerRun` steps (see `ScenePhotonTracer`).
    """

    name = "Volume Photon Tracer"

    def __init__(self, batchSize, source, target, wavelengthSource, response, rng, *, medium, nRuns=10, nScatteringPerRun=10,
                 objectId=0, capacity=None, callback=None, traceBBox=None, maxTime=1000.0 * u.ns, polarized=False,
                 blockSize=128, code=None):
        self._nRuns, self._nScatteringPerRun = nRuns, nScatteringPerRun
        self._nRNGPerRun = response.nRNGSamples + 4 * nScatteringPerRun
        VolumeForwardTracer.__init__(self, batchSize, source, target, wavelengthSource, response, rng, medium=medium, objectId=objectId,
                         capacity=capacity, callback=callback, nScattering=nRuns * nScatteringPerRun,
                         scatterCoefficient=float("NaN"), traceBBox=traceBBox, maxTime=maxTime, polarized=polarized,
                         disableDirectLighting=True, disableTargetSampling=True, blockSize=blockSize, code=code)
        # trace.py:2783-2797: one hit per photon, 4 draws per step
        self._maxHitsPerThread = 1
        self._nRNGSamples = source.nRNGForward + wavelengthSource.nRNGSamples + 4 * nRuns * nScatteringPerRun + response.nRNGSamples
        response.prepare(self._getTraceConfig())
        spec = self._program._spec
        spec.flags |= capi.FLAG_PHOTON
        spec.path_length = nRuns * nScatteringPerRun
        self._program = _Program(spec, None)

    nRuns = property(lambda self: self._nRuns, doc="number of runs; nRuns * nScatteringPerRun steps are simulated")
    nScatteringPerRun = property(lambda self: self._nScatteringPerRun, doc="scatter events per run")
    maxPathLength = property(lambda self: self._nRuns * self._nScatteringPerRun, doc="maximum number of events per photon")

    def _assemble(self, i: int) -> capi.RunParams:
        P = super()._assemble(i)
        P.photon_steps_per_run = self._nScatteringPerRun
        P.photon_rng_per_run = self._nRNGPerRun
        return P


_BACKWARD = ("VolumeBackwardTracer", "SceneBackwardTracer", "SceneBackwardTargetTracer", "DirectLightTracer", "BidirectionalPathTracer")


def __getattr__(name: str):
    if name in _BACKWARD:
        raise NotImplementedError(f"theia.trace.{name}: backward, direct and bidirectional tracers are not provided by the HIP backend "
                                  "(SURVEY.md 8: the forward tracing path -- VolumeForwardTracer, SceneForwardTracer and the photon tracers)")
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
