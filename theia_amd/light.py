"""
Light and wavelength sources of the supported path (reference:
src/theia/light.py:58-100 WavelengthSource, :258-352 Const/Uniform, :417-470
LightSource, :1105-1180 SphericalLightSource; device side
shader/lightsource.spherical.glsl, wavelengthsource.{const,uniform}.glsl).
"""
from __future__ import annotations

from ctypes import Structure, c_float, c_int64, c_uint32
from warnings import warn

import numpy as np

from . import _backend
from . import _capi as capi
from . import units as u
from ._program import _Program
from .hp.glsl import _layout, buffer_reference, vec2, vec3, vec4
from .hp.pipeline import PipelineStage, SourceCodeMixin
from .hp.queue import IOQueue
from .lookup import DeviceBlob, uploadTables

__all__ = ["WavelengthSource", "WavelengthSampleItem", "HostWavelengthSource", "StreamingHostWavelengthSource",
           "ConstWavelengthSource", "UniformWavelengthSource", "FunctionWavelengthSource", "LightSource", "LightSampleItem",
           "PolarizedLightSampleItem", "LightSampler", "HostLightSource", "StreamingHostLightSource", "ConeLightSource",
           "SphericalLightSource", "PencilLightSource", "CherenkovLightSource", "ParticleTrack", "CherenkovTrackLightSource",
           "MuonTrackLightSource", "ParticleCascadeLightSource", "frankTamm"]


def _xyz(n: int = 3):
    return c_float * n


class WavelengthSampleItem(Structure):
    """one sampled wavelength and its contribution (light.py:79-84)"""

    _fields_ = _layout(wavelength=c_float, contrib=c_float)


class LightSampleItem(Structure):
    """one sample of a light source (light.py:463-471, shader/lightsource.queue.glsl)"""

    _fields_ = _layout(position=_xyz(), direction=_xyz(), startTime=c_float, contrib=c_float)


class PolarizedLightSampleItem(Structure):
    """`LightSampleItem` with the Stokes vector and its reference direction (light.py:474-489)"""

    _fields_ = _layout(position=_xyz(), direction=_xyz(), stokes=_xyz(4), polarizationRef=_xyz(), startTime=c_float,
                       contrib=c_float)


class _Budgeted:
    """sources whose photons share a total `budget` (a plain stage parameter kept on the host side)"""

    _budget = 1.0

    def _set_budget(self, total) -> None:
        self._budget = total

    budget = property(lambda self: self._budget, _set_budget, doc="total contribution shared by the sampled photons")


class _Fed:
    """sources fed from the host: a queue of `capacity` items uploaded with every batch; `updateFn(source, i)`
    runs before a batch with config `i` is assembled and may refill it"""

    def _initFeed(self, item, capacity, updateFn) -> None:
        self._feed = (IOQueue(item, capacity, mode="update", skipCounter=True), capacity, updateFn)

    capacity = property(lambda self: self._feed[1], doc="most samples a run can draw")
    queue = property(lambda self: self._feed[0], doc="the items of the next batch")

    def update(self, i):
        refill = self._feed[2]
        if refill is not None:
            refill(self, i)
        SourceCodeMixin.update(self, i)


class WavelengthSource(SourceCodeMixin):
    """Base of wavelength samplers; `nRNGSamples` = draws per photon."""

    name = "Wavelength Sampler"

    def __init__(self, *, params={}, extra=set(), nRNGSamples=0):
        SourceCodeMixin.__init__(self, params, extra)
        self._draws = nRNGSamples

    nRNGSamples = property(lambda self: self._draws, doc="random numbers drawn per photon")


class HostWavelengthSource(_Fed, WavelengthSource):
    """
    Wavelength source passing samples from the host to the device (light.py:87-150,
    shader/wavelengthsource.host.glsl): photon `k` of a batch gets item `k` of
    `queue.view(i)` (`["wavelength"]`, `["contrib"]`). `updateFn(source, i)` is called before
    a batch with config `i` is assembled and may refill the queue. The queue is uploaded with
    every batch (PCIe: 8 B per photon).
    """

    name = "Host Wavelength Source"

    def __init__(self, capacity, *, extra=set(), updateFn=None):
        WavelengthSource.__init__(self, extra=extra)
        self._initFeed(WavelengthSampleItem, capacity, updateFn)


class _StreamMixin:
    """what the two streaming host sources share (light.py:184-255, 808-880): consecutive batches
    move through `data`; assigning new data rewinds the stream"""

    _sizeField = ""
    _label = ""

    def _initStream(self, columns, ignoreContrib, data) -> None:
        self._wanted = set(columns)
        if ignoreContrib:
            self._wanted.discard("contrib")
            for cfg in (0, 1):
                self.queue.view(cfg)["contrib"] = 1.0
        self.data = data

    def _rewind(self, table) -> None:
        self._cursor, self._rows, self._table = 0, 0, None
        if table is None:
            return
        absent = self._wanted.difference(table.keys())
        if absent:
            raise ValueError(f"provided data misses expected fields: {absent}")
        columns = {name: np.asarray(table[name]) for name in self._wanted}
        rows = np.size(table[self._sizeField], 0)
        if any(np.size(col, 0) != rows for col in columns.values()):
            raise ValueError("the fields of the provided data differ in length")
        self._rows, self._table = rows, columns

    data = property(lambda self: self._table, _rewind, lambda self: self._rewind(None), doc="data used as streaming source")

    @staticmethod
    def _updateData(source, i) -> None:
        if source._table is None:
            warn(f"{source._label}: New data was requested but no source was given!")
            return
        first = source._cursor
        if first >= source._rows:
            warn(f"{source._label} ran out of data!")
            return
        take = min(source.capacity, source._rows - first)
        source._cursor = first + take
        target = source.queue.view(i)
        for name, column in source._table.items():
            target[name][:take] = column[first:first + take]


class StreamingHostWavelengthSource(_StreamMixin, HostWavelengthSource):
    """Feeds pre-sampled wavelengths batch by batch from `data` (`{"wavelength": ..., "contrib": ...}`;
    `ignoreContrib=True`: contribution 1). Stage parameter `data`. Reference: light.py:152-255."""

    name = "Streaming Host Wavelength Source"
    _sizeField = "wavelength"
    _label = "StreamingHostWavelengthSource"

    def __init__(self, batchSize, *, ignoreContrib=False, data=None):
        HostWavelengthSource.__init__(self, batchSize, updateFn=self._updateData, extra={"data"})
        self._initStream({"wavelength", "contrib"}, ignoreContrib, data)


class FunctionWavelengthSource(WavelengthSource):
    """
    Importance-samples wavelengths from `fn(lambda in nm)` on `lambdaRange`: the distribution is
    integrated and inverted numerically on the host (scipy) into a look-up table of `numSamples`
    entries the device interpolates (light.py:354-414, shader/wavelengthsource.function.glsl);
    the contribution is the integral of `fn`. 1 draw.
    """

    name = "Function Wavelength Source"

    class WavelengthParams(Structure):
        _fields_ = _layout(_table=c_int64, _contrib=c_float)

    def __init__(self, fn, *, numSamples=1024, lambdaRange=(300.0, 700.0) * u.nm):
        WavelengthSource.__init__(self, nRNGSamples=1, params={"WavelengthParams": self.WavelengthParams})
        import warnings

        from scipy.integrate import quad
        from scipy.stats.sampling import NumericalInversePolynomial

        class _Density:
            pdf = staticmethod(fn)

        total = quad(fn, *lambdaRange)[0]
        with warnings.catch_warnings():  # scipy notes that it moved the expansion centre into the domain
            warnings.simplefilter("ignore", RuntimeWarning)
            quantile = NumericalInversePolynomial(_Density(), domain=tuple(lambdaRange)).ppf
        self._tableMemory, (address,) = uploadTables([quantile(np.linspace(0.0, 1.0, numSamples))])
        self.setParams(_table=address, _contrib=total)


class ConstWavelengthSource(WavelengthSource):
    """Every photon has the same wavelength (default 600 nm); draws nothing."""

    name = "Const Wavelength Source"

    class WavelengthParams(Structure):
        _fields_ = _layout(wavelength=c_float)

    def __init__(self, wavelength=600.0 * u.nm):
        WavelengthSource.__init__(self, params={"WavelengthParams": self.WavelengthParams})
        self.setParams(wavelength=wavelength)


class UniformWavelengthSource(WavelengthSource):
    """Wavelengths uniform in `lambdaRange`; one draw per photon. With
    ``normalize=False`` the contribution carries the range width."""

    name = "Uniform Wavelength Source"

    class SourceParams(Structure):
        _fields_ = _layout(lambdaRange=vec2, _contrib=c_float)

    normalize = True  # plain attribute, also a stage parameter

    def __init__(self, *, normalize=True, lambdaRange=(300.0, 700.0)):
        WavelengthSource.__init__(self, nRNGSamples=1, params={"WavelengthParams": self.SourceParams}, extra={"normalize"})
        self.setParams(lambdaRange=lambdaRange, normalize=normalize)

    def _finishParams(self, i):
        lo, hi = self.getParam("lambdaRange")
        width = abs(hi - lo)
        self.setParam("_contrib", 1.0 if self.normalize or width == 0.0 else width)


class LightSource(SourceCodeMixin):
    """Base of light sources."""

    name = "Light Source"

    def __init__(self, *, supportForward, supportBackward, params={}, extra=set(), nRNGForward=0, nRNGBackward=0):
        SourceCodeMixin.__init__(self, params, extra)
        self._modes = {"forward": (bool(supportForward), nRNGForward), "backward": (bool(supportBackward), nRNGBackward)}

    supportForward = property(lambda self: self._modes["forward"][0], doc="can start photons (forward tracing)")
    supportBackward = property(lambda self: self._modes["backward"][0], doc="can be sampled by backward tracers")
    nRNGForward = property(lambda self: self._modes["forward"][1], doc="draws per photon in forward mode")
    nRNGBackward = property(lambda self: self._modes["backward"][1], doc="draws per sample in backward mode")


class SphericalLightSource(_Budgeted, LightSource):
    """Isotropic, unpolarized point source: `position`, emission window
    `timeRange`, total `budget` shared by the sampled photons. 3 draws (direction
    2, start time 1)."""

    name = "Spherical Light Source"

    class LightParams(Structure):
        _fields_ = _layout(position=vec3, _contribFwd=c_float, _contribBwd=c_float, timeRange=vec2)

    def __init__(self, *, budget=1.0, timeRange=(0.0, 100.0) * u.ns, position=(0.0, 0.0, 0.0)):
        LightSource.__init__(self, supportForward=True, supportBackward=True, nRNGForward=3, nRNGBackward=1,
                             params={"LightParams": self.LightParams}, extra={"budget"})
        self.setParams(position=position, timeRange=timeRange, budget=budget)

    def _finishParams(self, i):
        LightSource._finishParams(self, i)
        # backward mode integrates over the sphere of directions
        self.setParams(_contribFwd=self.budget, _contribBwd=self.budget / (4.0 * np.pi))


class PencilLightSource(LightSource):
    """Pencil beam: every photon starts at `position` along `direction` at a time drawn from
    `timeRange`; `budget` is shared by the sampled photons. Polarized tracers take the beam's
    `stokes` vector and `polarizationRef` (unit, perpendicular to `direction`). 1 draw.
    Reference: light.py:1024-1110, shader/lightsource.pencil.glsl."""

    name = "Pencil Light Source"

    class LightParams(Structure):
        _fields_ = _layout(position=vec3, direction=vec3, budget=c_float, timeRange=vec2, stokes=vec4, polarizationRef=vec3)

    def __init__(self, **beam):
        LightSource.__init__(self, supportForward=True, supportBackward=False, nRNGForward=1, nRNGBackward=0,
                             params={"LightParams": self.LightParams})
        given = dict(position=(0.0, 0.0, 0.0), direction=(0.0, 0.0, 1.0), timeRange=(0.0, 100.0) * u.ns, budget=1.0,
                     stokes=(1.0, 0.0, 0.0, 0.0), polarizationRef=(0.0, 1.0, 0.0))
        unknown = set(beam) - set(given)
        if unknown:
            raise TypeError(f"PencilLightSource() got unexpected keyword arguments {sorted(unknown)}")
        given.update(beam)
        self.setParams(**given)


class ConeLightSource(_Budgeted, LightSource):
    """
    Point source radiating uniformly into the cone around `direction` with `cosOpeningAngle`;
    optionally polarized with a constant `stokes` vector whose reference `polarizationReference`
    is made perpendicular to every sampled direction. 3 draws. Parameters and stage parameters
    as in the reference (light.py:883-1021, shader/lightsource.cone.glsl).
    """

    name = "Cone Light Source"

    class LightParams(Structure):
        _fields_ = _layout(position=vec3, direction=vec3, cosOpeningAngle=c_float, _contribFwd=c_float, _contribBwd=c_float,
                           timeRange=vec2, polarizationReference=vec3, stokes=vec4)

    def __init__(self, *, polarized=True, budget=1.0, cosOpeningAngle=0.5, timeRange=(0.0, 100.0) * u.ns,
                 position=(0.0, 0.0, 0.0), direction=(1.0, 0.0, 0.0), stokes=(1.0, 0.0, 0.0, 0.0),
                 polarizationReference=(0.0, 1.0, 0.0)):
        along = float(np.dot(np.asarray(direction, dtype=float), np.asarray(polarizationReference, dtype=float)))
        if abs(along) > 1.0 - 1e-5:
            raise ValueError("direction and polarizationReference cannot be parallel!")
        if polarized and cosOpeningAngle <= 0.0:
            raise ValueError("Opening angles for polarized cone lights must be smaller 90 degrees!")
        LightSource.__init__(self, supportForward=True, supportBackward=True, nRNGForward=3, nRNGBackward=1,
                             params={"LightParams": self.LightParams}, extra={"budget"})
        self._polarized = bool(polarized)
        self.setParams(budget=budget, position=position, direction=direction, timeRange=timeRange, stokes=stokes,
                       cosOpeningAngle=cosOpeningAngle, polarizationReference=polarizationReference)

    polarized = property(lambda self: self._polarized, doc="whether stokes / polarizationReference are used")

    def _finishParams(self, i):
        LightSource._finishParams(self, i)
        # backward: divide by the solid angle of the cone
        cone = 2.0 * np.pi * (1.0 - self.cosOpeningAngle)
        self.setParams(_contribFwd=self.budget, _contribBwd=self.budget / cone)


class _FrankTamm:
    """the two compile-time switches of the Cherenkov family: the unit of the radiance (the two Cherenkov sources) and whether the
    Frank-Tamm factor is applied at all (muon track, cascade)"""

    _usePhotonCount = False
    _applyFrankTamm = True
    usePhotonCount = property(lambda self: self._usePhotonCount, doc="radiance in number of photons instead of eV")
    isFrankTammApplied = property(lambda self: self._applyFrankTamm, doc="whether the Frank-Tamm equation is applied")


class CherenkovLightSource(_FrankTamm, LightSource):
    """
    Cherenkov light of a particle at the speed of light on a straight track from `trackStart` (at `startTime`) to `trackEnd`
    (at `endTime`): photons start uniformly along the track on the cone of half angle acos(1 / n), linearly polarized in the
    plane of track and photon; their contribution is the Frank-Tamm radiance (eV, or photons with `usePhotonCount`) times the
    track length. 2 draws. Reference: light.py:1183-1268, shader/lightsource.cherenkov.simple.glsl:20-58.
    """

    name = "Cherenkov Light Source"

    class LightParams(Structure):
        _fields_ = _layout(trackStart=vec3, startTime=c_float, trackEnd=vec3, endTime=c_float, _trackDir=vec3, _trackDist=c_float)

    def __init__(self, *, trackStart=(0.0, 0.0, 0.0) * u.m, trackEnd=(100.0, 0.0, 0.0) * u.m, startTime=0.0 * u.ns,
                 endTime=100.0 * u.m / u.c, usePhotonCount=False):
        LightSource.__init__(self, supportForward=True, supportBackward=True, nRNGForward=2, nRNGBackward=0,
                             params={"LightParams": self.LightParams})
        self._usePhotonCount = bool(usePhotonCount)
        self.setParams(trackStart=trackStart, trackEnd=trackEnd, startTime=startTime, endTime=endTime)

    def _finishParams(self, i):
        LightSource._finishParams(self, i)
        along = np.asarray(self.trackEnd, dtype=np.float64) - np.asarray(self.trackStart, dtype=np.float64)
        dist = float(np.sqrt(np.square(along).sum()))
        self.setParams(_trackDir=tuple(along / dist), _trackDist=dist)


class ParticleTrack:
    """
    A polyline a particle travelled along: up to `capacity` vertices (x, y, z in m, t in ns), `length` = number of segments.
    The reference keeps it in a tensor mapped into host memory (light.py:1274-1352); here the vertices live in a numpy array
    and every update of the light source that uses the track copies them to the device (`address`).
    """

    class Header(Structure):
        _fields_ = [("length", c_uint32)]

    class Vertex(Structure):
        _fields_ = _layout(x=c_float, y=c_float, z=c_float, t=c_float)

    def __init__(self, capacity: int) -> None:
        if capacity < 2:
            raise ValueError("a track needs at least two vertices")
        self._capacity = int(capacity)
        self._host = np.zeros(1 + 4 * self._capacity, dtype=np.uint32)
        self._blob = DeviceBlob(self._host.nbytes)
        self._blob.upload(self._host)

    capacity = property(lambda self: self._capacity, doc="maximum number of vertices")
    address = property(lambda self: self._blob.address, doc="device address of the track")

    def __int__(self) -> int:
        return self._blob.address

    @property
    def length(self) -> int:
        """number of segments, i.e. number of vertices - 1"""
        return int(self._host[0])

    @length.setter
    def length(self, value: int) -> None:
        if not 0 <= value <= self._capacity - 1:
            raise ValueError("length must be between 0 and (capacity - 1)")
        self._host[0] = value

    def numpy(self, flat: bool = False) -> np.ndarray:
        """the vertices: a structured array (x, y, z, t), or with `flat` a float array of shape (capacity, 4)"""
        v = self._host[1:].view(np.float32).reshape(self._capacity, 4)
        return v if flat else v.view(np.dtype([(c, np.float32) for c in "xyzt"])).reshape(-1)

    def setVertices(self, data) -> None:
        """copies vertices of shape (n, 4) -- x, y, z, t -- into the track and sets its length to n - 1 segments"""
        data = np.asarray(data, dtype=np.float32)
        if data.ndim != 2 or data.shape[1] != 4 or not 2 <= len(data) <= self._capacity:
            raise ValueError("expected between 2 and `capacity` vertices (x, y, z, t)")
        self.numpy(True)[: len(data)] = data
        self.length = len(data) - 1

    def flush(self) -> None:
        """host -> device"""
        self._blob.upload(self._host)


class CherenkovTrackLightSource(_FrankTamm, LightSource):
    """
    Cherenkov light of a particle at the speed of light along a `ParticleTrack`: a segment is picked uniformly, the photon starts
    uniformly on it, cone and polarization as for `CherenkovLightSource`. 2 draws. `track` may be `None` until the source is used.
    Reference: light.py:1355-1410, shader/lightsource.cherenkov.track.glsl:23-78.
    """

    name = "Cherenkov Track Light Source"

    class TrackParams(Structure):
        _fields_ = _layout(_track=buffer_reference)

    def __init__(self, track=None, *, usePhotonCount=False):
        LightSource.__init__(self, supportForward=True, supportBackward=False, nRNGForward=2,
                             params={"TrackParams": self.TrackParams}, extra={"track"})
        self._usePhotonCount = bool(usePhotonCount)
        self.track = track

    def _set_track(self, track) -> None:
        if track is not None and not isinstance(track, ParticleTrack):
            raise TypeError("track must be a ParticleTrack or None")
        self._trackObj = track

    track = property(lambda self: self._trackObj, _set_track, doc="the track the light is sampled from")

    def _finishParams(self, i):
        LightSource._finishParams(self, i)
        if self._trackObj is None or self._trackObj.length < 1:
            raise RuntimeError("CherenkovTrackLightSource needs a ParticleTrack of at least one segment before it is sampled")
        self._trackObj.flush()
        self.setParams(_track=self._trackObj.address)


class MuonTrackLightSource(_FrankTamm, LightSource):
    """
    Cherenkov light of a high-energy muon and its secondaries below 500 MeV on a straight track (Raedel 2012): photons start
    uniformly between `startPosition` (at `startTime`) and `endPosition` (at `endTime`); their angle to the track follows the
    profile exp(-b x^a) x^(a-1), x = |cos(theta) - 1 / n|, with a, b and the light yield scale set by `muonEnergy`. 3 draws.
    With `applyFrankTamm=False` the Frank-Tamm factor is left to the wavelength source. Unpolarized.
    Reference: light.py:1413-1519, shader/lightsource.particles.muon.glsl:29-60.
    """

    name = "Muon Track Light Source"

    class TrackParameters(Structure):
        _fields_ = _layout(startPosition=vec3, startTime=c_float, endPosition=vec3, endTime=c_float, _energyScale=c_float,
                           _a_angular=c_float, _b_angular=c_float)

    def __init__(self, startPosition=(0.0, 0.0, 0.0), startTime=0.0, endPosition=(0.0, 0.0, 0.0), endTime=0.0,
                 muonEnergy=1.0 * u.GeV, applyFrankTamm=True):
        LightSource.__init__(self, supportForward=True, nRNGForward=3, supportBackward=True, nRNGBackward=1,
                             params={"MuonTrackParams": self.TrackParameters}, extra={"muonEnergy"})
        self.setParams(startPosition=startPosition, startTime=startTime, endPosition=endPosition, endTime=endTime)
        self.muonEnergy = muonEnergy
        self._applyFrankTamm = bool(applyFrankTamm)

    def _set_energy(self, energy) -> None:
        self._muonEnergy = energy
        # light yield of the secondaries and the angular profile as functions of the muon's energy (light.py:1505-1513)
        self.setParam("_energyScale", 1.1880 + 0.0206 * np.log(energy))
        self.setParam("_a_angular", 0.86634 - 7.5624e-3 * np.log10(energy))
        self.setParam("_b_angular", 2.5030 + 3.0533e-2 * np.log10(energy))

    muonEnergy = property(lambda self: self._muonEnergy, _set_energy, doc="energy of the muon producing the secondary particles")


class ParticleCascadeLightSource(_FrankTamm, LightSource):
    """
    Cherenkov light of an electromagnetic or hadronic cascade (Raedel 2012): photons start on the axis `startPosition` +
    z `direction`, z gamma distributed with shape `a_long` and scale `b_long` (m), at the time the primary passes; angular profile
    as for the muon track with `a_angular`, `b_angular`; contribution `effectiveLength` (times Frank-Tamm unless
    `applyFrankTamm=False`). The number of draws varies (rejection sampling of the gamma distribution); 12 are reserved.
    Unpolarized. Reference: light.py:1522-1653, shader/lightsource.particles.cascade.glsl:41-69, random.gamma.glsl.
    """

    name = "Particle Cascade Light Source"

    class CascadeParameters(Structure):
        _fields_ = _layout(startPosition=vec3, startTime=c_float, direction=vec3, effectiveLength=c_float, a_angular=c_float,
                           b_angular=c_float, a_long=c_float, b_long=c_float)

    def __init__(self, startPosition=(0.0, 0.0, 0.0), startTime=0.0 * u.ns, direction=(0.0, 0.0, 1.0), effectiveLength=1.0,
                 a_angular=0.0, b_angular=0.0, a_long=0.0, b_long=0.0, applyFrankTamm=True):
        LightSource.__init__(self, supportForward=True, supportBackward=True, nRNGForward=12, nRNGBackward=10,
                             params={"CascadeParams": self.CascadeParameters})
        self.setParams(startPosition=startPosition, startTime=startTime, direction=direction, effectiveLength=effectiveLength,
                       a_angular=a_angular, b_angular=b_angular, a_long=a_long, b_long=b_long)
        self._applyFrankTamm = bool(applyFrankTamm)


def frankTamm(wavelength, refractiveIndex, beta: float = 1.0) -> np.ndarray:
    """Frank-Tamm equation: photons per metre of track and nanometre of wavelength, 2 pi alpha / lambda^2 (1 - 1 / (beta n)^2)
    (light.py:1656-1677)"""
    lam = np.asarray(wavelength, dtype=np.float64) / u.nm
    n = beta * np.asarray(refractiveIndex, dtype=np.float64)
    alpha = 7.2973525693e-3  # fine-structure constant (CODATA 2018)
    return 2.0 * np.pi * alpha / lam**2 * (1.0 - 1.0 / n**2) * 1e9  # (one of the two nm^-1 becomes m^-1)


class HostLightSource(_Fed, LightSource):
    """
    Light source passing samples from the host to the device (light.py:692-772,
    shader/lightsource.host.glsl): photon `k` of a batch starts as item `k` of `queue.view(i)`
    (`["position"]`, `["direction"]`, `["startTime"]`, `["contrib"]`, with `polarized=True` also
    `["stokes"]`, `["polarizationRef"]`). Any light source can so be sampled on the host (or by a
    `LightSampler`) and traced here. `updateFn(source, i)` may refill the queue before a batch
    with config `i`. The queue is uploaded with every batch (PCIe: 32 / 60 B per photon).
    """

    name = "Host Light Source"

    def __init__(self, capacity, *, extra=set(), updateFn=None, polarized=False):
        LightSource.__init__(self, supportForward=True, supportBackward=False, extra=extra)
        self._polarized = bool(polarized)
        self._initFeed(PolarizedLightSampleItem if polarized else LightSampleItem, capacity, updateFn)

    polarized = property(lambda self: self._polarized, doc="whether the host also provides polarization information")


class StreamingHostLightSource(_StreamMixin, HostLightSource):
    """Feeds light rays batch by batch from `data` (a dict with the fields of the sample item;
    `ignoreContrib=True`: contribution 1). Stage parameter `data`. Reference: light.py:775-880."""

    name = "Streaming Host Light Source"
    _sizeField = "startTime"
    _label = "StreamingHostLightSource"

    def __init__(self, batchSize, *, ignoreContrib=False, polarized=False, data=None):
        HostLightSource.__init__(self, batchSize, polarized=polarized, updateFn=self._updateData, extra={"data"})
        self._initStream(self.queue.view(0).fields, ignoreContrib, data)


# ---- what the HIP backend needs to know about a (light source, wavelength source) pair --------

_CHERENKOV_FAMILY = {CherenkovLightSource: capi.FLAG_LIGHT_CHERENKOV, CherenkovTrackLightSource: capi.FLAG_LIGHT_CHERENKOV_TRACK,
                     MuonTrackLightSource: capi.FLAG_LIGHT_MUON_TRACK, ParticleCascadeLightSource: capi.FLAG_LIGHT_CASCADE}
_LIGHTS = (SphericalLightSource, PencilLightSource, ConeLightSource, HostLightSource) + tuple(_CHERENKOV_FAMILY)
_WAVELENGTHS = (ConstWavelengthSource, UniformWavelengthSource, FunctionWavelengthSource, HostWavelengthSource)


def sourceSpec(source: LightSource, wavelength: WavelengthSource) -> tuple[int, int, int]:
    """(THX_FLAG_* bits, light queue capacity, wavelength queue capacity) selecting the kernel branches"""
    flags, lightCap, lamCap = 0, 0, 0
    if isinstance(source, HostLightSource):
        flags |= capi.FLAG_LIGHT_HOST | (capi.FLAG_LIGHT_QUEUE_POLARIZED if source.polarized else 0)
        lightCap = source.capacity
    elif isinstance(source, ConeLightSource):
        flags |= capi.FLAG_LIGHT_CONE | (capi.FLAG_LIGHT_POLARIZED if source.polarized else 0)
    elif isinstance(source, PencilLightSource):
        flags |= capi.FLAG_LIGHT_PENCIL
    elif type(source) in _CHERENKOV_FAMILY:
        flags |= _CHERENKOV_FAMILY[type(source)]
        flags |= capi.FLAG_FRANK_TAMM_PHOTONS if source.usePhotonCount else 0
        flags |= 0 if source.isFrankTammApplied else capi.FLAG_FRANK_TAMM_IS
    if isinstance(wavelength, HostWavelengthSource):
        flags |= capi.FLAG_WAVELENGTH_HOST
        lamCap = wavelength.capacity
    elif isinstance(wavelength, FunctionWavelengthSource):
        flags |= capi.FLAG_WAVELENGTH_FUNCTION
    elif isinstance(wavelength, UniformWavelengthSource):
        flags |= capi.FLAG_WAVELENGTH_UNIFORM
    return flags, lightCap, lamCap


def fillSourceParams(P: capi.RunParams, i: int, source: LightSource, wavelength: WavelengthSource) -> None:
    """copies the sources' parameter blocks of config `i` into the C-ABI run parameters"""
    if type(source) in _CHERENKOV_FAMILY:
        _fillCherenkovFamily(P.light, i, source)
    elif not isinstance(source, HostLightSource):
        l = source._getParamsBlock("LightParams", i)
        for k, c in enumerate("xyz"):
            P.light.position[k] = getattr(l.position, c)
        P.light.t_min, P.light.t_max = l.timeRange.x, l.timeRange.y
        if isinstance(source, PencilLightSource):
            P.light.contribFwd, P.light.contribBwd = l.budget, 0.0
        else:
            P.light.contribFwd, P.light.contribBwd = l._contribFwd, l._contribBwd
        if isinstance(source, (PencilLightSource, ConeLightSource)):
            ref = l.polarizationRef if isinstance(source, PencilLightSource) else l.polarizationReference
            for k, c in enumerate("xyz"):
                P.light.direction[k] = getattr(l.direction, c)
                P.light.polRef[k] = getattr(ref, c)
            for k, c in enumerate("xyzw"):
                P.light.stokes[k] = getattr(l.stokes, c)
        if isinstance(source, ConeLightSource):
            P.light.cosOpeningAngle = l.cosOpeningAngle
    if isinstance(wavelength, HostWavelengthSource):
        return
    w = wavelength._getParamsBlock("WavelengthParams", i)
    if isinstance(wavelength, UniformWavelengthSource):
        P.wavelength.p0, P.wavelength.p1, P.wavelength.p2 = w.lambdaRange.x, w.lambdaRange.y, w._contrib
    elif isinstance(wavelength, FunctionWavelengthSource):
        P.wavelength.p2 = w._contrib
        P.wavelength.table[0], P.wavelength.table[1] = w._table & 0xFFFFFFFF, (w._table >> 32) & 0xFFFFFFFF
    else:
        P.wavelength.p0 = w.wavelength


def _fillCherenkovFamily(L: capi.LightParams, i: int, source: LightSource) -> None:
    """the parameter block of a Cherenkov-family source is the reference's uniform block, field by field"""
    def vec(dst, src):
        for k, c in enumerate("xyz"):
            dst[k] = getattr(src, c)

    if isinstance(source, CherenkovLightSource):
        b, d = source._getParamsBlock("LightParams", i), L.cherenkov
        vec(d.trackStart, b.trackStart), vec(d.trackEnd, b.trackEnd), vec(d.trackDir, b._trackDir)
        d.startTime, d.endTime, d.trackDist = b.startTime, b.endTime, b._trackDist
    elif isinstance(source, CherenkovTrackLightSource):
        b = source._getParamsBlock("TrackParams", i)
        L.cherenkovTrack.track[0], L.cherenkovTrack.track[1] = b._track.lo, b._track.hi
    elif isinstance(source, MuonTrackLightSource):
        b, d = source._getParamsBlock("MuonTrackParams", i), L.muonTrack
        vec(d.startPosition, b.startPosition), vec(d.endPosition, b.endPosition)
        d.startTime, d.endTime = b.startTime, b.endTime
        d.energyScale, d.a_angular, d.b_angular = b._energyScale, b._a_angular, b._b_angular
    else:
        b, d = source._getParamsBlock("CascadeParams", i), L.cascade
        vec(d.startPosition, b.startPosition), vec(d.direction, b.direction)
        d.startTime, d.effectiveLength = b.startTime, b.effectiveLength
        d.a_angular, d.b_angular, d.a_long, d.b_long = b.a_angular, b.b_angular, b.a_long, b.b_long


def uploadSourceQueues(program, i: int, source: LightSource, wavelength: WavelengthSource) -> None:
    """host sources: copy the queues of config `i` to the device before the batch is launched"""
    if isinstance(source, HostLightSource):
        program.write_light(i, source.queue.buffer(i))
    if isinstance(wavelength, HostWavelengthSource):
        program.write_wavelengths(i, wavelength.queue.buffer(i))


class LightSampler(PipelineStage):
    """
    Samples a light source and a wavelength source into two queues (light.py:492-689,
    shader/lightsource.sample.glsl). Parameters as in the reference: `source, wavelengthSource,
    capacity, *, rng=None, medium=None, polarized=False, retrieve=True, scene=None, batchSize=128,
    code=None`; stage parameters `count` (default: capacity), `baseCount`, `medium`.
    `lightQueue.view(i)` / `wavelengthQueue.view(i)` hold the samples of the most recent batch run
    with config `i`.
    """

    name = "Light Sampler"

    class SampleParams(Structure):
        _fields_ = _layout(count=c_uint32, baseCount=c_uint32, medium=buffer_reference)

    def __init__(self, source, wavelengthSource, capacity, *, batchSize=128, polarized=False, retrieve=True, rng=None,
                 scene=None, medium=None, code=None):
        if rng is None and source.nRNGForward + wavelengthSource.nRNGSamples > 0:
            raise ValueError("Light source requires a RNG but none was given!")
        PipelineStage.__init__(self, {"SampleParams": self.SampleParams})
        if not isinstance(source, _LIGHTS):
            raise NotImplementedError(f"{type(source).__name__} is not supported as light source by the HIP backend "
                                      f"(supported: {', '.join(c.__name__ for c in _LIGHTS)})")
        if not isinstance(wavelengthSource, _WAVELENGTHS):
            raise NotImplementedError(f"{type(wavelengthSource).__name__} is not supported as wavelength source by the HIP "
                                      f"backend (supported: {', '.join(c.__name__ for c in _WAVELENGTHS)})")
        if isinstance(source, HostLightSource) and source.polarized != polarized:
            # the reference's shader refuses to compile: "mismatch in light queue definition"
            # (lightsource.sample.glsl:28-33)
            raise RuntimeError("mismatch in light queue definition: HostLightSource and LightSampler differ in polarization")
        if medium is None and scene is not None:
            medium = scene.medium
        medium = medium or 0
        self._mediumOwner = medium  # (a `DeviceAddress` keeps the material store's memory alive as long as this sampler)
        self._batchSize, self._capacity, self._scene, self._source = batchSize, capacity, scene, source
        self._retrieve, self._polarized, self._rng, self._wavelengthSource = retrieve, polarized, rng, wavelengthSource
        self._code = b"" if code is None else code
        self.setParams(count=capacity, baseCount=0, medium=medium)
        self._lightQueue = IOQueue(PolarizedLightSampleItem if polarized else LightSampleItem, capacity, mode="retrieve")
        self._lamQueue = IOQueue(WavelengthSampleItem, capacity, mode="retrieve")
        flags, lightCap, lamCap = sourceSpec(source, wavelengthSource)
        flags |= capi.FLAG_POLARIZED if polarized else 0
        if rng is not None:
            from .random import rngFlags

            flags |= rngFlags(rng)
        self._program = _Program(_backend.TracerSpec(capi.TRACER_LIGHT_SAMPLER, flags, 0, batchSize, 1, capacity,
                                                     light_queue_capacity=lightCap, wavelength_queue_capacity=lamCap), None)

    batchSize = property(lambda self: self._batchSize, doc="number of samples drawn per work group (API parity)")
    capacity = property(lambda self: self._capacity, doc="maximum number of samples that can be drawn per run")
    code = property(lambda self: self._code, doc="kept for API parity; kernels are precompiled")
    lightQueue = property(lambda self: self._lightQueue, doc="queue storing the light samples")
    polarized = property(lambda self: self._polarized)
    retrieve = property(lambda self: self._retrieve)
    rng = property(lambda self: self._rng)
    scene = property(lambda self: self._scene)
    source = property(lambda self: self._source)
    wavelengthSource = property(lambda self: self._wavelengthSource)
    wavelengthQueue = property(lambda self: self._lamQueue, doc="queue holding the sampled wavelengths")
    program = property(lambda self: self._program)

    def collectStages(self):
        chain = [self._wavelengthSource, self._source, self]
        return chain if self._rng is None else [self._rng] + chain

    def _assemble(self, i: int) -> capi.RunParams:
        P = capi.RunParams()
        sp = self._getParamsBlock("SampleParams", i)
        if sp.count > self._capacity:
            raise ValueError("count must be at most capacity")
        P.dispatch.batchSize = sp.count
        P.dispatch.groupCountX = -(self._capacity // -self._batchSize)
        P.dispatch.groupCountY = P.dispatch.groupCountZ = 1
        P.photon_offset = sp.baseCount
        P.volume.medium[0], P.volume.medium[1] = sp.medium.lo, sp.medium.hi
        if self._rng is not None:
            from .random import fillRngParams

            fillRngParams(P.philox, self._rng, i)
        fillSourceParams(P, i, self._source, self._wavelengthSource)
        return P

    def run(self, i: int):
        def launch() -> None:
            uploadSourceQueues(self._program, i, self._source, self._wavelengthSource)
            self._program.run(i, self._assemble(i))
            if self._retrieve:
                self._program.read_samples(self._lightQueue.buffer(i), self._lamQueue.buffer(i))

        return [launch]
