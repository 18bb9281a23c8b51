"""
ctypes wrapper of the CPU oracle (oracle/libtheia_oracle.so).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py. Nothing under theia_amd/ imports this module.
Parity status: see oracle/README.md.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

FLAG_POLARIZED = 1 << 0
FLAG_DISABLE_DIRECT = 1 << 1
FLAG_DISABLE_MIS = 1 << 2
FLAG_DISABLE_TRANSMISSION = 1 << 3
FLAG_DISABLE_VOLUME_BORDER = 1 << 4
FLAG_HIT_USE_TRANSMITTED = 1 << 5
FLAG_WAVELENGTH_UNIFORM = 1 << 6
FLAG_LIGHT_PENCIL = 1 << 10
FLAG_TARGET_INNER = 1 << 11
FLAG_KERNEL_HISTOGRAM = 1 << 13
FLAG_BRUTE_FORCE = 1 << 7
FLAG_LIGHT_HOST = 1 << 14
FLAG_LIGHT_QUEUE_POLARIZED = 1 << 15
FLAG_WAVELENGTH_HOST = 1 << 16
FLAG_LIGHT_CONE = 1 << 17
FLAG_LIGHT_POLARIZED = 1 << 18
FLAG_WAVELENGTH_FUNCTION = 1 << 19
FLAG_TARGET_DISK = 1 << 20
FLAG_TARGET_FLAT = 1 << 21
FLAG_GUIDE_DISK = 1 << 22
FLAG_GUIDE_FLAT = 1 << 23
FLAG_RNG_SOBOL = 1 << 24
FLAG_TARGET_MISS_INFINITE = 1 << 25
TRACK_WORDS = 20
N_STATS = 20


class _P(C.Structure):
    _pack_ = 4


class PhiloxParams(_P):
    _fields_ = [("key", C.c_uint32 * 2), ("offset", C.c_uint32 * 4)]


class LightParams(_P):
    _fields_ = [("position", C.c_float * 3), ("contribFwd", C.c_float), ("contribBwd", C.c_float), ("t_min", C.c_float), ("t_max", C.c_float),
                ("direction", C.c_float * 3), ("stokes", C.c_float * 4), ("polRef", C.c_float * 3), ("cosOpeningAngle", C.c_float)]


class WavelengthParams(_P):
    _fields_ = [("p0", C.c_float), ("p1", C.c_float), ("p2", C.c_float), ("table", C.c_uint32 * 2)]


class TargetParams(_P):
    _fields_ = [("position", C.c_float * 3), ("radius", C.c_float), ("invPos", C.c_float * 3), ("invRad", C.c_float), ("hemisphereProb", C.c_float),
                ("normal", C.c_float * 3), ("objToWorld", C.c_float * 9), ("width", C.c_float), ("height", C.c_float)]


class GuideParams(_P):
    _fields_ = [("position", C.c_float * 3), ("radius", C.c_float), ("prob", C.c_float),
                ("normal", C.c_float * 3), ("objToWorld", C.c_float * 9), ("width", C.c_float), ("height", C.c_float)]


class PropagateParams(_P):
    _fields_ = [("scatterCoefficient", C.c_float), ("lowerBBoxCorner", C.c_float * 3), ("upperBBoxCorner", C.c_float * 3), ("maxTime", C.c_float), ("maxDist", C.c_float)]


class VolumeTraceParams(_P):
    _fields_ = [("medium", C.c_uint32 * 2), ("objectId", C.c_int32), ("propagation", PropagateParams)]


class SceneTraceParams(_P):
    _fields_ = [("targetId", C.c_int32), ("sceneMedium", C.c_uint32 * 2), ("propagation", PropagateParams)]


class ResponseParams(_P):
    _fields_ = [("t0", C.c_float), ("binSize", C.c_float), ("kernelBandwidth", C.c_float), ("kernelSupport", C.c_float)]


class Common(C.Structure):
    _fields_ = [
        ("flags", C.c_uint32), ("path_length", C.c_uint32), ("block_size", C.c_uint32), ("n_bins", C.c_uint32),
        ("batch_size", C.c_uint32), ("photon_offset", C.c_uint32), ("normalization", C.c_float),
        ("philox", PhiloxParams), ("light", LightParams), ("wavelength", WavelengthParams), ("response", ResponseParams),
    ]


class Mesh(C.Structure):
    _fields_ = [("vertices", C.POINTER(C.c_float)), ("n_vertices", C.c_uint32), ("indices", C.POINTER(C.c_uint32)), ("n_triangles", C.c_uint32)]


class Instance(C.Structure):
    _fields_ = [("mesh", C.c_uint32), ("transform", C.c_float * 12), ("customIndex", C.c_int32), ("material", C.c_uint64)]


def build() -> str:
    """Compiles the oracle (gcc) if needed and returns the library path."""
    so = os.path.join(_HERE, "libtheia_oracle.so")
    src = os.path.join(_HERE, "theia_oracle.c")
    if not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return so


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        fp, u32p, u64p = C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
        L.orc_rng_fill.argtypes = [C.POINTER(PhiloxParams), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, u32p, fp]
        L.orc_rng_fill.restype = None
        L.orc_reduce.argtypes = [fp, C.c_uint32, C.c_uint32, C.c_float, fp]
        L.orc_reduce.restype = None
        L.orc_scene_create.argtypes = [C.POINTER(Mesh), C.c_uint32, C.POINTER(Instance), C.c_uint32]
        L.orc_scene_create.restype = C.c_void_p
        L.orc_scene_destroy.argtypes = [C.c_void_p]
        L.orc_scene_destroy.restype = None
        L.orc_scene_query.argtypes = [C.c_void_p, fp, fp, C.c_float, C.c_int, fp]
        L.orc_scene_query.restype = C.c_int
        L.orc_trace_volume.argtypes = [C.POINTER(Common), C.POINTER(VolumeTraceParams), C.POINTER(TargetParams), C.c_int, fp, u64p, fp]
        L.orc_trace_volume.restype = C.c_int
        L.orc_trace_scene.argtypes = [C.POINTER(Common), C.POINTER(SceneTraceParams), C.POINTER(GuideParams), C.c_void_p, C.c_int, fp, u64p, fp]
        L.orc_trace_scene.restype = C.c_int
        L.orc_lookup.argtypes = [C.c_uint64, C.c_float, C.c_float]
        L.orc_lookup.restype = C.c_float
        L.orc_lookup_medium.argtypes = [C.c_uint64, C.c_float, fp]
        L.orc_lookup_medium.restype = None
        L.orc_scatter_dir.argtypes = [fp, C.c_float, C.c_float, fp]
        L.orc_scatter_dir.restype = None
        L.orc_fresnel.argtypes = [C.c_float, C.c_float, C.c_float, fp]
        L.orc_fresnel.restype = None
        L.orc_scatter_prob.argtypes = [C.c_uint64, fp, fp]
        L.orc_scatter_prob.restype = C.c_float
        L.orc_scatter_sample.argtypes = [C.c_uint64, fp, C.c_float, C.c_float, fp, fp]
        L.orc_scatter_sample.restype = None
        L.orc_eval_guide.argtypes = [C.POINTER(GuideParams), fp, fp, fp]
        L.orc_eval_guide.restype = None
        L.orc_sample_guide.argtypes = [C.POINTER(GuideParams), fp, C.c_float, C.c_float, fp]
        L.orc_sample_guide.restype = None
        L.orc_rotate_polref.argtypes = [fp, fp, fp, fp, fp]
        L.orc_rotate_polref.restype = None
        L.orc_align_polref.argtypes = [fp, fp, fp, fp]
        L.orc_align_polref.restype = None
        L.orc_polarizer.argtypes = [C.c_float, C.c_float, fp, fp]
        L.orc_polarizer.restype = None
        L.orc_phase_matrix_apply.argtypes = [C.c_uint64, C.c_float, C.c_float, C.c_float, fp, fp]
        L.orc_phase_matrix_apply.restype = None
        L.orc_sign_bit.argtypes = [C.c_float]
        L.orc_sign_bit.restype = C.c_float
        L.orc_cherenkov_units.argtypes = [C.c_uint32, fp, C.c_uint32, fp]
        L.orc_cherenkov_units.restype = None
        _LIB = L
    return _LIB


def _fp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def reduce(hists: np.ndarray, norm: float = 1.0) -> np.ndarray:
    """HistogramReducer (estimator.reduce.glsl:17-35) over hists[n_hist][n_bins]"""
    h = np.ascontiguousarray(hists, dtype=np.float32)
    out = np.empty(h.shape[1], dtype=np.float32)
    lib().orc_reduce(_fp(h), h.shape[0], h.shape[1], float(norm), _fp(out))
    return out


def rng_fill(key: int, offset: int, base_stream: int, n_streams: int, first_dim: int, n_samples: int):
    p = PhiloxParams()
    p.key[0], p.key[1] = key & 0xFFFFFFFF, (key >> 32) & 0xFFFFFFFF
    for i in range(4):
        p.offset[i] = (offset >> (32 * i)) & 0xFFFFFFFF
    u = np.empty((n_streams, n_samples), dtype=np.uint32)
    f = np.empty((n_streams, n_samples), dtype=np.float32)
    lib().orc_rng_fill(C.byref(p), base_stream, n_streams, first_dim, n_samples, u.ctypes.data_as(C.POINTER(C.c_uint32)), _fp(f))
    return u, f


class HostBlob:
    """Host memory standing in for a device allocation (addresses are host pointers)."""

    def __init__(self, nbytes: int) -> None:
        self.buf = np.zeros(max(int(nbytes), 16), dtype=np.uint8)
        self.address = int(self.buf.ctypes.data)

    def upload(self, data: bytes) -> None:
        raw = np.frombuffer(bytes(data), dtype=np.uint8)
        self.buf[: raw.size] = raw


class Scene:
    def __init__(self, meshes, instances) -> None:
        self._keep = []
        md = (Mesh * len(meshes))()
        for i, (v, f) in enumerate(meshes):
            v = np.ascontiguousarray(v, dtype=np.float32)
            f = np.ascontiguousarray(f, dtype=np.uint32)
            self._keep += [v, f]
            md[i].vertices = _fp(v)
            md[i].n_vertices = v.shape[0]
            md[i].indices = f.ctypes.data_as(C.POINTER(C.c_uint32))
            md[i].n_triangles = f.shape[0]
        idesc = (Instance * len(instances))()
        for i, (mesh, trafo, custom, material) in enumerate(instances):
            t = np.ascontiguousarray(trafo, dtype=np.float32).reshape(12)
            idesc[i].mesh = int(mesh)
            for k in range(12):
                idesc[i].transform[k] = float(t[k])
            idesc[i].customIndex = int(custom)
            idesc[i].material = int(material)
        self.handle = lib().orc_scene_create(md, len(meshes), idesc, len(instances))

    def query(self, origin, direction, tmax: float, brute_force: bool = False):
        o = np.ascontiguousarray(origin, dtype=np.float32)
        d = np.ascontiguousarray(direction, dtype=np.float32)
        out = np.zeros(5, dtype=np.float32)
        hit = lib().orc_scene_query(self.handle, _fp(o), _fp(d), float(tmax), int(brute_force), _fp(out))
        return bool(hit), out

    def __del__(self) -> None:
        try:
            if self.handle:
                lib().orc_scene_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


def set_hit_queue(queue: np.ndarray | None, capacity: int, polarized: bool) -> None:
    """HitRecorder response: the next traces append their hits to `queue` (uint32 array:
    count + columns, shader/response.queue.glsl layout) instead of binning; None switches back."""
    fn = lib().orc_set_hit_queue
    fn.argtypes = [C.c_void_p, C.c_uint32, C.c_int]
    fn.restype = None
    fn(queue.ctypes.data if queue is not None else None, int(capacity), int(bool(polarized)))


def set_time_queue(queue: np.ndarray | None, capacity: int, store_object_id: bool) -> None:
    """StoreTimeHitResponse: the next traces keep a hit's time with probability `contrib`"""
    fn = lib().orc_set_time_queue
    fn.argtypes = [C.c_void_p, C.c_uint32, C.c_int]
    fn.restype = None
    fn(queue.ctypes.data if queue is not None else None, int(capacity), int(bool(store_object_id)))


def set_photon_mode(steps_per_run: int, rng_per_run: int) -> None:
    """ScenePhotonTracer semantics for the next scene traces; (0, 0) switches back"""
    fn = lib().orc_set_photon_mode
    fn.argtypes = [C.c_uint32, C.c_uint32]
    fn.restype = None
    fn(int(steps_per_run), int(rng_per_run))


def trace_volume(common: Common, trace: VolumeTraceParams, target: TargetParams, *, threads: int = 0, track: bool = False):
    hist = np.zeros(common.n_bins, dtype=np.float32)
    stats = np.zeros(N_STATS, dtype=np.uint64)
    tr = None
    if track:
        tr = np.zeros((common.batch_size, common.path_length + 3, TRACK_WORDS), dtype=np.float32)
    rc = lib().orc_trace_volume(C.byref(common), C.byref(trace), C.byref(target), threads, _fp(hist),
                                stats.ctypes.data_as(C.POINTER(C.c_uint64)), _fp(tr) if tr is not None else None)
    if rc != 0:
        raise RuntimeError(f"orc_trace_volume failed: {rc}")
    return hist, stats, tr


def sobol_fill(seed: int, base_stream: int, n_streams: int, first_dim: int, n_samples: int, sample_dim: int = 1) -> np.ndarray:
    """RNGBufferSink over a SobolQRNG"""
    fn = lib().orc_sobol_fill
    fn.argtypes = [C.c_uint32] * 6 + [C.c_void_p]
    fn.restype = None
    shape = (int(n_streams), int(n_samples)) if sample_dim == 1 else (int(n_streams), int(n_samples), 2)
    out = np.zeros(shape, dtype=np.float32)
    fn(int(seed) & 0xFFFFFFFF, int(base_stream), int(n_streams), int(first_dim), int(n_samples), int(sample_dim), out.ctypes.data)
    return out


def probe_target(philox: PhiloxParams, flags: int, target: TargetParams, box_min, box_max, n: int) -> np.ndarray:
    """theia.testing.TargetSampler: [n][25] words (see theia_oracle.h)"""
    fn = lib().orc_probe_target
    fn.argtypes = [C.POINTER(PhiloxParams), C.c_uint32, C.POINTER(TargetParams), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_uint32, C.c_void_p]
    fn.restype = None
    out = np.zeros((int(n), 25), dtype=np.float32)
    lo, hi = (C.c_float * 3)(*box_min), (C.c_float * 3)(*box_max)
    fn(C.byref(philox), int(flags), C.byref(target), lo, hi, int(n), out.ctypes.data)
    return out


def probe_guide(philox: PhiloxParams, flags: int, guide: GuideParams, box_min, box_max, n: int) -> np.ndarray:
    """theia.testing.TargetGuideSampler: [n][13] floats"""
    fn = lib().orc_probe_guide
    fn.argtypes = [C.POINTER(PhiloxParams), C.c_uint32, C.POINTER(GuideParams), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_uint32, C.c_void_p]
    fn.restype = None
    out = np.zeros((int(n), 13), dtype=np.float32)
    lo, hi = (C.c_float * 3)(*box_min), (C.c_float * 3)(*box_max)
    fn(C.byref(philox), int(flags), C.byref(guide), lo, hi, int(n), out.ctypes.data)
    return out


def set_light_queue(queue: np.ndarray | None, capacity: int = 0) -> None:
    """HostLightSource: float32 SoA queue (no counter) the next traces / samplers read"""
    fn = lib().orc_set_light_queue
    fn.argtypes = [C.c_void_p, C.c_uint32]
    fn.restype = None
    fn(queue.ctypes.data if queue is not None else None, int(capacity))


def set_wavelength_queue(queue: np.ndarray | None, capacity: int = 0) -> None:
    """HostWavelengthSource: float32 SoA queue (wavelength[cap], contrib[cap])"""
    fn = lib().orc_set_wavelength_queue
    fn.argtypes = [C.c_void_p, C.c_uint32]
    fn.restype = None
    fn(queue.ctypes.data if queue is not None else None, int(capacity))


def sample_light(common: Common, medium: int, count: int, base: int, polarized: bool, capacity: int):
    """LightSampler: -> (light queue, wavelength queue) as uint32 arrays (count + columns)"""
    fn = lib().orc_sample_light
    fn.argtypes = [C.POINTER(Common), C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p]
    fn.restype = C.c_int
    light = np.zeros(1 + (15 if polarized else 8) * capacity, dtype=np.uint32)
    lam = np.zeros(1 + 2 * capacity, dtype=np.uint32)
    rc = fn(C.byref(common), int(medium), int(count), int(base), int(bool(polarized)), int(capacity), light.ctypes.data, lam.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"orc_sample_light failed: {rc}")
    return light, lam


def replay(common: Common, queue: np.ndarray, capacity: int, polarized: bool, *, threads: int = 0):
    """HitReplay: every item of `queue` (uint8/uint32 array, count + SoA columns) -> response"""
    fn = lib().orc_replay
    fn.argtypes = [C.POINTER(Common), C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_uint64)]
    fn.restype = C.c_int
    hist = np.zeros(common.n_bins, dtype=np.float32)
    stats = np.zeros(N_STATS, dtype=np.uint64)
    q = np.ascontiguousarray(queue)
    rc = fn(C.byref(common), q.ctypes.data, int(capacity), int(bool(polarized)), threads, _fp(hist),
            stats.ctypes.data_as(C.POINTER(C.c_uint64)))
    if rc != 0:
        raise RuntimeError(f"orc_replay failed: {rc}")
    return hist, stats


def trace_scene(common: Common, trace: SceneTraceParams, guide: GuideParams | None, scene: Scene, *, threads: int = 0, track: bool = False):
    hist = np.zeros(common.n_bins, dtype=np.float32)
    stats = np.zeros(N_STATS, dtype=np.uint64)
    tr = None
    if track:
        tr = np.zeros((common.batch_size, common.path_length + 3, TRACK_WORDS), dtype=np.float32)
    rc = lib().orc_trace_scene(C.byref(common), C.byref(trace), C.byref(guide) if guide is not None else None, scene.handle,
                               threads, _fp(hist), stats.ctypes.data_as(C.POINTER(C.c_uint64)), _fp(tr) if tr is not None else None)
    if rc != 0:
        raise RuntimeError(f"orc_trace_scene failed: {rc}")
    return hist, stats, tr
