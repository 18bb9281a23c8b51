"""
ctypes binding of ``libtheia_hip.so`` (C ABI declared in ``include/theia_hip.h``).

The library is the product's only compute path; loading fails loudly when the
shared object is missing (run ``python -c "import __graft_entry__ as g; g.build()"``
or ``make -C theia_amd/csrc``) and creating a context fails loudly when no HIP
device is present. There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_size_t, c_uint32, c_uint64, c_void_p

__all__ = [
    "lib", "lib_path", "check", "ThxError",
    "DispatchParams", "PhiloxParams", "LightParams", "WavelengthParams", "TargetParams",
    "GuideParams", "PropagateParams", "VolumeTraceParams", "SceneTraceParams", "ResponseParams",
    "RunParams", "TracerDesc", "MeshDesc", "InstanceDesc", "SceneInfo",
    "TRACER_VOLUME_FORWARD", "TRACER_SCENE_FORWARD", "TRACER_HIT_REPLAY", "N_STATS", "TRACK_WORDS", "STAT_NAMES",
    "FLAG_POLARIZED", "FLAG_DISABLE_DIRECT", "FLAG_DISABLE_MIS", "FLAG_DISABLE_TRANSMISSION",
    "FLAG_DISABLE_VOLUME_BORDER", "FLAG_HIT_USE_TRANSMITTED", "FLAG_WAVELENGTH_UNIFORM",
    "FLAG_STATISTICS", "FLAG_TRACK", "FLAG_LIGHT_PENCIL", "FLAG_TARGET_INNER", "FLAG_PHOTON", "FLAG_KERNEL_HISTOGRAM", "EXPORTED_SYMBOLS",
    "FLAG_LIGHT_HOST", "FLAG_LIGHT_QUEUE_POLARIZED", "FLAG_WAVELENGTH_HOST", "FLAG_LIGHT_CONE", "FLAG_LIGHT_POLARIZED",
    "FLAG_WAVELENGTH_FUNCTION", "TRACER_LIGHT_SAMPLER", "FLAG_TARGET_DISK", "FLAG_TARGET_FLAT", "FLAG_TARGET_MISS_INFINITE", "FLAG_GUIDE_DISK", "FLAG_GUIDE_FLAT", "FLAG_RNG_SOBOL",
]

TRACER_VOLUME_FORWARD = 0
TRACER_SCENE_FORWARD = 1
TRACER_HIT_REPLAY = 2
TRACER_LIGHT_SAMPLER = 3
RESPONSE_HISTOGRAM = 0
RESPONSE_RECORD = 1
RESPONSE_STORE_TIME = 2
FLAG_POLARIZED = 1 << 0
FLAG_DISABLE_DIRECT = 1 << 1
FLAG_DISABLE_MIS = 1 << 2
FLAG_DISABLE_TRANSMISSION = 1 << 3
FLAG_DISABLE_VOLUME_BORDER = 1 << 4
FLAG_HIT_USE_TRANSMITTED = 1 << 5
FLAG_WAVELENGTH_UNIFORM = 1 << 6
FLAG_STATISTICS = 1 << 8
FLAG_TRACK = 1 << 9
FLAG_LIGHT_PENCIL = 1 << 10
FLAG_TARGET_INNER = 1 << 11
FLAG_PHOTON = 1 << 12
FLAG_KERNEL_HISTOGRAM = 1 << 13
FLAG_LIGHT_HOST = 1 << 14
FLAG_LIGHT_QUEUE_POLARIZED = 1 << 15
FLAG_WAVELENGTH_HOST = 1 << 16
FLAG_LIGHT_CONE = 1 << 17
FLAG_LIGHT_POLARIZED = 1 << 18
FLAG_WAVELENGTH_FUNCTION = 1 << 19
FLAG_TARGET_DISK = 1 << 20
FLAG_TARGET_FLAT = 1 << 21
FLAG_TARGET_MISS_INFINITE = 1 << 25
FLAG_LIGHT_KIND_MASK = 7 << 26  # the Cherenkov family of light sources
FLAG_LIGHT_CHERENKOV = 1 << 26
FLAG_LIGHT_CHERENKOV_TRACK = 2 << 26
FLAG_LIGHT_MUON_TRACK = 3 << 26
FLAG_LIGHT_CASCADE = 4 << 26
FLAG_FRANK_TAMM_PHOTONS = 1 << 29
FLAG_FRANK_TAMM_IS = 1 << 30
# thx_probe_unit: function -> (id, input columns, output columns)
UNIT_PROBES = {"lookup": (0, 2, 1), "lookup_medium": (1, 1, 4), "scatter_dir": (2, 5, 3), "fresnel": (3, 3, 2), "scatter_prob": (4, 6, 1),
               "scatter_sample": (5, 5, 4), "rotate_polref": (6, 9, 5), "align_polref": (7, 9, 2), "phase_matrix": (8, 7, 4),
               "polarizer": (9, 6, 4), "sign_bit": (10, 1, 1),
               "emission_angle": (11, 4, 1), "emission_angle_range": (12, 6, 2), "emission_pdf": (13, 4, 1), "frank_tamm": (14, 2, 3)}
FLAG_GUIDE_DISK = 1 << 22
FLAG_GUIDE_FLAT = 1 << 23
FLAG_RNG_SOBOL = 1 << 24
N_STATS = 20
TRACK_WORDS = 20
MAX_SLOTS = 16  # THX_MAX_SLOTS: result slots of a tracer (0 / 1 = the pipeline configs)
STAT_NAMES = (
    "created", "scattered", "hit", "detected", "volume", "lost", "decayed", "absorbed", "missed",
    "maxIter", "error", "mismatch",
    "queries", "nodes", "triangles", "instances", "hits", "lut_reads", "responses", "steps",
)


class ThxError(RuntimeError):
    """Error reported by libtheia_hip.so"""

    def __init__(self, code: int, text: str) -> None:
        super().__init__(f"libtheia_hip error {code}: {text}")
        self.code = code


class _Packed(Structure):
    _pack_ = 4


class DispatchParams(_Packed):
    _fields_ = [("batchSize", c_uint32), ("groupCountX", c_uint32), ("groupCountY", c_uint32), ("groupCountZ", c_uint32)]


class PhiloxParams(_Packed):
    _fields_ = [("key", c_uint32 * 2), ("offset", c_uint32 * 4)]


class _BasicLight(_Packed):
    _fields_ = [("position", c_float * 3), ("contribFwd", c_float), ("contribBwd", c_float), ("t_min", c_float), ("t_max", c_float),
                ("direction", c_float * 3), ("stokes", c_float * 4), ("polRef", c_float * 3), ("cosOpeningAngle", c_float)]


class CherenkovParams(_Packed):
    _fields_ = [("trackStart", c_float * 3), ("startTime", c_float), ("trackEnd", c_float * 3), ("endTime", c_float),
                ("trackDir", c_float * 3), ("trackDist", c_float)]


class CherenkovTrackParams(_Packed):
    _fields_ = [("track", c_uint32 * 2)]


class MuonTrackParams(_Packed):
    _fields_ = [("startPosition", c_float * 3), ("startTime", c_float), ("endPosition", c_float * 3), ("endTime", c_float),
                ("energyScale", c_float), ("a_angular", c_float), ("b_angular", c_float)]


class CascadeParams(_Packed):
    _fields_ = [("startPosition", c_float * 3), ("startTime", c_float), ("direction", c_float * 3), ("effectiveLength", c_float),
                ("a_angular", c_float), ("b_angular", c_float), ("a_long", c_float), ("b_long", c_float)]


class LightParams(C.Union):
    """thx_light_params: one block for whichever light source the tracer was created with"""
    _pack_ = 4
    _anonymous_ = ("basic",)
    _fields_ = [("basic", _BasicLight), ("cherenkov", CherenkovParams), ("cherenkovTrack", CherenkovTrackParams),
                ("muonTrack", MuonTrackParams), ("cascade", CascadeParams)]


class WavelengthParams(_Packed):
    _fields_ = [("p0", c_float), ("p1", c_float), ("p2", c_float), ("table", c_uint32 * 2)]


class TargetParams(_Packed):
    _fields_ = [("position", c_float * 3), ("radius", c_float), ("invPos", c_float * 3), ("invRad", c_float), ("hemisphereProb", c_float),
                ("normal", c_float * 3), ("objToWorld", c_float * 9), ("width", c_float), ("height", c_float)]


class GuideParams(_Packed):
    _fields_ = [("position", c_float * 3), ("radius", c_float), ("prob", c_float),
                ("normal", c_float * 3), ("objToWorld", c_float * 9), ("width", c_float), ("height", c_float)]


class PropagateParams(_Packed):
    _fields_ = [("scatterCoefficient", c_float), ("lowerBBoxCorner", c_float * 3), ("upperBBoxCorner", c_float * 3), ("maxTime", c_float), ("maxDist", c_float)]


class VolumeTraceParams(_Packed):
    _fields_ = [("medium", c_uint32 * 2), ("objectId", c_int32), ("propagation", PropagateParams)]


class SceneTraceParams(_Packed):
    _fields_ = [("targetId", c_int32), ("sceneMedium", c_uint32 * 2), ("propagation", PropagateParams)]


class ResponseParams(_Packed):
    _fields_ = [("t0", c_float), ("binSize", c_float), ("kernelBandwidth", c_float), ("kernelSupport", c_float)]


class RunParams(_Packed):
    _fields_ = [
        ("dispatch", DispatchParams),
        ("photon_offset", c_uint32),
        ("philox", PhiloxParams),
        ("light", LightParams),
        ("wavelength", WavelengthParams),
        ("response", ResponseParams),
        ("volume", VolumeTraceParams),
        ("target", TargetParams),
        ("scene", SceneTraceParams),
        ("guide", GuideParams),
        ("photon_steps_per_run", c_uint32),
        ("photon_rng_per_run", c_uint32),
    ]


class TracerDesc(Structure):
    _fields_ = [
        ("kind", c_uint32), ("flags", c_uint32), ("path_length", c_uint32), ("block_size", c_uint32),
        ("n_bins", c_uint32), ("capacity", c_uint32), ("response_kind", c_uint32), ("hit_capacity", c_uint32),
        ("hit_polarized", c_uint32), ("light_queue_capacity", c_uint32), ("wavelength_queue_capacity", c_uint32),
        ("reserved", c_uint32 * 1),
    ]


class MeshDesc(Structure):
    _fields_ = [("vertices", POINTER(c_float)), ("n_vertices", c_uint32), ("indices", POINTER(c_uint32)), ("n_triangles", c_uint32)]


class InstanceDesc(Structure):
    _fields_ = [("mesh", c_uint32), ("transform", c_float * 12), ("customIndex", c_int32), ("material", c_uint64)]


class SceneInfo(Structure):
    _fields_ = [
        ("n_nodes", c_uint32), ("n_triangles", c_uint32), ("n_instances", c_uint32), ("max_depth", c_uint32),
        ("bytes_nodes", c_uint64), ("bytes_triangles", c_uint64), ("bytes_instances", c_uint64), ("bytes_total", c_uint64),
        ("layout_box_bytes", c_uint32), ("layout_triangle_bytes", c_uint32), ("layout_instance_bytes", c_uint32), ("layout_hit_bytes", c_uint32),
    ]


# every symbol include/theia_hip.h declares: name -> (restype, argtypes)
_PROTOS = {
    "thx_last_error": (c_char_p, []),
    "thx_device_count": (c_int, [POINTER(c_int)]),
    "thx_ctx_create": (c_int, [c_int, POINTER(c_void_p)]),
    "thx_ctx_destroy": (c_int, [c_void_p]),
    "thx_ctx_device_name": (c_int, [c_void_p, c_char_p, c_size_t]),
    "thx_ctx_device_info": (c_int, [c_void_p, POINTER(c_uint32), POINTER(c_uint32), POINTER(c_uint64)]),
    "thx_ctx_sync": (c_int, [c_void_p]),
    "thx_ctx_concurrent_kernels": (c_int, [c_void_p, POINTER(c_uint32)]),
    "thx_mem_alloc": (c_int, [c_void_p, c_size_t, POINTER(c_uint64)]),
    "thx_mem_free": (c_int, [c_void_p, c_uint64]),
    "thx_mem_upload": (c_int, [c_void_p, c_uint64, c_void_p, c_size_t]),
    "thx_mem_download": (c_int, [c_void_p, c_void_p, c_uint64, c_size_t]),
    "thx_host_register": (c_int, [c_void_p, c_void_p, c_size_t]),
    "thx_host_unregister": (c_int, [c_void_p, c_void_p]),
    "thx_scene_create": (c_int, [c_void_p, POINTER(MeshDesc), c_uint32, POINTER(InstanceDesc), c_uint32, POINTER(c_void_p)]),
    "thx_scene_destroy": (c_int, [c_void_p]),
    "thx_scene_get_info": (c_int, [c_void_p, POINTER(SceneInfo)]),
    "thx_scene_query": (c_int, [c_void_p, POINTER(c_float), c_uint32, POINTER(c_float)]),
    "thx_scene_query_ms": (c_int, [c_void_p, POINTER(c_float)]),
    "thx_tracer_create": (c_int, [c_void_p, POINTER(TracerDesc), c_void_p, POINTER(c_void_p)]),
    "thx_tracer_destroy": (c_int, [c_void_p]),
    "thx_tracer_run": (c_int, [c_void_p, c_uint32, POINTER(RunParams), c_void_p]),
    "thx_tracer_reduce": (c_int, [c_void_p, c_uint32, c_float, c_uint64, c_void_p]),
    "thx_tracer_read_hist": (c_int, [c_void_p, c_uint32, POINTER(c_float), c_uint32]),
    "thx_tracer_set_concurrency": (c_int, [c_void_p, c_uint32, c_uint32]),
    "thx_tracer_fetch_hist": (c_int, [c_void_p, c_uint32, c_void_p]),
    "thx_tracer_collect_hist": (c_int, [c_void_p, c_uint32, POINTER(c_float), c_uint32]),
    "thx_reduce_histograms": (c_int, [c_void_p, POINTER(c_float), c_uint32, c_uint32, c_float, POINTER(c_float)]),
    "thx_tracer_stats_device_ptr": (c_int, [c_void_p, POINTER(c_uint64)]),
    "thx_tracer_check": (c_int, [c_void_p]),
    "thx_tracer_hist_device_ptr": (c_int, [c_void_p, c_uint32, POINTER(c_uint64)]),
    "thx_tracer_read_stats": (c_int, [c_void_p, POINTER(c_uint64), c_uint32]),
    "thx_tracer_reset_stats": (c_int, [c_void_p]),
    "thx_tracer_read_stage_profile": (c_int, [c_void_p, POINTER(c_uint64)]),
    "thx_tracer_read_track": (c_int, [c_void_p, POINTER(c_float), c_size_t]),
    "thx_hit_queue_bytes": (c_size_t, [c_uint32, c_uint32]),
    "thx_time_queue_bytes": (c_size_t, [c_uint32, c_uint32]),
    "thx_tracer_read_hits": (c_int, [c_void_p, c_void_p, c_size_t, c_int]),
    "thx_tracer_read_hits_cfg": (c_int, [c_void_p, c_uint32, c_void_p, c_size_t, c_int]),
    "thx_tracer_hits_device_ptr": (c_int, [c_void_p, POINTER(c_uint64)]),
    "thx_tracer_write_hits": (c_int, [c_void_p, c_void_p, c_size_t]),
    "thx_tracer_write_light": (c_int, [c_void_p, c_uint32, c_void_p, c_size_t]),
    "thx_tracer_write_wavelengths": (c_int, [c_void_p, c_uint32, c_void_p, c_size_t]),
    "thx_tracer_read_samples": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_size_t]),
    "thx_tracer_launch_info": (c_int, [c_void_p, c_uint32, POINTER(c_uint32), POINTER(c_uint32), POINTER(c_uint32)]),
    "thx_tracer_last_kernel_ms": (c_int, [c_void_p, c_uint32, POINTER(c_float)]),
    "thx_tracer_kernel_times": (c_int, [c_void_p, POINTER(c_float), c_uint32, POINTER(c_uint32)]),
    "thx_probe_target": (c_int, [c_void_p, POINTER(PhiloxParams), c_uint32, POINTER(TargetParams), POINTER(c_float), POINTER(c_float), c_uint32, c_void_p]),
    "thx_probe_guide": (c_int, [c_void_p, POINTER(PhiloxParams), c_uint32, POINTER(GuideParams), POINTER(c_float), POINTER(c_float), c_uint32, c_void_p]),
    "thx_probe_unit": (c_int, [c_void_p, c_uint32, c_uint64, POINTER(c_float), c_uint32, c_uint32, POINTER(c_float), c_uint32]),
    "thx_rng_fill_sobol": (c_int, [c_void_p, c_uint32, c_uint32, c_uint32, c_uint32, c_uint32, c_uint32, POINTER(c_float)]),
    "thx_rng_fill": (c_int, [c_void_p, POINTER(PhiloxParams), c_uint32, c_uint32, c_uint32, c_uint32, POINTER(c_uint32), POINTER(c_float)]),
    "thx_comm_unique_id": (c_int, [c_void_p]),
    "thx_comm_init": (c_int, [c_void_p, c_void_p, c_int, c_int, POINTER(c_void_p)]),
    "thx_comm_destroy": (c_int, [c_void_p]),
    "thx_hist_allreduce": (c_int, [c_void_p, c_uint64, c_size_t, c_void_p]),
    "thx_stats_allreduce": (c_int, [c_void_p, c_uint64, c_size_t, c_void_p]),
    "thx_hist_gather_reduce": (c_int, [c_void_p, c_uint64, c_size_t, c_void_p]),
}
EXPORTED_SYMBOLS = tuple(_PROTOS.keys())

_lib = None


def lib_path() -> str:
    override = os.environ.get("THEIA_AMD_LIB")  # development: A/B builds of the same ABI
    if override:
        return override
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libtheia_hip.so")


def lib() -> C.CDLL:
    """Loads libtheia_hip.so (once). Raises if it has not been built."""
    global _lib
    if _lib is None:
        path = lib_path()
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} not found: the HIP extension has not been built. Run "
                "`make -C theia_amd/csrc` (or __graft_entry__.build()). theia_amd has no CPU fallback."
            )
        handle = C.CDLL(path)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(code: int) -> None:
    if code != 0:
        text = lib().thx_last_error()
        raise ThxError(code, text.decode("utf-8", "replace") if text else "unknown error")
