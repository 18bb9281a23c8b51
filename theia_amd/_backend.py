"""
Device backend used by the host-side stages: a thin object layer over the
C ABI (``include/theia_hip.h``). One context per process/device.

`get()` creates the HIP backend on first use and raises if the shared library
or a GPU is missing -- the product never computes on the CPU. The test-suite may
install a stand-in with the same interface through `install()` (used only by
``tests/`` to exercise host logic on machines without a GPU).
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

from . import _capi as capi

__all__ = ["HipBackend", "get", "install", "reset", "TracerSpec"]


@dataclass
class TracerSpec:
    kind: int
    flags: int
    path_length: int
    block_size: int
    n_bins: int
    capacity: int
    response_kind: int = 0   # capi.RESPONSE_*
    hit_capacity: int = 0    # HitRecorder queue size
    hit_polarized: bool = False
    light_queue_capacity: int = 0       # HostLightSource queue size
    wavelength_queue_capacity: int = 0  # HostWavelengthSource queue size


class HipBackend:
    """libtheia_hip.so on one MI355X."""

    name = "hip"

    def __init__(self, device: int | None = None) -> None:
        self._lib = capi.lib()
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", os.environ.get("THEIA_AMD_DEVICE", "0")))
            n = C.c_int(0)
            capi.check(self._lib.thx_device_count(C.byref(n)))
            if n.value > 0:
                device %= n.value
        ctx = C.c_void_p()
        capi.check(self._lib.thx_ctx_create(int(device), C.byref(ctx)))
        self._ctx = ctx
        self.device = int(device)

    # -- device ------------------------------------------------------------
    def device_name(self) -> str:
        buf = C.create_string_buffer(256)
        capi.check(self._lib.thx_ctx_device_name(self._ctx, buf, 256))
        return buf.value.decode()

    def device_info(self) -> dict:
        cus, lds, mem = C.c_uint32(), C.c_uint32(), C.c_uint64()
        capi.check(self._lib.thx_ctx_device_info(self._ctx, C.byref(cus), C.byref(lds), C.byref(mem)))
        return {"compute_units": cus.value, "lds_per_cu": lds.value, "hbm_bytes": mem.value}

    def sync(self) -> None:
        capi.check(self._lib.thx_ctx_sync(self._ctx))

    def concurrent_kernels(self) -> int:
        """kernels of different streams the runtime really runs side by side (measured once, ~2 ms; thx_ctx_concurrent_kernels)"""
        n = C.c_uint32()
        capi.check(self._lib.thx_ctx_concurrent_kernels(self._ctx, C.byref(n)))
        return int(n.value)

    # -- memory ------------------------------------------------------------
    def alloc(self, nbytes: int) -> int:
        addr = C.c_uint64()
        capi.check(self._lib.thx_mem_alloc(self._ctx, int(nbytes), C.byref(addr)))
        return int(addr.value)

    def free(self, addr: int) -> None:
        capi.check(self._lib.thx_mem_free(self._ctx, int(addr)))

    def upload(self, addr: int, data) -> None:
        buf = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8)) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data)
        capi.check(self._lib.thx_mem_upload(self._ctx, int(addr), buf.ctypes.data_as(C.c_void_p), buf.nbytes))

    def download(self, addr: int, nbytes: int) -> bytes:
        out = np.empty(int(nbytes), dtype=np.uint8)
        capi.check(self._lib.thx_mem_download(self._ctx, out.ctypes.data_as(C.c_void_p), int(addr), int(nbytes)))
        return out.tobytes()

    def host_register(self, array: np.ndarray) -> None:
        """page-locks a host array (device copies to / from it then run on the DMA engines, beside any kernel); idempotent"""
        capi.check(self._lib.thx_host_register(self._ctx, array.ctypes.data_as(C.c_void_p), array.nbytes))

    def host_unregister(self, array: np.ndarray) -> None:
        capi.check(self._lib.thx_host_unregister(self._ctx, array.ctypes.data_as(C.c_void_p)))

    # -- scene -------------------------------------------------------------
    def scene_create(self, meshes, instances) -> int:
        """meshes: [(vertices (N,6) f32, faces (M,3) u32)], instances:
        [(mesh index, transform (3,4), customIndex, material address)]"""
        keep = []
        md = (capi.MeshDesc * len(meshes))()
        for i, (v, f) in enumerate(meshes):
            v = np.ascontiguousarray(v, dtype=np.float32)
            f = np.ascontiguousarray(f, dtype=np.uint32)
            if v.ndim != 2 or v.shape[1] != 6 or f.ndim != 2 or f.shape[1] != 3:
                raise ValueError("mesh arrays must have shapes (N,6) and (M,3)")
            keep += [v, f]
            md[i].vertices = v.ctypes.data_as(C.POINTER(C.c_float))
            md[i].n_vertices = v.shape[0]
            md[i].indices = f.ctypes.data_as(C.POINTER(C.c_uint32))
            md[i].n_triangles = f.shape[0]
        idesc = (capi.InstanceDesc * len(instances))()
        for i, (mesh, trafo, custom, material) in enumerate(instances):
            t = np.ascontiguousarray(trafo, dtype=np.float32).reshape(12)
            idesc[i].mesh = int(mesh)
            for k in range(12):
                idesc[i].transform[k] = float(t[k])
            idesc[i].customIndex = int(custom)
            idesc[i].material = int(material)
        handle = C.c_void_p()
        capi.check(self._lib.thx_scene_create(self._ctx, md, len(meshes), idesc, len(instances), C.byref(handle)))
        return handle.value

    def scene_destroy(self, scene: int) -> None:
        capi.check(self._lib.thx_scene_destroy(C.c_void_p(scene)))

    def scene_info(self, scene: int) -> dict:
        info = capi.SceneInfo()
        capi.check(self._lib.thx_scene_get_info(C.c_void_p(scene), C.byref(info)))
        return {n: getattr(info, n) for n, _ in capi.SceneInfo._fields_}

    def scene_query(self, scene: int, origins, dirs, tmax) -> np.ndarray:
        o = np.asarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.asarray(dirs, dtype=np.float32).reshape(-1, 3)
        t = np.broadcast_to(np.asarray(tmax, dtype=np.float32), (o.shape[0],))
        rays = np.ascontiguousarray(np.concatenate([o, d, t[:, None]], axis=1), dtype=np.float32)
        out = np.empty((o.shape[0], 6), dtype=np.float32)
        capi.check(self._lib.thx_scene_query(C.c_void_p(scene), rays.ctypes.data_as(C.POINTER(C.c_float)), o.shape[0], out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    # -- tracer ------------------------------------------------------------
    def tracer_create(self, spec: TracerSpec, scene: int | None) -> int:
        desc = capi.TracerDesc(spec.kind, spec.flags, spec.path_length, spec.block_size, spec.n_bins, spec.capacity,
                               spec.response_kind, spec.hit_capacity, int(spec.hit_polarized), spec.light_queue_capacity,
                               spec.wavelength_queue_capacity)
        handle = C.c_void_p()
        capi.check(self._lib.thx_tracer_create(self._ctx, C.byref(desc), C.c_void_p(scene) if scene else None, C.byref(handle)))
        return handle.value

    def tracer_destroy(self, tracer: int) -> None:
        capi.check(self._lib.thx_tracer_destroy(C.c_void_p(tracer)))

    def tracer_run(self, tracer: int, cfg: int, params: capi.RunParams, stream: int = 0) -> None:
        capi.check(self._lib.thx_tracer_run(C.c_void_p(tracer), int(cfg), C.byref(params), C.c_void_p(stream) if stream else None))

    def tracer_reduce(self, tracer: int, cfg: int, norm: float, d_out: int = 0, stream: int = 0) -> None:
        capi.check(self._lib.thx_tracer_reduce(C.c_void_p(tracer), int(cfg), float(norm), int(d_out), C.c_void_p(stream) if stream else None))

    def probe_unit(self, name: str, inputs, addr: int = 0) -> np.ndarray:
        """one device function of the path applied to every row of `inputs` (thx_probe_unit)"""
        which, n_in, n_out = capi.UNIT_PROBES[name]
        x = np.ascontiguousarray(np.asarray(inputs, dtype=np.float32).reshape(-1, n_in))
        out = np.empty((x.shape[0], n_out), dtype=np.float32)
        capi.check(self._lib.thx_probe_unit(self._ctx, which, int(addr), x.ctypes.data_as(C.POINTER(C.c_float)), n_in, x.shape[0],
                                            out.ctypes.data_as(C.POINTER(C.c_float)), n_out))
        return out

    def tracer_check(self, tracer: int) -> None:
        """waits for the tracer's launches; raises ThxError if a kernel flagged a device error"""
        capi.check(self._lib.thx_tracer_check(C.c_void_p(tracer)))

    # -- multi-GPU exchange step (RCCL through the C ABI) -------------------
    def comm_unique_id(self) -> bytes:
        buf = C.create_string_buffer(128)
        capi.check(self._lib.thx_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, unique_id: bytes, rank: int, world: int) -> int:
        if len(unique_id) != 128:
            raise ValueError("an RCCL unique id has 128 bytes")
        out = C.c_void_p()
        capi.check(self._lib.thx_comm_init(self._ctx, C.create_string_buffer(unique_id, 128), int(rank), int(world), C.byref(out)))
        return out.value

    def comm_destroy(self, comm: int) -> None:
        capi.check(self._lib.thx_comm_destroy(C.c_void_p(comm)))

    def hist_allreduce(self, comm: int, d_hist: int, n_floats: int, stream: int = 0) -> None:
        capi.check(self._lib.thx_hist_allreduce(C.c_void_p(comm), int(d_hist), int(n_floats), C.c_void_p(stream) if stream else None))

    def stats_allreduce(self, comm: int, d_counters: int, n: int, stream: int = 0) -> None:
        capi.check(self._lib.thx_stats_allreduce(C.c_void_p(comm), int(d_counters), int(n), C.c_void_p(stream) if stream else None))

    def hist_gather_reduce(self, comm: int, d_hist: int, n_floats: int, stream: int = 0) -> None:
        """deterministic histogram sum: all-gather + fixed-order Kahan (thx_hist_gather_reduce)"""
        capi.check(self._lib.thx_hist_gather_reduce(C.c_void_p(comm), int(d_hist), int(n_floats), C.c_void_p(stream) if stream else None))

    def tracer_stats_ptr(self, tracer: int) -> int:
        out = C.c_uint64()
        capi.check(self._lib.thx_tracer_stats_device_ptr(C.c_void_p(tracer), C.byref(out)))
        return int(out.value)

    def reduce_histograms(self, hists, norm: float = 1.0) -> np.ndarray:
        """HistogramReducer on a host array [n_hist][n_bins] (thx_reduce_histograms)"""
        h = np.ascontiguousarray(hists, dtype=np.float32)
        out = np.empty(h.shape[1], dtype=np.float32)
        capi.check(self._lib.thx_reduce_histograms(self._ctx, h.ctypes.data_as(C.POINTER(C.c_float)), h.shape[0], h.shape[1], float(norm),
                                                   out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def tracer_read_hist(self, tracer: int, cfg: int, n_bins: int) -> np.ndarray:
        out = np.empty(int(n_bins), dtype=np.float32)
        capi.check(self._lib.thx_tracer_read_hist(C.c_void_p(tracer), int(cfg), out.ctypes.data_as(C.POINTER(C.c_float)), int(n_bins)))
        return out

    def tracer_set_concurrency(self, tracer: int, lanes: int, grid_divisor: int | None = None) -> None:
        """`lanes` batches in flight, each on a stream of its own and at most 1 / grid_divisor of the chip's workgroups
        (thx_tracer_set_concurrency; grid_divisor defaults to lanes)"""
        capi.check(self._lib.thx_tracer_set_concurrency(C.c_void_p(tracer), int(lanes), int(grid_divisor or lanes)))

    def tracer_fetch_hist(self, tracer: int, slot: int, stream: int = 0) -> None:
        """enqueues the readback of slot `slot`'s curve behind its reduction (asynchronous)"""
        capi.check(self._lib.thx_tracer_fetch_hist(C.c_void_p(tracer), int(slot), C.c_void_p(stream) if stream else None))

    def tracer_collect_hist(self, tracer: int, slot: int, out: np.ndarray) -> None:
        """waits for the readback enqueued by tracer_fetch_hist and copies the curve into `out` (float32, n_bins)"""
        capi.check(self._lib.thx_tracer_collect_hist(C.c_void_p(tracer), int(slot), out.ctypes.data_as(C.POINTER(C.c_float)), out.size))

    def tracer_hist_ptr(self, tracer: int, cfg: int) -> int:
        out = C.c_uint64()
        capi.check(self._lib.thx_tracer_hist_device_ptr(C.c_void_p(tracer), int(cfg), C.byref(out)))
        return int(out.value)

    def tracer_read_stats(self, tracer: int) -> np.ndarray:
        out = np.zeros(capi.N_STATS, dtype=np.uint64)
        capi.check(self._lib.thx_tracer_read_stats(C.c_void_p(tracer), out.ctypes.data_as(C.POINTER(C.c_uint64)), capi.N_STATS))
        return out

    def tracer_reset_stats(self, tracer: int) -> None:
        capi.check(self._lib.thx_tracer_reset_stats(C.c_void_p(tracer)))

    def tracer_read_track(self, tracer: int, batch: int, events: int) -> np.ndarray:
        out = np.empty((int(batch), int(events), capi.TRACK_WORDS), dtype=np.float32)
        capi.check(self._lib.thx_tracer_read_track(C.c_void_p(tracer), out.ctypes.data_as(C.POINTER(C.c_float)), out.size))
        return out

    def tracer_read_hits(self, tracer: int, out: np.ndarray, clear: bool = True, cfg: int | None = None) -> None:
        """copies the HitRecorder queue into `out` (uint8) and resets its counter; `cfg`: the queue of that pipeline config, waiting
        for its last launch only (thx_tracer_read_hits_cfg) -- else the queue of the most recent launch, after everything"""
        if cfg is None:
            capi.check(self._lib.thx_tracer_read_hits(C.c_void_p(tracer), out.ctypes.data_as(C.c_void_p), out.nbytes, int(clear)))
        else:
            capi.check(self._lib.thx_tracer_read_hits_cfg(C.c_void_p(tracer), int(cfg), out.ctypes.data_as(C.c_void_p), out.nbytes, int(clear)))

    def rng_fill_sobol(self, seed: int, base_stream: int, n_streams: int, first_dim: int, n_samples: int, sample_dim: int = 1) -> np.ndarray:
        """RNGBufferSink over a SobolQRNG: float32 [streams][samples] (or [...][2] for 2-D samples)"""
        shape = (int(n_streams), int(n_samples)) if sample_dim == 1 else (int(n_streams), int(n_samples), 2)
        out = np.zeros(shape, dtype=np.float32)
        capi.check(self._lib.thx_rng_fill_sobol(self._ctx, int(seed) & 0xFFFFFFFF, int(base_stream), int(n_streams), int(first_dim),
                                                int(n_samples), int(sample_dim), out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def probe_target(self, philox: capi.PhiloxParams, flags: int, target: capi.TargetParams, box_min, box_max, n: int) -> np.ndarray:
        """theia.testing.TargetSampler: [n][25] words (include/theia_hip.h)"""
        out = np.zeros((int(n), 25), dtype=np.float32)
        lo, hi = (C.c_float * 3)(*box_min), (C.c_float * 3)(*box_max)
        capi.check(self._lib.thx_probe_target(self._ctx, C.byref(philox), int(flags), C.byref(target), lo, hi, int(n), out.ctypes.data_as(C.c_void_p)))
        return out

    def probe_guide(self, philox: capi.PhiloxParams, flags: int, guide: capi.GuideParams, box_min, box_max, n: int) -> np.ndarray:
        """theia.testing.TargetGuideSampler: [n][13] floats"""
        out = np.zeros((int(n), 13), dtype=np.float32)
        lo, hi = (C.c_float * 3)(*box_min), (C.c_float * 3)(*box_max)
        capi.check(self._lib.thx_probe_guide(self._ctx, C.byref(philox), int(flags), C.byref(guide), lo, hi, int(n), out.ctypes.data_as(C.c_void_p)))
        return out

    def tracer_write_light(self, tracer: int, cfg: int, data: np.ndarray) -> None:
        """uploads a HostLightSource queue (uint8, SoA columns without counter) for pipeline config `cfg`"""
        data = np.ascontiguousarray(data)
        capi.check(self._lib.thx_tracer_write_light(C.c_void_p(tracer), int(cfg), data.ctypes.data_as(C.c_void_p), data.nbytes))

    def tracer_write_wavelengths(self, tracer: int, cfg: int, data: np.ndarray) -> None:
        data = np.ascontiguousarray(data)
        capi.check(self._lib.thx_tracer_write_wavelengths(C.c_void_p(tracer), int(cfg), data.ctypes.data_as(C.c_void_p), data.nbytes))

    def tracer_read_samples(self, tracer: int, light_out: np.ndarray, wavelength_out: np.ndarray) -> None:
        """copies the two output queues of a light sampler (uint8: count + SoA columns)"""
        capi.check(self._lib.thx_tracer_read_samples(C.c_void_p(tracer), light_out.ctypes.data_as(C.c_void_p), light_out.nbytes,
                                                     wavelength_out.ctypes.data_as(C.c_void_p), wavelength_out.nbytes))

    def tracer_write_hits(self, tracer: int, data: np.ndarray) -> None:
        """uploads the input queue (uint8, count + SoA columns) of a hit replay"""
        data = np.ascontiguousarray(data)
        capi.check(self._lib.thx_tracer_write_hits(C.c_void_p(tracer), data.ctypes.data_as(C.c_void_p), data.nbytes))

    def tracer_launch_info(self, tracer: int, batch: int) -> dict:
        g, b, l = C.c_uint32(), C.c_uint32(), C.c_uint32()
        capi.check(self._lib.thx_tracer_launch_info(C.c_void_p(tracer), int(batch), C.byref(g), C.byref(b), C.byref(l)))
        return {"grid": g.value, "block": b.value, "lds_bytes": l.value}

    def tracer_last_kernel_ms(self, tracer: int, cfg: int) -> float:
        ms = C.c_float()
        capi.check(self._lib.thx_tracer_last_kernel_ms(C.c_void_p(tracer), int(cfg), C.byref(ms)))
        return float(ms.value)

    def tracer_kernel_times(self, tracer: int, n: int = 64) -> np.ndarray:
        """durations (ms) of the last min(n, 64) tracing-kernel launches, oldest first"""
        out = np.zeros(int(n), dtype=np.float32)
        got = C.c_uint32()
        capi.check(self._lib.thx_tracer_kernel_times(C.c_void_p(tracer), out.ctypes.data_as(C.POINTER(C.c_float)), int(n), C.byref(got)))
        return out[: got.value]

    # -- rng probe ---------------------------------------------------------
    def rng_fill(self, key: int, offset: int, base_stream: int, n_streams: int, first_dim: int, n_samples: int):
        p = capi.PhiloxParams()
        p.key[0], p.key[1] = key & 0xFFFFFFFF, (key >> 32) & 0xFFFFFFFF
        for i in range(4):
            p.offset[i] = (offset >> (32 * i)) & 0xFFFFFFFF
        u = np.empty((n_streams, n_samples), dtype=np.uint32)
        f = np.empty((n_streams, n_samples), dtype=np.float32)
        capi.check(self._lib.thx_rng_fill(self._ctx, C.byref(p), base_stream, n_streams, first_dim, n_samples,
                                          u.ctypes.data_as(C.POINTER(C.c_uint32)), f.ctypes.data_as(C.POINTER(C.c_float))))
        return u, f

    def close(self) -> None:
        if self._ctx:
            self._lib.thx_ctx_destroy(self._ctx)
            self._ctx = None


_backend = None


def get():
    """The process-wide backend; created on first use (HIP only)."""
    global _backend
    if _backend is None:
        _backend = HipBackend()
    return _backend


def install(backend) -> None:
    """Installs a backend object. Used by tests/ to drive host logic without a GPU."""
    global _backend
    _backend = backend


def reset() -> None:
    global _backend
    if _backend is not None and hasattr(_backend, "close"):
        try:
            _backend.close()
        except Exception:
            pass
    _backend = None
