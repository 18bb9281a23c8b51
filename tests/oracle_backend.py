"""
TEST INFRASTRUCTURE: a stand-in for theia_amd's device backend that executes
on the CPU oracle (oracle/). It lets the `-m "not gpu"` suite drive the *host*
logic of theia_amd (stages, pipelines, parameter translation, sharding) on a
machine without a GPU, and gives the GPU parity tests the oracle's answer for
the same scenario description. It is never importable from the product.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_ROOT, "oracle"))
import oracle as orc  # noqa: E402

from theia_amd import _capi as capi  # noqa: E402


def _copy(dst, src) -> None:
    assert C.sizeof(dst) == C.sizeof(src), (type(dst), type(src))
    C.memmove(C.addressof(dst), C.addressof(src), C.sizeof(dst))


class _Tracer:
    def __init__(self, spec, scene) -> None:
        self.spec = spec
        self.scene = scene
        self.raw = [np.zeros(spec.n_bins, np.float32) for _ in range(capi.MAX_SLOTS)]  # result slots (0 / 1 = the pipeline configs)
        self.out = [np.zeros(spec.n_bins, np.float32) for _ in range(capi.MAX_SLOTS)]
        self.fetched = {}
        self.stats = np.zeros(orc.N_STATS, np.uint64)
        self.track = None
        self.hits = None  # HitRecorder queue of the batch being traced: uint32 count + columns
        self.hit_queues = {}  # ... by pipeline config (the tracing kinds keep one per config)
        self.last_cfg = 0


class OracleBackend:
    name = "oracle"

    def __init__(self, threads: int = 0, brute_force: bool = False) -> None:
        self.threads = threads
        self.brute_force = brute_force
        self._blobs: dict[int, orc.HostBlob] = {}
        self._scenes: dict[int, orc.Scene] = {}
        self._tracers: dict[int, _Tracer] = {}
        self._next = 1

    # device
    def device_name(self) -> str:
        return "CPU oracle (test stand-in)"

    def device_info(self) -> dict:
        return {"compute_units": os.cpu_count(), "lds_per_cu": 0, "hbm_bytes": 0}

    def sync(self) -> None:
        pass

    # memory
    def alloc(self, nbytes: int) -> int:
        blob = orc.HostBlob(nbytes)
        self._blobs[blob.address] = blob
        return blob.address

    def free(self, addr: int) -> None:
        self._blobs.pop(addr, None)

    def _inside(self, addr: int, nbytes: int):
        """(blob, offset) of an address range inside one of the allocations -- the stand-in for device pointer arithmetic"""
        for base, blob in self._blobs.items():
            if base <= addr and addr + nbytes <= base + blob.buf.nbytes:
                return blob, addr - base
        raise KeyError(f"address range {addr:#x} + {nbytes} lies in no allocation")

    def upload(self, addr: int, data) -> None:
        raw = np.frombuffer(bytes(data), dtype=np.uint8)
        blob, off = self._inside(addr, raw.nbytes)
        blob.buf[off:off + raw.nbytes] = raw

    def download(self, addr: int, nbytes: int) -> bytes:
        blob, off = self._inside(addr, nbytes)
        return blob.buf[off:off + nbytes].tobytes()

    # scene
    def scene_create(self, meshes, instances) -> int:
        h = self._next
        self._next += 1
        self._scenes[h] = orc.Scene(meshes, instances)
        return h

    def scene_destroy(self, scene: int) -> None:
        self._scenes.pop(scene, None)

    def scene_info(self, scene: int) -> dict:
        return {}

    def scene_query(self, scene: int, origins, dirs, tmax) -> np.ndarray:
        o = np.asarray(origins, np.float32).reshape(-1, 3)
        d = np.asarray(dirs, np.float32).reshape(-1, 3)
        t = np.broadcast_to(np.asarray(tmax, np.float32), (o.shape[0],))
        out = np.zeros((o.shape[0], 6), np.float32)
        for i in range(o.shape[0]):
            hit, r = self._scenes[scene].query(o[i], d[i], float(t[i]), self.brute_force)
            out[i, 0] = 1.0 if hit else 0.0
            out[i, 1:] = r
        return out

    # tracer
    def tracer_create(self, spec, scene) -> int:
        h = self._next
        self._next += 1
        self._tracers[h] = _Tracer(spec, self._scenes[scene] if scene else None)
        return h

    def tracer_destroy(self, tracer: int) -> None:
        self._tracers.pop(tracer, None)

    def tracer_run(self, tracer: int, cfg: int, params: capi.RunParams, stream: int = 0) -> None:
        t = self._tracers[tracer]
        spec = t.spec
        c = orc.Common()
        flags = (spec.flags & 0x7F) | (spec.flags & (capi.FLAG_LIGHT_PENCIL | capi.FLAG_TARGET_INNER | capi.FLAG_KERNEL_HISTOGRAM |
                                                     capi.FLAG_LIGHT_HOST | capi.FLAG_LIGHT_QUEUE_POLARIZED | capi.FLAG_WAVELENGTH_HOST |
                                                     capi.FLAG_LIGHT_CONE | capi.FLAG_LIGHT_POLARIZED | capi.FLAG_WAVELENGTH_FUNCTION |
                                                     capi.FLAG_TARGET_DISK | capi.FLAG_TARGET_FLAT | capi.FLAG_GUIDE_DISK | capi.FLAG_GUIDE_FLAT |
                                                     capi.FLAG_RNG_SOBOL | capi.FLAG_TARGET_MISS_INFINITE |
                                                     capi.FLAG_LIGHT_KIND_MASK | capi.FLAG_FRANK_TAMM_PHOTONS | capi.FLAG_FRANK_TAMM_IS))
        photon = bool(spec.flags & capi.FLAG_PHOTON)
        if self.brute_force:
            flags |= orc.FLAG_BRUTE_FORCE
        c.flags = flags
        c.path_length = spec.path_length
        c.block_size = spec.block_size
        c.n_bins = spec.n_bins
        c.batch_size = params.dispatch.batchSize
        c.photon_offset = params.photon_offset
        c.normalization = 1.0
        _copy(c.philox, params.philox)
        _copy(c.light, params.light)
        _copy(c.wavelength, params.wavelength)
        _copy(c.response, params.response)
        want_track = bool(spec.flags & capi.FLAG_TRACK)
        kind = getattr(spec, "response_kind", 0)
        record = kind in (capi.RESPONSE_RECORD, capi.RESPONSE_STORE_TIME)
        if record:
            key = cfg & 1 if spec.kind in (capi.TRACER_VOLUME_FORWARD, capi.TRACER_SCENE_FORWARD) else 0
            if key not in t.hit_queues:
                ncol = (2 if spec.hit_polarized else 1) if kind == capi.RESPONSE_STORE_TIME else (20 if spec.hit_polarized else 13)
                t.hit_queues[key] = np.zeros(1 + ncol * spec.hit_capacity, dtype=np.uint32)
            t.hits = t.hit_queues[key]
            t.last_cfg = key
            if kind == capi.RESPONSE_STORE_TIME:
                orc.set_time_queue(t.hits, spec.hit_capacity, spec.hit_polarized)
            else:
                orc.set_hit_queue(t.hits, spec.hit_capacity, spec.hit_polarized)
        if photon:
            orc.set_photon_mode(params.photon_steps_per_run, params.photon_rng_per_run)
        if spec.flags & capi.FLAG_LIGHT_HOST:
            orc.set_light_queue(t.light_in[cfg], spec.light_queue_capacity)
        if spec.flags & capi.FLAG_WAVELENGTH_HOST:
            orc.set_wavelength_queue(t.lam_in[cfg], spec.wavelength_queue_capacity)
        try:
            hist, stats, track = self._trace(spec, t, c, params, want_track)
        finally:
            if record:
                orc.set_hit_queue(None, 0, False)
                orc.set_time_queue(None, 0, False)
            if photon:
                orc.set_photon_mode(0, 0)
            orc.set_light_queue(None)
            orc.set_wavelength_queue(None)
        t.raw[cfg] = hist
        t.stats += stats
        t.track = track

    def rng_fill_sobol(self, seed, base_stream, n_streams, first_dim, n_samples, sample_dim=1):
        return orc.sobol_fill(seed, base_stream, n_streams, first_dim, n_samples, sample_dim)

    def probe_unit(self, name: str, inputs, addr: int = 0) -> np.ndarray:
        """the oracle's restatement of one shader function applied to every row (counterpart of thx_probe_unit)"""
        import ctypes as C

        L = orc.lib()
        n_in, n_out = {"lookup": (2, 1), "lookup_medium": (1, 4), "scatter_dir": (5, 3), "fresnel": (3, 2), "scatter_prob": (6, 1),
                       "scatter_sample": (5, 4), "rotate_polref": (9, 5), "align_polref": (9, 2), "phase_matrix": (7, 4), "polarizer": (6, 4),
                       "sign_bit": (1, 1), "emission_angle": (4, 1), "emission_angle_range": (6, 2), "emission_pdf": (4, 1),
                       "frank_tamm": (2, 3)}[name]
        x = np.ascontiguousarray(np.asarray(inputs, dtype=np.float32).reshape(-1, n_in))
        out = np.zeros((x.shape[0], n_out), dtype=np.float32)
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        family = {"emission_angle": 0, "emission_angle_range": 1, "emission_pdf": 2, "frank_tamm": 3}
        if name in family:  # (all rows in one call)
            L.orc_cherenkov_units(family[name], fp(x), x.shape[0], fp(out))
            return out
        for r, o in zip(x, out):
            if name == "lookup":
                o[0] = L.orc_lookup(addr, float(r[0]), float(r[1]))
            elif name == "lookup_medium":
                L.orc_lookup_medium(addr, float(r[0]), fp(o))
            elif name == "scatter_dir":
                L.orc_scatter_dir(fp(r[0:3]), float(r[3]), float(r[4]), fp(o))
            elif name == "fresnel":
                L.orc_fresnel(float(r[0]), float(r[1]), float(r[2]), fp(o))
            elif name == "scatter_prob":
                o[0] = L.orc_scatter_prob(addr, fp(r[0:3]), fp(r[3:6]))
            elif name == "scatter_sample":
                L.orc_scatter_sample(addr, fp(r[0:3]), float(r[3]), float(r[4]), fp(o[0:3]), fp(o[3:4]))
            elif name == "rotate_polref":
                L.orc_rotate_polref(fp(r[0:3]), fp(r[3:6]), fp(r[6:9]), fp(o[0:3]), fp(o[3:5]))
            elif name == "align_polref":
                L.orc_align_polref(fp(r[0:3]), fp(r[3:6]), fp(r[6:9]), fp(o))
            elif name == "phase_matrix":
                L.orc_phase_matrix_apply(addr, float(r[0]), float(r[1]), float(r[2]), fp(r[3:7]), fp(o))
            elif name == "polarizer":
                L.orc_polarizer(float(r[0]), float(r[1]), fp(r[2:6]), fp(o))
            else:
                o[0] = L.orc_sign_bit(float(r[0]))
        return out

    def probe_target(self, philox, flags, target, box_min, box_max, n):
        ph, tg = orc.PhiloxParams(), orc.TargetParams()
        _copy(ph, philox)
        _copy(tg, target)
        return orc.probe_target(ph, flags, tg, box_min, box_max, n)

    def probe_guide(self, philox, flags, guide, box_min, box_max, n):
        ph, g = orc.PhiloxParams(), orc.GuideParams()
        _copy(ph, philox)
        _copy(g, guide)
        return orc.probe_guide(ph, flags, g, box_min, box_max, n)

    def tracer_write_light(self, tracer: int, cfg: int, data: np.ndarray) -> None:
        t = self._tracers[tracer]
        t.light_in = getattr(t, "light_in", [None, None])
        t.light_in[cfg] = np.array(data, copy=True).view(np.uint8).view(np.float32)

    def tracer_write_wavelengths(self, tracer: int, cfg: int, data: np.ndarray) -> None:
        t = self._tracers[tracer]
        t.lam_in = getattr(t, "lam_in", [None, None])
        t.lam_in[cfg] = np.array(data, copy=True).view(np.uint8).view(np.float32)

    def tracer_read_samples(self, tracer: int, light_out: np.ndarray, wavelength_out: np.ndarray) -> None:
        t = self._tracers[tracer]
        light_out.reshape(-1).view(np.uint8)[:] = t.samples[0].view(np.uint8)
        wavelength_out.reshape(-1).view(np.uint8)[:] = t.samples[1].view(np.uint8)

    def tracer_write_hits(self, tracer: int, data: np.ndarray) -> None:
        self._tracers[tracer].replay_in = np.array(data, dtype=np.uint8, copy=True)

    def _trace(self, spec, t, c, params, want_track):
        if spec.kind == capi.TRACER_LIGHT_SAMPLER:
            medium = int(params.volume.medium[0]) | (int(params.volume.medium[1]) << 32)
            t.samples = orc.sample_light(c, medium, params.dispatch.batchSize, params.photon_offset,
                                         bool(spec.flags & capi.FLAG_POLARIZED), spec.capacity)
            return np.zeros(1, np.float32), np.zeros(orc.N_STATS, np.uint64), None
        if spec.kind == capi.TRACER_HIT_REPLAY:
            c.batch_size = spec.capacity
            hist, stats = orc.replay(c, t.replay_in, spec.capacity, bool(spec.flags & capi.FLAG_POLARIZED), threads=self.threads)
            return hist, stats, None
        if spec.kind == capi.TRACER_VOLUME_FORWARD:
            tp, tg = orc.VolumeTraceParams(), orc.TargetParams()
            _copy(tp, params.volume)
            _copy(tg, params.target)
            hist, stats, track = orc.trace_volume(c, tp, tg, threads=self.threads, track=want_track)
        else:
            tp, g = orc.SceneTraceParams(), orc.GuideParams()
            _copy(tp, params.scene)
            _copy(g, params.guide)
            guide = None if (spec.flags & capi.FLAG_DISABLE_MIS) else g
            hist, stats, track = orc.trace_scene(c, tp, guide, t.scene, threads=self.threads, track=want_track)
        return hist, stats, track

    def tracer_read_hits(self, tracer: int, out: np.ndarray, clear: bool = True, cfg=None) -> None:
        t = self._tracers[tracer]
        queue = t.hit_queues.get(t.last_cfg if cfg is None else cfg & 1, t.hits)
        raw = queue.view(np.uint8)
        n = min(out.nbytes, raw.nbytes)
        out.reshape(-1).view(np.uint8)[:n] = raw[:n]
        if clear:
            queue[0] = 0

    def tracer_reduce(self, tracer: int, cfg: int, norm: float, d_out: int = 0, stream: int = 0) -> None:
        t = self._tracers[tracer]
        t.out[cfg] = (t.raw[cfg] * np.float32(norm)).astype(np.float32)

    # the asynchronous readback of the HIP backend, synchronous here
    def tracer_set_concurrency(self, tracer: int, lanes: int, grid_divisor=None) -> None:
        if not 1 <= int(lanes) <= capi.MAX_SLOTS:
            raise ValueError("lanes must be in [1, MAX_SLOTS]")

    def tracer_fetch_hist(self, tracer: int, slot: int, stream: int = 0) -> None:
        t = self._tracers[tracer]
        t.fetched[slot] = t.out[slot].copy()

    def tracer_collect_hist(self, tracer: int, slot: int, out: np.ndarray) -> None:
        out[...] = self._tracers[tracer].fetched.pop(slot)

    def reduce_histograms(self, hists, norm: float = 1.0) -> np.ndarray:
        return orc.reduce(hists, norm)

    def tracer_read_hist(self, tracer: int, cfg: int, n_bins: int) -> np.ndarray:
        return self._tracers[tracer].out[cfg].copy()

    def tracer_hist_ptr(self, tracer: int, cfg: int) -> int:
        return int(self._tracers[tracer].out[cfg].ctypes.data)

    def tracer_read_stats(self, tracer: int) -> np.ndarray:
        return self._tracers[tracer].stats.copy()

    def tracer_reset_stats(self, tracer: int) -> None:
        self._tracers[tracer].stats[:] = 0

    def tracer_read_track(self, tracer: int, batch: int, events: int) -> np.ndarray:
        return self._tracers[tracer].track[:batch]

    def tracer_launch_info(self, tracer: int, batch: int) -> dict:
        return {"grid": 0, "block": 0, "lds_bytes": 0}

    def tracer_last_kernel_ms(self, tracer: int, cfg: int) -> float:
        return float("nan")

    def rng_fill(self, key, offset, base_stream, n_streams, first_dim, n_samples):
        return orc.rng_fill(key, offset, base_stream, n_streams, first_dim, n_samples)

    def close(self) -> None:
        self._tracers.clear()
        self._scenes.clear()
        self._blobs.clear()
