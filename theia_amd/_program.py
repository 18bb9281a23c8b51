"""The handle a tracer / replay / sampler stage owns on the backend (stands where `hp.Program` stood)."""
from __future__ import annotations

import numpy as np

from . import _backend
from . import _capi as capi

__all__ = ["_Program"]


class _Program:
    """The compiled-variant handle a tracer owns (stands where hp.Program stood).
    Created lazily so that building stages does not need a device."""

    def __init__(self, spec: _backend.TracerSpec, scene: Scene | None) -> None:
        self._spec = spec
        self._scene = scene
        self._handle = None
        self._be = None
        self.last_batch = 0
        self.lanes, self.grid_divisor = 1, 1  # what set_concurrency last asked the library for

    @property
    def created(self) -> bool:
        return self._handle is not None

    @property
    def handle(self) -> int:
        if self._handle is None:
            self._be = _backend.get()
            self._handle = self._be.tracer_create(self._spec, self._scene.handle if self._scene is not None else None)
        return self._handle

    @property
    def concurrent(self) -> bool:
        """batches of this program may be in flight side by side: everything a batch writes lives in a slot of its own
        (not so with a hit queue or a track buffer, of which the tracer has one)"""
        spec = self._spec
        return (spec.kind in (capi.TRACER_VOLUME_FORWARD, capi.TRACER_SCENE_FORWARD) and spec.response_kind == capi.RESPONSE_HISTOGRAM
                and not spec.flags & capi.FLAG_TRACK)

    @property
    def queue_per_config(self) -> bool:
        """a tracing program with a record / store-time response has one hit queue per pipeline config: the batch of one config runs
        while the other config's queue is read"""
        spec = self._spec
        return (spec.kind in (capi.TRACER_VOLUME_FORWARD, capi.TRACER_SCENE_FORWARD)
                and spec.response_kind in (capi.RESPONSE_RECORD, capi.RESPONSE_STORE_TIME))

    @property
    def max_in_flight(self) -> int:
        """batches of this program that can be on the device at once: one result slot each (histogram), one queue per config (hit
        queues), one (a track buffer, the samplers)"""
        if self.concurrent:
            return capi.MAX_SLOTS
        return 2 if self.queue_per_config and not self._spec.flags & capi.FLAG_TRACK else 1

    @property
    def backend(self):
        self.handle
        return self._be

    def run(self, cfg: int, params: capi.RunParams) -> None:
        self.last_batch = int(params.dispatch.batchSize)
        self.backend.tracer_run(self.handle, cfg, params)

    def reduce(self, cfg: int, norm: float, d_out: int = 0) -> None:
        self.backend.tracer_reduce(self.handle, cfg, norm, d_out)

    def set_concurrency(self, lanes: int, grid_divisor: int | None = None) -> None:
        """`lanes` batches in flight, each launch on 1 / grid_divisor of the chip. A tracer prepared for very large batches
        cannot cut its launches that small (the library bounds the ring positions of a workgroup): the divisor is halved
        until it fits -- such batches hide the tail of a launch by their size anyway."""
        div = int(grid_divisor or lanes)
        # no more lanes than the runtime runs kernels side by side (half of GPU_MAX_HW_QUEUES: 8 when this package could set the
        # variable before HIP initialised, 2 with the runtime's default) -- more would queue batches up behind each other on a
        # fraction of the chip, slower than whole-chip launches
        probe = getattr(self.backend, "concurrent_kernels", None)
        if lanes > 1 and probe is not None:
            lanes = max(1, min(int(lanes), int(probe())))
            div = max(1, min(div, lanes))
        self.lanes, self.grid_divisor = lanes, div
        while True:
            try:
                return self.backend.tracer_set_concurrency(self.handle, lanes, div)
            except capi.ThxError as err:
                if "grid_divisor too large" not in str(err) or div == 1:
                    raise
                div //= 2

    def fetch_hist(self, slot: int) -> None:
        self.backend.tracer_fetch_hist(self.handle, slot)

    def collect_hist(self, slot: int, out: np.ndarray) -> None:
        self.backend.tracer_collect_hist(self.handle, slot, out)

    def read_hist(self, cfg: int) -> np.ndarray:
        return self.backend.tracer_read_hist(self.handle, cfg, self._spec.n_bins)

    def hist_ptr(self, cfg: int) -> int:
        return self.backend.tracer_hist_ptr(self.handle, cfg)

    def read_hits(self, out: np.ndarray, clear: bool = True, cfg: int | None = None) -> None:
        self.backend.tracer_read_hits(self.handle, out, clear, cfg if self.queue_per_config else None)

    def write_hits(self, data: np.ndarray) -> None:
        self.backend.tracer_write_hits(self.handle, data)

    def write_light(self, cfg: int, data: np.ndarray) -> None:
        self.backend.tracer_write_light(self.handle, cfg, data)

    def write_wavelengths(self, cfg: int, data: np.ndarray) -> None:
        self.backend.tracer_write_wavelengths(self.handle, cfg, data)

    def read_samples(self, light_out: np.ndarray, wavelength_out: np.ndarray) -> None:
        self.backend.tracer_read_samples(self.handle, light_out, wavelength_out)

    def read_stats(self) -> np.ndarray:
        return self.backend.tracer_read_stats(self.handle)

    def reset_stats(self) -> None:
        self.backend.tracer_reset_stats(self.handle)

    def read_track(self, batch: int) -> np.ndarray:
        return self.backend.tracer_read_track(self.handle, batch, self._spec.path_length + 3)

    def launch_info(self, batch: int) -> dict:
        return self.backend.tracer_launch_info(self.handle, batch)

    def last_kernel_ms(self, cfg: int) -> float:
        return self.backend.tracer_last_kernel_ms(self.handle, cfg)

    def kernel_times(self, n: int = 64) -> np.ndarray:
        return self.backend.tracer_kernel_times(self.handle, n)

    def __del__(self) -> None:
        try:
            if self._handle is not None:
                self._be.tracer_destroy(self._handle)
                self._handle = None
        except Exception:
            pass
