"""
`hephaistos.queue` as far as theia's HitRecorder needs it: a structure-of-arrays
queue with a leading item counter, described by a ctypes `Structure` (reference
use: src/theia/response.py:219, 268-274; device layout shader/response.queue.glsl:6-33
-- `uint count`, then one column of `capacity` entries per scalar of the item, in
field order; a `c_float * 3` field is three columns).

Only the host side lives here: `QueueView` gives numpy views onto a queue in host memory,
`IOQueue` owns the two host buffers of the double-buffered pipeline: `mode="retrieve"` is
filled from the device after every batch (HitRecorder), `mode="update"` is written by the
user and uploaded before it (HitReplay, response.py:357-359). The device queue belongs to
the tracer / replay the stage is bound to.
"""
from __future__ import annotations

import ctypes
import mmap
import sys
from ctypes import Structure

import numpy as np

__all__ = ["QueueView", "IOQueue", "queueSize", "as_queue", "dumpQueue"]


def _columns(item: type[Structure]) -> list[tuple[str, np.dtype, int, int]]:
    """(field name, numpy dtype, number of components, index of the first column)"""
    out, col = [], 0
    for name, ctype in item._fields_:
        n = 1
        base = ctype
        if hasattr(ctype, "_length_"):
            n, base = ctype._length_, ctype._type_
        if ctypes.sizeof(base) != 4:
            raise TypeError(f"queue item field {name!r}: only 4-byte scalars are supported")
        out.append((name, np.dtype(base), n, col))
        col += n
    return out


def queueSize(item: type[Structure], capacity: int, *, skipCounter: bool = False) -> int:
    """bytes of a queue of `capacity` items"""
    ncol = sum(n for _, _, n, _ in _columns(item))
    return (0 if skipCounter else 4) + 4 * ncol * capacity


class QueueView:
    """
    numpy views onto a queue in memory. `view["position"]` is an `(capacity, 3)` array,
    `view["time"]` a `(capacity,)` array, `view.count` the item counter, `view[:n]` a view of
    the first `n` items (what the reference's tests do with `hits[: hits.count]`).
    """

    def __init__(self, data: np.ndarray, item: type[Structure], capacity: int, *, skipCounter: bool = False,
                 _slice: slice | None = None) -> None:
        self._data = data  # uint8, whole queue
        self._item = item
        self._capacity = capacity
        self._skipCounter = skipCounter
        self._cols = _columns(item)
        self._offset = 0 if skipCounter else 4
        self._slice = _slice if _slice is not None else slice(0, capacity)

    @property
    def capacity(self) -> int:
        return self._capacity

    @property
    def hasCounter(self) -> bool:
        return not self._skipCounter

    @property
    def count(self) -> int:
        """number of items the producer appended (clamped to the capacity)"""
        if self._skipCounter:
            return self._capacity
        return int(min(self._data[:4].view(np.uint32)[0], self._capacity))

    @count.setter
    def count(self, value: int) -> None:
        self._data[:4].view(np.uint32)[0] = value

    @property
    def fields(self) -> list[str]:
        return [name for name, *_ in self._cols]

    def __len__(self) -> int:
        start, stop, step = self._slice.indices(self._capacity)
        return len(range(start, stop, step))

    def __getitem__(self, key):
        if isinstance(key, str):
            for name, dtype, n, col in self._cols:
                if name == key:
                    base = self._data[self._offset + 4 * col * self._capacity:
                                      self._offset + 4 * (col + n) * self._capacity].view(dtype)
                    arr = base if n == 1 else np.lib.stride_tricks.as_strided(
                        base, shape=(self._capacity, n), strides=(4, 4 * self._capacity))
                    return arr[self._slice]
            raise KeyError(key)
        if isinstance(key, slice):
            start, stop, step = self._slice.indices(self._capacity)
            idx = range(start, stop, step)[key]
            return QueueView(self._data, self._item, self._capacity, skipCounter=self._skipCounter,
                             _slice=slice(idx.start, idx.stop, idx.step))
        if isinstance(key, (np.ndarray, list)):
            # items picked / reordered by index (`results[np.argsort(results["contrib"])]`): a copy
            sel = np.asarray(key)
            n = len(sel) if sel.dtype != np.bool_ else int(sel.sum())
            out = QueueView(np.zeros(queueSize(self._item, max(n, 1), skipCounter=self._skipCounter), dtype=np.uint8),
                            self._item, max(n, 1), skipCounter=self._skipCounter, _slice=slice(0, n))
            for name in self.fields:
                out[name] = self[name][sel]
            if not self._skipCounter:
                out.count = n
            return out
        raise TypeError("index a QueueView with a field name, a slice or an index array")

    def __setitem__(self, key: str, value) -> None:
        """`view["time"] = array`: writes a column (block) of the queue, converting to its type"""
        if not isinstance(key, str):
            raise TypeError("only whole fields of a QueueView can be assigned")
        for name, dtype, n, col in self._cols:
            if name == key:
                base = self._data[self._offset + 4 * col * self._capacity:
                                  self._offset + 4 * (col + n) * self._capacity].view(dtype)
                target = base.reshape(n, self._capacity)[:, self._slice]  # column major: one row per scalar
                value = np.asarray(value)
                if n == 1:
                    target[0, :] = value
                else:
                    target[:, :] = np.broadcast_to(value, (target.shape[1], n)).T
                return
        raise KeyError(key)

    def __iter__(self):
        return iter(self.fields)


def as_queue(data: np.ndarray, item: type[Structure], *, capacity: int, skipCounter: bool = False) -> QueueView:
    return QueueView(np.asarray(data).view(np.uint8).reshape(-1), item, capacity, skipCounter=skipCounter)


def dumpQueue(view: QueueView) -> dict[str, np.ndarray]:
    """copies of the valid items of a queue, one array per field (hephaistos.queue.dumpQueue)"""
    n = view.count if view.hasCounter else len(view)
    return {name: np.array(view[name][:n]) for name in view.fields}


def _hostBuffer(nbytes: int) -> np.ndarray:
    """zeroed host bytes in an anonymous mapping of their own (whole pages, page aligned): what `IOQueue.pin` page-locks is then
    exactly this memory -- page-locking marks pages as not inherited by a forked child, which must not happen to pages of the
    allocator's heap that a small numpy array shares with everything else. Unmapped when the last view onto it goes away."""
    return np.frombuffer(mmap.mmap(-1, max(int(nbytes), 1)), dtype=np.uint8, count=int(nbytes))


class IOQueue:
    """Host side of a queue that is filled on the device and retrieved after each batch."""

    def __init__(self, item: type[Structure], capacity: int | None = None, *, mode: str = "retrieve",
                 skipCounter: bool = False) -> None:
        if mode not in ("retrieve", "update"):
            raise NotImplementedError("IOQueue modes on the HIP backend: 'retrieve', 'update'")
        self._item = item
        self._mode = mode
        self._skipCounter = skipCounter
        self._capacity = 0
        self._buffers: list[np.ndarray] | None = None
        if capacity is not None:
            self.initialize(capacity)

    @property
    def item(self) -> type[Structure]:
        return self._item

    @property
    def mode(self) -> str:
        return self._mode

    @property
    def initialized(self) -> bool:
        return self._buffers is not None

    @property
    def capacity(self) -> int:
        return self._capacity

    @property
    def nbytes(self) -> int:
        return queueSize(self._item, self._capacity, skipCounter=self._skipCounter)

    def initialize(self, capacity: int) -> None:
        if self.initialized:
            raise RuntimeError("queue is already initialized")
        if capacity <= 0:
            raise ValueError("capacity must be positive")
        self._capacity = int(capacity)
        self._buffers = [_hostBuffer(self.nbytes) for _ in range(2)]
        if self._mode == "update" and not self._skipCounter:
            # a queue the user fills starts out full: the reference's tests write all `capacity`
            # items and never touch the counter (tests/test_response.py:24-37)
            for b in self._buffers:
                b[:4].view(np.uint32)[0] = self._capacity

    def pin(self, backend) -> None:
        """page-locks the two host buffers once (`thx_host_register`): the copies of a retrieve / update then run on the DMA engines;
        a backend without page-locking (the CPU stand-in of the tests) is skipped. Unpinned again when the queue goes away."""
        reg = getattr(backend, "host_register", None)
        if reg is None or self._buffers is None or getattr(self, "_pinned", None) is not None:
            return
        for b in self._buffers:
            reg(b)
        self._pinned = backend

    def __del__(self) -> None:
        backend = getattr(self, "_pinned", None)
        if backend is not None and self._buffers is not None and not sys.is_finalizing():
            for b in self._buffers:
                try:
                    backend.host_unregister(b)
                except Exception:
                    pass

    def buffer(self, i: int) -> np.ndarray:
        if self._buffers is None:
            raise RuntimeError("queue has not been initialized")
        return self._buffers[i]

    def view(self, i: int) -> QueueView:
        """view onto the queue as retrieved after the most recent batch run with config `i`"""
        return QueueView(self.buffer(i), self._item, self._capacity, skipCounter=self._skipCounter)
