"""
`hephaistos.queue` as far as theia's HitRecorder needs it: a structure-of-arrays
queue with a leading item counter, described by a ctypes `Structure` (reference
use: src/theia/response.py:219, 268-274; device layout shader/response.queue.glsl:6-33
-- `uint count`, then one column of `capacity` entries per scalar of the item, in
field order; a `c_float * 3` field is three columns).

Only the host side lives here: `QueueView` gives numpy views onto a retrieved queue,
`IOQueue(mode="retrieve")` owns the two host buffers of the double-buffered pipeline.
The device queue belongs to the tracer the recorder is bound to.
"""
from __future__ import annotations

import ctypes
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
                        base, shape=(self._capacity, n), strides=(4, 4 * self._capacity), writeable=False)
                    return arr[self._slice]
            raise KeyError(key)
        if isinstance(key, slice):
            start, stop, step = self._slice.indices(self._capacity)
            idx = range(start, stop, step)[key]
            return QueueView(self._data, self._item, self._capacity, skipCounter=self._skipCounter,
                             _slice=slice(idx.start, idx.stop, idx.step))
        raise TypeError("index a QueueView with a field name or a slice")

    def __iter__(self):
        return iter(self.fields)


def as_queue(data: np.ndarray, item: type[Structure], *, capacity: int, skipCounter: bool = False) -> QueueView:
    return QueueView(np.asarray(data).view(np.uint8).reshape(-1), item, capacity, skipCounter=skipCounter)


def dumpQueue(view: QueueView) -> dict[str, np.ndarray]:
    """copies of the valid items of a queue, one array per field (hephaistos.queue.dumpQueue)"""
    n = view.count if view.hasCounter else len(view)
    return {name: np.array(view[name][:n]) for name in view.fields}


class IOQueue:
    """Host side of a queue that is filled on the device and retrieved after each batch."""

    def __init__(self, item: type[Structure], *, mode: str = "retrieve", skipCounter: bool = False) -> None:
        if mode != "retrieve":
            raise NotImplementedError("only IOQueue(mode='retrieve') exists on the HIP backend")
        self._item = item
        self._mode = mode
        self._skipCounter = skipCounter
        self._capacity = 0
        self._buffers: list[np.ndarray] | None = None

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
        self._buffers = [np.zeros(self.nbytes, dtype=np.uint8) for _ in range(2)]

    def buffer(self, i: int) -> np.ndarray:
        if self._buffers is None:
            raise RuntimeError("queue has not been initialized")
        return self._buffers[i]

    def view(self, i: int) -> QueueView:
        """view onto the queue as retrieved after the most recent batch run with config `i`"""
        return QueueView(self.buffer(i), self._item, self._capacity, skipCounter=self._skipCounter)
