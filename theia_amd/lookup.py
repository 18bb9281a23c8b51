"""
Look-up tables as the kernels read them: an ``int32`` sample count per axis
followed by the ``float32`` samples (1-D: ``Table1D{int nx; float samples[];}``,
reference shader/lookup.glsl:4-12, host side src/theia/lookup.py:29-131).

Sampling points are equidistant on the table's range; the kernel interpolates
linearly (lookup.glsl:14-28).
"""
from __future__ import annotations

from typing import Literal

import numpy as np

from . import _backend

__all__ = ["Table", "getTableSize", "uploadTables", "sampleTable1D", "sampleTable2D", "evalTable"]


class Table:
    """Header + samples of one table (1-D or 2-D)."""

    ALIGNMENT = 4

    def __init__(self, data) -> None:
        self._data = np.ascontiguousarray(data, dtype=np.float32)
        self._header = np.asarray(self._data.shape, dtype=np.int32)

    @property
    def data(self) -> np.ndarray:
        return self._data

    @property
    def nbytes(self) -> int:
        return int(self._header.nbytes + self._data.nbytes)

    def tobytes(self) -> bytes:
        return self._header.tobytes() + self._data.tobytes()


def getTableSize(a) -> int:
    """Bytes a table of the given data/shape occupies on the device (0 for None)."""
    if a is None:
        return 0
    shape = a if isinstance(a, tuple) else np.shape(a)
    if len(shape) == 0:
        raise RuntimeError("table cannot have zero shape!")
    return 4 * (len(shape) + int(np.prod(shape)))


class DeviceBlob:
    """A device allocation that frees itself with its owner."""

    def __init__(self, nbytes: int) -> None:
        self._be = _backend.get()
        self.nbytes = int(nbytes)
        self.address = self._be.alloc(self.nbytes)

    def upload(self, data: bytes) -> None:
        self._be.upload(self.address, data)

    def __del__(self) -> None:
        try:
            if self.address:
                self._be.free(self.address)
                self.address = 0
        except Exception:
            pass


class DeviceAddress(int):
    """A device address that keeps the allocation it points into alive. theia hands addresses around as plain integers
    (`store.media["water"]`, `scene.medium`); here the integer also holds a reference to its `DeviceBlob`, so that a function which
    builds a material store, makes a tracer from its addresses and returns only the tracer does not leave the tracer reading freed
    memory. Arithmetic gives plain integers."""

    def __new__(cls, address: int, owner) -> "DeviceAddress":
        self = super().__new__(cls, int(address))
        self.owner = owner
        return self

    def __reduce__(self):
        return (int, (int(self),))


def uploadTables(data: list) -> tuple[DeviceBlob, list[int]]:
    """Uploads the given arrays as consecutive tables; returns the allocation and
    the device address of every table (src/theia/lookup.py:101-131)."""
    tables = [Table(d) for d in data]
    blob = DeviceBlob(sum(t.nbytes for t in tables))
    addresses, offset, payload = [], 0, b""
    for t in tables:
        addresses.append(blob.address + offset)
        payload += t.tobytes()
        offset += t.nbytes
    blob.upload(payload)
    return blob, addresses


def sampleTable1D(data, nx: int = 1024, *, boundary=None, mode: Literal["linear", "cubic"] = "linear") -> Table:
    """Resamples (x, f(x)) rows onto `nx` equidistant points."""
    data = np.asarray(data, dtype=np.float64)
    lo, hi = (data[:, 0].min(), data[:, 0].max()) if boundary is None else boundary
    x = np.linspace(lo, hi, nx)
    if mode == "linear":
        return Table(np.interp(x, data[:, 0], data[:, 1]))
    if mode == "cubic":
        from scipy.interpolate import CubicSpline

        return Table(CubicSpline(data[:, 0], data[:, 1])(x))
    raise RuntimeError("Unknown interpolation mode!")


def sampleTable2D(data, nx: int = 1024, ny: int = 1024, *, boundaries=None, mode: Literal["linear", "cubic"] = "linear") -> Table:
    """Resamples scattered (x, y, f(x, y)) rows onto a regular `ny` x `nx` grid (rows: y) by triangulation -- piecewise linear, or
    Clough-Tocher cubic; `boundaries`: ((x_min, x_max), (y_min, y_max)), either may be None = the data's own range. No kernel of this
    backend reads 2-D tables (neither does a shader of the reference); the table is there for user code (lookup.py:181-238)."""
    from scipy.interpolate import CloughTocher2DInterpolator, LinearNDInterpolator

    data = np.asarray(data, dtype=np.float64)
    if boundaries is not None and not (isinstance(boundaries, tuple) and len(boundaries) == 2):
        raise RuntimeError("Cant parse given boundaries!")
    axes = []
    for column, count, bounds in ((0, nx, None if boundaries is None else boundaries[0]), (1, ny, None if boundaries is None else boundaries[1])):
        lo, hi = (data[:, column].min(), data[:, column].max()) if bounds is None else bounds
        axes.append(np.linspace(lo, hi, count))
    if mode not in ("linear", "cubic"):
        raise RuntimeError("Unknown interpolation mode!")
    model = LinearNDInterpolator if mode == "linear" else CloughTocher2DInterpolator
    gx, gy = np.meshgrid(*axes)
    return Table(model(data[:, :2], data[:, 2])(gx, gy))


def evalTable(f, *axes) -> Table:
    """Samples `f` on a regular grid; each axis is a sample count (range [0,1])
    or a tuple (min, max, count)."""
    grids = []
    for spec in axes:
        if isinstance(spec, int):
            grids.append(np.linspace(0.0, 1.0, spec))
        elif isinstance(spec, tuple) and len(spec) == 3:
            grids.append(np.linspace(*spec))
        else:
            raise RuntimeError(f"Cannot parse dimension: {spec}")
    mesh = np.meshgrid(*grids, indexing="ij")
    return Table(f(*mesh))
