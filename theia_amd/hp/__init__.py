"""
The slice of the `hephaistos` API that theia's forward tracers, examples and
tests touch, implemented on the HIP backend: device queries,
`Mesh`, buffers / tensors and the commands that copy between them (`memory`),
and the `pipeline` / `glsl` / `queue` submodules. Programs and compilers (GLSL
at run time) have no counterpart here -- component GLSL injection is replaced
by compiled kernel variants -- and say so when constructed.
"""
from __future__ import annotations

import numpy as np

from .. import _backend
from . import glsl, memory, pipeline, queue  # noqa: F401
from .memory import *  # noqa: F401,F403
from .pipeline import Command  # noqa: F401

__all__ = [
    "Mesh", "getCurrentDevice", "isRaytracingSupported", "isRaytracingEnabled", "enableRaytracing",
    "suitableDeviceAvailable", "destroyResources", "selectDevice", "enumerateDevices", "enableAtomics", "Command", "glsl", "pipeline", "queue",
    "memory",
] + memory.__all__


class Device:
    def __init__(self, name: str, info: dict) -> None:
        self.name = name
        self.info = info
        self.isDiscrete = True
        self.supportsRaytracing = True

    def __repr__(self) -> str:
        return f"Device({self.name!r}, CUs={self.info.get('compute_units')}, HBM={self.info.get('hbm_bytes', 0) / 2**30:.0f} GiB)"

    __str__ = __repr__


def getCurrentDevice() -> Device:
    """The device the backend runs on (creates the context on first use)."""
    be = _backend.get()
    return Device(be.device_name(), be.device_info())


def enumerateDevices() -> list[Device]:
    return [getCurrentDevice()]


def selectDevice(index: int, *, force: bool = False) -> None:
    """Selects the GPU; must run before anything touched the backend."""
    _backend.reset()
    _backend.install(_backend.HipBackend(index))


def suitableDeviceAvailable() -> bool:
    try:
        _backend.get()
        return True
    except Exception:
        return False


def isRaytracingSupported(*_a) -> bool:
    """BVH traversal is part of libtheia_hip.so; any supported device can do it."""
    return True


def isRaytracingEnabled() -> bool:
    return True


def enableRaytracing(*_a, **_k) -> None:
    pass


def enableAtomics(*_a, **_k) -> None:
    pass


def destroyResources() -> None:
    """Device memory is reference counted by its Python owners; nothing to do."""


class Mesh:
    """Vertex data ``(N,6)`` float32 (position + normal) and triangle indices ``(M,3)`` uint32."""

    def __init__(self) -> None:
        self._vertices = np.zeros((0, 6), dtype=np.float32)
        self._indices = np.zeros((0, 3), dtype=np.uint32)

    @property
    def vertices(self) -> np.ndarray:
        return self._vertices

    @vertices.setter
    def vertices(self, value) -> None:
        self._vertices = np.ascontiguousarray(value, dtype=np.float32)

    @property
    def indices(self) -> np.ndarray:
        return self._indices

    @indices.setter
    def indices(self, value) -> None:
        self._indices = np.ascontiguousarray(value, dtype=np.uint32)
