"""
Scene assembly: transforms, bounding boxes, mesh store, instances and the
`Scene` that owns the device acceleration structure (reference:
src/theia/scene.py:42-296 Transform, :299-417 boxes, :420-449 loadMesh,
:452-605 MeshInstance/MeshStore, :608-710 Scene). The reference hands meshes to
the Vulkan driver for BLAS/TLAS builds; here `Scene` calls `thx_scene_create`,
which builds an own two-level BVH (theia_amd/csrc/bvh_build.cpp).
"""
from __future__ import annotations

import itertools
from ctypes import Structure, c_float, c_uint64
from pathlib import Path
from typing import Iterable, Mapping

import numpy as np

from . import _backend
from . import units as u
from .hp import Mesh
from .hp.glsl import vec3
from .material import MaterialStore
from .meshio import load_mesh_arrays

__all__ = ["loadMesh", "MeshInstance", "MeshStore", "RectBBox", "Scene", "SphereBBox", "Transform"]


class Transform:
    """Affine 3x4 transformation with composition helpers (`A @ B` applies B first)."""

    def __init__(self, matrix=None) -> None:
        self._arr = np.identity(4)
        if matrix is not None:
            matrix = np.asarray_chkfinite(matrix)
            if matrix.shape != (3, 4):
                raise ValueError("matrix must be of shape (3,4)!")
            self._arr[:3, :] = matrix

    def apply(self, points):
        """points (N,3) -> transformed points"""
        return points @ self._arr[:3, :3].T + self._arr[:3, 3]

    def applyVec(self, vector):
        """like `apply` but without the translation"""
        return vector @ self._arr[:3, :3].T

    def copy(self) -> "Transform":
        return Transform(self._arr[:3, :].copy())

    def freeze(self) -> "Transform":
        self._arr.setflags(write=False)
        return self

    def inverse(self) -> "Transform":
        inv = Transform()
        inv._arr = np.linalg.inv(self._arr)
        return inv

    def numpy(self) -> np.ndarray:
        """(3,4) float32 as consumed by mesh instances"""
        return np.ascontiguousarray(self._arr[:3, :], dtype=np.float32)

    @property
    def innerMatrix(self) -> np.ndarray:
        return self.numpy()[:3, :3]

    @property
    def offset(self) -> np.ndarray:
        return self.numpy()[:3, 3]

    @staticmethod
    def Rotation(dx: float, dy: float, dz: float, angle: float) -> "Transform":
        """Rotation about the axis (dx,dy,dz) by `angle` degrees (Rodrigues)."""
        axis = np.array([dx, dy, dz], dtype=np.float64)
        axis /= np.sqrt(np.dot(axis, axis))
        K = np.array([[0.0, -axis[2], axis[1]], [axis[2], 0.0, -axis[0]], [-axis[1], axis[0], 0.0]])
        a = np.deg2rad(angle)
        res = Transform()
        res._arr[:3, :3] += np.sin(a) * K + (1.0 - np.cos(a)) * (K @ K)
        return res

    def rotate(self, dx: float, dy: float, dz: float, angle: float) -> "Transform":
        return Transform.Rotation(dx, dy, dz, angle) @ self

    @staticmethod
    def Scale(x: float, y: float | None = None, z: float | None = None) -> "Transform":
        res = Transform()
        res._arr[0, 0] = x
        res._arr[1, 1] = x if y is None else y
        res._arr[2, 2] = x if z is None else z
        return res

    def scale(self, x: float, y: float | None = None, z: float | None = None) -> "Transform":
        return Transform.Scale(x, y, z) @ self

    @staticmethod
    def Translation(x: float, y: float, z: float) -> "Transform":
        res = Transform()
        res._arr[:3, 3] = (x, y, z)
        return res

    def translate(self, x: float, y: float, z: float) -> "Transform":
        return Transform.Translation(x, y, z) @ self

    @staticmethod
    def TRS(*, scale=1.0, rotate=None, translate=(0.0, 0.0, 0.0)) -> "Transform":
        """scale, then rotate (axis + angle), then translate"""
        result = Transform()
        result = result.scale(*scale) if type(scale) is tuple else result.scale(scale)
        if rotate is not None:
            result = result.rotate(*rotate)
        return result.translate(*translate)

    @staticmethod
    def View(*, position=(0.0, 0.0, 0.0), direction=(0.0, 0.0, 1.0), up=(0.0, 1.0, 0.0)) -> "Transform":
        position, direction, up = np.array(position), np.array(direction), np.array(up)
        unit = lambda v: v / np.sqrt(np.square(v).sum(-1))
        z = unit(direction)
        x = unit(np.cross(up, direction))
        y = unit(np.cross(z, x))
        return Transform(np.stack([x, y, z, position], -1))

    @staticmethod
    def LookAt(*, position=(0.0, 0.0, 0.0), target=(0.0, 0.0, 1.0), up=(0.0, 1.0, 0.0)) -> "Transform":
        position, target = np.array(position), np.array(target)
        return Transform.View(position=position, direction=(position - target), up=np.array(up))

    def __matmul__(self, other: "Transform") -> "Transform":
        if type(other) != Transform:
            raise TypeError(other)
        result = Transform()
        result._arr = self._arr @ other._arr
        return result

    def __imatmul__(self, other: "Transform") -> "Transform":
        if type(other) != Transform:
            raise TypeError(other)
        if not self._arr.flags.writeable:
            raise RuntimeError("Tried to change frozen Transform!")
        self._arr = other._arr @ self._arr
        return self

    def __repr__(self) -> str:
        return "Transform(" + np.array2string(self.numpy(), separator=",", prefix="Transform(", suffix=")") + ")"

    def __str__(self) -> str:
        return np.array_str(self.numpy(), precision=5, suppress_small=True)


class RectBBox:
    """Axis aligned box given by its minimal and maximal corner."""

    class GLSL(Structure):
        _fields_ = [("lowerCorner", vec3), ("upperCorner", vec3)]

    def __init__(self, lowerCorner, upperCorner) -> None:
        self._glsl = self.GLSL()
        self.lowerCorner = lowerCorner
        self.upperCorner = upperCorner

    @property
    def glsl(self) -> "RectBBox.GLSL":
        return self._glsl

    @property
    def diagonal(self) -> float:
        d = np.subtract(self.upperCorner, self.lowerCorner)
        return np.sqrt(np.square(d).sum())

    @property
    def lowerCorner(self) -> tuple[float, float, float]:
        return tuple(self._glsl.lowerCorner)

    @lowerCorner.setter
    def lowerCorner(self, value) -> None:
        self._glsl.lowerCorner = vec3(*value)

    @property
    def upperCorner(self) -> tuple[float, float, float]:
        return tuple(self._glsl.upperCorner)

    @upperCorner.setter
    def upperCorner(self, value) -> None:
        self._glsl.upperCorner = vec3(*value)

    def transform(self, trafo: Transform) -> "RectBBox":
        """Box around this box after `trafo`."""
        lo, hi = self.lowerCorner, self.upperCorner
        corners = np.array(list(itertools.product((hi[0], lo[0]), (hi[1], lo[1]), (hi[2], lo[2]))))
        corners = trafo.apply(corners)
        return RectBBox(tuple(corners.min(0)), tuple(corners.max(0)))


class SphereBBox:
    """Sphere given by center and radius."""

    class GLSL(Structure):
        _fields_ = [("center", vec3), ("radius", c_float)]

    def __init__(self, center, radius: float) -> None:
        self._glsl = self.GLSL()
        self.center = center
        self.radius = radius

    @property
    def glsl(self) -> "SphereBBox.GLSL":
        return self._glsl

    @property
    def center(self) -> tuple[float, float, float]:
        return tuple(self._glsl.center)

    @center.setter
    def center(self, value) -> None:
        self._glsl.center = vec3(*value)

    @property
    def radius(self) -> float:
        return self._glsl.radius

    @radius.setter
    def radius(self, value: float) -> None:
        self._glsl.radius = value


def loadMesh(filepath) -> Mesh:
    """Loads an STL/PLY file into a `Mesh` (position + vertex normal per vertex)."""
    vertices, faces = load_mesh_arrays(filepath)
    mesh = Mesh()
    mesh.vertices = vertices
    mesh.indices = faces
    return mesh


class _GeometryInstance:
    """What hp.GeometryInstance carries: mesh reference, 3x4 transform, custom index."""

    def __init__(self, meshIndex: int) -> None:
        self.meshIndex = meshIndex
        self.transform = np.ascontiguousarray(np.identity(4)[:3, :], dtype=np.float32)
        self.customIndex = 0
        self.mask = 0xFF


class MeshInstance:
    """A placed mesh: geometry reference, transformation, material name, detector id."""

    def __init__(self, instance: _GeometryInstance, store: "MeshStore", triangleCount: int, bbox: RectBBox, material: str) -> None:
        self._instance = instance
        self._store = store
        self._triangleCount = triangleCount
        self._localBbox = bbox
        self._bbox = self.localBBox.transform(self.transform)
        self.material = material

    @property
    def bbox(self) -> RectBBox:
        return self._bbox

    @property
    def localBBox(self) -> RectBBox:
        return self._localBbox

    @property
    def instance(self) -> _GeometryInstance:
        return self._instance

    @property
    def store(self) -> "MeshStore":
        return self._store

    @property
    def material(self) -> str:
        return self._material

    @material.setter
    def material(self, value: str) -> None:
        self._material = value

    @property
    def triangleCount(self) -> int:
        return self._triangleCount

    @property
    def transform(self) -> Transform:
        return Transform(self.instance.transform)

    @transform.setter
    def transform(self, value) -> None:
        self.instance.transform = np.ascontiguousarray(value, dtype=np.float32).reshape(3, 4)
        self._bbox = self.localBBox.transform(self.transform)


class MeshStore:
    """Named meshes (objects or file paths) that instances refer to."""

    def __init__(self, meshes: Mapping) -> None:
        self._keys = list(meshes.keys())
        self._meshes = [loadMesh(v) if isinstance(v, (str, Path)) else v for v in meshes.values()]
        for key, mesh in zip(self._keys, self._meshes):
            if mesh.vertices.ndim != 2 or mesh.vertices.shape[1] != 6 or len(mesh.indices) == 0:
                raise ValueError(f'Mesh "{key}" must provide (N,6) vertices and (M,3) indices')
        self._bbox = [RectBBox(tuple(m.vertices[:, :3].min(0)), tuple(m.vertices[:, :3].max(0))) for m in self._meshes]

    @property
    def meshes(self) -> list[Mesh]:
        return self._meshes

    def createInstance(self, key: str, material: str, transform: Transform | None = None, *, detectorId: int = 0,
                       scale: float | None = None) -> MeshInstance:
        """New instance of mesh `key` with the given material name; `scale` is the
        unit of the vertex positions (default 1 m) and is applied before `transform`."""
        idx = self._keys.index(key)
        mesh = self._meshes[idx]
        inst = MeshInstance(_GeometryInstance(idx), self, len(mesh.indices), self._bbox[idx], material)
        inst.instance.customIndex = detectorId
        if scale is None:
            scale = 1.0 * u.m
        if scale != 1.0 or transform is not None:
            trafo = Transform.Scale(scale, scale, scale)
            if transform is not None:
                trafo = transform @ trafo
            inst.transform = trafo.numpy()
        return inst


class Scene:
    """
    Everything rays are traced against: instances, the material address of each,
    the surrounding `medium` (device address, 0 = vacuum) and the box `bbox`
    limiting traced rays (default +-1 km).
    """

    class GLSLGeometry(Structure):
        _fields_ = [("vertices", c_uint64), ("indices", c_uint64), ("material", c_uint64)]

    def __init__(self, instances: Iterable[MeshInstance], materials, *, medium: int = 0, bbox: RectBBox | None = None) -> None:
        instances = list(instances)
        if len(instances) == 0:
            raise ValueError("No instances given. Scene cannot be empty!")
        if isinstance(materials, MaterialStore):
            materials = materials.material
        # distinct meshes in use -> dense mesh list
        meshes, mesh_slot = [], {}
        descs = []
        for inst in instances:
            key = (id(inst.store), inst.instance.meshIndex)
            if key not in mesh_slot:
                m = inst.store.meshes[inst.instance.meshIndex]
                mesh_slot[key] = len(meshes)
                meshes.append((m.vertices, m.indices))
            if inst.material is not None:
                if inst.material not in materials:
                    raise ValueError(f'Unknown material "{inst.material}"')
                mat = materials.get(inst.material)
            else:
                mat = 0
            descs.append((mesh_slot[key], inst.instance.transform, inst.instance.customIndex, mat))
        self._backend = _backend.get()
        self._handle = self._backend.scene_create(meshes, descs)
        self._instances = instances
        self._keepalive = materials  # the proxy keeps the MaterialStore blob alive
        self.medium = medium
        if bbox is None:
            bbox = RectBBox((-1.0 * u.km,) * 3, (1.0 * u.km,) * 3)
        self.bbox = bbox

    @property
    def bbox(self) -> RectBBox:
        return self._bbox

    @bbox.setter
    def bbox(self, value: RectBBox) -> None:
        self._bbox = value

    @property
    def handle(self) -> int:
        """backend handle of the acceleration structure"""
        return self._handle

    @property
    def info(self) -> dict:
        """node / triangle / instance counts and byte sizes of the built hierarchy"""
        return self._backend.scene_info(self._handle)

    @property
    def medium(self) -> int:
        return self._medium

    @medium.setter
    def medium(self, value: int) -> None:
        self._medium = value

    def __del__(self) -> None:
        try:
            if self._handle:
                self._backend.scene_destroy(self._handle)
                self._handle = None
        except Exception:
            pass
