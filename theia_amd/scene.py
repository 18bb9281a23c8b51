"""
Scene assembly: transforms, bounding boxes, mesh store, instances and the
`Scene` that owns the device acceleration structure (reference:
src/theia/scene.py:42-296 Transform, :299-417 boxes, :420-449 loadMesh,
:452-605 MeshInstance/MeshStore, :608-710 Scene). The reference hands meshes to
the Vulkan driver for BLAS/TLAS builds; here `Scene` calls `thx_scene_create`,
which builds an own two-level BVH (theia_amd/csrc/bvh_build.cpp).
"""
from __future__ import annotations

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


def _unit(v) -> np.ndarray:
    v = np.asarray(v, dtype=np.float64)
    return v / np.linalg.norm(v)


class Transform:
    """
    Affine map x -> L x + t, kept as its linear part `L` (3x3) and shift `t` (float64). `A @ B`
    is "first B, then A"; the chaining methods (`rotate`, `scale`, `translate`) return a new
    transform that applies the extra step AFTER this one. API of reference src/theia/scene.py:42-296.
    """

    __slots__ = ("_L", "_t", "_locked")

    def __init__(self, matrix=None) -> None:
        self._locked = False
        if matrix is None:
            self._L, self._t = np.eye(3), np.zeros(3)
            return
        m = np.array(matrix, dtype=np.float64)
        if m.shape != (3, 4):
            raise ValueError(f"a Transform takes a 3x4 matrix, got shape {m.shape}")
        if not np.isfinite(m).all():
            raise ValueError("a Transform must not contain inf or NaN")
        self._L, self._t = m[:, :3].copy(), m[:, 3].copy()

    @classmethod
    def _of(cls, linear, shift) -> "Transform":
        out = cls()
        out._L = np.array(linear, dtype=np.float64)
        out._t = np.array(shift, dtype=np.float64)
        return out

    # ---- evaluation
    def apply(self, points):
        """maps points, shape (..., 3)"""
        return np.asarray(points) @ self._L.T + self._t

    def applyVec(self, vector):
        """maps directions, shape (..., 3): the shift does not act on them"""
        return np.asarray(vector) @ self._L.T

    def numpy(self) -> np.ndarray:
        """(3,4) float32 [L | t], the form mesh instances store"""
        return np.concatenate([self._L, self._t[:, None]], axis=1).astype(np.float32)

    @property
    def innerMatrix(self) -> np.ndarray:
        return self._L.astype(np.float32)

    @property
    def offset(self) -> np.ndarray:
        return self._t.astype(np.float32)

    def copy(self) -> "Transform":
        return Transform._of(self._L, self._t)

    def freeze(self) -> "Transform":
        """in-place composition (`@=`) raises from now on"""
        self._locked = True
        return self

    def inverse(self) -> "Transform":
        Li = np.linalg.inv(self._L)
        return Transform._of(Li, -Li @ self._t)

    # ---- elementary maps
    @staticmethod
    def Rotation(dx: float, dy: float, dz: float, angle: float) -> "Transform":
        """right-handed turn by `angle` degrees about the axis (dx, dy, dz)"""
        a = _unit((dx, dy, dz))
        c, s = np.cos(np.radians(angle)), np.sin(np.radians(angle))
        cross = np.array([[0.0, -a[2], a[1]], [a[2], 0.0, -a[0]], [-a[1], a[0], 0.0]])
        return Transform._of(c * np.eye(3) + s * cross + (1.0 - c) * np.outer(a, a), np.zeros(3))

    @staticmethod
    def Scale(x: float, y: float | None = None, z: float | None = None) -> "Transform":
        """per-axis factors; a single factor scales uniformly"""
        return Transform._of(np.diag([x, x if y is None else y, x if z is None else z]), np.zeros(3))

    @staticmethod
    def Translation(x: float, y: float, z: float) -> "Transform":
        return Transform._of(np.eye(3), (x, y, z))

    def rotate(self, dx: float, dy: float, dz: float, angle: float) -> "Transform":
        return Transform.Rotation(dx, dy, dz, angle) @ self

    def scale(self, x: float, y: float | None = None, z: float | None = None) -> "Transform":
        return Transform.Scale(x, y, z) @ self

    def translate(self, x: float, y: float, z: float) -> "Transform":
        return Transform.Translation(x, y, z) @ self

    @staticmethod
    def TRS(*, scale=1.0, rotate=None, translate=(0.0, 0.0, 0.0)) -> "Transform":
        """object placement: scale (number or 3-tuple), then rotate (axis x, y, z, degrees), then shift"""
        factors = scale if isinstance(scale, tuple) else (scale,)
        placed = Transform.Scale(*factors)
        if rotate is not None:
            placed = Transform.Rotation(*rotate) @ placed
        return Transform.Translation(*translate) @ placed

    @staticmethod
    def View(*, position=(0.0, 0.0, 0.0), direction=(0.0, 0.0, 1.0), up=(0.0, 1.0, 0.0)) -> "Transform":
        """frame whose local z axis points along `direction`, local y axis as close to `up` as possible"""
        ez = _unit(direction)
        ex = _unit(np.cross(np.asarray(up, dtype=np.float64), np.asarray(direction, dtype=np.float64)))
        ey = _unit(np.cross(ez, ex))
        return Transform._of(np.column_stack([ex, ey, ez]), position)

    @staticmethod
    def LookAt(*, position=(0.0, 0.0, 0.0), target=(0.0, 0.0, 1.0), up=(0.0, 1.0, 0.0)) -> "Transform":
        away = np.subtract(position, target).astype(np.float64)
        return Transform.View(position=position, direction=away, up=up)

    # ---- composition
    def _compose(self, first: "Transform") -> tuple[np.ndarray, np.ndarray]:
        """self after `first`"""
        return self._L @ first._L, self._L @ first._t + self._t

    def __matmul__(self, other: "Transform") -> "Transform":
        if not isinstance(other, Transform):
            return NotImplemented
        return Transform._of(*self._compose(other))

    def __imatmul__(self, other: "Transform") -> "Transform":
        """`t @= o`: o is applied after t (the chaining order)"""
        if not isinstance(other, Transform):
            return NotImplemented
        if self._locked:
            raise RuntimeError("this Transform is frozen and cannot be modified in place")
        self._L, self._t = other._compose(self)
        return self

    def __repr__(self) -> str:
        rows = "; ".join(" ".join(f"{v:.6g}" for v in row) for row in self.numpy())
        return f"Transform[{rows}]"

    __str__ = __repr__


class _Vec3Field:
    """tuple-valued attribute backed by a numpy row (what RectBBox / SphereBBox expose)"""

    def __init__(self, row: int) -> None:
        self.row = row

    def __get__(self, obj, owner=None):
        if obj is None:
            return self
        return tuple(float(v) for v in obj._data[self.row])

    def __set__(self, obj, value) -> None:
        obj._data[self.row] = np.asarray(value, dtype=np.float32).reshape(3)


class RectBBox:
    """Axis-aligned box between `lowerCorner` and `upperCorner` (reference scene.py:299-366)."""

    class GLSL(Structure):
        _fields_ = [("lowerCorner", vec3), ("upperCorner", vec3)]

    lowerCorner = _Vec3Field(0)
    upperCorner = _Vec3Field(1)

    def __init__(self, lowerCorner, upperCorner) -> None:
        self._data = np.zeros((2, 3), dtype=np.float32)
        self.lowerCorner, self.upperCorner = lowerCorner, upperCorner

    @property
    def glsl(self) -> "RectBBox.GLSL":
        """the box as the ctypes block the parameter structs embed"""
        return self.GLSL(vec3(*self.lowerCorner), vec3(*self.upperCorner))

    @property
    def diagonal(self) -> float:
        return float(np.linalg.norm(self._data[1].astype(np.float64) - self._data[0]))

    def transform(self, trafo: Transform) -> "RectBBox":
        """the axis-aligned box around the eight mapped corners"""
        pick = (np.arange(8)[:, None] >> np.arange(3)) & 1          # (8,3) of 0/1: which corner per axis
        moved = trafo.apply(self._data[pick, np.arange(3)].astype(np.float64))
        return RectBBox(tuple(moved.min(axis=0)), tuple(moved.max(axis=0)))

    def __repr__(self) -> str:
        return f"RectBBox({self.lowerCorner}, {self.upperCorner})"


class SphereBBox:
    """Bounding sphere (reference scene.py:369-417)."""

    class GLSL(Structure):
        _fields_ = [("center", vec3), ("radius", c_float)]

    center = _Vec3Field(0)

    def __init__(self, center, radius: float) -> None:
        self._data = np.zeros((1, 3), dtype=np.float32)
        self.center = center
        self.radius = float(radius)

    @property
    def glsl(self) -> "SphereBBox.GLSL":
        return self.GLSL(vec3(*self.center), self.radius)

    def __repr__(self) -> str:
        return f"SphereBBox({self.center}, {self.radius})"


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
    """A placed mesh: geometry reference (`instance`), the `store` it came from, its placement
    (`transform`, settable), the `material` name (settable) and its boxes before / after placement
    (reference scene.py:452-526)."""

    def __init__(self, instance: _GeometryInstance, store: "MeshStore", triangleCount: int, bbox: RectBBox, material: str) -> None:
        self._fixed = (instance, store, int(triangleCount), bbox)
        self.material = material
        self._refit()

    instance = property(lambda self: self._fixed[0], doc="geometry reference, 3x4 transform and custom index")
    store = property(lambda self: self._fixed[1], doc="the MeshStore the mesh lives in")
    triangleCount = property(lambda self: self._fixed[2], doc="triangles of the mesh")
    localBBox = property(lambda self: self._fixed[3], doc="box of the mesh in its own frame")
    bbox = property(lambda self: self._placed, doc="box of the placed mesh")

    def _refit(self) -> None:
        self._placed = self.localBBox.transform(self.transform)

    @property
    def transform(self) -> Transform:
        return Transform(self.instance.transform)

    @transform.setter
    def transform(self, matrix) -> None:
        self.instance.transform = np.array(matrix, dtype=np.float32, order="C").reshape(3, 4)
        self._refit()


class MeshStore:
    """Named meshes (Mesh objects, or paths of STL / PLY files) that instances refer to
    (reference scene.py:529-605)."""

    def __init__(self, meshes: Mapping) -> None:
        self._slot = {name: k for k, name in enumerate(meshes)}
        self._meshes = []
        self._bbox = []
        for name, source in meshes.items():
            mesh = loadMesh(source) if isinstance(source, (str, Path)) else source
            v = np.asarray(mesh.vertices)
            if v.ndim != 2 or v.shape[1] != 6 or len(mesh.indices) == 0:
                raise ValueError(f'Mesh "{name}" must provide (N,6) vertices and (M,3) indices')
            self._meshes.append(mesh)
            self._bbox.append(RectBBox(tuple(v[:, :3].min(axis=0)), tuple(v[:, :3].max(axis=0))))

    @property
    def meshes(self) -> list[Mesh]:
        return self._meshes

    def createInstance(self, key: str, material: str, transform: Transform | None = None, *, detectorId: int = 0,
                       scale: float | None = None) -> MeshInstance:
        """New instance of mesh `key` with the given material name; `scale` is the
        unit of the vertex positions (default 1 m) and is applied before `transform`."""
        if key not in self._slot:
            raise KeyError(f'no mesh named "{key}" in this store')
        k = self._slot[key]
        geometry = _GeometryInstance(k)
        geometry.customIndex = detectorId
        placed = MeshInstance(geometry, self, len(self._meshes[k].indices), self._bbox[k], material)
        unit = 1.0 * u.m if scale is None else scale
        if transform is not None or unit != 1.0:
            placement = Transform.Scale(unit)
            placed.transform = (placement if transform is None else transform @ placement).numpy()
        return placed


class Scene:
    """
    Everything rays are traced against: instances, the material address of each,
    the surrounding `medium` (device address, 0 = vacuum) and the box `bbox`
    limiting traced rays (default +-1 km).
    """

    class GLSLGeometry(Structure):
        _fields_ = [("vertices", c_uint64), ("indices", c_uint64), ("material", c_uint64)]

    def __init__(self, instances: Iterable[MeshInstance], materials, *, medium: int = 0, bbox: RectBBox | None = None) -> None:
        instances = [*instances]
        if not instances:
            raise ValueError("a Scene needs at least one instance (got an empty list)")
        materials = materials.material if isinstance(materials, MaterialStore) else materials
        # distinct meshes in use -> dense mesh list
        meshes, mesh_slot = [], {}
        descs = []
        for inst in instances:
            key = (id(inst.store), inst.instance.meshIndex)
            if key not in mesh_slot:
                m = inst.store.meshes[inst.instance.meshIndex]
                mesh_slot[key] = len(meshes)
                meshes.append((m.vertices, m.indices))
            mat = 0  # no material: address 0
            if inst.material is not None:
                mat = materials.get(inst.material) if inst.material in materials else None
                if mat is None:
                    raise ValueError(f'Unknown material "{inst.material}" on an instance of the scene')
            descs.append((mesh_slot[key], inst.instance.transform, inst.instance.customIndex, mat))
        self._backend = _backend.get()
        self._handle = self._backend.scene_create(meshes, descs)
        self._instances = instances
        self._keepalive = materials  # (its values are `DeviceAddress`es: they keep the MaterialStore's blob alive)
        #: device address of the medium surrounding the instances (0 = vacuum); plain attribute
        self.medium = medium
        #: rays leaving this box are lost; default: a cube of 2 km around the origin
        half = 1.0 * u.km
        self.bbox = bbox if bbox is not None else RectBBox((-half, -half, -half), (half, half, half))

    @property
    def handle(self) -> int:
        """backend handle of the acceleration structure"""
        return self._handle

    @property
    def info(self) -> dict:
        """node / triangle / instance counts and byte sizes of the built hierarchy"""
        return self._backend.scene_info(self._handle)

    def __del__(self) -> None:
        try:
            if self._handle:
                self._backend.scene_destroy(self._handle)
                self._handle = None
        except Exception:
            pass
