"""
Analytic sphere detector of the volume tracer and sphere guide of the scene
tracer (reference: src/theia/target.py:45-139, :440-525; device side
shader/target.sphere.glsl, target_guide.sphere.glsl).
"""
from __future__ import annotations

from ctypes import Structure, c_float

import numpy as np

from . import units as u
from .hp.glsl import mat3, vec3
from .hp.pipeline import SourceCodeMixin

__all__ = ["Target", "SphereTarget", "InnerSphereTarget", "FlatTarget", "DiskTarget", "TargetGuide", "SphereTargetGuide",
           "FlatTargetGuide", "DiskTargetGuide"]


class Target(SourceCodeMixin):
    name = "Target"

    def __init__(self, *, nRNGSamples: int, params: dict = {}, extra: set = set()) -> None:
        super().__init__(params, extra)
        self._nRNGSamples = nRNGSamples

    @property
    def nRNGSamples(self) -> int:
        return self._nRNGSamples


class SphereTarget(Target):
    """Sphere sampled over the cap visible from the observer; hits are reported
    on the unit sphere of the target's local frame."""

    name = "Sphere Target"

    class TargetParams(Structure):
        _fields_ = [("position", vec3), ("radius", c_float), ("_invPos", vec3), ("_invRad", c_float), ("_hemisphereProb", c_float)]

    def __init__(self, *, position=(0.0, 0.0, 0.0), radius: float = 1.0 * u.m) -> None:
        super().__init__(nRNGSamples=2, params={"TargetParams": self.TargetParams})
        self.setParams(position=position, radius=radius)

    def _finishParams(self, i: int) -> None:
        r = self.getParam("radius")
        x, y, z = self.getParam("position")
        self.setParams(_hemisphereProb=1.0 / (2.0 * np.pi * r**2), _invPos=(-x / r, -y / r, -z / r), _invRad=1.0 / r)


class InnerSphereTarget(Target):
    """Sphere seen from the inside (the light is inside the target); sampled uniformly over its
    surface. Hits are reported on the unit sphere of the target's local frame with the normal
    pointing to the centre. Reference: target.py:142-199, shader/target.sphere.inner.glsl."""

    name = "Inner Sphere Target"

    class TargetParams(Structure):
        _fields_ = [("position", vec3), ("radius", c_float), ("_invPos", vec3), ("_invRad", c_float), ("_prob", c_float)]

    def __init__(self, *, position=(0.0, 0.0, 0.0), radius: float = 1.0 * u.m) -> None:
        super().__init__(nRNGSamples=2, params={"TargetParams": self.TargetParams})
        self.setParams(position=position, radius=radius)

    def _finishParams(self, i: int) -> None:
        r = self.getParam("radius")
        x, y, z = self.getParam("position")
        self.setParams(_prob=1.0 / (4.0 * np.pi * r**2), _invPos=(-x / r, -y / r, -z / r), _invRad=1.0 / r)


def _viewMatrix(direction, up) -> np.ndarray:
    """inner (rotation) matrix of `Transform.View(direction=direction, up=up)`: columns x, y, z of the
    object frame in world space; `up=None`: any frame around `direction` (PBRT 4th ed., ch. 3.3.3)"""
    unit = lambda v: v / np.sqrt(np.square(v).sum(-1))
    z = unit(np.array(direction, dtype=np.float64))
    if up is None:
        vx, vy, vz = z
        sign = np.copysign(1.0, vz)
        a = -1.0 / (sign + vz)
        b = vx * vy * a
        x = np.array([1.0 + sign * vx**2 * a, sign * b, -sign * vx])
        y = np.array([b, sign + vy**2 * a, -vy])
    else:
        x = unit(np.cross(np.array(up, dtype=np.float64), z))
        y = unit(np.cross(z, x))
    return np.stack([x, y, z], -1)


class FlatTarget(Target):
    """
    Rectangular target: in its local frame it lies in the xy plane centred at the origin, `width`
    along x, `length` along y, the normal `direction` along +z, `up` along +y; both sides detect
    and are told apart by the sign of the local normal. Parameters / stage parameters as in the
    reference (target.py:200-317, shader/target.flat.glsl): `width, length, position, direction, up`.
    """

    name = "Flat Target"

    class TargetParams(Structure):
        _fields_ = [("width", c_float), ("length", c_float), ("position", vec3), ("_normal", vec3), ("_prob", c_float),
                    ("_objToWorld", mat3)]

    def __init__(self, *, width: float = 1.0 * u.cm, length: float = 1.0 * u.cm, position=(0.0, 0.0, 0.0),
                 direction=(0.0, 0.0, 1.0), up=(0.0, 1.0, 0.0), missIsInfinite: bool = False) -> None:
        super().__init__(nRNGSamples=2, params={"TargetParams": self.TargetParams}, extra={"direction", "up"})
        self.missIsInfinite = missIsInfinite  # extension, see `targetFlags`
        self.direction = direction
        self.up = up
        self.setParams(width=width, length=length, position=position)

    def _finishParams(self, i: int) -> None:
        super()._finishParams(i)
        m = _viewMatrix(self.direction, self.up)
        self.setParam("_objToWorld", m.T)  # column major
        self.setParam("_normal", m[:, 2])
        self.setParam("_prob", 1.0 / (self.width * self.length))


class DiskTarget(Target):
    """
    Circular target: in its local frame it lies in the xy plane centred at the origin with the
    `normal` along +z and `up` along +y; both sides detect. Parameters / stage parameters as in
    the reference (target.py:320-425, shader/target.disk.glsl): `position, radius, normal, up`.
    """

    name = "Disk Target"

    class TargetParams(Structure):
        _fields_ = [("position", vec3), ("radius", c_float), ("_normal", vec3), ("_prob", c_float), ("_objToWorld", mat3)]

    def __init__(self, *, position=(0.0, 0.0, 0.0), radius: float = 1.0 * u.m, normal=(0.0, 0.0, 1.0), up=None,
                 missIsInfinite: bool = False) -> None:
        super().__init__(nRNGSamples=2, params={"TargetParams": self.TargetParams}, extra={"normal", "up"})
        self.missIsInfinite = missIsInfinite  # extension, see `targetFlags`
        self.normal = normal
        self.up = up
        self.setParams(position=position, radius=radius)

    def _finishParams(self, i: int) -> None:
        super()._finishParams(i)
        m = _viewMatrix(self.normal, self.up)
        self.setParam("_objToWorld", m.T)  # column major
        self.setParam("_normal", m[:, 2])  # normalised
        self.setParam("_prob", 1.0 / (np.pi * self.getParam("radius") ** 2))


class TargetGuide(SourceCodeMixin):
    name = "Target Guide"

    def __init__(self, *, nRNGSamples: int, params: dict = {}, extra: set = set()) -> None:
        super().__init__(params, extra)
        self._nRNGSamples = nRNGSamples

    @property
    def nRNGSamples(self) -> int:
        return self._nRNGSamples


class SphereTargetGuide(TargetGuide):
    """Guides scattered rays into the cone subtended by a sphere (samples its far side)."""

    name = "Sphere Target Guide"

    class TargetGuideParams(Structure):
        _fields_ = [("position", vec3), ("radius", c_float), ("_prob", c_float)]

    def __init__(self, *, position=(0.0, 0.0, 0.0), radius: float = 1.0 * u.m) -> None:
        super().__init__(nRNGSamples=2, params={"TargetGuideParams": self.TargetGuideParams})
        self.setParams(position=position, radius=radius)

    def _finishParams(self, i: int) -> None:
        r = self.getParam("radius")
        self.setParam("_prob", 1.0 / (2.0 * np.pi * r**2))


class FlatTargetGuide(TargetGuide):
    """
    Rectangular target guide (`width` orthogonal to `normal` and `up`, `height` along `up`); only
    the face the normal points to is sampled -- from behind it the next event estimate is skipped.
    Parameters / stage parameters as in the reference (target.py:527-636,
    shader/target_guide.flat.glsl): `width, height, position, normal, up`.
    """

    name = "Flat Target Guide"

    class TargetGuideParams(Structure):
        _fields_ = [("width", c_float), ("height", c_float), ("position", vec3), ("_normal", vec3), ("_prob", c_float),
                    ("_objToWorld", mat3)]

    def __init__(self, *, width: float = 1.0 * u.m, height: float = 1.0 * u.m, position=(0.0, 0.0, 0.0),
                 normal=(0.0, 0.0, 1.0), up=(0.0, 1.0, 0.0)) -> None:
        super().__init__(nRNGSamples=2, params={"TargetGuideParams": self.TargetGuideParams}, extra={"normal", "up"})
        self.normal = normal
        self.up = up
        self.setParams(width=width, height=height, position=position)

    def _finishParams(self, i: int) -> None:
        super()._finishParams(i)
        m = _viewMatrix(self.normal, self.up)
        self.setParam("_objToWorld", m.T)  # column major
        self.setParam("_normal", m[:, 2])
        self.setParam("_prob", 1.0 / (self.width * self.height))


class DiskTargetGuide(TargetGuide):
    """
    Circular target guide; only the face the `normal` points to is sampled. Parameters / stage
    parameters as in the reference (target.py:639-735, shader/target_guide.disk.glsl):
    `position, radius, normal`.
    """

    name = "Disk Target Guide"

    class TargetGuideParams(Structure):
        _fields_ = [("position", vec3), ("radius", c_float), ("_normal", vec3), ("_prob", c_float), ("_objToWorld", mat3)]

    def __init__(self, *, position=(0.0, 0.0, 0.0), radius: float = 1.0 * u.m, normal=(0.0, 0.0, 1.0)) -> None:
        super().__init__(nRNGSamples=2, params={"TargetGuideParams": self.TargetGuideParams}, extra={"normal"})
        self.normal = normal
        self.setParams(position=position, radius=radius)

    def _finishParams(self, i: int) -> None:
        r = self.getParam("radius")
        self.setParam("_prob", 1.0 / (np.pi * r**2))
        m = _viewMatrix(self.normal, None)
        self.setParam("_normal", m[:, 2])
        self.setParam("_objToWorld", m)  # as the reference passes it (target.py:734-735)


# ---- what the HIP backend needs to know about targets and guides -----------------------------

_TARGETS = (SphereTarget, InnerSphereTarget, DiskTarget, FlatTarget)
_GUIDES = (SphereTargetGuide, DiskTargetGuide, FlatTargetGuide)


def targetFlags(target) -> int:
    """THX_FLAG_* bits selecting the target's kernel branch. Plane targets (`DiskTarget`, `FlatTarget`)
    behave like the reference's shaders by default: `intersectTarget` reports the distance to the
    target's PLANE even where the outline is missed or the plane lies behind the ray, and the volume
    tracers stop the ray there (target.disk.glsl:60, target.flat.glsl:66, tracer.volume.forward.glsl:167).
    `missIsInfinite=True` on the target (an extension of this backend) reports such a miss as
    infinitely far instead."""
    from . import _capi as capi

    flags = {InnerSphereTarget: capi.FLAG_TARGET_INNER, DiskTarget: capi.FLAG_TARGET_DISK, FlatTarget: capi.FLAG_TARGET_FLAT}.get(type(target), 0)
    if getattr(target, "missIsInfinite", False):
        flags |= capi.FLAG_TARGET_MISS_INFINITE
    return flags


def guideFlags(guide) -> int:
    from . import _capi as capi

    return {DiskTargetGuide: capi.FLAG_GUIDE_DISK, FlatTargetGuide: capi.FLAG_GUIDE_FLAT}.get(type(guide), 0)


def fillTargetParams(dst, target, i: int) -> None:
    """copies a target's parameter block of config `i` into `thx_target_params`"""
    g = target._getParamsBlock("TargetParams", i)
    for k, c in enumerate("xyz"):
        dst.position[k] = getattr(g.position, c)
    if isinstance(target, (DiskTarget, FlatTarget)):
        for k, c in enumerate("xyz"):
            dst.normal[k] = getattr(g._normal, c)
        for k in range(9):
            dst.objToWorld[k] = g._objToWorld.data[k]
        dst.hemisphereProb = g._prob
        if isinstance(target, DiskTarget):
            dst.radius = g.radius
        else:
            dst.width, dst.height = g.width, g.length
        return
    for k, c in enumerate("xyz"):
        dst.invPos[k] = getattr(g._invPos, c)
    dst.radius, dst.invRad = g.radius, g._invRad
    dst.hemisphereProb = g._prob if isinstance(target, InnerSphereTarget) else g._hemisphereProb


def fillGuideParams(dst, guide, i: int) -> None:
    """copies a target guide's parameter block of config `i` into `thx_guide_params`"""
    g = guide._getParamsBlock("TargetGuideParams", i)
    for k, c in enumerate("xyz"):
        dst.position[k] = getattr(g.position, c)
    dst.prob = g._prob
    if isinstance(guide, SphereTargetGuide):
        dst.radius = g.radius
        return
    for k, c in enumerate("xyz"):
        dst.normal[k] = getattr(g._normal, c)
    for k in range(9):
        dst.objToWorld[k] = g._objToWorld.data[k]
    if isinstance(guide, DiskTargetGuide):
        dst.radius = g.radius
    else:
        dst.width, dst.height = g.width, g.height
