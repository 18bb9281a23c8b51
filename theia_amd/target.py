"""
Detector shapes: the analytic targets of the volume tracers (sphere seen from outside / inside, disk, rectangle) and
the target guides of the scene tracers (sphere, disk, rectangle). Host side only: parameter blocks with the
reference's layouts (src/theia/target.py:45-735) and their derived fields; the device side is
`theia_amd/csrc/dev_trace.hpp` (shader/target.*.glsl, target_guide.*.glsl).
"""
from __future__ import annotations

from ctypes import Structure, c_float

import numpy as np

from . import units as u
from .hp.glsl import _layout, mat3, vec3
from .hp.pipeline import SourceCodeMixin

__all__ = ["Target", "SphereTarget", "InnerSphereTarget", "FlatTarget", "DiskTarget", "TargetGuide", "SphereTargetGuide",
           "FlatTargetGuide", "DiskTargetGuide"]


class _Sampled(SourceCodeMixin):
    """what targets and target guides have in common: a stage with parameter blocks that draws a fixed number of
    random numbers per sample"""

    def __init__(self, *, nRNGSamples, params={}, extra=set()):
        super().__init__(params, extra)
        self._draws = int(nRNGSamples)

    nRNGSamples = property(lambda self: self._draws, doc="random numbers drawn per sample")


class Target(_Sampled):
    """Detector shape of the volume tracers (reference target.py:45-75)."""

    name = "Target"


class TargetGuide(_Sampled):
    """Shape the scene tracers aim their next-event samples at (reference target.py:440-467)."""

    name = "Target Guide"


def _frame(axis, up=None) -> np.ndarray:
    """3x3 rotation whose columns are the x, y, z axes of an object frame in world space, z along `axis`. With `up`
    the y axis is as close to it as possible (= the rotation part of `Transform.View`); without, x and y are the
    branch-free orthonormal basis of Duff et al. 2017 -- the same one the shaders build (math.glsl:45-55)."""
    z = np.asarray(axis, dtype=np.float64)
    z = z / np.linalg.norm(z)
    if up is not None:
        x = np.cross(np.asarray(up, dtype=np.float64), z)
        x /= np.linalg.norm(x)
        y = np.cross(z, x)
        return np.column_stack([x, y / np.linalg.norm(y), z])
    s = np.copysign(1.0, z[2])
    a = -1.0 / (s + z[2])
    b = z[0] * z[1] * a
    return np.column_stack([(1.0 + s * z[0] * z[0] * a, s * b, -s * z[0]), (b, s + z[1] * z[1] * a, -z[1]), z])


class _SphereShape:
    """`position`, `radius` and the unit-sphere frame `_invPos`, `_invRad` hits are reported in"""

    def _unit_frame(self) -> dict:
        r = self.getParam("radius")
        return {"_invPos": tuple(-c / r for c in self.getParam("position")), "_invRad": 1.0 / r}


class SphereTarget(_SphereShape, Target):
    """Sphere seen from outside, sampled over the cap visible from the observer; hits are reported on the unit sphere
    of the target's own frame. Reference target.py:78-139, shader/target.sphere.glsl."""

    name = "Sphere Target"

    class TargetParams(Structure):
        _fields_ = _layout(position=vec3, radius=c_float, _invPos=vec3, _invRad=c_float, _hemisphereProb=c_float)

    def __init__(self, *, position=(0.0, 0.0, 0.0), radius=1.0 * u.m):
        Target.__init__(self, nRNGSamples=2, params={"TargetParams": self.TargetParams})
        self.setParams(position=position, radius=radius)

    def _finishParams(self, i):
        half_area = 2.0 * np.pi * self.getParam("radius") ** 2
        self.setParams(_hemisphereProb=1.0 / half_area, **self._unit_frame())


class InnerSphereTarget(_SphereShape, Target):
    """Sphere seen from the inside (the light is inside the target), sampled uniformly over its whole surface; hits on
    the unit sphere of the target's own frame, normal towards the centre. Reference target.py:142-199,
    shader/target.sphere.inner.glsl."""

    name = "Inner Sphere Target"

    class TargetParams(Structure):
        _fields_ = _layout(position=vec3, radius=c_float, _invPos=vec3, _invRad=c_float, _prob=c_float)

    def __init__(self, *, position=(0.0, 0.0, 0.0), radius=1.0 * u.m):
        Target.__init__(self, nRNGSamples=2, params={"TargetParams": self.TargetParams})
        self.setParams(position=position, radius=radius)

    def _finishParams(self, i):
        area = 4.0 * np.pi * self.getParam("radius") ** 2
        self.setParams(_prob=1.0 / area, **self._unit_frame())


class _PlaneShape:
    """flat shapes: the object frame (z = normal) as a column-major 3x3 block plus the uniform density 1 / area"""

    def _plane(self, axis, up, area: float, *, transposed: bool = True) -> None:
        m = _frame(axis, up)
        self.setParams(_objToWorld=m.T if transposed else m, _normal=m[:, 2], _prob=1.0 / area)


class FlatTarget(_PlaneShape, Target):
    """
    Rectangular target: in its local frame it lies in the xy plane centred at the origin, `width`
    along x, `length` along y, the normal `direction` along +z, `up` along +y; both sides detect
    and are told apart by the sign of the local normal. Parameters / stage parameters as in the
    reference (target.py:200-317, shader/target.flat.glsl): `width, length, position, direction, up`.
    `missIsInfinite` is this backend's extension (see `targetFlags`).
    """

    name = "Flat Target"

    class TargetParams(Structure):
        _fields_ = _layout(width=c_float, length=c_float, position=vec3, _normal=vec3, _prob=c_float, _objToWorld=mat3)

    def __init__(self, *, width=1.0 * u.cm, length=1.0 * u.cm, position=(0.0, 0.0, 0.0),
                 direction=(0.0, 0.0, 1.0), up=(0.0, 1.0, 0.0), missIsInfinite=False):
        Target.__init__(self, nRNGSamples=2, params={"TargetParams": self.TargetParams}, extra={"direction", "up"})
        self.missIsInfinite, self.direction, self.up = missIsInfinite, direction, up
        self.setParams(width=width, length=length, position=position)

    def _finishParams(self, i):
        Target._finishParams(self, i)
        self._plane(self.direction, self.up, self.width * self.length)


class DiskTarget(_PlaneShape, Target):
    """
    Circular target: in its local frame it lies in the xy plane centred at the origin with the
    `normal` along +z and `up` along +y (None: any frame about the normal); both sides detect. Parameters / stage
    parameters as in the reference (target.py:320-425, shader/target.disk.glsl): `position, radius, normal, up`.
    `missIsInfinite` is this backend's extension (see `targetFlags`).
    """

    name = "Disk Target"

    class TargetParams(Structure):
        _fields_ = _layout(position=vec3, radius=c_float, _normal=vec3, _prob=c_float, _objToWorld=mat3)

    def __init__(self, *, position=(0.0, 0.0, 0.0), radius=1.0 * u.m, normal=(0.0, 0.0, 1.0), up=None,
                 missIsInfinite=False):
        Target.__init__(self, nRNGSamples=2, params={"TargetParams": self.TargetParams}, extra={"normal", "up"})
        self.missIsInfinite, self.normal, self.up = missIsInfinite, normal, up
        self.setParams(position=position, radius=radius)

    def _finishParams(self, i):
        Target._finishParams(self, i)
        self._plane(self.normal, self.up, np.pi * self.getParam("radius") ** 2)


class SphereTargetGuide(TargetGuide):
    """Guides scattered rays into the cone a sphere subtends (samples its far side). Reference target.py:470-525,
    shader/target_guide.sphere.glsl."""

    name = "Sphere Target Guide"

    class TargetGuideParams(Structure):
        _fields_ = _layout(position=vec3, radius=c_float, _prob=c_float)

    def __init__(self, *, position=(0.0, 0.0, 0.0), radius=1.0 * u.m):
        TargetGuide.__init__(self, nRNGSamples=2, params={"TargetGuideParams": self.TargetGuideParams})
        self.setParams(position=position, radius=radius)

    def _finishParams(self, i):
        self.setParam("_prob", 1.0 / (2.0 * np.pi * self.getParam("radius") ** 2))


class FlatTargetGuide(_PlaneShape, TargetGuide):
    """
    Rectangular target guide (`width` orthogonal to `normal` and `up`, `height` along `up`); only
    the face the normal points to is sampled -- from behind it the next event estimate is skipped.
    Parameters / stage parameters as in the reference (target.py:527-636,
    shader/target_guide.flat.glsl): `width, height, position, normal, up`.
    """

    name = "Flat Target Guide"

    class TargetGuideParams(Structure):
        _fields_ = _layout(width=c_float, height=c_float, position=vec3, _normal=vec3, _prob=c_float, _objToWorld=mat3)

    def __init__(self, *, width=1.0 * u.m, height=1.0 * u.m, position=(0.0, 0.0, 0.0),
                 normal=(0.0, 0.0, 1.0), up=(0.0, 1.0, 0.0)):
        TargetGuide.__init__(self, nRNGSamples=2, params={"TargetGuideParams": self.TargetGuideParams}, extra={"normal", "up"})
        self.normal, self.up = normal, up
        self.setParams(width=width, height=height, position=position)

    def _finishParams(self, i):
        TargetGuide._finishParams(self, i)
        self._plane(self.normal, self.up, self.width * self.height)


class DiskTargetGuide(_PlaneShape, TargetGuide):
    """
    Circular target guide; only the face the `normal` points to is sampled. Parameters / stage
    parameters as in the reference (target.py:639-735, shader/target_guide.disk.glsl):
    `position, radius, normal`. The frame matrix is handed to the device UNtransposed, unlike in every other
    plane shape: that is what the reference does (target.py:734-735), kept bug for bug -- it only matters for
    normals with a z component.
    """

    name = "Disk Target Guide"

    class TargetGuideParams(Structure):
        _fields_ = _layout(position=vec3, radius=c_float, _normal=vec3, _prob=c_float, _objToWorld=mat3)

    def __init__(self, *, position=(0.0, 0.0, 0.0), radius=1.0 * u.m, normal=(0.0, 0.0, 1.0)):
        TargetGuide.__init__(self, nRNGSamples=2, params={"TargetGuideParams": self.TargetGuideParams}, extra={"normal"})
        self.normal = normal
        self.setParams(position=position, radius=radius)

    def _finishParams(self, i):
        self._plane(self.normal, None, np.pi * self.getParam("radius") ** 2, transposed=False)


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
