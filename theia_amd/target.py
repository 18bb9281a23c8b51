"""
Analytic sphere detector of the volume tracer and sphere guide of the scene
tracer (reference: src/theia/target.py:45-139, :440-525; device side
shader/target.sphere.glsl, target_guide.sphere.glsl).
"""
from __future__ import annotations

from ctypes import Structure, c_float

import numpy as np

from . import units as u
from .hp.glsl import vec3
from .hp.pipeline import SourceCodeMixin

__all__ = ["Target", "SphereTarget", "InnerSphereTarget", "TargetGuide", "SphereTargetGuide"]


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
