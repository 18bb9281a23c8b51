"""
Light and wavelength sources of the supported path (reference:
src/theia/light.py:58-100 WavelengthSource, :258-352 Const/Uniform, :417-470
LightSource, :1105-1180 SphericalLightSource; device side
shader/lightsource.spherical.glsl, wavelengthsource.{const,uniform}.glsl).
"""
from __future__ import annotations

from ctypes import Structure, c_float

import numpy as np

from . import units as u
from .hp.glsl import vec2, vec3, vec4
from .hp.pipeline import SourceCodeMixin

__all__ = ["WavelengthSource", "ConstWavelengthSource", "UniformWavelengthSource", "LightSource", "SphericalLightSource",
           "PencilLightSource"]


class WavelengthSource(SourceCodeMixin):
    """Base of wavelength samplers; `nRNGSamples` = draws per photon."""

    name = "Wavelength Sampler"

    def __init__(self, *, nRNGSamples: int = 0, params: dict = {}, extra: set = set()) -> None:
        super().__init__(params, extra)
        self._nRNGSamples = nRNGSamples

    @property
    def nRNGSamples(self) -> int:
        return self._nRNGSamples


class ConstWavelengthSource(WavelengthSource):
    """Every photon has the same wavelength (default 600 nm); draws nothing."""

    name = "Const Wavelength Source"

    class WavelengthParams(Structure):
        _fields_ = [("wavelength", c_float)]

    def __init__(self, wavelength: float = 600.0 * u.nm) -> None:
        super().__init__(params={"WavelengthParams": self.WavelengthParams})
        self.setParams(wavelength=wavelength)


class UniformWavelengthSource(WavelengthSource):
    """Wavelengths uniform in `lambdaRange`; one draw per photon. With
    ``normalize=False`` the contribution carries the range width."""

    name = "Uniform Wavelength Source"

    class SourceParams(Structure):
        _fields_ = [("lambdaRange", vec2), ("_contrib", c_float)]

    def __init__(self, *, lambdaRange: tuple[float, float] = (300.0, 700.0), normalize: bool = True) -> None:
        super().__init__(nRNGSamples=1, params={"WavelengthParams": self.SourceParams}, extra={"normalize"})
        self.setParams(lambdaRange=lambdaRange, normalize=normalize)

    @property
    def normalize(self) -> bool:
        return self._normalize

    @normalize.setter
    def normalize(self, value: bool) -> None:
        self._normalize = value

    def _finishParams(self, i: int) -> None:
        lo, hi = self.getParam("lambdaRange")
        width = hi - lo
        contrib = 1.0
        if width != 0.0 and not self.normalize:
            contrib *= abs(width)
        self.setParam("_contrib", contrib)


class LightSource(SourceCodeMixin):
    """Base of light sources."""

    name = "Light Source"

    def __init__(self, *, supportForward: bool, supportBackward: bool, nRNGForward: int = 0, nRNGBackward: int = 0,
                 params: dict = {}, extra: set = set()) -> None:
        super().__init__(params, extra)
        self._forward, self._backward = supportForward, supportBackward
        self._rngForward, self._rngBackward = nRNGForward, nRNGBackward

    @property
    def supportForward(self) -> bool:
        return self._forward

    @property
    def supportBackward(self) -> bool:
        return self._backward

    @property
    def nRNGForward(self) -> int:
        return self._rngForward

    @property
    def nRNGBackward(self) -> int:
        return self._rngBackward


class SphericalLightSource(LightSource):
    """Isotropic, unpolarized point source: `position`, emission window
    `timeRange`, total `budget` shared by the sampled photons. 3 draws (direction
    2, start time 1)."""

    name = "Spherical Light Source"

    class LightParams(Structure):
        _fields_ = [("position", vec3), ("_contribFwd", c_float), ("_contribBwd", c_float), ("timeRange", vec2)]

    def __init__(self, *, position=(0.0, 0.0, 0.0), timeRange=(0.0, 100.0) * u.ns, budget: float = 1.0) -> None:
        super().__init__(supportForward=True, supportBackward=True, nRNGForward=3, nRNGBackward=1,
                         params={"LightParams": SphericalLightSource.LightParams}, extra={"budget"})
        self.budget = budget
        self.setParams(position=position, timeRange=timeRange)

    @property
    def budget(self) -> float:
        return self._budget

    @budget.setter
    def budget(self, value: float) -> None:
        self._budget = value

    def _finishParams(self, i: int) -> None:
        super()._finishParams(i)
        self.setParam("_contribFwd", self.budget)
        # backward mode integrates over the sphere of directions
        self.setParam("_contribBwd", self.budget / (4.0 * np.pi))


class PencilLightSource(LightSource):
    """Pencil beam: every photon starts at `position` along `direction` at a time drawn from
    `timeRange`; `budget` is shared by the sampled photons. Polarized tracers take the beam's
    `stokes` vector and `polarizationRef` (unit, perpendicular to `direction`). 1 draw.
    Reference: light.py:1024-1110, shader/lightsource.pencil.glsl."""

    name = "Pencil Light Source"

    class LightParams(Structure):
        _fields_ = [("position", vec3), ("direction", vec3), ("budget", c_float), ("timeRange", vec2), ("stokes", vec4),
                    ("polarizationRef", vec3)]

    def __init__(self, *, position=(0.0, 0.0, 0.0), direction=(0.0, 0.0, 1.0), timeRange=(0.0, 100.0) * u.ns,
                 budget: float = 1.0, stokes=(1.0, 0.0, 0.0, 0.0), polarizationRef=(0.0, 1.0, 0.0)) -> None:
        super().__init__(supportForward=True, supportBackward=False, nRNGForward=1, nRNGBackward=0,
                         params={"LightParams": PencilLightSource.LightParams})
        self.setParams(position=position, direction=direction, timeRange=timeRange, budget=budget, stokes=stokes,
                       polarizationRef=polarizationRef)
