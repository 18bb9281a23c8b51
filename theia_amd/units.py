"""
Unit scalars. Internal units are metre, nanosecond, nanometre, radian and GeV;
multiplying a number (or tuple) by a unit converts it to internal units,
dividing converts back.

Mirrors the names of the reference's `theia.units` (src/theia/units.py:9-37,
values :197-231; `c` = 0.299792458 m/ns at :211).
"""
from __future__ import annotations

from math import pi as _pi

__all__ = [
    "convert", "km", "m", "cm", "mm", "inch", "ft", "yd", "mi", "per_km", "per_m", "per_cm",
    "per_mm", "c", "s", "ms", "us", "ns", "ps", "um", "nm", "rad", "deg", "eV", "keV", "GeV",
    "TeV", "PeV", "Unit", "InverseUnit",
]


def _map(value, fn):
    if type(value) is tuple:
        return tuple(fn(v) for v in value)
    return fn(value)


class Unit:
    """Conversion factor from the annotated unit to the internal one."""

    # let numpy hand whole arrays to __rmul__/__rtruediv__
    __array_priority__ = 9000

    def __init__(self, dimension: str, scale: float = 1.0) -> None:
        self._dim = dimension
        self._scale = float(scale)

    @property
    def dimension(self) -> str:
        return self._dim

    @property
    def scale(self) -> float:
        return self._scale

    def __rmul__(self, value):
        k = self.scale
        return _map(value, lambda v: v * k)

    def __rtruediv__(self, value):
        k = self.scale
        return _map(value, lambda v: v / k)

    def __repr__(self) -> str:
        return f"Unit({self._dim!r}, {self.scale!r})"


class InverseUnit(Unit):
    """1/unit, e.g. per metre."""

    def __init__(self, unit: Unit) -> None:
        super().__init__(unit.dimension, 1.0 / unit.scale)


class _Compound(Unit):
    def __init__(self, const: float, nom: list[Unit], denom: list[Unit]) -> None:
        k = const
        for n in nom:
            k *= n.scale
        for d in denom:
            k /= d.scale
        super().__init__("compound", k)


km = Unit("length", 1000.0)
m = Unit("length", 1.0)
cm = Unit("length", 0.01)
mm = Unit("length", 0.001)
inch = Unit("length", 0.0254)
ft = Unit("length", 0.3048)
yd = Unit("length", 0.9144)
mi = Unit("length", 1609.344)
per_km = InverseUnit(km)
per_m = InverseUnit(m)
per_cm = InverseUnit(cm)
per_mm = InverseUnit(mm)
s = Unit("time", 1.0e9)
ms = Unit("time", 1.0e6)
us = Unit("time", 1.0e3)
ns = Unit("time", 1.0)
ps = Unit("time", 1.0e-3)
c = _Compound(299792458.0, [m], [s])
um = Unit("wavelength", 1000.0)
nm = Unit("wavelength", 1.0)
rad = Unit("angle", 1.0)
deg = Unit("angle", _pi / 180.0)
eV = Unit("energy", 1e-6)
keV = Unit("energy", 1e-3)
GeV = Unit("energy", 1.0)
TeV = Unit("energy", 1e3)
PeV = Unit("energy", 1e6)


def convert(data, unit: Unit):
    """`data / unit`, spelled out."""
    return data / unit
