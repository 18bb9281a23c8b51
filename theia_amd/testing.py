"""Models shared by tests and benchmarks (reference: src/theia/testing.py:641-656)."""
from __future__ import annotations

from .material import HenyeyGreensteinPhaseFunction, KokhanovskyOceanWaterPhaseMatrix, MediumModel, WaterBaseModel

__all__ = ["WaterTestModel"]


class WaterTestModel(WaterBaseModel, HenyeyGreensteinPhaseFunction, KokhanovskyOceanWaterPhaseMatrix, MediumModel):
    """Sea water at 5 degC / 1000 dbar / 35 PSU, HG g=0.6, Kokhanovsky phase matrix (Voss fit)."""

    ModelName = "water"

    def __init__(self) -> None:
        WaterBaseModel.__init__(self, 5.0, 1000.0, 35.0)
        HenyeyGreensteinPhaseFunction.__init__(self, 0.6)
        KokhanovskyOceanWaterPhaseMatrix.__init__(self, p90=0.66, theta0=0.25, alpha=4.0, xi=25.6)
