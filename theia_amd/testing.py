"""Models shared by tests and benchmarks (reference: src/theia/testing.py:641-656)."""
from __future__ import annotations

import numpy as np

from . import _backend
from . import _capi as capi
from .hp.pipeline import PipelineStage
from .scene import RectBBox
from .target import _GUIDES, _TARGETS, fillGuideParams, fillTargetParams, guideFlags, targetFlags
from .material import HenyeyGreensteinPhaseFunction, KokhanovskyOceanWaterPhaseMatrix, MediumModel, WaterBaseModel

__all__ = ["WaterTestModel", "TargetSampler", "TargetGuideSampler"]


class WaterTestModel(WaterBaseModel, HenyeyGreensteinPhaseFunction, KokhanovskyOceanWaterPhaseMatrix, MediumModel):
    """Sea water at 5 degC / 1000 dbar / 35 PSU, HG g=0.6, Kokhanovsky phase matrix (Voss fit)."""

    ModelName = "water"

    def __init__(self) -> None:
        WaterBaseModel.__init__(self, 5.0, 1000.0, 35.0)
        HenyeyGreensteinPhaseFunction.__init__(self, 0.6)
        KokhanovskyOceanWaterPhaseMatrix.__init__(self, p90=0.66, theta0=0.25, alpha=4.0, xi=25.6)


def _rngFlags(rng) -> int:
    from .random import rngFlags

    return rngFlags(rng)


def _philoxParams(rng, i: int) -> capi.PhiloxParams:
    from .random import fillRngParams

    P = capi.PhiloxParams()
    fillRngParams(P, rng, i)
    return P


class TargetSampler(PipelineStage):
    """
    Test program sampling a target (src/theia/testing.py:537-640, shader/target.sample.glsl): every
    stream draws an observer inside `sampleBox` and a direction, samples the target from there and
    intersects it along the direction. `getResults(i)` -> dict of arrays `observer, direction,
    samplePos, sampleNrm, sampleProb, sampleValid, sampleError, hitPos, hitNrm, hitProb, hitValid,
    hitError, occluded` (the error fields flag inconsistent object-space data, bit by bit).
    """

    name = "Target Sampler"

    def __init__(self, capacity: int, target, *, rng, sampleBox: RectBBox | None = None) -> None:
        super().__init__()
        if not isinstance(target, _TARGETS):
            raise NotImplementedError(f"{type(target).__name__} is not supported as target by the HIP backend "
                                      f"(supported: {', '.join(c.__name__ for c in _TARGETS)})")
        self._target, self._rng, self._batch = target, rng, capacity
        self._box = sampleBox if sampleBox is not None else RectBBox((-20.0,) * 3, (20.0,) * 3)
        self._result = [None, None]

    batchSize = property(lambda self: self._batch, doc="number of samples to draw per batch")
    target = property(lambda self: self._target)
    rng = property(lambda self: self._rng)

    def collectStages(self) -> list[PipelineStage]:
        return [self.rng, self.target, self]

    def getResults(self, i: int) -> dict:
        d = self._result[i]
        u32, i32 = (lambda a: a.view(np.uint32)), (lambda a: a.view(np.int32))
        return {"observer": d[:, 0:3], "direction": d[:, 3:6], "samplePos": d[:, 6:9], "sampleNrm": d[:, 9:12],
                "sampleProb": d[:, 12], "sampleValid": u32(d[:, 13]), "sampleError": i32(d[:, 14]), "hitPos": d[:, 15:18],
                "hitNrm": d[:, 18:21], "hitProb": d[:, 21], "hitValid": u32(d[:, 22]), "hitError": i32(d[:, 23]),
                "occluded": u32(d[:, 24])}

    def run(self, i: int):
        def launch() -> None:
            T = capi.TargetParams()
            fillTargetParams(T, self._target, i)
            self._result[i] = _backend.get().probe_target(_philoxParams(self._rng, i), targetFlags(self._target) | _rngFlags(self._rng), T,
                                                          self._box.lowerCorner, self._box.upperCorner, self._batch)

        return [launch]


class TargetGuideSampler(PipelineStage):
    """
    Test program sampling a target guide (src/theia/testing.py:439-534,
    shader/target_guide.sample.glsl). `getResults(i)` -> dict of arrays `observer, sampleDir,
    sampleDist, sampleProb, evalDir, evalDist, evalProb`.
    """

    name = "Target Guide Sampler"

    def __init__(self, capacity: int, guide, *, rng, sampleBox: RectBBox | None = None) -> None:
        super().__init__()
        if not isinstance(guide, _GUIDES):
            raise NotImplementedError(f"{type(guide).__name__} is not supported as target guide by the HIP backend "
                                      f"(supported: {', '.join(c.__name__ for c in _GUIDES)})")
        self._guide, self._rng, self._batch = guide, rng, capacity
        self._box = sampleBox if sampleBox is not None else RectBBox((-20.0,) * 3, (20.0,) * 3)
        self._result = [None, None]

    batchSize = property(lambda self: self._batch, doc="number of samples to draw per batch")
    guide = property(lambda self: self._guide)
    rng = property(lambda self: self._rng)

    def collectStages(self) -> list[PipelineStage]:
        return [self.rng, self.guide, self]

    def getResults(self, i: int) -> dict:
        d = self._result[i]
        return {"observer": d[:, 0:3], "sampleDir": d[:, 3:6], "sampleDist": d[:, 6], "sampleProb": d[:, 7],
                "evalDir": d[:, 8:11], "evalDist": d[:, 11], "evalProb": d[:, 12]}

    def run(self, i: int):
        def launch() -> None:
            G = capi.GuideParams()
            fillGuideParams(G, self._guide, i)
            self._result[i] = _backend.get().probe_guide(_philoxParams(self._rng, i), guideFlags(self._guide) | _rngFlags(self._rng), G,
                                                         self._box.lowerCorner, self._box.upperCorner, self._batch)

        return [launch]
