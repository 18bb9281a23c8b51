"""
Run-until-converged driver for light curves (reference: src/theia/task.py:22-196): a dynamic
task keeps issuing batches of a histogram pipeline and averages them until the standard error of
the mean of the curve's integral falls below `atol + rtol * integral`.
"""
from __future__ import annotations

import warnings
from typing import Callable

import numpy as np

from .hp.pipeline import DynamicTask
from .response import HistogramHitResponse

__all__ = ["ConvergeHistogramTask"]


class ConvergeHistogramTask(DynamicTask):
    """
    Issues `initialBatchCount` batches with `params` applied to the pipeline, then
    `extraBatchCount` more at a time until the error estimate drops below the threshold or
    `maxBatchCount` batches were used (`converged` tells which). `result` is the running mean
    of the batches' histograms, `error` the standard error of the mean of their integrals with
    a small-sample correction (sqrt(M2 / (n - 1.5)) / sqrt(n)).
    """

    def __init__(self, response: HistogramHitResponse, params: dict, pipeline: str | None = None, *,
                 initialBatchCount: int = 4, extraBatchCount: int = 2, maxBatchCount: int = 50, atol: float = 0.1,
                 rtol: float = 5e-5, finishedCallback: Callable[["ConvergeHistogramTask"], None] | None = None) -> None:
        super().__init__(params, pipeline, initialBatchCount=initialBatchCount)
        if not response.retrieve:
            raise ValueError("Task requires response to retrieve results!")
        if initialBatchCount < 2:
            raise ValueError("initialBatchCount must be at least 2!")
        if extraBatchCount < 1:
            raise ValueError("extraBatchCount must be at least1!")
        self._response = response
        self._extraCount = extraBatchCount
        self._maxBatchCount = maxBatchCount
        self._rtol, self._atol = rtol, atol
        self._totalBatches = 0
        self._converged = False
        self._callback = finishedCallback
        self._result = np.zeros(response.nBins)
        self._integralMean = 0.0  # running mean of the curve integrals
        self._m2 = 0.0            # Welford: sum of squared deviations

    converged = property(lambda self: self._converged, doc="True if the result has converged")
    atol = property(lambda self: self._atol, doc="absolute error threshold")
    rtol = property(lambda self: self._rtol, doc="relative error threshold")
    batchCountLimit = property(lambda self: self._maxBatchCount, doc="maximum allowed amount of batches")
    response = property(lambda self: self._response, doc="response producing the histograms")
    result = property(lambda self: self._result, doc="mean histogram; check `converged`")
    totalBatches = property(lambda self: self._totalBatches, doc="batches used so far")

    @property
    def error(self) -> float:
        """standard error of the mean of the curve integral (pessimistic for few batches)"""
        n = self._totalBatches
        return float(np.sqrt(self._m2 / (n - 1.5)) / np.sqrt(n))

    def onTaskFinished(self) -> None:
        if self._callback is not None:
            self._callback(self)

    def processBatch(self, config: int) -> int:
        self._totalBatches += 1
        hist = self._response.result(config)
        assert hist is not None
        self._result += (hist - self._result) / self._totalBatches
        integral = float(hist.sum())
        delta = integral - self._integralMean
        self._integralMean += delta / self._totalBatches
        self._m2 += delta * (integral - self._integralMean)
        if self.batchesRemaining > 0:  # decide once everything issued so far is in
            return 0
        if self.error <= self._atol + self._rtol * self._integralMean:
            self._converged = True
            return 0
        n = min(max(self._maxBatchCount - self._totalBatches, 0), self._extraCount)
        if n == 0:
            warnings.warn(f"Failed to converge histogram (error: {self.error:.3e})!\nPipeline: {self.pipeline}\n"
                          f"Params: {self.parameters}")
        return n
