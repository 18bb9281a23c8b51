"""
Run-until-converged driver for light curves (API of reference src/theia/task.py:22-196): a
dynamic task keeps issuing batches of a histogram pipeline and averages them until the
standard error of the curve's integral falls below `atol + rtol * integral`.

Host logic only. The statistics live in `_RunningCurve` (Welford update of the integrals, running
mean of the bins); the task adds the batch policy on top.
"""
from __future__ import annotations

import math
import warnings
import numpy as np

from .hp.pipeline import DynamicTask
from .response import HistogramHitResponse

__all__ = ["ConvergeHistogramTask"]


class _RunningCurve:
    """mean light curve over the batches seen so far + spread of the per-batch integrals"""

    def __init__(self, n_bins: int) -> None:
        self.mean = np.zeros(n_bins)
        self.count = 0
        self.integral = 0.0   # mean of the integrals
        self.sq_dev = 0.0     # sum of squared deviations of the integrals from their mean

    def add(self, curve: np.ndarray) -> None:
        self.count += 1
        k = self.count
        self.mean += (curve - self.mean) / k
        total = float(np.sum(curve))
        before = total - self.integral
        self.integral += before / k
        self.sq_dev += before * (total - self.integral)

    def standard_error(self) -> float:
        """s / sqrt(n) with s^2 = sq_dev / (n - 1.5): the small-sample correction of the reference
        (task.py:113-120), pessimistic for a handful of batches"""
        k = self.count
        if k < 2:
            return math.inf
        return math.sqrt(self.sq_dev / (k - 1.5) / k)


class ConvergeHistogramTask(DynamicTask):
    """
    `initialBatchCount` batches with `params` applied to the pipeline first; whenever all issued
    batches are in and the error is still above `atol + rtol * integral`, `extraBatchCount` more,
    up to `maxBatchCount` in total. `converged` says how it ended, `result` is the mean curve,
    `error` the standard error of its integral. `finishedCallback(task)` runs at the end.
    """

    def __init__(self, response, params, pipeline=None, *, atol=0.1, rtol=5e-5, initialBatchCount=4,
                 extraBatchCount=2, maxBatchCount=50, finishedCallback=None):
        # response: HistogramHitResponse, params: dict of stage parameters, pipeline: name or None,
        # finishedCallback: callable taking the task
        for name, least, given in (("initialBatchCount", 2, initialBatchCount), ("extraBatchCount", 1, extraBatchCount)):
            if given < least:
                raise ValueError(f"{name} = {given} is too small (minimum {least})")
        if not response.retrieve:
            raise ValueError("the response does not copy its histogram to the host (retrieve=False)")
        DynamicTask.__init__(self, params, pipeline, initialBatchCount=initialBatchCount)
        self._response = response
        self._policy = {"extra": int(extraBatchCount), "limit": int(maxBatchCount), "atol": float(atol), "rtol": float(rtol)}
        self._notify = finishedCallback
        self._stat = _RunningCurve(response.nBins)
        self._done = False

    # read-only views
    response = property(lambda self: self._response, doc="the histogram response the curves come from")
    atol = property(lambda self: self._policy["atol"], doc="absolute part of the error threshold")
    rtol = property(lambda self: self._policy["rtol"], doc="relative part of the error threshold")
    batchCountLimit = property(lambda self: self._policy["limit"], doc="most batches the task will issue")
    totalBatches = property(lambda self: self._stat.count, doc="batches averaged so far")
    converged = property(lambda self: self._done, doc="True once the error fell below the threshold")
    result = property(lambda self: self._stat.mean, doc="mean light curve")
    error = property(lambda self: self._stat.standard_error(), doc="standard error of the curve's integral")

    def _threshold(self) -> float:
        return self.atol + self.rtol * self._stat.integral

    def processBatch(self, config):
        curve = self._response.result(config)
        if curve is None:
            raise RuntimeError("the response returned no histogram for a finished batch")
        self._stat.add(curve)
        if self.batchesRemaining > 0:
            return 0  # judge only once every issued batch has arrived
        if self.error <= self._threshold():
            self._done = True
            return 0
        room = self.batchCountLimit - self.totalBatches
        more = max(0, min(room, self._policy["extra"]))
        if more == 0:
            warnings.warn(f"light curve did not converge within {self.batchCountLimit} batches: error {self.error:.3e} > "
                          f"{self._threshold():.3e} (pipeline {self.pipeline!r}, parameters {self.parameters})")
        return more

    def onTaskFinished(self) -> None:
        if self._notify is not None:
            self._notify(self)
