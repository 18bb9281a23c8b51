"""
Pipeline protocol the host components are written against: `PipelineStage`
(ctypes parameter blocks, double buffered for configs 0/1), `SourceCodeMixin`,
`Pipeline`, `PipelineScheduler`, `runPipeline`.

Restates the contract of `hephaistos.pipeline` as documented by the reference
(docs/pipeline/pipeline.md:22-92, docs/pipeline/extend.md:216-346): stages expose
public fields of their blocks as attributes and through get/setParams, names
starting with `_` are private, `extra` names are python-side properties,
`_finishParams(i)` runs right before config `i` is committed, `run(i)` returns
the commands of config `i`, pipeline parameters are addressed as
``stage__param``. Commands here are callables that enqueue work on the HIP
stream instead of Vulkan command buffers; a stage whose work can be in flight
while the host goes on returns `Command` objects (an `enqueue` half that only
queues device work and a `collect` half that waits for it), which is what lets
`PipelineScheduler` keep several batches on the device at once as the reference's
scheduler does (docs/pipeline/pipeline.md:74-92).
"""
from __future__ import annotations

import atexit
import ctypes
import threading
import weakref
from ctypes import Array, Structure
from queue import Queue
from typing import Any, Callable, Iterable

__all__ = ["Command", "DeviceCommand", "PipelineStage", "SourceCodeMixin", "Pipeline", "PipelineScheduler", "runPipeline", "DynamicTask",
           "DynamicTaskScheduler"]

Command = Callable[[], None]

# schedulers whose threads still run: stopped before the interpreter shuts down, so that no thread of ours is inside a device
# call (or about to make one) while the HIP runtime and the interpreter are being taken apart
_liveSchedulers: "weakref.WeakSet" = weakref.WeakSet()


@atexit.register
def _stopSchedulers() -> None:
    for s in list(_liveSchedulers):
        try:
            s.destroy()
        except Exception:
            pass


class DeviceCommand:
    """
    A command in two halves. `enqueue(slot)` queues the device work of one batch and returns without waiting;
    `collect(slot)` (optional) waits for exactly that work and hands its results to the host side of the stage.
    `slot` names the device-side buffers the batch uses: the pipeline config itself when the pipeline runs one
    batch at a time, one of `MAX_SLOTS` when a scheduler keeps more than two batches in flight (the two configs
    then only double-buffer the HOST side: parameter blocks and result arrays). Calling the command runs both
    halves on the config's own slot, which is what `Pipeline.run` / `runPipeline` do.
    """

    __slots__ = ("config", "enqueue", "collect")

    def __init__(self, config: int, enqueue: Callable[[int], None], collect: Callable[[int], None] | None = None) -> None:
        self.config, self.enqueue, self.collect = config, enqueue, collect

    def __call__(self) -> None:
        self.enqueue(self.config)
        if self.collect is not None:
            self.collect(self.config)


def _to_python(value: Any) -> Any:
    if isinstance(value, Structure):
        if isinstance(getattr(type(value), "value", None), property):
            return value.value
        if hasattr(value, "__iter__"):
            return tuple(_to_python(v) for v in value)
        if hasattr(value, "data") and isinstance(value.data, Array):
            return tuple(value.data)
        return value
    if isinstance(value, Array):
        return tuple(_to_python(v) for v in value)
    return value


def _assign(struct: Structure, name: str, ftype, value: Any) -> None:
    if issubclass(ftype, Structure):
        field = getattr(struct, name)
        if isinstance(value, ftype):
            ctypes.memmove(ctypes.addressof(field), ctypes.addressof(value), ctypes.sizeof(ftype))
        elif isinstance(getattr(ftype, "value", None), property):
            field.value = int(value)
        elif hasattr(field, "data") and isinstance(field.data, Array):
            import numpy as np

            flat = np.asarray(value, dtype=np.float32).reshape(-1)
            for k in range(len(field.data)):
                field.data[k] = float(flat[k])
        else:
            for (fname, *_), v in zip(ftype._fields_, value):
                setattr(field, fname, v)
    elif issubclass(ftype, Array):
        setattr(struct, name, ftype(*value))
    else:
        setattr(struct, name, value)


class PipelineStage:
    """Base of everything that takes part in a pipeline."""

    name = "stage"

    def __init__(self, params: dict[str, type[Structure]] = {}, extra: set[str] = set()) -> None:
        object.__setattr__(self, "_blocks", {n: t() for n, t in params.items()})
        object.__setattr__(self, "_configs", [{n: t() for n, t in params.items()} for _ in range(2)])
        object.__setattr__(self, "_extra", set(extra))
        fields: dict[str, tuple[str, Any]] = {}
        for bname, btype in params.items():
            for fname, ftype, *_ in btype._fields_:
                fields[fname] = (bname, ftype)
        object.__setattr__(self, "_fieldmap", fields)

    # attribute access to public fields ------------------------------------
    def __getattr__(self, name: str):
        fields = self.__dict__.get("_fieldmap")
        if fields is not None and name in fields and not name.startswith("_"):
            return self.getParam(name)
        raise AttributeError(f"'{type(self).__name__}' object has no attribute '{name}'")

    def __setattr__(self, name: str, value) -> None:
        fields = self.__dict__.get("_fieldmap")
        if (fields is not None and name in fields and not name.startswith("_")
                and not isinstance(getattr(type(self), name, None), property)):
            self.setParam(name, value)
        else:
            object.__setattr__(self, name, value)

    # parameter api --------------------------------------------------------
    @property
    def fields(self) -> tuple[str, ...]:
        return tuple(self._fieldmap.keys())

    def getParam(self, name: str):
        if name in self._fieldmap:
            block, _ = self._fieldmap[name]
            return _to_python(getattr(self._blocks[block], name))
        if name in self._extra:
            return getattr(self, name)
        raise ValueError(f"No parameter with name {name}")

    def setParam(self, name: str, value) -> None:
        if name in self._fieldmap:
            block, ftype = self._fieldmap[name]
            _assign(self._blocks[block], name, ftype, value)
        elif name in self._extra:
            setattr(self, name, value)
        else:
            raise ValueError(f"No parameter with name {name}")

    def getParams(self) -> dict[str, Any]:
        out = {n: self.getParam(n) for n in self._fieldmap if not n.startswith("_")}
        out.update({n: getattr(self, n) for n in self._extra})
        return out

    def setParams(self, **kwargs) -> None:
        for name, value in kwargs.items():
            if name in self._fieldmap or name in self._extra:
                self.setParam(name, value)

    def _finishParams(self, i: int) -> None:
        """Hook: called before the local state is committed to config `i`."""

    def update(self, i: int) -> None:
        self._finishParams(i)
        for name, block in self._blocks.items():
            ctypes.memmove(ctypes.addressof(self._configs[i][name]), ctypes.addressof(block), ctypes.sizeof(block))

    def _getParamsBlock(self, name: str, i: int) -> Structure:
        """The committed parameter block of config `i` (the reference's UBO tensor)."""
        return self._configs[i][name]

    def _bindParams(self, program, i: int) -> None:
        """Binds this stage's own blocks; the HIP backend reads them at launch."""

    def bindParams(self, program, i: int) -> None:
        self._bindParams(program, i)

    def run(self, i: int) -> list[Command]:
        return []


class SourceCodeMixin(PipelineStage):
    """Components that contribute device code. The HIP backend selects compiled
    kernel variants by component *type*; `sourceCode` is kept for API parity and
    names the device implementation."""

    @property
    def sourceCode(self) -> str:
        return f"// {type(self).__name__}: implemented in libtheia_hip.so (theia_amd/csrc)\n"


class Pipeline:
    """Ordered stages sharing the double-buffered configs."""

    def __init__(self, stages: Iterable) -> None:
        named = []
        used: dict[str, int] = {}
        for entry in stages:
            if isinstance(entry, tuple):
                name, stage = entry
            else:
                name, stage = entry.name, entry
            if name in used:
                used[name] += 1
                name = f"{name}{used[name]}"
            else:
                used[name] = 1
            named.append((name, stage))
        self._stages = named
        self._byName = dict(named)
        self._commands = [[cmd for _, st in named for cmd in st.run(i)] for i in range(2)]

    @property
    def stages(self) -> list[tuple[str, PipelineStage]]:
        return list(self._stages)

    def getParams(self) -> dict[str, Any]:
        return {f"{n}__{p}": v for n, st in self._stages for p, v in st.getParams().items()}

    def setParams(self, **params) -> None:
        grouped: dict[str, dict[str, Any]] = {}
        for key, value in params.items():
            stage, sep, param = key.partition("__")
            if not sep:
                # no stage given: every stage that has a parameter of that name gets it (the
                # reference's tests rely on it: tests/test_light.py:119 {"data": ..., "count": ...})
                hit = False
                for name, st in self._stages:
                    if key in st.fields or key in st._extra:
                        grouped.setdefault(name, {})[key] = value
                        hit = True
                if not hit:
                    raise ValueError(f'No stage has a parameter "{key}"')
                continue
            grouped.setdefault(stage, {})[param] = value
        for stage, kv in grouped.items():
            if stage not in self._byName:
                raise ValueError(f'There is no stage with name "{stage}"')
            self._byName[stage].setParams(**kv)

    def update(self, i: int) -> None:
        for _, st in self._stages:
            st.update(i)

    def run(self, i: int, *, update: bool = True) -> None:
        if update:
            self.update(i)
        for cmd in self._commands[i]:
            cmd()

    # the two halves of run(), for schedulers that keep batches in flight ----
    @property
    def asynchronous(self) -> bool:
        """True if every command of the pipeline is a `DeviceCommand`: batches can then be queued on the device
        ahead of the one being processed, each in a slot of its own"""
        return all(isinstance(c, DeviceCommand) for cmds in self._commands for c in cmds)

    @property
    def maxInFlight(self) -> int:
        """batches of this pipeline that can be on the device at once: the smallest limit any stage states (`maxInFlight`: a tracer with a
        hit queue has one queue per config), 16 if none does"""
        return min([int(getattr(st, "maxInFlight")) for _, st in self._stages if getattr(st, "maxInFlight", None) is not None] + [16])

    def setConcurrency(self, lanes: int, gridDivisor: int | None = None) -> None:
        """tells the stages that launch kernels how many batches will be in flight (`Tracer.setConcurrency`)"""
        for _, st in self._stages:
            fn = getattr(st, "setConcurrency", None)
            if fn is not None:
                fn(lanes, gridDivisor)

    def submit(self, i: int, slot: int | None = None) -> None:
        """queues the commands of config `i` (already updated) on the device; plain callables run whole"""
        slot = i if slot is None else slot
        for cmd in self._commands[i]:
            if isinstance(cmd, DeviceCommand):
                cmd.enqueue(slot)
            else:
                cmd()

    def collect(self, i: int, slot: int | None = None) -> None:
        """waits for what `submit(i, slot)` queued and moves its results into the stages' buffers of config `i`"""
        slot = i if slot is None else slot
        for cmd in self._commands[i]:
            if isinstance(cmd, DeviceCommand) and cmd.collect is not None:
                cmd.collect(slot)


def runPipeline(stages: Iterable, i: int = 0, *, update: bool = True) -> None:
    """Creates a throw-away pipeline from the stages and runs config `i` once."""
    Pipeline([s if isinstance(s, tuple) else (f"stage{k}", s) for k, s in enumerate(stages)]).run(i, update=update)


def _env_int(name: str, default: int) -> int:
    import os

    try:
        return int(os.environ.get(name, default))
    except ValueError:
        return default


class PipelineScheduler:
    """
    Feeds a list of tasks (parameter dicts, or `(pipeline name, dict[, args])` tuples) through a pipeline and
    calls `processFn(config, batch, args)` after each one, in order. As in the reference
    (docs/pipeline/pipeline.md:74-92) three threads work on different batches at the same time:

    * update   applies a task's parameters and commits them to config `batch % 2` (`Pipeline.update`),
    * submit   queues the batch on the device (`Pipeline.submit`: kernel launches, reduction, readback -- all
               asynchronous),
    * process  waits for the batch's results (`Pipeline.collect`) and calls `processFn`.

    A config's host side (parameter blocks, `result(config)` arrays) is reused every second batch: batch k + 2 is
    not updated before batch k's parameters went to the device, and its results are not written before
    `processFn` of batch k returned -- `result(config)` is valid inside `processFn`, as in the reference.

    Batches in flight. A pipeline made of `DeviceCommand`s only (the tracers with a histogram response) keeps up
    to `inFlight` batches on the device, each in a result slot of its own (`THX_MAX_SLOTS`), on `lanes` streams
    and `1 / gridDivisor` of the chip per launch (`thx_tracer_set_concurrency`): the tail of a launch -- the
    life of its last photons, 0.22 ms on config 2 whatever the batch size -- is covered by the other batches
    (config 2, 20 batches of 10^6 photons: 0.42 -> 0.26 ms per batch, tools/overlap_bench.py). Any other
    pipeline runs with the two configs as its slots (one batch on the device while the previous one is
    processed). Defaults: lanes 8, gridDivisor 4, inFlight 16 (THEIA_AMD_LANES / THEIA_AMD_GRID_DIVISOR /
    THEIA_AMD_IN_FLIGHT); slot s runs on lane s % lanes.
    """

    MAX_SLOTS = 16  # THX_MAX_SLOTS of include/theia_hip.h

    def __init__(self, pipeline, *, queueSize: int = 0, processFn: Callable[[int, int, Any], None] | None = None,
                 lanes: int | None = None, gridDivisor: int | None = None, inFlight: int | None = None) -> None:
        self._pipelines = pipeline if isinstance(pipeline, dict) else {None: pipeline}
        self._process = processFn
        lanes = _env_int("THEIA_AMD_LANES", 8) if lanes is None else int(lanes)
        gridDivisor = _env_int("THEIA_AMD_GRID_DIVISOR", 4) if gridDivisor is None else int(gridDivisor)
        self._lanes = max(1, min(lanes, self.MAX_SLOTS))
        self._divisor = max(1, min(gridDivisor, self._lanes))
        deep = self._lanes > 1 and all(p.asynchronous for p in self._pipelines.values())
        limit = min([p.maxInFlight for p in self._pipelines.values()] + [self.MAX_SLOTS])  # (hit queues: one per config, i.e. 2)
        if deep and limit < self._lanes:
            self._lanes = self._divisor = 1  # whole-chip launches, two batches in flight: one runs while the other's queue is read
        # (more slots than lanes: the submit thread then never waits for processFn of the batch whose slot it needs next)
        inFlight = _env_int("THEIA_AMD_IN_FLIGHT", self.MAX_SLOTS) if inFlight is None else int(inFlight)
        self._depth = min(limit, max(self._lanes, min(inFlight, self.MAX_SLOTS))) if deep else 2
        self._deep = deep
        self._sliced = deep and self._lanes > 1
        if self._sliced:
            for p in self._pipelines.values():
                p.setConcurrency(self._lanes, self._divisor)
        self._tasks: Queue = Queue(queueSize)   # schedule() -> update thread
        self._toSubmit: Queue = Queue()          # update -> submit
        self._toProcess: Queue = Queue()         # submit -> process
        self._cv = threading.Condition()
        self._total = 0
        self._submitted = 0   # batches whose parameters went to the device
        self._processed = 0   # batches whose processFn returned: their slot and config are free again
        self._finished = 0    # tasks done with, processed or dropped after an error (what wait() counts)
        self._error: BaseException | None = None
        self._alive = True
        _liveSchedulers.add(self)
        self._threads = [threading.Thread(target=fn, name=f"theia-scheduler-{name}", daemon=True)
                         for name, fn in (("update", self._updateLoop), ("submit", self._submitLoop), ("process", self._processLoop))]
        for t in self._threads:
            t.start()

    tasksFinished = property(lambda self: self._finished)
    totalTasks = property(lambda self: self._total)
    tasksScheduled = property(lambda self: self._total - self._finished)
    batchesInFlight = property(lambda self: self._depth, doc="batches this scheduler keeps on the device at once")

    def schedule(self, tasks: Iterable, *, timeout: float | None = None) -> None:
        for task in tasks:
            with self._cv:
                self._total += 1
            self._tasks.put(task, timeout=timeout)

    def wait(self) -> None:
        with self._cv:
            self._cv.wait_for(lambda: self._finished >= self._total)  # (after an error the queued tasks are dropped: this ends soon)
        if self._error is not None:
            err, self._error = self._error, None
            raise err

    def destroy(self) -> None:
        if self._alive:
            self._alive = False
            self._tasks.put(None)
            for t in self._threads:
                t.join(timeout=5.0)
            if self._sliced:
                for p in self._pipelines.values():
                    try:
                        p.setConcurrency(1, 1)
                    except Exception:
                        pass

    def __del__(self) -> None:
        try:
            self.destroy()
        except Exception:
            pass

    # -- the three threads; a batch is (index, pipeline, config, slot, args) --------------------------------
    def _fail(self, exc: BaseException) -> None:
        with self._cv:
            if self._error is None:
                self._error = exc
            self._cv.notify_all()

    def _done(self, processed: bool = False) -> None:
        with self._cv:
            self._finished += 1
            self._processed += 1 if processed else 0
            self._cv.notify_all()

    def _resolve(self, task):
        name, params, args = None, task, None
        if isinstance(task, tuple):
            name, params = task[0], task[1]
            args = task[2] if len(task) > 2 else None
        if name not in self._pipelines:
            if name is None and len(self._pipelines) == 1:
                name = next(iter(self._pipelines))
            else:
                raise ValueError(f"There is no pipeline with name {name}")
        return self._pipelines[name], params, args

    def _updateLoop(self) -> None:
        batch = 0
        while True:
            task = self._tasks.get()
            if task is None:
                self._toSubmit.put(None)
                return
            if self._error is not None:  # drop what is queued behind a failed batch
                self._done()
                continue
            try:
                pipe, params, args = self._resolve(task)
                cfg = batch % 2
                # the config's parameter blocks are free once batch - 2 went to the device; pipelines whose commands
                # finish on the host (hit queues, tracks) also need its results processed
                with self._cv:
                    need = batch - 1
                    self._cv.wait_for(lambda: (self._submitted if self._deep else self._processed) >= need or self._error is not None)
                if self._error is not None:
                    self._done()
                    continue
                if params:
                    pipe.setParams(**params)
                pipe.update(cfg)
                self._toSubmit.put((batch, pipe, cfg, batch % self._depth if self._deep else cfg, args))
                batch += 1
            except BaseException as exc:
                self._fail(exc)
                self._done()

    def _submitLoop(self) -> None:
        while True:
            item = self._toSubmit.get()
            if item is None:
                self._toProcess.put(None)
                return
            batch, pipe, cfg, slot, _ = item
            try:
                with self._cv:  # the slot's device buffers are free once batch - depth was processed
                    need = batch - self._depth + 1
                    self._cv.wait_for(lambda: self._processed >= need or self._error is not None)
                if self._error is None:
                    pipe.submit(cfg, slot)
            except BaseException as exc:
                self._fail(exc)
            with self._cv:
                self._submitted += 1
                self._cv.notify_all()
            self._toProcess.put(item)

    def _processLoop(self) -> None:
        while True:
            item = self._toProcess.get()
            if item is None:
                return
            batch, pipe, cfg, slot, args = item
            try:
                if self._error is None:
                    pipe.collect(cfg, slot)
                    if self._process is not None:
                        self._process(cfg, batch, args)
            except BaseException as exc:  # surfaced by wait()
                self._fail(exc)
            self._done(processed=True)


class DynamicTask:
    """
    A task that decides, batch by batch, how many more batches of a pipeline it needs
    (hephaistos.pipeline.DynamicTask as theia's `ConvergeHistogramTask` uses it, task.py:22-196):
    `parameters` are applied to the pipeline for its batches, `processBatch(config)` is called
    after every batch and returns the number of batches to issue in addition, `onTaskFinished()`
    once no batch is left. `batchesRemaining` = issued and not yet processed.
    """

    def __init__(self, parameters: dict, pipeline: str | None = None, *, initialBatchCount: int = 1) -> None:
        if initialBatchCount < 1:
            raise ValueError("initialBatchCount must be at least 1")
        self._parameters = dict(parameters)
        self._pipeline = pipeline
        self._initial = initialBatchCount
        self._remaining = 0
        self._finished = False

    @property
    def parameters(self) -> dict:
        return self._parameters

    @property
    def pipeline(self) -> str | None:
        return self._pipeline

    @property
    def initialBatchCount(self) -> int:
        return self._initial

    @property
    def batchesRemaining(self) -> int:
        """batches issued for this task that have not been processed yet"""
        return self._remaining

    @property
    def isFinished(self) -> bool:
        return self._finished

    def processBatch(self, config: int) -> int:  # pragma: no cover - interface
        raise NotImplementedError

    def onTaskFinished(self) -> None:
        pass


class DynamicTaskScheduler:
    """Runs `DynamicTask`s on one or several pipelines (configs alternate 0/1): `scheduleTask(task)`, `waitAll()`, `destroy()`.
    One background thread; the batches a task has asked for are queued on the device ahead of the one being processed -- up to
    `inFlight` of them for a pipeline made of `DeviceCommand`s (slots, lanes and the share of the chip per launch as in
    `PipelineScheduler`), one at a time for any other -- so the device works while `processBatch` runs and the tail of one launch
    is covered by the next. `processBatch(config)` is called in the order the batches were issued, with `result(config)` holding
    that batch."""

    def __init__(self, pipeline, *, queueSize: int = 0, lanes: int | None = None, gridDivisor: int | None = None,
                 inFlight: int | None = None) -> None:
        self._pipelines = pipeline if isinstance(pipeline, dict) else {None: pipeline}
        lanes = _env_int("THEIA_AMD_LANES", 8) if lanes is None else int(lanes)
        gridDivisor = _env_int("THEIA_AMD_GRID_DIVISOR", 4) if gridDivisor is None else int(gridDivisor)
        self._lanes = max(1, min(lanes, PipelineScheduler.MAX_SLOTS))
        self._divisor = max(1, min(gridDivisor, self._lanes))
        self._deep = self._lanes > 1 and all(p.asynchronous for p in self._pipelines.values())
        limit = min([p.maxInFlight for p in self._pipelines.values()] + [PipelineScheduler.MAX_SLOTS])
        if self._deep and limit < self._lanes:
            self._lanes = self._divisor = 1  # (hit queues: one per config -- whole-chip launches, two batches in flight)
        inFlight = _env_int("THEIA_AMD_IN_FLIGHT", PipelineScheduler.MAX_SLOTS) if inFlight is None else int(inFlight)
        self._depth = min(limit, max(self._lanes, min(inFlight, PipelineScheduler.MAX_SLOTS))) if self._deep else 1
        self._sliced = self._deep and self._lanes > 1
        if self._sliced:
            for p in self._pipelines.values():
                p.setConcurrency(self._lanes, self._divisor)
        self._queue: Queue = Queue(queueSize)
        self._cv = threading.Condition()
        self._open = 0  # tasks scheduled and not finished
        self._error: BaseException | None = None
        self._alive = True
        _liveSchedulers.add(self)
        self._thread = threading.Thread(target=self._worker, name="theia-task-scheduler", daemon=True)
        self._thread.start()

    batchesInFlight = property(lambda self: self._depth, doc="batches of a task this scheduler keeps on the device at once")

    def scheduleTask(self, task: DynamicTask) -> None:
        with self._cv:
            self._open += 1
        self._queue.put(task)

    def scheduleTasks(self, tasks: Iterable) -> None:
        for t in tasks:
            self.scheduleTask(t)

    def waitAll(self) -> None:
        with self._cv:
            self._cv.wait_for(lambda: self._open == 0 or self._error is not None)
        if self._error is not None:
            err, self._error = self._error, None
            raise err

    def destroy(self) -> None:
        if self._alive:
            self._alive = False
            self._queue.put(None)
            self._thread.join(timeout=5.0)
            if self._sliced:
                for p in self._pipelines.values():
                    try:
                        p.setConcurrency(1, 1)
                    except Exception:
                        pass

    def __del__(self) -> None:
        try:
            self.destroy()
        except Exception:
            pass

    def _worker(self) -> None:
        batch = 0
        while True:
            task = self._queue.get()
            if task is None:
                return
            flying: list[tuple[int, int]] = []  # (config, slot) of the batches on the device, oldest first
            pipe = None
            try:
                name = task.pipeline
                if name not in self._pipelines:
                    if name is None and len(self._pipelines) == 1:
                        name = next(iter(self._pipelines))
                    else:
                        raise ValueError(f"There is no pipeline with name {name}")
                pipe = self._pipelines[name]
                if task.parameters:
                    pipe.setParams(**task.parameters)
                todo = task.initialBatchCount  # asked for and not yet on the device
                task._remaining = todo
                while todo > 0 or flying:
                    # a config's result array is written when its batch is collected, so two batches of one config may be in
                    # flight; its parameter block is consumed by the launch itself
                    while todo > 0 and len(flying) < self._depth:
                        cfg = batch % 2
                        slot = batch % self._depth if self._deep else cfg
                        pipe.update(cfg)
                        pipe.submit(cfg, slot)
                        flying.append((cfg, slot))
                        batch += 1
                        todo -= 1
                    cfg, slot = flying.pop(0)
                    pipe.collect(cfg, slot)
                    task._remaining -= 1
                    more = int(task.processBatch(cfg) or 0)
                    if more > 0:
                        todo += more
                        task._remaining += more
                task._finished = True
                task.onTaskFinished()
            except BaseException as exc:  # surfaced by waitAll()
                self._error = exc
                for cfg, slot in flying:  # leave no readback behind
                    try:
                        pipe.collect(cfg, slot)
                    except BaseException:
                        pass
            finally:
                with self._cv:
                    self._open -= 1
                    self._cv.notify_all()
