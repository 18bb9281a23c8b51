"""
`PipelineScheduler` as the reference documents it (/root/reference/docs/pipeline/pipeline.md:74-92): update, submission and
processing of different batches overlap on three threads, `processFn(config, batch, args)` runs in task order, `result(config)`
is valid inside it, a config's parameters are not touched while the device may still read them. Host logic only: the
pipelines here are made of recording stand-in stages (and of the real stages on the oracle stand-in backend).
"""
import threading
import time

import numpy as np
import pytest

import scenarios as S
from theia_amd.hp import pipeline as pl


class _Recorder(pl.PipelineStage):
    """a stage whose device side is a python list: enqueue appends, collect reads -- with optional delays that make
    thread interleavings show"""

    name = "rec"

    def __init__(self, asynchronous=True, submit_delay=0.0, collect_delay=0.0):
        super().__init__({}, {"value"})
        self.value = 0
        self.committed = [None, None]
        self.device = {}            # slot -> value a batch was submitted with
        self.results = [None, None]  # config -> value collected
        self.log = []                # (what, config, slot, thread name)
        self.asynchronous = asynchronous
        self.delays = (submit_delay, collect_delay)
        self.concurrency = None

    def setConcurrency(self, lanes, gridDivisor=None):
        self.concurrency = (lanes, gridDivisor)

    def update(self, i):
        super().update(i)
        self.committed[i] = self.value
        self.log.append(("update", i, None, threading.current_thread().name))

    def run(self, i):
        def enqueue(slot=i):
            time.sleep(self.delays[0])
            assert slot not in self.device, f"slot {slot} reused before its batch was collected"
            self.device[slot] = self.committed[i]
            self.log.append(("submit", i, slot, threading.current_thread().name))

        def collect(slot=i):
            time.sleep(self.delays[1])
            self.results[i] = self.device.pop(slot)
            self.log.append(("collect", i, slot, threading.current_thread().name))

        if self.asynchronous:
            return [pl.DeviceCommand(i, enqueue, collect)]
        return [lambda: (enqueue(), collect())]


def _run(stage, n, **kw):
    seen = []
    sched = pl.PipelineScheduler({"rec": pl.Pipeline([stage])}, processFn=lambda c, b, a: seen.append((c, b, a, stage.results[c])), **kw)
    sched.schedule([("rec", {"rec__value": 100 + k}, f"task{k}") if k % 2 else {"rec__value": 100 + k} for k in range(n)])
    sched.wait()
    return sched, seen


@pytest.mark.parametrize("asynchronous", [True, False])
def test_order_configs_and_results(asynchronous):
    stage = _Recorder(asynchronous)
    sched, seen = _run(stage, 37)
    sched.destroy()
    # processFn in task order, config = batch % 2, args handed through, and the value of THAT task in result(config)
    assert [(c, b) for c, b, _, _ in seen] == [(k % 2, k) for k in range(37)]
    assert [a for _, _, a, _ in seen] == [f"task{k}" if k % 2 else None for k in range(37)]
    assert [v for _, _, _, v in seen] == [100 + k for k in range(37)]
    assert sched.tasksFinished == sched.totalTasks == 37 and sched.tasksScheduled == 0
    # an asynchronous pipeline keeps more batches in flight than the two configs, a blocking one exactly the configs
    assert sched.batchesInFlight == (16 if asynchronous else 2)
    assert stage.concurrency == ((1, 1) if asynchronous else None)  # set at construction, reset by destroy()
    slots = {s for what, _, s, _ in stage.log if what == "submit"}
    assert slots == (set(range(16)) if asynchronous else {0, 1})


def test_three_threads_overlap():
    """update / submit / process run on threads of their own, and batches really are in flight together: with 20 ms of
    'device work' behind collect and 16 slots the submit thread is many batches ahead of the process thread"""
    stage = _Recorder(True, collect_delay=0.02)
    sched, seen = _run(stage, 12)
    sched.destroy()
    names = {what: {t for w, _, _, t in stage.log if w == what} for what in ("update", "submit", "collect")}
    assert names["update"] == {"theia-scheduler-update"} and names["submit"] == {"theia-scheduler-submit"}
    assert names["collect"] == {"theia-scheduler-process"}
    first_collect = next(k for k, e in enumerate(stage.log) if e[0] == "collect")
    assert sum(1 for e in stage.log[:first_collect] if e[0] == "submit") >= 2
    ahead, submitted, collected = 0, 0, 0
    for what, *_ in stage.log:
        submitted += what == "submit"
        collected += what == "collect"
        ahead = max(ahead, submitted - collected)
    assert ahead >= 8, ahead


def test_a_config_is_not_updated_while_its_batch_waits_for_the_device():
    """batch k + 2 shares the config of batch k: its update must come after batch k's parameters went to the device
    (asynchronous pipelines) resp. after batch k was processed (blocking ones)"""
    for asynchronous, gate in ((True, "submit"), (False, "collect")):
        stage = _Recorder(asynchronous, submit_delay=0.003)
        sched, _ = _run(stage, 10)
        sched.destroy()
        events = [(w, c) for w, c, _, _ in stage.log]
        for cfg in (0, 1):
            mine = [w for w, c in events if c == cfg and w in ("update", gate)]
            assert mine == ["update", gate] * 5, (asynchronous, cfg, mine)


def test_in_flight_limits():
    stage = _Recorder(True, collect_delay=0.002)
    sched, seen = _run(stage, 20, lanes=2, gridDivisor=2, inFlight=3)
    assert stage.concurrency == (2, 2) and sched.batchesInFlight == 3
    sched.destroy()
    assert {s for w, _, s, _ in stage.log if w == "submit"} == {0, 1, 2} and len(seen) == 20
    stage = _Recorder(True)
    sched, seen = _run(stage, 5, lanes=1)  # one lane: the reference's two configs
    assert sched.batchesInFlight == 2 and stage.concurrency is None
    sched.destroy()


def test_errors_surface_and_the_queue_drains():
    stage = _Recorder(True)

    def boom(config, batch, args):
        if batch == 3:
            raise RuntimeError("boom")

    sched = pl.PipelineScheduler(pl.Pipeline([stage]), processFn=boom)
    sched.schedule([{}] * 40)
    with pytest.raises(RuntimeError, match="boom"):
        sched.wait()
    assert sched.tasksFinished == 40
    sched.schedule([{}] * 3)  # usable again afterwards
    sched.wait()
    sched.destroy()
    with pytest.raises(ValueError, match="no pipeline"):
        s2 = pl.PipelineScheduler({"a": pl.Pipeline([_Recorder()])})
        s2.schedule([("b", {})])
        try:
            s2.wait()
        finally:
            s2.destroy()


def test_pipelined_equals_serial_on_the_real_stages(oracle_backend):
    """the tracer + histogram response as DeviceCommands: scheduler (16 slots), scheduler (two configs) and Pipeline.run give
    the same curves for the same tasks"""
    curves = {}
    for mode, kw in (("deep", {}), ("two", {"lanes": 1})):
        sc = S.config1_volume(3000)
        got = []
        sched = pl.PipelineScheduler(pl.Pipeline(sc.stages()), processFn=lambda c, b, a: got.append(sc.response.result(c).copy()), **kw)
        assert pl.Pipeline(sc.stages()).asynchronous
        sched.schedule([{}] * 5 + [{"lightSource__budget": 3.0e5}])
        sched.wait()
        sched.destroy()
        curves[mode] = np.array(got)
    sc = S.config1_volume(3000)
    pipe = pl.Pipeline(sc.stages())
    serial = []
    for k in range(6):
        if k == 5:
            pipe.setParams(lightSource__budget=3.0e5)
        pipe.run(k % 2)
        serial.append(sc.response.result(k % 2).copy())
    assert np.array_equal(curves["deep"], curves["two"]) and np.array_equal(curves["deep"], np.array(serial))
    assert curves["deep"][5].sum() > 2 * curves["deep"][0].sum()


def test_recording_tracers_run_two_batches_one_queue_per_config(oracle_backend):
    """a hit queue exists once per pipeline config: a recording tracer's pipeline keeps exactly two batches in flight (one runs
    while the other's queue is read), on whole-chip launches; a track buffer exists once per tracer: one batch at a time. The hits
    of every batch are those of a serial loop."""
    from theia_amd.response import HitRecorder

    def hits_of(view):
        n = view.count
        return np.column_stack([np.asarray(view["time"][:n]), np.asarray(view["contrib"][:n])])

    rec = HitRecorder()
    sc = S.config2_sphere(700, response=rec)
    pipe = pl.Pipeline(sc.stages())
    assert pipe.asynchronous and pipe.maxInFlight == 2
    got = []
    sched = pl.PipelineScheduler(pipe, processFn=lambda c, b, a: got.append(hits_of(rec.queue.view(c))))
    assert sched.batchesInFlight == 2
    sched.schedule([{}] * 5)
    sched.wait()
    sched.destroy()
    rec2 = HitRecorder()
    sc2 = S.config2_sphere(700, response=rec2)
    pipe2 = pl.Pipeline(sc2.stages())
    for k in range(5):
        pipe2.run(k % 2)
        want = hits_of(rec2.queue.view(k % 2))
        assert len(want) > 50 and np.array_equal(np.sort(got[k], axis=0), np.sort(want, axis=0)), k
    assert not np.array_equal(np.sort(got[0], axis=0), np.sort(got[1], axis=0))  # different photons per batch
    tracked = pl.Pipeline(S.config2_sphere(500, track=True).stages())
    assert tracked.maxInFlight == 1


class _CountingTask(pl.DynamicTask):
    """asks for `extra` more batches each time all issued ones are in, `rounds` times"""

    def __init__(self, stage, initial, extra, rounds):
        super().__init__({"rec__value": 7}, "rec", initialBatchCount=initial)
        self.stage, self.extra, self.rounds = stage, extra, rounds
        self.seen, self.remaining_at_call, self.finished_calls = [], [], 0

    def processBatch(self, config):
        self.seen.append((config, self.stage.results[config]))
        self.remaining_at_call.append(self.batchesRemaining)
        if self.batchesRemaining == 0 and self.rounds > 0:
            self.rounds -= 1
            return self.extra
        return 0

    def onTaskFinished(self):
        self.finished_calls += 1


@pytest.mark.parametrize("asynchronous", [True, False])
def test_dynamic_task_scheduler_keeps_a_tasks_batches_in_flight(asynchronous):
    """`DynamicTaskScheduler`: the batches a task has asked for go to the device ahead of the one being processed (asynchronous
    pipelines), `processBatch` sees them in issue order with `batchesRemaining` = asked for and not yet processed"""
    stage = _Recorder(asynchronous, collect_delay=0.002)
    sched = pl.DynamicTaskScheduler({"rec": pl.Pipeline([stage])}, inFlight=5)
    assert sched.batchesInFlight == (8 if asynchronous else 1)  # at least one slot per lane
    sched.destroy()
    stage = _Recorder(asynchronous, collect_delay=0.002)
    sched = pl.DynamicTaskScheduler({"rec": pl.Pipeline([stage])}, lanes=3, gridDivisor=2, inFlight=5)
    task = _CountingTask(stage, initial=9, extra=4, rounds=2)
    sched.scheduleTask(task)
    sched.waitAll()
    sched.destroy()
    n = 9 + 4 + 4
    assert task.isFinished and task.finished_calls == 1 and len(task.seen) == n
    assert [c for c, _ in task.seen] == [k % 2 for k in range(n)] and all(v == 7 for _, v in task.seen)
    assert task.remaining_at_call == list(range(8, -1, -1)) + [3, 2, 1, 0] * 2
    ahead, submitted, collected = 0, 0, 0
    for what, *_ in stage.log:
        submitted += what == "submit"
        collected += what == "collect"
        ahead = max(ahead, submitted - collected)
    assert ahead == (5 if asynchronous else 1)
    assert stage.concurrency == ((1, 1) if asynchronous else None)
    # errors of processBatch surface in waitAll
    class Boom(pl.DynamicTask):
        def processBatch(self, config):
            raise RuntimeError("boom")
    sched = pl.DynamicTaskScheduler({"rec": pl.Pipeline([_Recorder(asynchronous)])})
    sched.scheduleTask(Boom({}, "rec", initialBatchCount=3))
    with pytest.raises(RuntimeError, match="boom"):
        sched.waitAll()
    sched.destroy()
