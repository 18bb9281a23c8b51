#!/usr/bin/env python3
"""
bench.py -- headline benchmark of the photon-tracing hot path on MI355X.

Metric (BASELINE.json): Mphoton-steps/s at 10^7 photons, plus the achieved
algorithmic HBM GB/s of the tracing kernel against the 8 TB/s roofline.

One "step" of this script = one batch of `--batch` photons (default 10^7, the
size the metric is quoted at) traced by SceneForwardTracer through the C ABI
(one kernel launch + one histogram reduction per step). The workload is
BASELINE config 2: one sphere.stl (detector, r = 5 m) in a scattering medium,
light 6 m away, sphere target guide (MIS), maxPathLength 10, unpolarized
(SURVEY.md 8(d)). Inputs (scene, BVH, medium tables) are resident in HBM before
the timed region. Photon-steps are counted with the EventStatisticCallback
semantics by an instrumented, untimed replay of exactly the timed batches.

The JSON line carries, besides the contract fields:
  roofline                  what binds the kernel on the BASELINE scenes is instruction issue, not HBM (they are cache
                            resident): bound "issue", achieved = vector instructions per second of the dominant kernel
                            (PMC count per launch, profiles/issue.json / live HIP-event kernel time), peak = what 1024 SIMDs
                            issue at the measured four-wavefront rates for that instruction mix (profiles/r03_valu_rates.txt),
                            frac = achieved / peak, lane utilisation beside it; `traffic` = HBM bytes per launch (PMC)
  roofline.contract         the north star's algorithmic-HBM figure: SURVEY.md 8(d) bytes per counted unit x the units the
                            instrumented kernel counted / kernel time against 8 TB/s (`contract_frac`; may exceed 1 on a
                            cache-resident scene) and `frac_layout`, the same counts priced with this build's own records
  secondary                 BASELINE configs 3 and 5 at the same photon count; the 5.2 M-triangle mesh, the one workload here
                            that IS bound by HBM (bound "hbm": PMC bytes per launch / kernel time against 8 TB/s); and the
                            flow of the reference's example 02 end to end: 20 tasks of 2^19 and of 10^6 photons through
                            `PipelineScheduler` with a processFn that copies the curve, by wall clock (N = 1 only)
  cpu_baseline              the C oracle on the host cores: all cores, and one thread

Multi-GPU (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`):
one rank per GPU, every rank traces its own contiguous photon-index range of the
global batch (weak scaling: `--batch` photons per GPU per step), no collective on
the data path, ONE RCCL all-reduce of the per-step time histograms at the end.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (RCCL between the ranks of one node: this pool's driver supports dmabuf IPC only)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")  # before torch / HIP initialise: the scheduler-flow lines keep 8 launches side by side (theia_amd/__init__.py)
ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s measured copy
# bytes per counted unit of the roofline contract
CONTRACT_BYTES = {"nodes": 32.0, "triangles": 48.0, "instances": 112.0, "hits": 132.0, "lut_reads": 8.0}
# Cycles a SIMD needs per vector instruction of a class with four wavefronts resident (tools/valu_rates.hip ->
# profiles/r03_valu_rates.txt), the FASTEST member of each class: v_add_f32 1.76, v_mul / v_fma_f32 1.96, transcendentals 5.26,
# 32-bit integer (v_xor / v_and 1.76; multiplies and three-operand forms need 2.95), 64-bit integer 2.95, conversions 2.74,
# double precision 2.95, everything else (moves, selects, compares, permutes, lane reads) priced like the cheapest instruction.
# The peak built from these is what the vector unit could at most have issued of THIS mix: `frac` is a lower bound.
ISSUE_CYCLES = {"add_f32": 1.76, "mul_f32": 1.96, "fma_f32": 1.96, "trans_f32": 5.26, "int32": 1.76, "int64": 2.95, "cvt": 2.74, "f64": 2.95, "other": 1.76}
N_SIMD = 1024          # 256 CUs x 4
CLOCK_HZ = 2.4e9       # MI355X peak engine clock (MI355X_MICROARCH.md)
COUNTERS = ("queries", "nodes", "triangles", "instances", "hits", "lut_reads", "responses", "steps")  # statistics slots 12..19

WORKLOADS = {
    "config1": "BASELINE config 1: VolumeForwardTracer, homogeneous HG medium (example 01)",
    "config2": "BASELINE config 2: SceneForwardTracer, single sphere.stl (1280 tris) in scattering medium, sphere guide/MIS, maxPathLength 10, unpolarized",
    "config3": "BASELINE config 3: SceneForwardTracer, BK7 cube + torus detector in sea water, polarized",
    "config5": "BASELINE config 5: SceneForwardTracer, suzanne x1000 instances + sphere detector, polarized",
    "bigmesh": "config 2 with the detector sphere subdivided to 5 242 880 triangles (not a BASELINE config: out-of-cache BVH stress)",
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)  # a fresh GPU needs a few launches before its clocks settle
    ap.add_argument("--batch", type=int, default=10_000_000, help="photons per GPU per step")
    ap.add_argument("--workload", default="config2", choices=list(WORKLOADS))
    ap.add_argument("--polarized", type=int, default=-1, help="override the workload's polarization (0/1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the config 3 / config 5 / big mesh / scheduler flow lines")
    ap.add_argument("--no-flow", action="store_true", help="skip the scheduler-flow lines (their quarter-chip launches are launches of the headline's kernel: a "
                                                          "rocprofv3 --stats average over that kernel name would mix them with the whole-chip ones)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="target duration of each CPU-oracle sample")
    return ap.parse_args()


def make_scenario(S, name: str, batch: int, polarized: int, **kw):
    pol = {}
    if polarized >= 0:
        pol["polarized"] = bool(polarized)
    if name == "config1":
        return S.config1_volume(batch, **pol, **kw)
    if name == "config2":
        return S.config2_sphere(batch, **pol, **kw)
    if name == "config3":
        return S.config3_cube_torus(batch, **pol, **kw)
    if name == "bigmesh":  # config 2 with the sphere subdivided to 5.2 M triangles: a BVH beyond every cache
        return S.config2_sphere(batch, subdivide=6, **pol, **kw)
    return S.config5_suzanne_grid(batch, **pol, **kw)


def unit_counts(delta) -> dict:
    return {n: float(delta[12 + i]) for i, n in enumerate(COUNTERS)}


def priced(units: dict, bytes_per: dict, grid: int, n_bins: int, launches: int) -> float:
    """bytes of `launches` launches: counted units x bytes per unit + histogram flush (slab write + reducer read)"""
    return sum(bytes_per[k] * units[k] for k in bytes_per) + launches * grid * n_bins * 4.0 * 2.0


def allreduce(dist, t, op=None):
    """in-place all-reduce; the gloo rehearsal (ranks sharing a GPU) goes through host memory"""
    kw = {} if op is None else {"op": op}
    if t.is_cuda and dist.get_backend() == "gloo":
        c = t.cpu()
        dist.all_reduce(c, **kw)
        t.copy_(c)
    else:
        dist.all_reduce(t, **kw)


class Measurement:
    """W warm-up + K timed batches of one workload on this rank's GPU, and the instrumented replay that
    counts their photon-steps and traversal units"""

    def __init__(self, S, hip, torch, workload: str, polarized: int, B: int, K: int, W: int, rank: int, world: int, dist, dev, stream):
        import numpy as np
        from theia_amd.hp.pipeline import Pipeline

        self.workload, self.K, self.W, self.B = workload, K, W, B
        norm = 1.0 / (B * world)

        def prepared(stats: bool):
            sc = make_scenario(S, workload, B, polarized, stats=stats)
            sc.response.normalization = norm
            sc.tracer.photonOffset = rank * B  # contiguous index range of this rank
            return sc, Pipeline(sc.stages())

        # ---- untimed, instrumented replay of the W + K batches: exact step and unit counts
        sc_i, pipe_i = prepared(True)
        for k in range(W):
            pipe_i.run(k % 2)
        before = sc_i.tracer.program.read_stats().astype(np.int64)
        for k in range(W, W + K):
            pipe_i.run(k % 2)
        self.delta = sc_i.tracer.program.read_stats().astype(np.int64) - before
        events = self.delta[:12]
        self.steps_local = int(self.delta[19])
        # cross-check with the reference's definition: one step = one loop event other than created / maxIter
        assert self.steps_local == int(events[1:9].sum() + events[10]), (self.steps_local, events)
        self.scene_info = sc_i.scene.info if hasattr(sc_i, "scene") else {}
        del sc_i, pipe_i

        # ---- timed region
        sc, pipe = prepared(False)
        tr, prog = sc.tracer, sc.tracer.program
        self.n_bins = n_bins = sc.response.nBins
        curves = torch.zeros((K, n_bins), dtype=torch.float32, device=dev)

        def step(k: int, out_ptr: int) -> None:
            cfg = k % 2
            pipe.update(cfg)  # host side of one batch: RNG auto-advance, parameter commit
            hip.tracer_run(prog.handle, cfg, tr._assemble(cfg), stream)
            hip.tracer_reduce(prog.handle, cfg, norm, out_ptr, stream)

        scratch = torch.zeros((n_bins,), dtype=torch.float32, device=dev)
        for k in range(W):
            step(k, scratch.data_ptr())
        if dist is not None:
            torch.cuda.synchronize()
            allreduce(dist, scratch)  # warm the communicator outside the timed region
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(K):
            step(W + k, curves[k].data_ptr())
        if dist is not None:
            # the tracer works on the library's own (non-blocking) stream when torch's current stream
            # is the default one: finish the reductions into `curves` before RCCL reads them
            torch.cuda.synchronize()
            allreduce(dist, curves)  # the single collective: K x n_bins floats
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        self.elapsed = time.perf_counter() - t0
        # results went into an external buffer the library never reads back: ask it whether a kernel flagged
        # an error (scheduler watchdog), and look at the curves themselves
        hip.tracer_check(prog.handle)
        if not bool(torch.isfinite(curves).all()) or float(curves.sum()) <= 0.0:
            raise RuntimeError(f"{workload}: the timed batches produced an invalid light curve")

        kt = prog.kernel_times(min(K, 64))  # live HIP-event durations of the timed launches
        self.kernel_ms = float(np.mean(kt)) if len(kt) else float("nan")
        self.info = prog.launch_info(B)
        self.kernel = "thx::traceKernel (volume forward)" if workload == "config1" else (
            "thx::sceneStagedKernel" if self.info.get("block") != 256 else "thx::traceKernel (scene forward)")

    def roofline(self, steps_local: int) -> dict:
        """the north star's contract figure (SURVEY.md 8(d) bytes x counted units / kernel time against the HBM peak) and the
        same counts priced with this build's own records"""
        units = unit_counts(self.delta)
        grid, K = self.info["grid"], self.K
        contract = priced(units, CONTRACT_BYTES, grid, self.n_bins, K) / K
        si = self.scene_info
        layout_per = {"nodes": float(si.get("layout_box_bytes", 0)), "triangles": float(si.get("layout_triangle_bytes", 0)),
                      "instances": float(si.get("layout_instance_bytes", 0)), "hits": float(si.get("layout_hit_bytes", 0)), "lut_reads": 8.0}
        layout = priced(units, layout_per, grid, self.n_bins, K) / K
        sec = self.kernel_ms * 1e-3
        return {
            "kernel": self.kernel,
            "kernel_ms": self.kernel_ms,
            "bound": "hbm, algorithmic bytes (SURVEY.md 8(d))",
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "achieved": contract / sec / 1e9,
            "contract_frac": contract / sec / 1e9 / HBM_PEAK_GBS,
            "algorithmic_bytes_per_launch": contract,
            "algorithmic_bytes_per_step": contract * K / max(steps_local, 1),
            "achieved_layout": layout / sec / 1e9,
            "frac_layout": layout / sec / 1e9 / HBM_PEAK_GBS,
            "layout_bytes_per_step": layout * K / max(steps_local, 1),
            "units_per_step": {k: v / max(steps_local, 1) for k, v in units.items() if k != "steps"},
            "unit_bytes": {"contract": CONTRACT_BYTES, "layout": layout_per, "hist_flush_per_launch": grid * self.n_bins * 8.0},
        }


def profile_json(name: str, workload: str):
    """per-workload entry of a committed PMC summary (profiles/traffic.json, profiles/issue.json): collected in separate
    rocprofv3 --pmc passes over this script (tools/prof_round.sh), not in this run"""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name))).get(workload)
    except Exception:
        return None


def issue_roofline(issue, kernel_ms: float) -> dict:
    """Vector-instruction issue of the dominant kernel against what 1024 SIMDs can issue of its instruction mix at the
    measured four-wavefront rates. `issue` = profiles/issue.json entry (PMC counts per launch of 10^7 photons)."""
    if not issue or not issue.get("vector_instructions"):
        return {"bound": "issue", "unit": "G vector instructions/s", "achieved": None, "peak": None, "frac": None}
    valu = float(issue["vector_instructions"])
    classes = {k: float(v) for k, v in (issue.get("classes") or {}).items() if k in ISSUE_CYCLES}
    other = max(valu - sum(classes.values()), 0.0)
    cyc = (sum(n * ISSUE_CYCLES[k] for k, n in classes.items()) + other * ISSUE_CYCLES["other"]) / valu if classes else ISSUE_CYCLES["fma_f32"]
    achieved = valu / (kernel_ms * 1e-3) / 1e9
    peak = N_SIMD * CLOCK_HZ / cyc / 1e9
    lanes = issue.get("lane_utilisation")
    return {
        "bound": "issue",
        "unit": "G vector instructions/s",
        "achieved": achieved,
        "peak": peak,
        "frac": achieved / peak,
        "lane_utilisation": lanes,
        "lane_throughput_frac": achieved / peak * lanes if lanes else None,
        "mix_cycles_per_instruction": cyc,
        "simd_cycles_per_vector_instruction": N_SIMD * CLOCK_HZ * kernel_ms * 1e-3 / valu,
        "instructions_per_launch": {"vector": valu, "vector_by_class": dict(classes, other=other) if classes else None,
                                    "scalar": issue.get("scalar_instructions"), "branch": issue.get("branch_instructions"),
                                    "vector_memory_reads": issue.get("vector_memory_reads"), "lds": issue.get("lds_instructions")},
        "class_cycles": ISSUE_CYCLES,
        "source": issue.get("source"),
    }


def scheduler_flow(S, B: int, N: int, kernel_rate: float, reps: int = 3) -> dict:
    """The flow of the reference's example 02 (examples/02_scene_tracing.ipynb cells 20-24: `tasks = [{}] * 20`, a
    processFn that copies `response.result(config)`) on BASELINE config 2, end to end through `PipelineScheduler`:
    wall clock from schedule() to the end of wait(), best of `reps` after one warm-up; steps from an instrumented
    replay of the same tasks."""
    import numpy as np

    from theia_amd.hp import pipeline as pl

    sc_i = S.config2_sphere(B, stats=True)
    S.run_batches(sc_i, N)
    steps = int(sc_i.tracer.program.read_stats()[19])
    del sc_i
    sc = S.config2_sphere(B)
    curves = []
    sched = pl.PipelineScheduler(pl.Pipeline(sc.stages()), processFn=lambda c, b, a: curves.append(sc.response.result(c).copy()))
    best = float("inf")
    for rep in range(reps + 1):
        curves.clear()
        sc.rng.setParams(offset=0)  # the same photons every time (and the ones the replay counted)
        t0 = time.perf_counter()
        sched.schedule([{}] * N)
        sched.wait()
        dt = time.perf_counter() - t0
        if rep:
            best = min(best, dt)
    prog = sc.tracer.program  # (what the library was asked for: the scheduler's lanes, clamped to what the runtime runs side by side)
    lanes, div, depth = prog.lanes, prog.grid_divisor, sched.batchesInFlight
    sched.destroy()
    if len(curves) != N or not all(np.isfinite(c).all() and c.sum() > 0 for c in curves):
        raise RuntimeError("scheduler flow: invalid light curves")
    return {
        "workload": f"example 02 flow on BASELINE config 2: {N} tasks of {B} photons through PipelineScheduler, processFn copies the curve",
        "photons_per_batch": B, "batches": N, "value": steps / best / 1e6, "unit": "Mphoton-steps/s", "wall_ms": best * 1e3,
        "ms_per_batch": best / N * 1e3, "steps_per_photon": steps / (B * N), "lanes": lanes, "grid_divisor": div, "batches_in_flight": depth,
        "vs_kernel_rate_at_1e7": steps / best / kernel_rate if kernel_rate else None,
        "timing": "wall clock, schedule() to the end of wait(), best of 3 after a warm-up run",
    }


def cpu_baseline(S, _backend, workload: str, polarized: int, seconds: float) -> dict:
    """the C oracle (a port, not the reference's llvmpipe path) on a bounded sample: all cores, then one thread"""
    from oracle_backend import OracleBackend

    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores, 16)  # CPU share of a one-GPU box

    def sample(threads: int) -> dict:
        ob = OracleBackend(threads=threads)
        _backend.install(ob)

        def cpu_run(n: int):
            c = make_scenario(S, workload, n, polarized, stats=True)
            t = time.perf_counter()
            S.run_batches(c, 1)
            dt = time.perf_counter() - t
            return int(c.tracer.program.read_stats()[19]), dt

        cpu_run(2_000)  # page in the library, spin up the OpenMP team
        probe = 20_000 if threads == 1 else 50_000
        _s0, d0 = cpu_run(probe)
        n = int(min(max(seconds * probe / max(d0, 1e-6), probe), 50_000_000))
        s1, d1 = cpu_run(n)
        ob.close()
        return {"value": s1 / d1 / 1e6, "photons": n, "photon_steps": s1, "seconds": d1}

    many, one = sample(cores), sample(1)
    return {
        "value": many["value"],
        "unit": "Mphoton-steps/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{many['photons']} photons of the same workload ({many['photon_steps']} photon-steps, {many['seconds']:.1f} s) on the C oracle with "
                  f"OpenMP on {cores} threads; the reference's Vulkan/llvmpipe path cannot run here (no Vulkan, no hephaistos)",
        "single_thread": {"value": one["value"], "unit": "Mphoton-steps/s", "cores": 1,
                          "sample": f"{one['photons']} photons ({one['photon_steps']} photon-steps, {one['seconds']:.1f} s), one thread"},
    }


def main() -> None:
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with: python -m torch.distributed.run --nnodes=1 "
                     "--nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
        args.gpus = world

    import torch

    from theia_amd import _backend, distributed
    from theia_amd import workloads as S

    # rehearsal on a box with fewer GPUs than ranks (THX_BENCH_BACKEND=gloo): ranks share devices
    ndev = max(torch.cuda.device_count(), 1)
    backend = os.environ.get("THX_BENCH_BACKEND", "nccl")
    if backend == "nccl" and world > ndev:
        sys.exit(f"bench.py: {world} ranks need {world} GPUs for RCCL, found {ndev}")
    local %= ndev
    hip = _backend.HipBackend(local)
    _backend.install(hip)
    torch.cuda.set_device(local)
    dist = distributed.init_from_env(backend) if world > 1 else None
    dev = torch.device("cuda", local)
    stream = torch.cuda.current_stream().cuda_stream

    K, W, B = args.steps, args.warmup, args.batch
    m = Measurement(S, hip, torch, args.workload, args.polarized, B, K, W, rank, world, dist, dev, stream)

    t_max = torch.tensor([m.elapsed], dtype=torch.float64, device=dev)
    total_steps = torch.tensor([m.steps_local], dtype=torch.float64, device=dev)
    if dist is not None:
        allreduce(dist, t_max, dist.ReduceOp.MAX)
        allreduce(dist, total_steps, dist.ReduceOp.SUM)
    elapsed = float(t_max.item())
    steps_all = float(total_steps.item())
    # every rank's own clock and kernel time (stragglers show here; `ms_per_step` is the maximum over ranks)
    per_rank = torch.zeros((world, 2), dtype=torch.float64, device=dev)
    per_rank[rank, 0], per_rank[rank, 1] = m.elapsed / K * 1e3, m.kernel_ms
    if dist is not None:
        allreduce(dist, per_rank)
    per_rank = per_rank.cpu().tolist()
    global_batch = B * world

    traffic = (profile_json("traffic.json", args.workload) or {}).get("hbm_bytes_per_launch")
    contract = m.roofline(m.steps_local)
    roof = issue_roofline(profile_json("issue.json", args.workload), m.kernel_ms)
    roof.update({"kernel": m.kernel, "kernel_ms": m.kernel_ms, "traffic": traffic, "contract": contract,
                 "contract_frac": contract["contract_frac"], "frac_layout": contract["frac_layout"]})
    roof["note"] = ("The BASELINE scenes are cache resident (`traffic` = HBM bytes per launch from the PMC counters, 1 % of the algorithmic "
                    "bytes), so HBM does not bind the kernel; instruction issue of divergent code does. `frac` = vector instructions per second "
                    "(PMC count per launch / live kernel time) over what 1024 SIMDs issue of that instruction mix at the rates measured with "
                    "four wavefronts per SIMD, every class priced at its fastest member: a lower bound of how busy the vector units are; "
                    "x `lane_utilisation` = share of the peak LANE throughput doing photon work. `contract` holds the north star's "
                    "algorithmic-HBM figure (SURVEY.md 8(d) bytes x counted units / kernel time against 8 TB/s): `contract_frac` exceeds 1 "
                    "where the build moves fewer bytes than the contract prices (16-B quantised boxes read from a copy in LDS) -- it is "
                    "not a fraction of anything the hardware does; `frac_layout` prices the same counts with the bytes this build's loads "
                    "ask for. The workload that IS bound by HBM is in `secondary` (5.2 M triangles, bound hbm).")

    result = {
        "metric": "Mphoton-steps/sec at 10^7 photons; achieved HBM GB/s vs roofline",
        "value": steps_all / elapsed / 1e6,
        "unit": "Mphoton-steps/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": elapsed / K * 1e3,
        "ms_per_step_per_rank": [r[0] for r in per_rank],
        "kernel_ms_per_rank": [r[1] for r in per_rank],
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": WORKLOADS[args.workload],
            "photons_per_step_per_gpu": B,
            "photons_total": global_batch * K,
            "photon_steps_total": steps_all,
            "steps_per_photon": steps_all / (global_batch * K),
            "sharding": (f"photon index ranges x{world}, one {'RCCL' if backend == 'nccl' else backend + ' (rehearsal: ranks share GPUs)'} "
                         f"all-reduce of {K}x{m.n_bins} f32") if world > 1 else "single GPU",
            "launch": m.info,
            "scene": m.scene_info,
        },
        "roofline": roof,
    }

    # ---- the other single-GPU BASELINE configs at the same photon count, the HBM-bound mesh and the end-to-end
    # scheduler flow (reported, not the headline)
    if rank == 0 and world == 1 and not args.no_secondary and args.workload == "config2":
        secondary = []
        for name in ("config3", "config5", "bigmesh"):
            k2 = max(2, min(K, 3))
            s = Measurement(S, hip, torch, name, -1, B, k2, 3, 0, 1, None, dev, stream)  # (three warm-up launches: a 6-ms kernel right after an idle gap runs 3 % slow)
            r = s.roofline(s.steps_local)
            entry = {
                "workload": WORKLOADS[name], "photons_per_step": B, "steps": k2, "value": s.steps_local / s.elapsed / 1e6,
                "unit": "Mphoton-steps/s", "ms_per_step": s.elapsed / k2 * 1e3, "steps_per_photon": s.steps_local / (B * k2),
                "kernel_ms": r["kernel_ms"], "bytes_per_step": r["algorithmic_bytes_per_step"], "contract_frac": r["contract_frac"],
                "layout_bytes_per_step": r["layout_bytes_per_step"], "frac_layout": r["frac_layout"], "units_per_step": r["units_per_step"],
                "launch": s.info,
            }
            tr = (profile_json("traffic.json", name) or {}).get("hbm_bytes_per_launch")
            if name == "bigmesh":  # ~0.4 GB of BVH: the walk's gathers come out of HBM
                entry["scene"] = s.scene_info
                entry["roofline"] = {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "traffic": tr,
                                     "achieved": tr / (r["kernel_ms"] * 1e-3) / 1e9 if tr else None,
                                     "frac": tr / (r["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS if tr else None,
                                     "note": "achieved = HBM bytes per launch from the PMC counters (2 x FETCH_SIZE + WRITE_SIZE, separate passes: "
                                             "profiles/traffic.json) / live kernel time; 16-B gathers of 4-wide quantised records (one 64-B line per visit), ~0.4 GB of BVH"}
            else:
                entry["roofline"] = dict(issue_roofline(profile_json("issue.json", name), r["kernel_ms"]), traffic=tr)
            secondary.append(entry)
            del s
        kernel_rate = m.steps_local / (K * m.kernel_ms * 1e-3)  # photon-steps per second of the headline's kernel alone
        for fb, nb in ((1 << 19, 20), (1_000_000, 20), (B, 8)):
            if not args.no_flow:
                secondary.append(scheduler_flow(S, fb, nb, kernel_rate))
        result["secondary"] = secondary

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(S, _backend, args.workload, args.polarized, args.cpu_seconds)
        _backend.install(hip)

    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
