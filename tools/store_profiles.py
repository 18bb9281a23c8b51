#!/usr/bin/env python3
"""Copies what tools/prof_round.sh left under gpurun_out/<tag>/ into profiles/ under the round's names:
   tools/store_profiles.py <tag> <round, e.g. r03>"""
import json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, f"{rnd}_bench_1e7.json"))
shutil.copy(os.path.join(src, "kernel_stats.csv"), os.path.join(dst, f"{rnd}_bench_1e7_kernel_stats.csv"))
note = (f"gpurun {tag} (kernels of round {rnd[1:].lstrip('0')}): tools/prof_traffic.sh (separate FETCH_SIZE / WRITE_SIZE passes over `bench.py --workload W --steps 3`), "
        "2 x FETCH_SIZE (gfx950 correction of MI355X_MICROARCH.md) + WRITE_SIZE, KB per launch of 10^7 photons, production kernel. The writes are the cold-state "
        "stores of the shading stages and the records of walks set aside, written back by the L2.")
out = {}
for w in ("config2", "config3", "config5", "bigmesh"):
    f = os.path.join(src, f"traffic_{w}.json")
    if not os.path.exists(f):
        continue
    d = json.loads(open(f).read().strip().splitlines()[-1])[w]
    out[w] = {"hbm_bytes_per_launch": d["hbm_bytes_per_launch"], "FETCH_SIZE_KB": round(d["FETCH_SIZE_KB"], 1), "WRITE_SIZE_KB": round(d["WRITE_SIZE_KB"], 1), "source": note}
json.dump(out, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
with open(os.path.join(dst, f"{rnd}_pmc_staged.txt"), "w") as fh:
    fh.write("rocprofv3 --pmc (five passes, kernel-trace only) over `python3 bench.py --no-cpu-baseline --no-secondary --steps 3 --warmup 1 [--workload W]`, MI355X,\n"
             f"10^7 photons per launch; per-launch means, tools/prof_pmc_quick.sh (tools/prof_round.sh {tag}). \"staged\" = production kernel, \"staged_debug\" = instrumented.\n"
             "Derived: lane utilisation = SQ_THREAD_CYCLES_VALU / (64 SQ_INSTS_VALU); vector instructions per SIMD and cycle = SQ_INSTS_VALU / (1024 SIMDs x kernel time x 2.4 GHz),\n"
             "to be held against the 1.8-2.95 cycles per instruction a SIMD needs with four wavefronts resident (profiles/r03_valu_rates.txt).\n\n")
    for w in ("config2", "config3", "config5"):
        f = os.path.join(src, f"pmc_{w}.txt")
        if os.path.exists(f):
            fh.write(w + "\n" + "".join(l for l in open(f) if l.strip()) + "\n")
# what the vector units do, per workload: derived from the PMC passes and the kernel times of the bench line
import re
b = json.load(open(os.path.join(src, "bench.json")))
kernel_ms = {"config2": b["roofline"]["kernel_ms"]}
for sec in b.get("secondary", []):
    for w in ("config 3", "config 5"):
        if w in sec["workload"] and "kernel_ms" in sec:
            kernel_ms[w.replace(" ", "")] = sec["kernel_ms"]
issue = {}
CLASSES = {"add_f32": "SQ_INSTS_VALU_ADD_F32", "mul_f32": "SQ_INSTS_VALU_MUL_F32", "fma_f32": "SQ_INSTS_VALU_FMA_F32", "trans_f32": "SQ_INSTS_VALU_TRANS_F32",
           "int32": "SQ_INSTS_VALU_INT32", "int64": "SQ_INSTS_VALU_INT64", "cvt": "SQ_INSTS_VALU_CVT"}
for w in ("config2", "config3", "config5"):
    f = os.path.join(src, f"pmc_{w}.txt")
    if not os.path.exists(f) or w not in kernel_ms:
        continue
    t = open(f).read()
    st = t[t.index("== staged\n"):]
    def g(n, default=None):
        m = re.search(n + r"\s+n=\s*\d+ mean=([0-9.e+]+)", st)
        return float(m.group(1)) if m else default
    valu, thr = g("SQ_INSTS_VALU"), g("SQ_THREAD_CYCLES_VALU")
    classes = {k: g(c, 0.0) for k, c in CLASSES.items()}
    classes["f64"] = g("SQ_INSTS_VALU_ADD_F64", 0.0) + g("SQ_INSTS_VALU_MUL_F64", 0.0) + g("SQ_INSTS_VALU_FMA_F64", 0.0)
    gui = g("GRBM_GUI_ACTIVE")
    issue[w] = {"vector_instructions": valu, "classes": classes, "scalar_instructions": g("SQ_INSTS_SALU"), "branch_instructions": g("SQ_INSTS_BRANCH"),
                "vector_memory_reads": g("SQ_INSTS_VMEM_RD"), "lds_instructions": g("SQ_INSTS_LDS"), "lane_utilisation": thr / (64.0 * valu),
                "gpu_cycles_per_launch": gui, "kernel_ms_profiled": kernel_ms[w],
                "source": f"profiles/{rnd}_pmc_staged.txt (rocprofv3 --pmc, five passes, production kernel, per launch of 10^7 photons: SQ_INSTS_VALU and its "
                          "class counters SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F32 / _INT32 / _INT64 / _CVT / _*_F64, SQ_THREAD_CYCLES_VALU); issue cost per class: "
                          "profiles/r03_valu_rates.txt (four wavefronts per SIMD)"}
json.dump(issue, open(os.path.join(dst, "issue.json"), "w"), indent=1)
print("bench:", round(b["value"], 1), "Mphoton-steps/s, kernel", round(b["roofline"]["kernel_ms"], 3), "ms; traffic", {k: v["hbm_bytes_per_launch"] for k, v in out.items()})
