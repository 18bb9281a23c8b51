#!/usr/bin/env python3
"""Per-stage cycle shares of the staged scene kernel (needs a -DTHX_STAGE_PROFILE build)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from theia_amd import workloads as S
from theia_amd import _backend, _capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
which = sys.argv[2] if len(sys.argv) > 2 else "config2"
make = {"config2": S.config2_sphere, "config3": S.config3_cube_torus, "config5": S.config5_suzanne_grid,
        "mid": lambda batch: S.config2_sphere(batch, subdivide=3), "bigmesh": lambda batch: S.config2_sphere(batch, subdivide=6)}[which]
sc = make(batch=n)
S.run_batches(sc, 2)
be = _backend.get()
out = np.zeros(48, np.uint64)
_capi.check(be._lib.thx_tracer_read_stage_profile(C.c_void_p(sc.tracer.program.handle), out.ctypes.data_as(C.POINTER(C.c_uint64))))
S.run_batches(sc, 1)
ms = sc.tracer.program.last_kernel_ms(0)
_capi.check(be._lib.thx_tracer_read_stage_profile(C.c_void_p(sc.tracer.program.handle), out.ctypes.data_as(C.POINTER(C.c_uint64))))
names = ["BIRTH", "A", "T0", "BM", "M", "TS", "C"]
tot = out[:8].sum()
info = sc.tracer.program.launch_info(n)
print(which, "kernel ms", ms, info, "total cycles/WG", tot / info["grid"])
for i, nm in enumerate(names):
    cyc, rounds, items = int(out[i]), int(out[16 + i]), int(out[32 + i])
    if rounds:
        print(f"{nm:6s} share={cyc / tot:6.1%} rounds/WG={rounds / info['grid']:7.1f} items/round={items / rounds:7.1f} cycles/round={cyc / rounds:9.0f} cycles/item={cyc / max(items,1):8.1f}")

claims = int(out[16:23].sum())
print(f"idle share={int(out[7]) / tot:6.1%} idle cycles/claim={int(out[7]) / claims:.0f} empty polls/claim={int(out[11]) / claims:.2f}")
print(f"per claim: failed looks {int(out[12]) / claims:.0f}  scan+claim {int(out[13]) / claims:.0f}  entries {int(out[14]) / claims:.0f} cycles")
if out[10]:
    r = int(out[16 + 2])
    print(f"T0 detail: waves/round={int(out[10]) / r:.1f} walk cycles/wave={int(out[8]) / int(out[10]):.0f} post cycles/wave={int(out[9]) / int(out[10]):.0f} round={int(out[2]) / r:.0f}")

if out[27]:
    w = int(out[10])
    print(f"walk detail per wave-batch: advance rounds={int(out[27]) / w:.1f} inner trips (lane 0)={int(out[28]) / w:.1f} "
          f"inner cycles={int(out[24]) / w:.0f} triangle-leaf cycles={int(out[25]) / w:.0f} instance/exit cycles={int(out[26]) / w:.0f}")

if out[30]:
    print(f"walk occupancy: lanes per round={int(out[29]) / int(out[30]):.1f} of 64 (wave rounds per batch={int(out[30]) / max(int(out[10]), 1):.1f}), "
          f"lanes per inner trip={int(out[31]) / max(int(out[39]), 1):.1f} (wave trips per batch={int(out[39]) / max(int(out[10]), 1):.1f})")

if out[40] or out[42]:
    rays = int(out[40]) + int(out[42]); w = max(int(out[10]), 1)
    print(f"trips per ray: primary {int(out[41]) / max(int(out[40]), 1):.1f} ({int(out[40]) / rays:.0%} of the rays), shadow {int(out[43]) / max(int(out[42]), 1):.1f}; "
          f"longest ray of a batch {int(out[44]) / w:.1f}; rays with <=4 / 5-16 / 17-32 / >32 trips: "
          f"{int(out[45]) / rays:.0%} / {int(out[46]) / rays:.0%} / {int(out[47]) / rays:.0%} / {(rays - int(out[45]) - int(out[46]) - int(out[47])) / rays:.0%}")

if out[23]:
    g = info["grid"]
    print(f"workgroup run times: mean {int(out[15]) / g / 100.0:.1f} us, longest {int(out[23]) / 100.0:.1f} us "
          f"(+{int(out[23]) * g / max(int(out[15]), 1) - 1.0:.1%})")
