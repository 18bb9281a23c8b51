#!/usr/bin/env python3
"""
Static check of the library's device code for ONE miscompilation pattern (found in round 4):

    .LBB7_223:  v_cmp_lt_u32  vcc, 8, v10
                s_and_saveexec_b64 s[18:19], vcc      ; EXEC = lanes with more than eight saved stack entries
                s_cbranch_execz .LBB7_251             ; nobody: skip -- with EXEC = 0
                ...
    .LBB7_251:  scratch_store_dword off, v67, off offset:160   ; 4-byte Folded Spill   <-- under the NARROW (here: empty) mask
                s_or_b64 exec, exec, s[18:19]                  ; the region's lanes come back only here

hipcc 7.2 placed a register spill at the top of a control-flow join IN FRONT OF the instruction that gives the region's
lanes back (the end-of-region `s_or_b64 exec, exec, sN`): the store runs for the lanes of the inner branch only -- for none
at all if the branch was skipped through `s_cbranch_execz` -- and every later reload hands the other lanes whatever the
scratch slot held before. That is what made single instrumented kernel variants of round 3 trace photons differently, trip
the scheduler watchdog or fault with a wild address while their siblings stayed exact: v67 above is the LDS address of a
photon's state. A vector instruction can never legitimately sit between a branch target and the exec restore that opens the
block (the restore is the first thing a join block does), so the pattern is checked for directly:

    for every `s_or_b64 exec, exec, s[..]`: walking back from it, a scratch access (spill store / reload) must not be reached
    before a branch, another write of EXEC or the start of the kernel -- if a branch target lies in between.

    python tools/exec_restore_check.py [libtheia_hip.so]      exit code 1 and a listing if the pattern occurs
"""
from __future__ import annotations

import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
_INSN = re.compile(r"^\t(\S+)(?:\s+(.*?))?\s*//\s*([0-9A-Fa-f]+):")
_TARGET = re.compile(r"<[^>]*\+0x([0-9a-fA-F]+)>\s*$")
_HEAD = re.compile(r"^([0-9a-fA-F]+) <(\S+)>:")


def code_objects(so: str, tmp: str) -> list[str]:
    """the gfx950 code objects embedded in the shared library's .hip_fatbin section"""
    fat = os.path.join(tmp, "fatbin")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", so, os.path.join(tmp, "copy.so")])
    data = open(fat, "rb").read()
    magic, out, pos = b"__CLANG_OFFLOAD_BUNDLE__", [], 0
    while (i := data.find(magic, pos)) >= 0:
        entries = struct.unpack_from("<Q", data, i + 24)[0]
        off = i + 32
        for _ in range(entries):
            o, size, tlen = struct.unpack_from("<QQQ", data, off)
            triple = data[off + 24:off + 24 + tlen].decode()
            off += 24 + tlen
            if "gfx950" in triple and size:
                path = os.path.join(tmp, f"co{len(out)}.elf")
                open(path, "wb").write(data[i + o:i + o + size])
                out.append(path)
        pos = i + len(magic)
    return out


def _is_vector(op: str) -> bool:
    # Spill stores and reloads are scratch accesses. (Vector ALU instructions and other memory accesses do sit in front of such a
    # restore legitimately: a flow block copies values for the lanes that came ITS way before the others rejoin. A scratch access
    # there is legitimate only for a stack object of the program's own; the kernels of this library keep none in such a place.)
    return op.startswith("scratch_")


def _is_spill_slot(op: str, args: str) -> bool:
    """spill slots are addressed absolutely (`scratch_store_dword off, v67, off offset:160`); a stack array of the program's own that
    is indexed at run time -- the per-stage counters of the instrumented variants, incremented inside short branches -- by a register"""
    ops = [a.strip().split(" ")[0] for a in (args or "").split(",")]
    return (ops[0] == "off" and ops[-1] == "off") if op.startswith("scratch_store") else (len(ops) >= 3 and ops[1] == "off" and ops[2] == "off")


def _writes_exec(op: str, args: str) -> bool:
    first = (args or "").split(",")[0].strip()
    return first in ("exec", "exec_lo", "exec_hi") or "saveexec" in op or op.startswith("v_cmpx")


def check(elf: str) -> list[str]:
    """violations found in one code object, as readable excerpts"""
    return check_text(subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", elf]).decode())


def check_text(text: str) -> list[str]:
    """... in the text of `llvm-objdump -d --no-show-raw-insn`"""
    findings = []
    kernel, base, insns = None, 0, []  # insns: (address, op, args)

    def scan() -> None:
        if not insns:
            return
        targets = set()
        for addr, op, args, line in insns:
            if op.startswith(("s_cbranch", "s_branch")):
                m = _TARGET.search(line)
                if m:
                    targets.add(base + int(m.group(1), 16))
        index = {a: k for k, (a, _, _, _) in enumerate(insns)}
        for k, (addr, op, args, line) in enumerate(insns):
            if not (op == "s_or_b64" and args.replace(" ", "").startswith("exec,exec,")):
                continue
            crossed_vector = []
            j = k - 1
            while j >= 0:
                a, o, g, _ = insns[j]
                if o.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")) or _writes_exec(o, g):
                    break
                if _is_vector(o) and _is_spill_slot(o, g):
                    crossed_vector.append((a, o, g))
                if a in targets:  # insns[j] is the first instruction of a block other blocks jump to
                    if crossed_vector:
                        lo, hi = max(0, j - 1), min(len(insns), k + 2)
                        findings.append(f"{kernel}: scratch access(es) between the branch target at 0x{a:x} and the exec restore at 0x{addr:x}:\n" +
                                        "\n".join(f"      0x{x[0]:x}: {x[1]} {x[2]}" for x in insns[lo:hi]))
                    break
                j -= 1

    for line in text.split("\n"):
        h = _HEAD.match(line)
        if h:
            scan()
            kernel, base, insns = h.group(2), int(h.group(1), 16), []
            continue
        m = _INSN.match(line)
        if m:
            insns.append((int(m.group(3), 16), m.group(1), (m.group(2) or "").strip(), line))
    scan()
    return findings


def check_library(so: str) -> list[str]:
    with tempfile.TemporaryDirectory() as tmp:
        found = []
        for co in code_objects(so, tmp):
            found += check(co)
        return found


if __name__ == "__main__":
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "theia_amd", "lib", "libtheia_hip.so")
    bad = check_library(lib)
    for b in bad:
        print(b)
    print(f"{lib}: {len(bad)} spill(s) / reload(s) in front of the exec restore of a control-flow join")
    sys.exit(1 if bad else 0)
