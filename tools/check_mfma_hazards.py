#!/usr/bin/env python3
"""Scan every kernel of libmppi_hip.so for the hazard hipcc cannot see when MFMAs live inside inline asm: a vector / LDS / memory
instruction that READS OR WRITES a register of an MFMA's destination before the MFMA's result has landed.

   tools/check_mfma_hazards.py [name-substring]          exit status 1 if anything is found

Rule (CDNA3/4 ISA, "manually inserted wait states", XDL write VGPR -> VALU / VMEM / LDS read or write of the same VGPR):
passes + 2 wait states, passes = 16 for a 32x32 f32 MFMA (v_mfma_f32_32x32x2_f32), 8 for the 32x32x16 bf16 and 16x16 forms,
4 for 4x4 (hipcc's own spacing after its builtin MFMAs, measured on this library: 18 for the f32 32x32x2, 12 for the bf16 32x32x16).
An instruction is one wait state, `s_nop N` is N + 1, an intervening MFMA on another accumulator its own pass count (the matrix
pipe runs one MFMA at a time, in order). NOT a hazard: the next MFMA accumulating into the same
destination (SrcC = vDst, back to back), and MFMAs reading it as SrcC. An MFMA reading a pending destination as A or B IS
one. The scan follows the fall-through path: labels and conditional branches do not end the window, an unconditional branch
does (the window is then simply not checked further: this is a lower bound on what is found, which is what a tripwire needs)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from valu_static_mix import LLVM, code_objects  # noqa: E402

REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), r) for r in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def passes(op):
    """pipeline passes of an MFMA (4 clocks each): how long it occupies the matrix pipe"""
    m = re.search(r"_(\d+)x(\d+)x(\d+)", op)
    mm, _, kk = (int(v) for v in m.groups()) if m else (32, 32, 2)
    if mm == 32:
        return 16 if kk <= 2 else 8   # f32 32x32x2 (and x1): 16; the 8/16-deep narrow-type forms: hipcc spaces its builtins by 12 = 8 + 3 + 1
    if mm == 16:
        return 8 if kk <= 4 and "f32_16x16x4_f32" in op else 4
    return 2


def wait_states_needed(op):
    return passes(op) + 2 if passes(op) == 16 else passes(op) + 3


def scan(name, lines):
    found = []
    pend = []  # [dst regs, remaining wait states, mfma text]
    for ln in lines:
        txt = ln.split("//")[0].strip()
        if not txt:
            continue
        if txt.endswith(":"):  # a label: the fall-through path arrives here with everything still pending
            continue
        if txt.startswith("s_branch") or txt.startswith("s_endpgm") or txt.startswith("s_setpc"):
            pend = []
            continue
        # (a conditional branch falls through when not taken: it counts as the one wait state it is — r03: an inline-asm v_max
        # behind `MFMA; s_cbranch` read a builtin MFMA's result too early and the scan, which used to stop at branches, missed it)
        op, _, rest = txt.partition(" ")
        ops = [o.strip() for o in rest.split(",")]
        step = 1
        if op == "s_nop":
            step = int(ops[0], 0) + 1
        elif op.startswith("v_mfma") or op.startswith("v_smfmac"):
            dst, srcs = regs(ops[0]), ops[1:]
            for p in pend:
                if p[1] > 0:
                    if regs(srcs[0]) & p[0] or regs(srcs[1]) & p[0]:
                        found.append((name, p[2], txt, p[1], "MFMA reads a pending destination as A/B"))
            pend = [p for p in pend if not (p[0] == dst)]  # same accumulator: the chain (D -> C) is interlocked
            for p in pend:
                p[1] -= passes(op)  # an MFMA issued behind another one starts when the pipe is free: the older result has landed by then
            pend = [p for p in pend if p[1] > 0]
            pend.append([dst, wait_states_needed(op), txt])
            continue
        elif op[0] == "v" or op.startswith(("ds_", "global_", "buffer_", "flat_", "scratch_")):
            used = set()
            for o in ops:
                used |= regs(o)
            for p in pend:
                if p[1] > 0 and used & p[0]:
                    found.append((name, p[2], txt, p[1], "touches an MFMA destination %d wait states early" % p[1]))
        for p in pend:
            p[1] -= step
        pend = [p for p in pend if p[1] > 0]
    return found


def main():
    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    so = os.environ.get("MPPI_SO_PATH") or os.path.join(ROOT, "mppi-tf_amd", "libmppi_hip.so")
    found, n_kernels, n_mfma = [], 0, 0
    for co in code_objects(so, os.path.join(ROOT, "build", "co")):
        dis = subprocess.check_output([LLVM + "llvm-objdump", "-d", "--demangle", co], text=True, errors="replace")
        cur, lines = None, []
        for line in dis.split("\n") + ["0 <end>:"]:
            m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
            if m:
                if cur and pat in cur and "v_mfma" in "".join(lines):
                    n_kernels += 1
                    n_mfma += sum("v_mfma" in l for l in lines)
                    found += scan(cur.split("(")[0], lines)
                cur, lines = m.group(1), []
            else:
                lines.append(line)
    print("%d kernels with MFMAs, %d MFMA instructions scanned, %d hazards" % (n_kernels, n_mfma, len(found)))
    for f in found[:40]:
        print("  %s\n      %s\n      -> %s   [%s]" % (f[0], f[1], f[2], f[4]))
    return 1 if found else 0


if __name__ == "__main__":
    sys.exit(main())
