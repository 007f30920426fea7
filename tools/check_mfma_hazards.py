#!/usr/bin/env python3
"""Stage 1's matrix instructions are inline asm (sig16.hip: hipcc would otherwise rename accumulator tiles between MFMAs), so
hipcc does not know that their results take passes to arrive and pads nothing behind them.  This reads the kernels' assembly
and reports every v_accvgpr_read / _mov / _write that touches the accumulator tile of an MFMA fewer than WAIT wait states
behind it, and every MFMA that accumulates onto a tile fewer than WRITE_WAIT wait states behind a v_accvgpr_write / _mov of it
(straight-line distance; `s_nop N` counts N + 1).  Round 6: a change of the kernel's template arguments made hipcc
move a tile between the drain's MFMA and its wait states - keys of vectors with an odd number of k-tiles were wrong until the
live audit refused them.  No GPU needed (tests/test_round6_host.py runs it):
    python tools/check_mfma_hazards.py            -> exit code 1 and the offending lines when there is one"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WAIT = 11          # wait states a 4-pass MFMA's result needs before v_accvgpr_read / a VALU may touch it (the drain pads 13)
WRITE_WAIT = 3     # ... and a v_accvgpr_write / _mov needs in front of an MFMA that accumulates onto the tile it wrote
UNITS = ("sig16.hip",)


def assembly(unit: str) -> str:
    csrc = os.path.join(ROOT, "lshrs_amd", "csrc")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "unit.s")
        subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                        "-I" + os.path.join(ROOT, "include"), "-I" + csrc, "-S", "--cuda-device-only", os.path.join(csrc, unit),
                        "-o", out], check=True, capture_output=True)
        with open(out) as fh:
            return fh.read()


def regs(operand: str):
    m = re.fullmatch(r"a\[(\d+):(\d+)\]", operand)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"a(\d+)", operand)
    return {int(m.group(1))} if m else set()


def scan(text: str):
    findings, kernel, pending, written = [], None, [], []   # pending: [registers, wait states still owed, line]; written: the same for writes
    for no, raw in enumerate(text.splitlines(), 1):
        line = raw.split(";")[0].strip()
        if not line:
            continue
        if line.endswith(":") and not line.startswith("."):
            kernel, pending, written = line[:-1], [], []
            continue
        if line.startswith(".") or line.endswith(":"):
            continue
        op, _, rest = line.partition(" ")
        ops = [o.strip() for o in rest.split(",")] if rest else []
        if op.startswith("v_accvgpr_"):
            touched = set()
            for o in ops:
                touched |= regs(o)
            for need, owed, at in pending:
                if owed > 0 and touched & need:
                    findings.append((kernel, no, raw.strip(), at, WAIT - owed))
        if op.startswith("v_mfma") and ops:
            used = regs(ops[0]) | regs(ops[-1])
            for need, owed, at in written:
                if owed > 0 and used & need:
                    findings.append((kernel, no, raw.strip(), at, WRITE_WAIT - owed))
        states = 1
        if op == "s_nop" and ops:
            states = int(ops[0], 0) + 1
        pending = [[r, owed - states, at] for r, owed, at in pending if owed - states > 0]
        written = [[r, owed - states, at] for r, owed, at in written if owed - states > 0]
        if op.startswith("v_mfma") and ops:
            pending.append([regs(ops[0]), WAIT, no])
        if op in ("v_accvgpr_write_b32", "v_accvgpr_mov_b32") and ops:
            written.append([regs(ops[0]), WRITE_WAIT, no])
    return findings


def main() -> int:
    bad = []
    for unit in UNITS:
        bad += [(unit,) + f for f in scan(assembly(unit))]
    for unit, kernel, no, text, at, dist in bad:
        name = subprocess.run(["c++filt", kernel], capture_output=True, text=True).stdout.strip()
        print(f"{unit}: {name}: line {no}: `{text}` {dist} wait states behind line {at}, which writes the tile it touches")
    print(f"{len(bad)} hazard(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
