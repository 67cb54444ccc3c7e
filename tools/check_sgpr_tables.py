#!/usr/bin/env python3
"""Build-time check of the hand-issued scalar table loads of the uniform-epoch tiles.

`chi_batch_uniform` (rajepy_amd/csrc/rjp_device.h) requests a burst's step table with an inline-asm
`s_load_dwordx16` whose outputs the compiler regards as valid at once, while the data only
arrives at the later `s_waitcnt lgkmcnt(0)` statement.  That is right as long as nothing READS
the destination registers in between -- a copy or a spill (`v_writelane_b32`, `s_mov_b32/b64`)
inserted there by the register allocator would capture stale values without any build-time
signal.  This script compiles the K1 slices to ISA and verifies, for every
`s_load_dwordx16` of every kernel, that no instruction up to the next `s_waitcnt` that waits for
lgkmcnt(0) names one of its destination SGPRs as a source.

    python tools/check_sgpr_tables.py          # exit status 1 on a violation
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "rajepy_amd", "csrc", "ff_scan_inst.hip")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-disable-machine-licm",
         "-S", "--cuda-device-only"]


def sregs(tok):
    """SGPR numbers named by an operand token (s5, s[4:7], ...)."""
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"s(\d+)", tok)
    return {int(m.group(1))} if m else set()


def check(asm_text, where):
    bad, nloads = [], 0
    kernel = "?"
    lines = asm_text.split("\n")
    for i, line in enumerate(lines):
        m = re.match(r"^(_ZN3rjp\w+):", line)
        if m:
            kernel = m.group(1)
        ins = line.strip()
        if not ins.startswith("s_load_dwordx16"):
            continue
        nloads += 1
        dst = sregs(ins.split()[1].rstrip(","))
        for j in range(i + 1, min(i + 4000, len(lines))):
            nxt = lines[j].strip()
            if not nxt or nxt.startswith((";", ".", "//")):
                continue
            if nxt.endswith(":"):
                continue
            if nxt.startswith("s_waitcnt") and "lgkmcnt(0)" in nxt:
                break
            if nxt.startswith("s_endpgm"):
                bad.append((where, kernel, i + 1, "no s_waitcnt lgkmcnt(0) after the load"))
                break
            toks = [t.rstrip(",") for t in nxt.split()[1:]]
            # sources = every operand but the first (the destination), plus the first for
            # instructions without a destination (stores, writelane's value operand is a source)
            srcs = toks[1:] if len(toks) > 1 else toks
            used = set()
            for t in srcs:
                used |= sregs(t)
            if used & dst and not nxt.startswith("s_load_dwordx16"):
                bad.append((where, kernel, j + 1, nxt))
                break
    return bad, nloads


def main():
    bad, total = [], 0
    with tempfile.TemporaryDirectory() as tmp:
        for inst in (0, 1, 2):              # the f64 tau and compact slices hold the long tiles
            out = os.path.join(tmp, "inst%d.s" % inst)
            subprocess.run(["hipcc"] + FLAGS + ["-DRJP_INST=%d" % inst, "-o", out, SRC], check=True,
                           stderr=subprocess.DEVNULL)
            b, n = check(open(out).read(), "RJP_INST=%d" % inst)
            bad += b
            total += n
    if bad:
        for b in bad:
            print("VIOLATION %s %s line %d: %s" % b)
        sys.exit(1)
    print("%d hand-issued s_load_dwordx16 table loads: no reader before their s_waitcnt lgkmcnt(0)"
          % total)


if __name__ == "__main__":
    main()
