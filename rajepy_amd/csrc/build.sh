#!/bin/bash
# Builds librjprt.so for gfx950 in-tree (rajepy_amd/librjprt.so) through csrc/Makefile (ten
# translation units, compiled in parallel).  hipcc cross-compiles without a GPU.  Usage: build.sh [--report | --debug-switches]
#   --report          prints per-kernel register use
#   --debug-switches  builds rajepy_amd/librjprt_dbg.so with -DRJP_DEBUG_SWITCHES: the only build
#                     that reads the RJP_YSPLIT / RJP_FORCE_VEC1 / RJP_NO_UNIFORM / RJP_NO_TILE32
#                     experiment variables (load it with RJP_DEBUG=1 RJP_LIB=<path>)
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="$here/../librjprt.so"
# -disable-machine-licm: the backend otherwise hoists every FP64 constant (VOP3 takes no
# literal, so each is a register pair) out of the scan loops and then spills them -- K3 carried
# 192-312 B/lane of scratch reloads INSIDE its channel loop that way.  Without the hoisting
# constants are rematerialised where used: K3 -17 % (0 spills in the loop), the 32-epoch K1
# tile 169 -> 164 VGPRs = 3 waves/SIMD instead of 2 (-3 %), the single-epoch K1 unchanged
# (A/B in profiles/r02_build_flags_ab.md).
jobs="$(nproc 2>/dev/null || echo 4)"
if [[ "${1:-}" == "--report" ]]; then
  obj="$here/../../build/rjprt_report"
  make -s -C "$here" -j"$jobs" OUT="$out" OBJDIR="$obj" REPORT=1 EXTRA="-Rpass-analysis=kernel-resource-usage"
  mkdir -p "$here/../../gpurun_out"
  cat "$obj"/*.remarks > "$here/../../gpurun_out/resource_usage.txt"
  python3 - "$here/../../gpurun_out/resource_usage.txt" <<'PY'
import re, sys, subprocess
txt = open(sys.argv[1]).read()
rows = []
cur = None
for line in txt.splitlines():
    m = re.search(r"remark: +(Function Name|VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|SGPRs Spill|VGPRs Spill): (.*?) \[-R", line)
    if not m: continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": v}; rows.append(cur)
    elif cur is not None:
        cur[k.split(" [")[0]] = v
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
print("%-6s %-6s %-8s %-4s %-6s %s" % ("VGPR", "SGPR", "scratch", "occ", "LDS", "kernel"))
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n).replace("void rjp::", "")
    print("%-6s %-6s %-8s %-4s %-6s %s" % (r.get("VGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize"), r.get("Occupancy"), r.get("LDS Size"), n))
PY
  # the hand-issued scalar table loads of the uniform-epoch tiles: nothing may read their
  # destination SGPRs before the s_waitcnt that covers them (tools/check_sgpr_tables.py)
  python3 "$here/../../tools/check_sgpr_tables.py"
elif [[ "${1:-}" == "--debug-switches" ]]; then
  out="$here/../librjprt_dbg.so"
  make -s -C "$here" -j"$jobs" OUT="$out" OBJDIR="$here/../../build/rjprt_dbg" EXTRA="-DRJP_DEBUG_SWITCHES ${RJP_EXTRA:-}"
else
  make -s -C "$here" -j"$jobs" OUT="$out" EXTRA="${RJP_EXTRA:-}"
fi
echo "built $out"
