#!/bin/bash
# Counter evidence for the launch-time moment path (cfg5: 32 epochs of 512x4096x512):
# kernel-trace stats + SQ / LDS counter passes of moments_kernel, each pass its own rocprofv3
# run with the program directly after `--`.  usage: tools/prof_moments.sh <tag>
# (writes gpurun_out/<tag>_mom_*; summarise with tools/sq_summary.py)
set -eo pipefail
tag="${1:-r03}"
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
out="$root/gpurun_out"
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
export PROBE_NO_EM=1
K1=(python3 "$root/tools/k1_probe.py" cfg5 f64 32)

pass() {   # name, counters..., then the probe command after --
  local name="$1"; shift
  local pmc=()
  while [[ "$1" != "--" ]]; do pmc+=("$1"); shift; done
  shift
  rocprofv3 --kernel-trace --pmc "${pmc[@]}" -d "$out/${tag}_mom_${name}" -o run --output-format csv \
    -- "$@" > "$out/${tag}_mom_${name}.log" 2>&1
  echo "pass $name done"
}

rocprofv3 --kernel-trace --stats -d "$out/${tag}_mom_stats" -o run --output-format csv -- "${K1[@]}" > "$out/${tag}_mom_stats.log" 2>&1
echo "stats done"
pass sq1 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -- "${K1[@]}"
pass sq2 SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES -- "${K1[@]}"
pass grbm GRBM_GUI_ACTIVE -- "${K1[@]}"
echo "all passes done"
