#!/bin/bash
# Counter evidence for K3 (cfg3, rrl_scan_kernel<double,256,true>) and the 32-epoch K1 tile
# (cfg5): kernel-trace stats + SQ counter passes, each pass its own rocprofv3 run with the
# program directly after `--`.  usage: tools/prof_k3_k1.sh <tag> [k1|k3]  (writes gpurun_out/<tag>_*)
set -eo pipefail
tag="${1:-r02}"
only="${2:-both}"
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
out="$root/gpurun_out"
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > "$out/${tag}_counters_available.txt" 2>&1 || true

pass() {   # name, counters..., then the probe command after --
  local name="$1"; shift
  local pmc=()
  while [[ "$1" != "--" ]]; do pmc+=("$1"); shift; done
  shift
  rocprofv3 --kernel-trace --pmc "${pmc[@]}" -d "$out/${tag}_${name}" -o run --output-format csv \
    -- "$@" > "$out/${tag}_${name}.log" 2>&1
  echo "pass $name done"
}

K3=(python3 "$root/tools/k3_probe.py" cfg3 f64)
K1=(python3 "$root/tools/k1_probe.py" cfg5 f64 32)

if [[ $only != k1 ]]; then
rocprofv3 --kernel-trace --stats -d "$out/${tag}_k3_stats" -o run --output-format csv -- "${K3[@]}" > "$out/${tag}_k3_stats.log" 2>&1
echo "k3 stats done"
pass k3_sq1 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -- "${K3[@]}"
pass k3_sq2 SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVES -- "${K3[@]}"
pass k3_grbm GRBM_GUI_ACTIVE -- "${K3[@]}"
fi
[[ $only == k3 ]] && { echo "all passes done"; exit 0; }

export PROBE_NO_EM=1
rocprofv3 --kernel-trace --stats -d "$out/${tag}_k1e32_stats" -o run --output-format csv -- "${K1[@]}" > "$out/${tag}_k1e32_stats.log" 2>&1
echo "k1 stats done"
pass k1e32_sq1 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -- "${K1[@]}"
pass k1e32_grbm GRBM_GUI_ACTIVE -- "${K1[@]}"
echo "all passes done"
