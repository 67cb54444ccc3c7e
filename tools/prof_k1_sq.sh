#!/bin/bash
# SQ counter passes of the K1 stage as tools/k1_probe.py runs it (no EM maps, default layout):
#   tools/prof_k1_sq.sh <tag> <cfg> <epochs>     -> gpurun_out/<tag>_{sq1,sq2,grbm}
# each pass its own rocprofv3 run, the program directly after `--`; summarise with
# tools/sq_summary.py <tag> <kernel-substring> <work items> gpurun_out/<tag>_sq1 ...
set -eo pipefail
tag="${1:?tag}"; cfg="${2:-cfg4}"; nep="${3:-1}"
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
out="$root/gpurun_out"
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
export PROBE_NO_EM=1
K1=(python3 "$root/tools/k1_probe.py" "$cfg" f64 "$nep")
pass() {
  local name="$1"; shift
  local pmc=()
  while [[ "$1" != "--" ]]; do pmc+=("$1"); shift; done
  shift
  rocprofv3 --kernel-trace --pmc "${pmc[@]}" -d "$out/${tag}_${name}" -o run --output-format csv \
    -- "$@" > "$out/${tag}_${name}.log" 2>&1
  echo "pass $name done"
}
pass sq1 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -- "${K1[@]}"
pass sq2 SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_WAVES -- "${K1[@]}"
pass grbm GRBM_GUI_ACTIVE -- "${K1[@]}"
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS -- "${K1[@]}"
echo "all passes done"
