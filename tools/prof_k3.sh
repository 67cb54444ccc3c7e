#!/bin/bash
# SQ counter passes of K3 on cfg3: tools/prof_k3.sh <tag> [lib.so]
set -eo pipefail
tag="$1"; lib="${2:-}"
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
out="$root/gpurun_out"
if [[ -n "$lib" ]]; then export RJP_DEBUG=1 RJP_LIB="$root/$lib"; fi
export TMPDIR=/tmp
cd /tmp
K3=(python3 "$root/tools/k3_probe.py" cfg3 f64)
pass() { local name="$1"; shift; local pmc=(); while [[ "$1" != "--" ]]; do pmc+=("$1"); shift; done; shift
  rocprofv3 --kernel-trace --pmc "${pmc[@]}" -d "$out/${tag}_${name}" -o run --output-format csv -- "$@" > "$out/${tag}_${name}.log" 2>&1; echo "pass $name done"; }
pass k3_sq1 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -- "${K3[@]}"
pass k3_sq2 SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVES -- "${K3[@]}"
pass k3_grbm GRBM_GUI_ACTIVE -- "${K3[@]}"
