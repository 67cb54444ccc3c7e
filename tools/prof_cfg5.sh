#!/bin/bash
# Round-4 profile set of cfg5 (32 epochs x 64 channels on 512x4096x512), for both moment paths:
#   lds = moments_kernel (LDS atomics) + moments_eval_kernel     (bench.py --config cfg5)
#   lt  = lt_moments_kernel on the launch-time-ordered layout    (bench.py --config cfg5 --lt)
# per path: the bench line, rocprofv3 kernel stats of the same command, FETCH_SIZE / WRITE_SIZE
# passes (separate runs, program directly after `--`) and SQ / GRBM counter passes of the scan
# alone (tools/k1_probe.py).  usage: tools/prof_cfg5.sh <round-tag>   (GPU box, repo root)
set -eo pipefail
tag="${1:-r04}"
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
out="$root/gpurun_out"
prof="$root/profiles"
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
export PROBE_NO_EM=1
cells=$((512*4096*512))
for variant in lds lt; do
  if [[ $variant == lt ]]; then flags=(--lt); kern="lt_moments_kernel"; sfx="_lt"; export PROBE_LT=20
  else flags=(); kern="moments_kernel"; sfx=""; unset PROBE_LT; fi
  B=(python3 "$root/bench.py" --config cfg5 "${flags[@]}" --no-cpu-baseline)
  "${B[@]}" > "$out/${tag}_cfg5_f64${sfx}_bench.json" 2> "$out/${tag}_cfg5_f64${sfx}_bench.err"
  echo "bench $variant done"
  rocprofv3 --kernel-trace --stats -d "$out/${tag}_cfg5${sfx}_stats" -o run --output-format csv -- \
    "${B[@]}" > "$out/${tag}_cfg5_f64${sfx}_bench_under_rocprof.json" 2> "$out/${tag}_cfg5${sfx}_stats.log"
  echo "stats $variant done"
  S=("${B[@]}" --steps 8 --warmup 2 --sustained-seconds 0)
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$out/${tag}_cfg5${sfx}_fetch" -o run --output-format csv -- "${S[@]}" > "$out/${tag}_cfg5${sfx}_fetch.log" 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$out/${tag}_cfg5${sfx}_write" -o run --output-format csv -- "${S[@]}" > "$out/${tag}_cfg5${sfx}_write.log" 2>&1
  echo "pmc $variant done"
  K1=(python3 "$root/tools/k1_probe.py" cfg5 f64 32)
  pass() { local name="$1"; shift; local pmc=(); while [[ "$1" != "--" ]]; do pmc+=("$1"); shift; done; shift
    rocprofv3 --kernel-trace --pmc "${pmc[@]}" -d "$out/${tag}_cfg5${sfx}_${name}" -o run --output-format csv -- "$@" > "$out/${tag}_cfg5${sfx}_${name}.log" 2>&1; echo "pass $name done"; }
  pass sq1 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -- "${K1[@]}"
  pass sq2 SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVES -- "${K1[@]}"
  pass grbm GRBM_GUI_ACTIVE -- "${K1[@]}"
  cp "$out/${tag}_cfg5_f64${sfx}_bench.json" "$prof/${tag}_cfg5_f64${sfx}_bench.json"
  cp "$out/${tag}_cfg5_f64${sfx}_bench_under_rocprof.json" "$prof/${tag}_cfg5_f64${sfx}_bench_under_rocprof.json"
  st="$(find "$out/${tag}_cfg5${sfx}_stats" -name '*kernel_stats.csv' | head -1)"
  [[ -n "$st" ]] && cp "$st" "$prof/${tag}_cfg5_f64${sfx}_kernel_stats.csv"
  python3 "$root/tools/pmc_summary.py" "${tag}_cfg5_f64${sfx}" "$kern" "$out/${tag}_cfg5${sfx}_fetch" "$out/${tag}_cfg5${sfx}_write" \
    > "$out/${tag}_cfg5_f64${sfx}_pmc.log"
  python3 "$root/tools/sq_summary.py" "${tag}_cfg5${sfx}_moments" "$kern" "$cells" \
    "$out/${tag}_cfg5${sfx}_sq1" "$out/${tag}_cfg5${sfx}_sq2" "$out/${tag}_cfg5${sfx}_grbm" > "$out/${tag}_cfg5${sfx}_sq.log" 2>&1 || true
  cp "$prof/${tag}_cfg5_f64${sfx}_pmc.json" "$out/" 2>/dev/null || true
  cp "$prof/${tag}_cfg5_f64${sfx}_kernel_stats.csv" "$out/" 2>/dev/null || true
  cp "$prof/${tag}_cfg5${sfx}_moments_sq.json" "$out/" 2>/dev/null || true
done
echo "all done"
