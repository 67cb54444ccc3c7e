#!/bin/bash
# The round's judged profile set for the headline configuration (cfg4 f64): for the tau layout
# (the default: a0, ts), the tau layout with the EM map (--em: a0, em0, ts) and the wide layout
# (SURVEY 8(d)'s five fields) -- bench line, rocprofv3 kernel stats of the same command,
# FETCH_SIZE / WRITE_SIZE passes (separate runs, program directly after `--`).
# usage: tools/prof_round.sh <round-tag>     (on the GPU box, from the repo root)
# Writes gpurun_out/<tag>_* and the summaries profiles/<tag>_cfg4_f64<sfx>_{pmc.json,kernel_stats.csv,...}.
set -eo pipefail
tag="${1:-r03}"
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
out="$root/gpurun_out"
prof="$root/profiles"
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
for variant in tau2 tau3 wide5; do
  case $variant in
    # (the kernel of the timed step, by its full template arguments: <T, VEC, ET, MODE, BURSTS,
    # UNIF, LAY, EM> -- the run also launches the other layouts' kernels for its side figures)
    # (round 4: single-epoch scans of the tau layout take the burst factor from an LDS table)
    tau2) flags=(); kern="ff_scan_table_kernel<6, false>" ;;
    tau3) flags=(--em); kern="ff_scan_table_kernel<4, true>" ;;
    wide5) flags=(--layout wide --em); kern="ff_scan_table_wide_kernel" ;;
  esac
  sfx="_${variant}"
  B=(python3 "$root/bench.py" "${flags[@]}")
  if [[ $variant == tau2 ]]; then
    "${B[@]}" > "$out/${tag}_cfg4_f64${sfx}_bench.json" 2> "$out/${tag}_cfg4_f64${sfx}_bench.err"
  else
    "${B[@]}" --no-cpu-baseline --no-other-configs > "$out/${tag}_cfg4_f64${sfx}_bench.json" 2> "$out/${tag}_cfg4_f64${sfx}_bench.err"
  fi
  echo "bench $variant done"
  # (--no-other-configs: the `configs` object of the default line would add cfg2-size launches
  # of the same kernels to the per-kernel averages)
  # (--no-api-level: that leg runs the step WHILE the previous step's cubes travel to the host --
  # a blit kernel of ~10 ms on the same CUs and HBM: its eight scan launches take 15-20 ms each
  # and showed up as the 20 ms MaxNs of round 4's summaries, profiles/r05_outlier_launches.json;
  # they are no measure of the kernel and stay out of the per-kernel averages)
  rocprofv3 --kernel-trace --stats -d "$out/${tag}_cfg4${sfx}_stats" -o run --output-format csv -- \
    "${B[@]}" --no-cpu-baseline --no-other-configs --no-api-level > "$out/${tag}_cfg4_f64${sfx}_bench_under_rocprof.json" 2> "$out/${tag}_cfg4${sfx}_stats.log"
  echo "stats $variant done"
  S=("${B[@]}" --steps 8 --warmup 2 --no-cpu-baseline --no-api-level --sustained-seconds 0 --no-other-configs)
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$out/${tag}_cfg4${sfx}_fetch" -o run --output-format csv -- "${S[@]}" > "$out/${tag}_cfg4${sfx}_fetch.log" 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$out/${tag}_cfg4${sfx}_write" -o run --output-format csv -- "${S[@]}" > "$out/${tag}_cfg4${sfx}_write.log" 2>&1
  echo "pmc $variant done"
  # summaries into profiles/ (tracked)
  cp "$out/${tag}_cfg4_f64${sfx}_bench.json" "$prof/${tag}_cfg4_f64${sfx}_bench.json"
  cp "$out/${tag}_cfg4_f64${sfx}_bench_under_rocprof.json" "$prof/${tag}_cfg4_f64${sfx}_bench_under_rocprof.json"
  st="$(find "$out/${tag}_cfg4${sfx}_stats" -name '*kernel_stats.csv' | head -1)"
  [[ -n "$st" ]] && cp "$st" "$prof/${tag}_cfg4_f64${sfx}_kernel_stats.csv"
  python3 "$root/tools/pmc_summary.py" "${tag}_cfg4_f64${sfx}" "$kern" "$out/${tag}_cfg4${sfx}_fetch" "$out/${tag}_cfg4${sfx}_write" \
    > "$out/${tag}_cfg4_f64${sfx}_pmc.log"
  # (the summary lands in profiles/ on the box: copy it where gpurun merges from)
  cp "$prof/${tag}_cfg4_f64${sfx}_pmc.json" "$out/"
  cp "$prof/${tag}_cfg4_f64${sfx}_kernel_stats.csv" "$out/" 2>/dev/null || true
done
echo "all done"
