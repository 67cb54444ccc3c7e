#!/bin/bash
# The round's judged profile set for the headline configuration (cfg4 f64), compact and wide
# layouts: bench line, rocprofv3 kernel stats of the same command, FETCH_SIZE / WRITE_SIZE
# passes (separate runs, program directly after `--`).  usage: tools/prof_round.sh <round-tag>
# Writes gpurun_out/<tag>_*; copy the summaries into profiles/.
set -eo pipefail
tag="${1:-r02}"
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
out="$root/gpurun_out"
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
for layout in compact wide; do
  sfx=""; [[ $layout == wide ]] && sfx="_wide5"
  B=(python3 "$root/bench.py" --layout $layout)
  "${B[@]}" > "$out/${tag}_cfg4_f64${sfx}_bench.json" 2> "$out/${tag}_cfg4_f64${sfx}_bench.err"
  echo "bench $layout done"
  rocprofv3 --kernel-trace --stats -d "$out/${tag}_cfg4${sfx}_stats" -o run --output-format csv -- \
    "${B[@]}" --no-cpu-baseline > "$out/${tag}_cfg4_f64${sfx}_bench_under_rocprof.json" 2> "$out/${tag}_cfg4${sfx}_stats.log"
  echo "stats $layout done"
  S=("${B[@]}" --steps 8 --warmup 2 --no-cpu-baseline --no-api-level --sustained-seconds 0)
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$out/${tag}_cfg4${sfx}_fetch" -o run --output-format csv -- "${S[@]}" > "$out/${tag}_cfg4${sfx}_fetch.log" 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$out/${tag}_cfg4${sfx}_write" -o run --output-format csv -- "${S[@]}" > "$out/${tag}_cfg4${sfx}_write.log" 2>&1
  echo "pmc $layout done"
done
echo "all done"
