#!/bin/bash
# The judged profile set of cfg3 (512x2048x512 x 256 H66a channels, K3): bench line, rocprofv3
# kernel stats of the same command, FETCH_SIZE / WRITE_SIZE passes (separate runs, the program
# directly after `--`).   usage: tools/prof_cfg3.sh <round-tag>   (on the GPU box, from the repo root)
set -eo pipefail
tag="${1:-r04}"
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
out="$root/gpurun_out"
prof="$root/profiles"
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
B=(python3 "$root/bench.py" --config cfg3 --steps 4 --warmup 1 --no-cpu-baseline --no-api-level --sustained-seconds 0 --no-other-configs)
"${B[@]}" > "$out/${tag}_cfg3_f64_bench.json" 2> "$out/${tag}_cfg3_f64_bench.err"
echo "bench done"
rocprofv3 --kernel-trace --stats -d "$out/${tag}_cfg3_stats" -o run --output-format csv -- "${B[@]}" \
  > "$out/${tag}_cfg3_f64_bench_under_rocprof.json" 2> "$out/${tag}_cfg3_stats.log"
echo "stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$out/${tag}_cfg3_fetch" -o run --output-format csv -- "${B[@]}" > "$out/${tag}_cfg3_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$out/${tag}_cfg3_write" -o run --output-format csv -- "${B[@]}" > "$out/${tag}_cfg3_write.log" 2>&1
echo "pmc done"
cp "$out/${tag}_cfg3_f64_bench.json" "$prof/${tag}_cfg3_f64_bench.json"
cp "$out/${tag}_cfg3_f64_bench_under_rocprof.json" "$prof/${tag}_cfg3_f64_bench_under_rocprof.json"
st="$(find "$out/${tag}_cfg3_stats" -name '*kernel_stats.csv' | head -1)"
[[ -n "$st" ]] && cp "$st" "$prof/${tag}_cfg3_f64_kernel_stats.csv"
python3 "$root/tools/pmc_summary.py" "${tag}_cfg3_f64" "rrl_scan_kernel" "$out/${tag}_cfg3_fetch" "$out/${tag}_cfg3_write" > "$out/${tag}_cfg3_f64_pmc.log"
cp "$prof/${tag}_cfg3_f64_pmc.json" "$out/"
cp "$prof/${tag}_cfg3_f64_kernel_stats.csv" "$out/" 2>/dev/null || true
echo "all done"
