#!/bin/bash
# A/B of library builds on the GPU box: tools/ab_libs.sh <lib.so> [<lib.so> ...]
# ("default" = rajepy_amd/librjprt.so).  Prints K3 (cfg3) and K1 (cfg4 single epoch, cfg5
# 32-epoch tile, cfg2, cfg4 power-law Gaunt) times per build.
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
for lib in "$@"; do
  if [[ "$lib" == "default" ]]; then unset RJP_LIB RJP_DEBUG; else export RJP_DEBUG=1 RJP_LIB="$root/$lib"; fi
  echo "=== $lib"
  python3 "$root/tools/k3_probe.py" cfg3 f64 || exit 1
  python3 "$root/tools/k1_probe.py" cfg4 f64 1 || exit 1
  PROBE_NO_EM=1 python3 "$root/tools/k1_probe.py" cfg5 f64 32 || exit 1
  python3 "$root/tools/k1_probe.py" cfg2 f64 1 || exit 1
  PROBE_POWERLAW=1 python3 "$root/tools/k1_probe.py" cfg4 f64 1 || exit 1
done
