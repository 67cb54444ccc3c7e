#!/bin/bash
# A/B of two builds of librjprt on the wide / compact / tau layouts in ONE box: tools/k1_wide_ab.sh libA.so libB.so
for rep in 1 2; do
for lib in "$@"; do
  for lay in wide compact tau; do
    PROBE_LAYOUT=$lay RJP_DEBUG=1 RJP_LIB=$PWD/rajepy_amd/$lib python tools/k1_probe.py cfg4 f64 1 2>/dev/null
  done
done
done
