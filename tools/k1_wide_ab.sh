#!/bin/bash
# A/B of two builds of librjprt in ONE box on the headline scan (tau layout, no EM: 2 fields),
# the same with the EM map (3 fields) and the wide layout: tools/k1_wide_ab.sh libA.so libB.so
for rep in 1 2 3; do
for lib in "$@"; do
  PROBE_NO_EM=1 PROBE_LAYOUT=tau RJP_DEBUG=1 RJP_LIB=$PWD/rajepy_amd/$lib python tools/k1_probe.py cfg4 f64 1 2>/dev/null
  PROBE_LAYOUT=tau RJP_DEBUG=1 RJP_LIB=$PWD/rajepy_amd/$lib python tools/k1_probe.py cfg4 f64 1 2>/dev/null
  [ "$rep" = 1 ] && PROBE_LAYOUT=wide RJP_DEBUG=1 RJP_LIB=$PWD/rajepy_amd/$lib python tools/k1_probe.py cfg4 f64 1 2>/dev/null
done
done
