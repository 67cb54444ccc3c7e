set -e
for lib in librjprt_dbg.so librjprt_u2.so librjprt_u8.so; do
  for ys in auto 4 16 32; do
    if [ "$ys" = auto ]; then unset RJP_YSPLIT; else export RJP_YSPLIT=$ys; fi
    PROBE_NO_EM=1 RJP_DEBUG=1 RJP_LIB=$PWD/rajepy_amd/$lib python tools/k1_probe.py cfg4 f64 1
  done
done
unset RJP_YSPLIT
for lib in librjprt_dbg.so librjprt_u2.so librjprt_u8.so; do
  RJP_DEBUG=1 RJP_LIB=$PWD/rajepy_amd/$lib python tools/k1_probe.py cfg4 f64 1
  PROBE_NO_EM=1 RJP_DEBUG=1 RJP_LIB=$PWD/rajepy_amd/$lib python tools/k1_probe.py cfg2 f64 1
done
PROBE_NO_EM=1 python tools/k1_probe.py cfg5 f64 32
python tools/k1_probe.py cfg5 f64 32
PROBE_NO_EM=1 PROBE_LAYOUT=compact python tools/k1_probe.py cfg5 f64 32
PROBE_NO_EM=1 python tools/k1_probe.py cfg4 f64 8
PROBE_NO_EM=1 PROBE_POWERLAW=1 python tools/k1_probe.py cfg4 f64 1
