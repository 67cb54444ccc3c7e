#!/bin/bash
# K2 A/B in one box: default build vs a variant (tools/k2_ab.sh librjprt_k2nt.so)
for rep in 1 2; do
  python tools/k2_probe.py 2>/dev/null
  RJP_DEBUG=1 RJP_LIB=$PWD/rajepy_amd/$1 python tools/k2_probe.py 2>/dev/null | sed "s/^/[$1] /"
done
python bench.py --no-cpu-baseline --sustained-seconds 2 --no-api-level 2>/dev/null | python -c "import sys,json; r=json.load(sys.stdin); print('default  step %.3f ms K1 %.3f' % (r['ms_per_step'], r['roofline']['ms_per_launch']))"
RJP_DEBUG=1 RJP_LIB=$PWD/rajepy_amd/$1 python bench.py --no-cpu-baseline --sustained-seconds 2 --no-api-level 2>/dev/null | python -c "import sys,json; r=json.load(sys.stdin); print('variant  step %.3f ms K1 %.3f' % (r['ms_per_step'], r['roofline']['ms_per_launch']))"
