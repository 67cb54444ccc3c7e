#!/usr/bin/env python3
"""Same-process A/B of the launch-time-ordered layout's bin count on cfg5's 32-epoch sweep:
python tools/lt_bins_ab.py 32 20 16   (each K built and timed in turn, three rounds)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from rajepy_amd import engine as E

Ks = [int(a) for a in sys.argv[1:]] or [32, 20]
eng = E.RTEngine(0)
eng.cache_moments = False
shape = bench.CONFIGS["cfg5"][0]
fields = eng.synth_fields(shape, 20240505, 0, E.RJP_F64, csize_au=0.5, wide=False,
                          tau_mode=E.RJP_GFF_SCALAR)
ej = bench.EXAMPLE_BURSTS
red, blue = [], []
for t0, hl, chi, which in zip(ej["t_0"], ej["hl"], ej["chi"], ej["which"]):
    sig = hl * bench.YEAR * 2. / (2. * np.sqrt(2. * np.log(2.)))
    for jet, lst in (("R", red), ("B", blue)):
        if jet in str(which):
            lst.append((t0 * bench.YEAR, chi - 1., sig))
bursts = E.make_bursts(red, blue)
ep = list(np.linspace(0.0, 5.0, 32) * bench.YEAR)
for rnd in range(3):
    for K in Ks:
        info = eng.build_lt(fields, K)
        eng.time_ff_scan(fields, bursts, ep, E.RJP_GFF_SCALAR, reps=2, want_em=False, want_tavg=False)
        ms = min(eng.time_ff_scan(fields, bursts, ep, E.RJP_GFF_SCALAR, reps=5, want_em=False,
                                  want_tavg=False) for _ in range(3))
        print("round %d K=%d path %s rows %d cells@%#x (mod 1 GiB %#x) rowoff@%#x: %.3f ms" % (
            rnd, K, eng.last_scan_path(), info["rows"], info["cells"].data_ptr(),
            info["cells"].data_ptr() % (1 << 30), info["rowoff"].data_ptr(), ms), flush=True)
