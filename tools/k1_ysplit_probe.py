#!/usr/bin/env python3
"""Times the single-epoch K1 of cfg4 under several forced y-split counts in ONE process (one
placement of the fields): RJP_DEBUG=1 RJP_LIB=.../librjprt_dbg.so python tools/k1_ysplit_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from rajepy_amd import engine as E

shape = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg4"][0]
eng = E.RTEngine(0)
ej = bench.EXAMPLE_BURSTS
red, blue = [], []
for t0, hl, chi, which in zip(ej["t_0"], ej["hl"], ej["chi"], ej["which"]):
    sig = hl * bench.YEAR * 2. / (2. * np.sqrt(2. * np.log(2.)))
    for jet, lst in (("R", red), ("B", blue)):
        if jet in str(which):
            lst.append((t0 * bench.YEAR, chi - 1., sig))
bursts = E.make_bursts(red, blue)
f = eng.synth_fields(shape, 20240504, 0, E.RJP_F64, csize_au=0.5, wide=False)
for rnd in range(2):
    for ys in (8, 7, 9, 6, 10, 11, 12, 13, 5, 8):
        os.environ["RJP_YSPLIT"] = str(ys)
        eng.time_ff_scan(f, bursts, [bench.YEAR], E.RJP_GFF_SCALAR, reps=2)
        ms = [eng.time_ff_scan(f, bursts, [bench.YEAR], E.RJP_GFF_SCALAR, reps=5) for _ in range(2)]
        print("ysplit %2d (rows per range %4d): %.3f %.3f ms" % (ys, -(-shape[1] // ys), *ms), flush=True)
