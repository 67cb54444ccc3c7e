#!/usr/bin/env python3
"""y-splits of the single-epoch scan on the SAME buffers in ONE process (debug-switch build:
RJP_YSPLIT is read on every call): the scan as shipped, and its chi-free skeleton (no bursts:
a0 + em0, the same two streams per cell with next to no arithmetic).
    RJP_DEBUG=1 RJP_LIB=rajepy_amd/librjprt_dbg.so python tools/k1_ysplit_ab.py [cfg4]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from rajepy_amd import engine as E

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
shape = bench.CONFIGS[cfg][0]
eng = E.RTEngine(0)
fields = eng.synth_fields(shape, 20240504, 0, E.RJP_F64, csize_au=0.5, tau_mode=E.RJP_GFF_SCALAR)
ej = bench.EXAMPLE_BURSTS
red, blue = [], []
for t0, hl, chi, which in zip(ej["t_0"], ej["hl"], ej["chi"], ej["which"]):
    sig = hl * bench.YEAR * 2. / (2. * np.sqrt(2. * np.log(2.)))
    for jet, lst in (("R", red), ("B", blue)):
        if jet in str(which):
            lst.append((t0 * bench.YEAR, chi - 1., sig))
bursts = E.make_bursts(red, blue)
LAYOUT = os.environ.get("PROBE_LAYOUT", "tau")          # tau | compact | wide
WANT_EM = os.environ.get("PROBE_EM") is not None
if LAYOUT in ("compact", "wide"):
    fields.a0 = None
if LAYOUT == "wide":
    fields.em0 = None
ep = [1.0 * bench.YEAR]
for rnd in range(2):
    for ys in os.environ.get("PROBE_YS", "0,1,2,3,4,6,8,12,16").split(","):
        os.environ["RJP_YSPLIT"] = ys
        out = []
        for flag in (False, None):
            b = bursts if flag is not None else None
            em = flag is None or WANT_EM           # the chi-free skeleton streams a0 + em0
            eng.time_ff_scan(fields, b, ep, E.RJP_GFF_SCALAR, reps=2, want_em=em, want_tavg=False)
            out.append(eng.time_ff_scan(fields, b, ep, E.RJP_GFF_SCALAR, reps=8, want_em=em,
                                        want_tavg=False))
        print("round %d ysplit %2s: scan %.3f ms  no-chi skeleton %.3f ms" % (
            rnd, ys if ys != "0" else "auto", out[0], out[1]), flush=True)
