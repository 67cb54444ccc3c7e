#!/usr/bin/env python3
"""What the K1 launch skeleton streams without the burst factor: the BURSTS = false kernel on the
tau layout reads a0 alone (8 B/cell); with the EM map a0 and em0 (16 B/cell, the same two
streams per cell as the timed step, no chi at all).  python tools/k1_skeleton_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from rajepy_amd import engine as E
shape = bench.CONFIGS["cfg4"][0]
n = shape[0] * shape[1] * shape[2]
eng = E.RTEngine(0)
f = eng.synth_fields(shape, 20240504, 0, E.RJP_F64, csize_au=0.5, tau_mode=E.RJP_GFF_SCALAR)
for em, nb in ((False, 8), (True, 16)):
    eng.time_ff_scan(f, None, [0.0], E.RJP_GFF_SCALAR, reps=2, want_em=em, want_tavg=False)
    ms = min(eng.time_ff_scan(f, None, [0.0], E.RJP_GFF_SCALAR, reps=8, want_em=em, want_tavg=False)
             for _ in range(3))
    print("no bursts, em=%s: %.3f ms  %.0f GB/s of %d B/cell" % (em, ms, n * nb / ms / 1e6, nb))
