#!/usr/bin/env python3
"""Does K1's streaming rate depend on where its three fields sit relative to each other?
Re-creates the compact fields of cfg4 several times in one process, shifting `temp` and `ts`
by a byte offset inside over-allocated buffers, and times the single-epoch scan each time."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from rajepy_amd import engine as E

shape = bench.CONFIGS["cfg4"][0]
n = shape[0] * shape[1] * shape[2]
eng = E.RTEngine(0)
ej = bench.EXAMPLE_BURSTS
red, blue = [], []
for t0, hl, chi, which in zip(ej["t_0"], ej["hl"], ej["chi"], ej["which"]):
    sig = hl * bench.YEAR * 2. / (2. * np.sqrt(2. * np.log(2.)))
    for jet, lst in (("R", red), ("B", blue)):
        if jet in str(which):
            lst.append((t0 * bench.YEAR, chi - 1., sig))
bursts = E.make_bursts(red, blue)
base = eng.synth_fields(shape, 20240504, 0, E.RJP_F64, csize_au=0.5, wide=False)
pad = 1 << 24                                   # doubles of slack per buffer
bufs = [torch.empty(n + pad, dtype=torch.float64, device=eng.device) for _ in range(3)]
print("buffer bases mod 2^21:", [b.data_ptr() % (1 << 21) for b in bufs], flush=True)
for off_t, off_s in ((0, 0), (32, 64), (128, 256), (512, 1024), (2048, 4096), (4096, 8192),
                     (8192, 16384), (65536, 131072), (1 << 20, 1 << 21), (0, 0)):
    em0 = bufs[0][:n]
    temp = bufs[1][off_t // 8: off_t // 8 + n]
    ts = bufs[2][off_s // 8: off_s // 8 + n]
    em0.copy_(base.em0); temp.copy_(base.temp); ts.copy_(base.ts)
    f = E.DeviceFields(shape, E.RJP_F64, 0.5, None, None, temp, None, ts)
    f.em0 = em0
    eng.time_ff_scan(f, bursts, [bench.YEAR], E.RJP_GFF_SCALAR, reps=2)
    ms = [eng.time_ff_scan(f, bursts, [bench.YEAR], E.RJP_GFF_SCALAR, reps=5) for _ in range(3)]
    print("offsets temp +%d B, ts +%d B: %.3f %.3f %.3f ms" % (off_t, off_s, *ms), flush=True)
