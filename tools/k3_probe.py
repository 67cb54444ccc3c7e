#!/usr/bin/env python3
"""Times K3 (rjp_rrl_scan) alone: python tools/k3_probe.py cfg3 f64 [nchan]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from rajepy_amd import _lib, engine as E
from rajepy_amd.maths import rrls

cfg, storage = sys.argv[1], sys.argv[2]
shape, nchan = bench.CONFIGS[cfg][0], bench.CONFIGS[cfg][1]
if len(sys.argv) > 3:
    nchan = int(sys.argv[3])
eng = E.RTEngine(0)
dtype = E.RJP_F64 if storage == "f64" else E.RJP_F32
fields = eng.synth_fields(shape, 20240504, 0, dtype, csize_au=0.5, with_vy=True)
ej = bench.EXAMPLE_BURSTS
red, blue = [], []
for t0, hl, chi, which in zip(ej["t_0"], ej["hl"], ej["chi"], ej["which"]):
    sig = hl * bench.YEAR * 2. / (2. * np.sqrt(2. * np.log(2.)))
    for jet, lst in (("R", red), ("B", blue)):
        if jet in str(which):
            lst.append((t0 * bench.YEAR, chi - 1., sig))
bursts = E.make_bursts(red, blue)
lc = rrls.line_constants("H66a")
line = _lib.Line(**lc)
freqs = lc["nu_rest"] - nchan * 1e5 / 2. + 1e5 / 2. + np.arange(nchan) * 1e5
eng.rrl_scan(fields, bursts, bench.YEAR, line, freqs)
torch.cuda.synchronize()
best = 1e30
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    tau = eng.rrl_scan(fields, bursts, bench.YEAR, line, freqs)
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1))
n = shape[0] * shape[1] * shape[2]
print("%s %s F=%d: %.1f ms  %.3e Voigt/s  checksum %.10e" % (
    cfg, storage, nchan, best, n * nchan / best * 1e3, float(tau.sum().item())))
