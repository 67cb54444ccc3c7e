#!/usr/bin/env python3
"""Where K3's time goes: the same grid with the electron density scaled so that every cell
sits in one regime of the Voigt y parameter, and with the band moved off the line."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from rajepy_amd import _lib, engine as E
from rajepy_amd.maths import rrls

shape, nchan = (256, 1024, 256), 256
eng = E.RTEngine(0)
lc = rrls.line_constants("H66a")
line = _lib.Line(**lc)
base = lc["nu_rest"] - nchan * 1e5 / 2. + 1e5 / 2. + np.arange(nchan) * 1e5


def run(tag, scale, offset):
    f = eng.synth_fields(shape, 20240504, 0, E.RJP_F64, csize_au=0.5, with_vy=True)
    f.nd.mul_(scale)
    freqs = base + offset
    eng.rrl_scan(f, None, bench.YEAR, line, freqs)
    eng.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(3):
        eng.rrl_scan(f, None, bench.YEAR, line, freqs)
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / 3
    n = shape[0] * shape[1] * shape[2]
    print("%-46s %8.1f ms  %.3e Voigt/s" % (tag, ms, n * nchan / ms * 1e3), flush=True)


run("as generated (y ~ 3e-3 .. 10), band on the line", 1.0, 0.0)
run("n x 10 (y >= 0.03)", 10.0, 0.0)
run("n x 3", 3.0, 0.0)
run("n x 30 (y >= 0.1)", 30.0, 0.0)
run("n x 100 (y >= 0.3: plain lattice only)", 100.0, 0.0)
run("n / 100 (y <= 0.1: mostly shifted lattice)", 0.01, 0.0)
run("n / 1e4 (y <= 1e-3: shifted lattice only)", 1e-4, 0.0)
run("as generated, band 40 MHz off the line (far field)", 1.0, 4e7)
run("n x 100, band 40 MHz off the line", 100.0, 4e7)
