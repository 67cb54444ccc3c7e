#!/usr/bin/env python3
"""Times K2 (rjp_ff_maps) alone on cfg4-size maps: python tools/k2_probe.py [nchan]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from rajepy_amd import engine as E

nchan = int(sys.argv[1]) if len(sys.argv) > 1 else 256
P = 512 * 512
eng = E.RTEngine(0)
sumA = torch.rand(1, P, dtype=torch.float64, device=eng.device) * 1e12
tavg = torch.full((P,), 1e4, dtype=torch.float64, device=eng.device)
freqs = np.geomspace(1e9, 5e10, nchan)
ctau, cflux = E.ff_channel_coeffs(freqs, 0.5, 120., E.RJP_GFF_POWERLAW)
out = (eng._f64(1, nchan, P), eng._f64(1, nchan, P), eng._f64(1, nchan))
for want in ("tau+flux+ftot", "tau+flux", "flux+ftot", "ftot"):
    o = (out[0] if "tau" in want else None, out[1] if "flux" in want else None,
         out[2] if "ftot" in want else None)
    eng.ff_maps(sumA, tavg, ctau, cflux, out=o)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(20):
        eng.ff_maps(sumA, tavg, ctau, cflux, out=o)
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / 20
    nb = P * nchan * 8 * (("tau" in want) + ("flux" in want))
    print("%-14s %.3f ms  %.0f GB/s written" % (want, ms, nb / ms / 1e6))
