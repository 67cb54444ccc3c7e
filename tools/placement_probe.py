#!/usr/bin/env python3
"""Does the time of the single-epoch K1 scan depend on WHERE a0 and ts lie relative to each other?
One slab holds both; ts is placed `skew` bytes past the end of a0 (python tools/placement_probe.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from rajepy_amd import engine as E

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
shape = bench.CONFIGS[cfg][0]
n = shape[0] * shape[1] * shape[2]
eng = E.RTEngine(0)
eng.cache_moments = False
f = eng.synth_fields(shape, 20240504, 0, E.RJP_F64, csize_au=0.5, wide=False,
                     tau_mode=E.RJP_GFF_SCALAR, with_em0=False)
ej = bench.EXAMPLE_BURSTS
red, blue = [], []
for t0, hl, chi, which in zip(ej["t_0"], ej["hl"], ej["chi"], ej["which"]):
    sig = hl * bench.YEAR * 2. / (2. * np.sqrt(2. * np.log(2.)))
    for jet, lst in (("R", red), ("B", blue)):
        if jet in str(which):
            lst.append((t0 * bench.YEAR, chi - 1., sig))
bursts = E.make_bursts(red, blue)
ep = [1.0 * bench.YEAR]

def t():
    eng.time_ff_scan(f, bursts, ep, E.RJP_GFF_SCALAR, reps=2, want_em=False, want_tavg=False)
    return min(eng.time_ff_scan(f, bursts, ep, E.RJP_GFF_SCALAR, reps=10, want_em=False,
                                want_tavg=False) for _ in range(3))

a0_0, ts_0 = f.a0, f.ts
print("separate tensors: a0@%#x ts@%#x (ts - a0 = %#x): %.3f ms  path %s" % (
    a0_0.data_ptr(), ts_0.data_ptr(), ts_0.data_ptr() - a0_0.data_ptr(), t(), eng.last_scan_path()),
    flush=True)
PAD = 1 << 27                      # elements of slack (1 GiB)
slab = torch.empty(2 * n + 2 * PAD, dtype=torch.float64, device=a0_0.device)
for base in (0, 1 << 17):          # elements: 0, 1 MiB
    for skew in (0, 32, 128, 512, 2048, 8192, 1 << 15, 1 << 17, 1 << 19, 1 << 21, 1 << 23, 1 << 25,
                 (1 << 25) + 2048 + 128):
        a0 = slab[base:base + n]
        ts = slab[base + n + skew:base + 2 * n + skew]
        a0.copy_(a0_0); ts.copy_(ts_0)
        f.a0, f.ts = a0, ts
        f.ts_range = None; f._ts_range_of = None
        print("base %#10x skew %#11x B: %.3f ms" % (base * 8, skew * 8, t()), flush=True)
