#!/usr/bin/env python3
"""Does the RELATIVE placement of the two streams of the single-epoch scan (a0, ts) set its
time?  At 512x4096x512 each field is exactly 2^33 bytes: buffers allocated back to back put
a0[i] and ts[i] a power of two apart.  Here ts is copied into one large buffer at chosen byte
offsets and the same scan is timed for each, in one process.
    python tools/k1_placement_probe.py"""
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
fields = eng.synth_fields(shape, 20240504, 0, E.RJP_F64, csize_au=0.5, tau_mode=E.RJP_GFF_SCALAR,
                          wide=False)
ej = bench.EXAMPLE_BURSTS
red, blue = [], []
for t0, hl, chi, which in zip(ej["t_0"], ej["hl"], ej["chi"], ej["which"]):
    sig = hl * bench.YEAR * 2. / (2. * np.sqrt(2. * np.log(2.)))
    for jet, lst in (("R", red), ("B", blue)):
        if jet in str(which):
            lst.append((t0 * bench.YEAR, chi - 1., sig))
bursts = E.make_bursts(red, blue)
ep = [1.0 * bench.YEAR]
ts0 = fields.ts
pad = (1 << 28) // 8                       # 256 MiB of slack, in doubles
big = torch.empty(n + pad, dtype=torch.float64, device=eng.device)
a_ptr = fields.a0.data_ptr()


def timed():
    eng.time_ff_scan(fields, bursts, ep, E.RJP_GFF_SCALAR, reps=2, want_em=False, want_tavg=False)
    return min(eng.time_ff_scan(fields, bursts, ep, E.RJP_GFF_SCALAR, reps=8, want_em=False,
                                want_tavg=False) for _ in range(2))


fields.ts = ts0
d = ts0.data_ptr() - a_ptr
print("as allocated: ts - a0 = %d B = 2^33 * %.6f  (mod 2 MiB: %d, mod 4 KiB: %d): %.3f ms" % (
    d, d / 2.0 ** 33, d % (1 << 21), d % 4096, timed()), flush=True)
for off_bytes in (0, 256, 512, 1024, 2048, 4096, 8192, 16384, 65536, 262144, 1 << 20,
                  (1 << 20) + 4096, 1 << 21, (1 << 21) + 2048, 1 << 24, (1 << 24) + 4096 + 256):
    view = big[off_bytes // 8: off_bytes // 8 + n]
    view.copy_(ts0)
    fields.ts = view
    fields.ts_range = fields._ts_range_of = None
    d = view.data_ptr() - a_ptr
    print("offset %9d B: ts - a0 mod 2^33 = %d (mod 2 MiB %7d, mod 4 KiB %4d): %.3f ms" % (
        off_bytes, d % (1 << 33), d % (1 << 21), d % 4096, timed()), flush=True)
