#!/usr/bin/env python3
"""Does K1's single-epoch time depend on WHERE the driver puts the fields?  Allocates the
three compact fields of cfg4 several times in one process -- as three separate allocations,
with perturbing allocations in between, and as slices of one arena -- and times the scan."""
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


def run(tag, em0, temp, ts):
    em0.copy_(base.em0); temp.copy_(base.temp); ts.copy_(base.ts)
    f = E.DeviceFields(shape, E.RJP_F64, 0.5, None, None, temp, None, ts)
    f.em0 = em0
    eng.time_ff_scan(f, bursts, [bench.YEAR], E.RJP_GFF_SCALAR, reps=2)
    ms = [eng.time_ff_scan(f, bursts, [bench.YEAR], E.RJP_GFF_SCALAR, reps=5) for _ in range(3)]
    print("%-34s %.3f %.3f %.3f ms   ptr%%1GiB = %s" % (
        tag, *ms, [hex(t.data_ptr() % (1 << 30)) for t in (em0, temp, ts)]), flush=True)


print("the synth fields themselves:", flush=True)
run("synth (own allocations)", base.em0.clone(), base.temp.clone(), base.ts.clone())
junk = []
for trial in range(4):
    bufs = [torch.empty(n, dtype=torch.float64, device=eng.device) for _ in range(3)]
    run("separate allocations #%d" % trial, *bufs)
    # perturb the allocator: keep an odd-sized block alive, drop the fields
    junk.append(torch.empty((trial + 1) * 77777777, dtype=torch.uint8, device=eng.device))
    del bufs
    torch.cuda.empty_cache()
for trial in range(3):
    arena = torch.empty(3 * n + 3 * 4096, dtype=torch.float64, device=eng.device)
    pad = trial * 512          # doubles between the fields: 0, 4 KiB, 8 KiB
    run("one arena, gap %d B" % (pad * 8), arena[:n], arena[n + pad: 2 * n + pad],
        arena[2 * n + 2 * pad: 3 * n + 2 * pad])
    del arena
    torch.cuda.empty_cache()
