#!/usr/bin/env python3
"""Does the launch-time-ordered pass depend on where its cell buffer lies?  The buffer is copied to
several offsets of one slab and to freshly allocated blocks (python tools/lt_placement_probe.py [K])."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from rajepy_amd import engine as E

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
shape = bench.CONFIGS["cfg5"][0]
eng = E.RTEngine(0)
eng.cache_moments = False
f = eng.synth_fields(shape, 20240505, 0, E.RJP_F64, csize_au=0.5, wide=False,
                     tau_mode=E.RJP_GFF_SCALAR, with_em0=False)
ej = bench.EXAMPLE_BURSTS
red, blue = [], []
for t0, hl, chi, which in zip(ej["t_0"], ej["hl"], ej["chi"], ej["which"]):
    sig = hl * bench.YEAR * 2. / (2. * np.sqrt(2. * np.log(2.)))
    for jet, lst in (("R", red), ("B", blue)):
        if jet in str(which):
            lst.append((t0 * bench.YEAR, chi - 1., sig))
bursts = E.make_bursts(red, blue)
ep = list(np.linspace(0.0, 5.0, 32) * bench.YEAR)

def t():
    eng.time_ff_scan(f, bursts, ep, E.RJP_GFF_SCALAR, reps=2, want_em=False, want_tavg=False)
    return min(eng.time_ff_scan(f, bursts, ep, E.RJP_GFF_SCALAR, reps=5, want_em=False,
                                want_tavg=False) for _ in range(3))

lt = eng.build_lt(f, K)
cells0 = lt["cells"]
m = cells0.numel()
print("built: cells@%#x: %.3f ms %s" % (cells0.data_ptr(), t(), eng.last_scan_path()), flush=True)
def stream_ms(blk):
    """a plain read stream of the same bytes (torch's reduction), best of 3"""
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); blk.sum(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best

keep = [cells0]
free, total = torch.cuda.mem_get_info()
nblk = int((free - (8 << 30)) // (m * 8))
print("free %.1f GB -> %d more blocks of %.1f GB" % (free / 1e9, nblk, m * 8 / 1e9), flush=True)
for i in range(nblk):
    blk = torch.empty(m, dtype=torch.float64, device=cells0.device)
    blk.copy_(cells0)
    lt["cells"] = blk
    print("block %2d @%#x: lt pass %.3f ms   plain sum %.3f ms" % (i, blk.data_ptr(), t(), stream_ms(blk)),
          flush=True)
    keep.append(blk)
