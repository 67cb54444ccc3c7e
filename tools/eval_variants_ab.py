#!/usr/bin/env python3
"""A/B of builds on the contraction alone: cfg5's 32-epoch sweep from CACHED moment maps
(rjp_fields.d_mom_cache filled once by the default build), same buffers, one process.
    python tools/eval_variants_ab.py lib1.so [lib2.so ...]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
from rajepy_amd import _lib, engine as E

libs = [("default", None)] + [(os.path.basename(p), os.path.abspath(p)) for p in sys.argv[1:]]
eng = E.RTEngine(0)
handles = []
for name, path in libs:
    if path is None:
        handles.append((name, eng.lib, eng.ctx))
        continue
    lb = C.CDLL(path)
    for fn_name, (res, args) in _lib.SIGNATURES.items():
        fn = getattr(lb, fn_name)
        fn.restype, fn.argtypes = res, args
    ctx = C.c_void_p()
    assert lb.rjp_ctx_create(0, C.byref(ctx)) == 0
    handles.append((name, lb, ctx))
shape = bench.CONFIGS["cfg5"][0]
fields = eng.synth_fields(shape, 20240504, 0, E.RJP_F64, csize_au=0.5, wide=False,
                          tau_mode=E.RJP_GFF_SCALAR)
ej = bench.EXAMPLE_BURSTS
red, blue = [], []
for t0, hl, chi, which in zip(ej["t_0"], ej["hl"], ej["chi"], ej["which"]):
    sig = hl * bench.YEAR * 2. / (2. * np.sqrt(2. * np.log(2.)))
    for jet, lst in (("R", red), ("B", blue)):
        if jet in str(which):
            lst.append((t0 * bench.YEAR, chi - 1., sig))
bursts = E.make_bursts(red, blue)
eng.launch_time_range(fields)
nx, ny, nz = shape
ep = [float(t) for t in np.linspace(0., 5., 32) * bench.YEAR]
epa = _lib.dbl_array(ep)
sumA = eng._f64(32, fields.npix)
work = eng._workspace(eng.lib.rjp_ff_scan_workspace(nx, ny, nz, 32))
cache = torch.empty(eng.lib.rjp_moment_cache_bytes(nx, nz) // 8, dtype=torch.float64, device=eng.device)
fs = fields.struct()
fs.d_mom_cache = cache.data_ptr()


def time(lib, ctx, reps=20):
    ms = C.c_double()
    st = lib.rjp_time_ff_scan(ctx, C.byref(fs), C.byref(bursts), epa, 32, 0, sumA.data_ptr(), None,
                              None, work.data_ptr(), work.numel(), eng._stream(), reps, C.byref(ms))
    assert st == 0, lib.rjp_last_error(ctx)
    return ms.value


fulls = {name: [] for name, _, _ in handles}     # pass + contraction, filling the cache
for _ in range(3):
    for name, lb, ctx in handles:
        fs.mom_cache_K = fs.mom_cache_N = 0
        fulls[name].append(time(lb, ctx, 5))
for k, v in fulls.items():
    print("%-28s pass + contraction: mean %.4f ms  min %.4f ms" % (k, np.mean(v), np.min(v)))
full = time(eng.lib, eng.ctx, 3)                 # the cache as the default build leaves it
fs.mom_cache_K, fs.mom_cache_N = 53, 12
rows = {name: [] for name, _, _ in handles}
for _ in range(5):
    for name, lb, ctx in handles:
        rows[name].append(time(lb, ctx))
        assert lb.rjp_last_scan_path(ctx, None, None) == 4
print("pass + contraction (default build): %.4f ms" % full)
for k, v in rows.items():
    v = np.array(v)
    print("%-28s contraction alone: mean %.4f ms  min %.4f ms" % (k, v.mean(), v.min()))
