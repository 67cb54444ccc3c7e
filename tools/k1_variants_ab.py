#!/usr/bin/env python3
"""A/B of several builds of librjprt on the SAME device buffers in ONE process, single-epoch cfg4
scan WITH the launch-time range in rjp_fields (so the LDS-table path is eligible) and, as the
reference, the default build without it (the Gaussians).
    python tools/k1_variants_ab.py lib1.so [lib2.so ...]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import bench
from rajepy_amd import _lib, engine as E

MODE = os.environ.get("AB_MODE", "tau")        # tau | em (a0, em0, ts) | wide (five model fields)
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
shape = bench.CONFIGS["cfg4"][0]
fields = eng.synth_fields(shape, 20240504, 0, E.RJP_F64, csize_au=0.5, wide=(MODE == "wide"),
                          tau_mode=E.RJP_GFF_SCALAR)
if MODE == "wide":
    fields.a0 = fields.em0 = None
ej = bench.EXAMPLE_BURSTS
red, blue = [], []
for t0, hl, chi, which in zip(ej["t_0"], ej["hl"], ej["chi"], ej["which"]):
    sig = hl * bench.YEAR * 2. / (2. * np.sqrt(2. * np.log(2.)))
    for jet, lst in (("R", red), ("B", blue)):
        if jet in str(which):
            lst.append((t0 * bench.YEAR, chi - 1., sig))
bursts = E.make_bursts(red, blue)
eng.launch_time_range(fields)
fs_tab = fields.struct()
fs_gau = fields.struct()
fs_gau.ts_lo = fs_gau.ts_hi = 0.0
nx, ny, nz = shape
sumA = eng._f64(1, fields.npix)
emap = eng._f64(1, fields.npix) if MODE != "tau" else None
tavg = eng._f64(fields.npix) if MODE == "wide" else None
ptr = lambda t: t.data_ptr() if t is not None else None
nfld = {"tau": 2, "em": 3, "wide": 5}[MODE]
work = eng._workspace(eng.lib.rjp_ff_scan_workspace(nx, ny, nz, 1))
epa = _lib.dbl_array([1.0 * bench.YEAR])


def time(lib, ctx, fs, reps=20):
    ms = C.c_double()
    st = lib.rjp_time_ff_scan(ctx, C.byref(fs), C.byref(bursts), epa, 1, 0, sumA.data_ptr(),
                              ptr(emap), ptr(tavg), work.data_ptr(), work.numel(), eng._stream(),
                              reps, C.byref(ms))
    assert st == 0, lib.rjp_last_error(ctx)
    return ms.value


rows = {"gaussians(default)": []}
for name, _, _ in handles:
    rows[name] = []
for _ in range(5):
    rows["gaussians(default)"].append(time(eng.lib, eng.ctx, fs_gau))
    for name, lb, ctx in handles:
        rows[name].append(time(lb, ctx, fs_tab))
for k, v in rows.items():
    v = np.array(v)
    print("%-28s mean %.4f ms  min %.4f ms  (%.3f of 8 TB/s)" % (
        k, v.mean(), v.min(), 8.0 * nfld * nx * ny * nz / (v.min() * 1e-3) / 8e12))
