#!/usr/bin/env python3
"""A/B of two builds of librjprt on the SAME device buffers in ONE process.

K1's time depends on where the driver places the fields in physical memory (a few per cent
between allocations, profiles/r02_box_spread.md), so two processes cannot resolve a 1-3 %
difference between builds; here both libraries scan the very same allocation, alternately.

    python tools/k1_inproc_ab.py rajepy_amd/librjprt_prev.so [cfg4] [n_epochs] [--em]
(the default build is A, the named one B)
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import bench
from rajepy_amd import _lib, engine as E


def main():
    other = os.path.abspath(sys.argv[1])
    cfg = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else "cfg4"
    nep = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].isdigit() else 1
    want_em = "--em" in sys.argv
    layout = os.environ.get("PROBE_LAYOUT", "tau")
    eng = E.RTEngine(0)
    libb = C.CDLL(other)
    for name, (res, args) in _lib.SIGNATURES.items():
        fn = getattr(libb, name)
        fn.restype, fn.argtypes = res, args
    assert libb.rjp_version() == _lib.RJP_VERSION
    ctxb = C.c_void_p()
    assert libb.rjp_ctx_create(0, C.byref(ctxb)) == 0
    shape = bench.CONFIGS[cfg][0]
    mode = E.RJP_GFF_SCALAR
    fields = eng.synth_fields(shape, 20240504, 0, E.RJP_F64, csize_au=0.5,
                              tau_mode=mode if layout == "tau" else None)
    if layout == "wide":
        fields.em0 = None
    ej = bench.EXAMPLE_BURSTS
    red, blue = [], []
    for t0, hl, chi, which in zip(ej["t_0"], ej["hl"], ej["chi"], ej["which"]):
        sig = hl * bench.YEAR * 2. / (2. * np.sqrt(2. * np.log(2.)))
        for jet, lst in (("R", red), ("B", blue)):
            if jet in str(which):
                lst.append((t0 * bench.YEAR, chi - 1., sig))
    bursts = E.make_bursts(red, blue)
    ep = list(np.linspace(0.5, 4.5, nep) * bench.YEAR) if nep > 1 else [1.0 * bench.YEAR]
    P = fields.npix
    nx, ny, nz = shape
    sumA = eng._f64(nep, P)
    em = eng._f64(nep, P) if want_em else None
    work = eng._workspace(eng.lib.rjp_ff_scan_workspace(nx, ny, nz, nep))
    eng.launch_time_range(fields)       # (single-epoch scans then take the LDS table path)
    fs = fields.struct()
    epa = _lib.dbl_array(ep)

    def time(lib, ctx, reps=10):
        ms = C.c_double()
        st = lib.rjp_time_ff_scan(ctx, C.byref(fs), C.byref(bursts), epa, nep, mode,
                                  sumA.data_ptr(), em.data_ptr() if em is not None else None,
                                  None, work.data_ptr(), work.numel(), eng._stream(), reps,
                                  C.byref(ms))
        assert st == 0, lib.rjp_last_error(ctx)
        return ms.value
    time(eng.lib, eng.ctx, 2), time(libb, ctxb, 2)
    a, b = [], []
    for _ in range(6):
        a.append(time(eng.lib, eng.ctx))
        b.append(time(libb, ctxb))
    fa, fb = np.array(a), np.array(b)
    print("%s %s E=%d em=%s  A(default) %.4f ms (min %.4f)  B(%s) %.4f ms (min %.4f)  A/B = %.4f"
          % (cfg, layout, nep, want_em, fa.mean(), fa.min(), os.path.basename(other), fb.mean(),
             fb.min(), fa.mean() / fb.mean()))


if __name__ == "__main__":
    main()
