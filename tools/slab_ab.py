#!/usr/bin/env python3
"""Same-buffer A/B of library builds on ONE RANK'S SHARE of an N-way x-slab split (VERDICT r04
item 1b: "pick nsplit / y-range thresholds for 64-row slabs from a same-buffer A/B at that
size").  All builds scan the very same allocation alternately, in one process:

    python tools/slab_ab.py <ways> [--full] lib1.so lib2.so ...      ("default" = the shipped build)

legs: single-epoch table scan (cfg4's share), 32-epoch sweep through the LDS moments and on the
launch-time-ordered layout (cfg5's share).  --full: the whole 512x4096x512 grid instead."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import bench
from rajepy_amd import _lib, engine as E


def load(path):
    lib = C.CDLL(os.path.abspath(path))
    for name, (res, args) in _lib.SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    assert lib.rjp_version() == _lib.RJP_VERSION
    ctx = C.c_void_p()
    assert lib.rjp_ctx_create(0, C.byref(ctx)) == 0
    return lib, ctx


def main():
    argv = [a for a in sys.argv[1:] if a != "--full"]
    ways = int(argv[0])
    full = "--full" in sys.argv
    eng = E.RTEngine(0)
    eng.cache_moments = False
    libs = [("default", eng.lib, eng.ctx)] + [(os.path.basename(p),) + load(p) for p in argv[1:]
                                              if p != "default"]
    nx, ny, nz = bench.CONFIGS["cfg4"][0]
    shape = (nx if full else nx // ways, ny, nz)
    mode = E.RJP_GFF_SCALAR
    fields = eng.synth_fields(shape, bench.SEED, 0, E.RJP_F64, csize_au=0.5, wide=False,
                              tau_mode=mode)
    ej = bench.EXAMPLE_BURSTS
    red, blue = [], []
    for t0, hl, chi, which in zip(ej["t_0"], ej["hl"], ej["chi"], ej["which"]):
        sig = hl * bench.YEAR * 2. / (2. * np.sqrt(2. * np.log(2.)))
        for jet, lst in (("R", red), ("B", blue)):
            if jet in str(which):
                lst.append((t0 * bench.YEAR, chi - 1., sig))
    bursts = E.make_bursts(red, blue)
    eng.launch_time_range(fields)
    P = fields.npix
    # (the largest workspace any of the builds asks for: their split rules may differ)
    work = eng._workspace(max(max(lib.rjp_ff_scan_workspace(*shape, 1),
                                  lib.rjp_ff_scan_workspace(*shape, 32)) for _, lib, _ in libs))
    sumA = eng._f64(32, P)

    def time(lib, ctx, fs, ep, reps):
        ms = C.c_double()
        epa = _lib.dbl_array(ep)
        st = lib.rjp_time_ff_scan(ctx, C.byref(fs), C.byref(bursts), epa, len(ep), mode,
                                  sumA.data_ptr(), None, None, work.data_ptr(), work.numel(),
                                  eng._stream(), reps, C.byref(ms))
        assert st == 0, lib.rjp_last_error(ctx)
        return ms.value, lib.rjp_last_scan_path(ctx, None, None)

    e1 = [1.0 * bench.YEAR]
    e32 = [float(t) for t in np.linspace(0., 5., 32) * bench.YEAR]
    legs = [("table scan, 1 epoch", e1, False, 30), ("LDS moments, 32 epochs", e32, False, 10),
            ("lt layout, 32 epochs", e32, True, 10)]
    print("shape", shape)
    for name, ep, lt, reps in legs:
        if lt:
            eng.build_lt(fields, 20)
        fs = fields.struct()
        fs.occupied_cells = -1                     # (skip the tiles-or-moments cost model)
        res = {n: [] for n, _, _ in libs}
        paths = {}
        for n, lib, ctx in libs:
            time(lib, ctx, fs, ep, 2)
        for _ in range(5):
            for n, lib, ctx in libs:
                ms, path = time(lib, ctx, fs, ep, reps)
                res[n].append(ms)
                paths[n] = path
        base = np.mean(res["default"])
        for n, _, _ in libs:
            v = np.array(res[n])
            print("%-26s %-28s path %d  mean %.4f ms  min %.4f  vs default %.4f"
                  % (name, n, paths[n], v.mean(), v.min(), v.mean() / base))
        fields.lt = None


if __name__ == "__main__":
    main()
