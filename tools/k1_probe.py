#!/usr/bin/env python3
"""Times K1 (rjp_ff_scan) alone on a synthetic grid: python tools/k1_probe.py cfg4 f64 [E]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from rajepy_amd import engine as E

cfg, storage = sys.argv[1], sys.argv[2]
nep = int(sys.argv[3]) if len(sys.argv) > 3 else 1
want_em = os.environ.get("PROBE_NO_EM") is None
MODE = E.RJP_GFF_POWERLAW if os.environ.get("PROBE_POWERLAW") else E.RJP_GFF_SCALAR
shape = bench.CONFIGS[cfg][0]
eng = E.RTEngine(0)
dtype = E.RJP_F64 if storage == "f64" else E.RJP_F32
LAYOUT = os.environ.get("PROBE_LAYOUT", "tau")          # tau | compact | wide
fields = eng.synth_fields(shape, 20240504, 1 if os.environ.get("PROBE_POWERLAW") else 0, dtype, csize_au=0.5,
                          wide=(cfg != "cfg4x8"),
                          tau_mode=MODE if (LAYOUT == "tau" and storage == "f64") else None)
if LAYOUT == "wide":
    fields.em0 = None
if os.environ.get("PROBE_LT"):              # launch-time-ordered layout, K bins per jet
    info = eng.build_lt(fields, int(os.environ["PROBE_LT"]))
    print("lt layout: K=%d rows=%d build %.1f ms" % (info["K"], info["rows"], info["build_ms"]))
ej = bench.EXAMPLE_BURSTS
red, blue = [], []
for t0, hl, chi, which in zip(ej["t_0"], ej["hl"], ej["chi"], ej["which"]):
    sig = hl * bench.YEAR * 2. / (2. * np.sqrt(2. * np.log(2.)))
    for jet, lst in (("R", red), ("B", blue)):
        if jet in str(which):
            lst.append((t0 * bench.YEAR, chi - 1., sig))
bursts = E.make_bursts(red, blue)
ep = list(np.linspace(0.5, 4.5, nep) * bench.YEAR)
eng.time_ff_scan(fields, bursts, ep, MODE, reps=2, want_em=want_em, want_tavg=False)
ms = min(eng.time_ff_scan(fields, bursts, ep, MODE, reps=5, want_em=want_em, want_tavg=False)
         for _ in range(3))
n = shape[0] * shape[1] * shape[2]
npass = -(-nep // (32 if nep >= 32 else 16 if nep >= 16 else 8)) if nep > 1 else 1
nf = fields.scan_fields(MODE, want_em)
gb = npass * nf * n * int(dtype) / 1e9
print("path", eng.last_scan_path(), eng.last_moment_shape)
print("%s %s %s E=%d lib=%s ysplit=%s: %.3f ms  %.0f GB/s (alg)  %.3f ms/epoch" % (
    cfg, storage, "%d fields" % nf, nep, os.path.basename(os.environ.get("RJP_LIB", "default")),
    os.environ.get("RJP_YSPLIT", "auto"), ms, gb / ms * 1e3, ms / nep))
