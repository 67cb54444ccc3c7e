#!/usr/bin/env python3
"""K1 without bursts (no launch times read, no exp): the pure streaming rate of the scan --
2 fields in the compact layout, 4 in the wide one (RJP_NO_COMPACT=1)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from rajepy_amd import engine as E
cfg, storage = sys.argv[1], sys.argv[2]
shape = bench.CONFIGS[cfg][0]
eng = E.RTEngine(0)
dtype = E.RJP_F64 if storage == "f64" else E.RJP_F32
fields = eng.synth_fields(shape, 20240504, 0, dtype, csize_au=0.5)
eng.time_ff_scan(fields, None, [0.0], E.RJP_GFF_SCALAR, reps=2)
ms = min(eng.time_ff_scan(fields, None, [0.0], E.RJP_GFF_SCALAR, reps=5) for _ in range(3))
n = shape[0] * shape[1] * shape[2]
nf = 2 if fields.em0 is not None else 4
print("%s %s no bursts: %.3f ms  %.0f GB/s (%d fields)" % (cfg, storage, ms, nf * n * int(dtype) / ms / 1e6, nf))
