#!/usr/bin/env python3
"""Light curve of a REAL jet model over decades: the example jet at 512x4096x512 (built on the
device, occupied y-ranges attached), 64 irregularly spaced epochs from 0 to 40 yr -> direct
8-epoch tiles.  Beyond ~18 yr every burst of the example model has underflowed to exactly zero
in every cell; the multi-epoch tiles skip such bursts per wave.
usage: python tools/long_sweep_probe.py [dense]   (dense = without the y-ranges)"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from rajepy_amd import classes, logger, engine as E
from tests.test_host_logic import example_params

YEAR = 31536000.0
eng = E.RTEngine(0)
p = example_params()
p["grid"].update(n_x=512, n_y=4096, n_z=512)
jm = classes.JetModel(p, log=logger.Log(os.path.join(tempfile.mkdtemp(), "a.log"), verbose=False),
                      engine=eng)
dev = jm.device_fields
if len(sys.argv) > 1 and sys.argv[1] == "dense":
    dev.ylo = dev.yhi = None
b = jm._rjp_bursts()
ep = list(np.geomspace(0.05, 40., 64) * YEAR)
for want_em in (True, False):
    eng.ff_scan(dev, b, ep, jm.gff_mode, want_em=want_em)
    eng.synchronize()
    best = 1e9
    for _ in range(3):
        ms = eng.time_ff_scan(dev, b, ep, jm.gff_mode, reps=3, want_em=want_em)
        best = min(best, ms)
    sumA, _, _ = eng.ff_scan(dev, b, ep, jm.gff_mode, want_em=want_em)
    print("%s lib=%s ybounds=%s want_em=%s: 64 epochs 0.05-40 yr: %.3f ms  checksum %.17e" % (
        "example jet 512x4096x512", os.path.basename(os.environ.get("RJP_LIB", "default")),
        dev.ylo is not None, want_em, best, float(sumA.sum().item())), flush=True)
