#!/usr/bin/env python3
"""Times K4 (rjp_build_fields through JetModel.device_fields) on the example jet scaled to a
grid: python tools/k4_probe.py [nx ny nz]"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rajepy_amd import classes, logger
from rajepy_amd.engine import RTEngine
from tests.test_host_logic import example_params

shape = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (512, 4096, 512)
eng = RTEngine(0)
tmp = tempfile.mkdtemp()
for rep in range(3):
    p = example_params()
    p["grid"].update(n_x=shape[0], n_y=shape[1], n_z=shape[2])
    jm = classes.JetModel(p, log=logger.Log(os.path.join(tmp, "a%d.log" % rep), verbose=False),
                          engine=eng)
    eng.synchronize()
    t0 = time.perf_counter()
    dev = jm.device_fields
    eng.synchronize()
    print("K4 build %dx%dx%d: %.4f s" % (*shape, time.perf_counter() - t0), flush=True)
    del jm, dev
