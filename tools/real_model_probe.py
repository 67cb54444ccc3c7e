#!/usr/bin/env python3
"""The hot path on REAL (K4-built) models at sizes far beyond the reference's example grid:
the example jet's parameters on a refined grid (same physical box, smaller cells), at two
inclinations.  Launch times of a real model vary smoothly along a sightline -- the opposite of
the synthetic set's uncorrelated ones -- so this is where the epoch tiles and the launch-time
moment path are compared on coherent data.
    python tools/real_model_probe.py [refine=5]   (one JSON line per model)"""
import json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from rajepy_amd import classes, logger, engine as E
from tests.test_host_logic import example_params

YEAR = 31557600.0
refine = int(sys.argv[1]) if len(sys.argv) > 1 else 5
nep = int(sys.argv[2]) if len(sys.argv) > 2 else 32


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


for inc in (90., 60., 30.):
    p = example_params()
    p["geometry"]["inc"] = inc
    g = p["grid"]
    g["n_x"], g["n_y"], g["n_z"] = g["n_x"] * refine, g["n_y"] * refine, g["n_z"] * refine
    g["c_size"] = g["c_size"] / refine
    tmp = tempfile.mkdtemp()
    log = logger.Log(os.path.join(tmp, "run.log"), verbose=False)
    t0 = time.perf_counter()
    jm = classes.JetModel(p, log=log)
    dev = jm.device_fields
    torch.cuda.synchronize()
    rec = {"inc": inc, "grid": [jm.nx, jm.ny, jm.nz], "cells": jm.nx * jm.ny * jm.nz,
           "construct_ms": (time.perf_counter() - t0) * 1e3}
    eng = jm.engine
    eng.cache_moments = False
    rec["occupied_cells"] = int(dev.occupied_cells) if dev.occupied_cells is not None else None
    lo, hi = eng.launch_time_range(dev)
    rec["ts_range_yr"] = [lo / YEAR, hi / YEAR]
    freqs = np.logspace(9, np.log10(5e10), 64)
    times = np.linspace(0., 5., nep) * YEAR
    jm.time = 1.0 * YEAR
    rec["tau_64ch_ms"] = timed(lambda: jm.optical_depth_ff(freqs))
    bursts = jm._rjp_bursts()
    rec["k1_1epoch_ms"] = eng.time_ff_scan(dev, bursts, [jm.time], jm.gff_mode, reps=5,
                                           want_em=False, want_tavg=False)
    out = {}
    for name, um, fm in (("tiles", False, False), ("moments", True, True), ("auto", True, False)):
        eng.use_moments, eng.force_moments = um, fm
        eng.time_ff_scan(dev, bursts, list(times), jm.gff_mode, reps=1, want_em=False,
                         want_tavg=False)
        rec["k1_%depoch_%s_ms" % (nep, name)] = eng.time_ff_scan(
            dev, bursts, list(times), jm.gff_mode, reps=3, want_em=False, want_tavg=False)
        rec["path_" + name] = eng.last_scan_path()[0]
        s, _, _ = eng.ff_scan(dev, bursts, list(times), jm.gff_mode, want_em=False,
                              want_tavg=False)
        out[name] = s.cpu().numpy()
        rec["lightcurve_%s_ms" % name] = timed(lambda: jm.flux_vs_time(times, freqs))
    eng.use_moments, eng.force_moments = True, False
    a, b = out["tiles"], out["moments"]
    ok = np.isfinite(a) & (a != 0)
    rec["moments_vs_tiles_max_rel"] = float(np.max(np.abs(b[ok] - a[ok]) / np.abs(a[ok]))) if ok.any() else None
    rec["moments_vs_tiles_pattern_equal"] = bool(np.array_equal(np.isnan(a), np.isnan(b)) and
                                                 np.array_equal(a == 0, b == 0))
    print(json.dumps(rec), flush=True)
    del jm, dev
    torch.cuda.empty_cache()
