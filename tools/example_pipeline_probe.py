#!/usr/bin/env python3
"""End-to-end wall time of the reference's own example: the example jet on its as-shipped
grid (l_z = 2 arcsec -> 108 x 110 x 588 cells) through `Pipeline.execute(simobserve=False)`
with the run table of files/example-pipeline-params.py (121 continuum epochs, linspace(0, 5)
yr, 6 GHz, 0.5 GHz bandwidth in 0.2 GHz channels; EM / Tau / Flux FITS products per epoch).
The reference spends ~31 s per run on the 50 x 400 x 50 grid.
    python tools/example_pipeline_probe.py"""
import json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from rajepy_amd import classes, logger
from tests.test_host_logic import example_params

out = {}
for l_z in (None, 2.):
    tmp = tempfile.mkdtemp()
    p = example_params()
    p["grid"]["l_z"] = l_z
    pp = {"min_el": 20., "dcys": {"model_dcy": os.path.join(tmp, "out")},
          "continuum": {"times": np.linspace(0., 5., 24 * 5 + 1), "freqs": np.array([6.]) * 1e9,
                        "t_obs": np.array([59400]), "tscps": np.array([("EMERLIN", "0")]),
                        "t_ints": np.array([5]), "bws": np.array([.5e9]),
                        "chanws": np.array([2.e8])},
          "rrls": {"times": np.array([]), "lines": np.array(["H58a"]),
                   "t_obs": np.array([30000]), "tscps": np.array([("VLA", "A")]),
                   "t_ints": np.array([60]), "bws": np.array([1e8]), "chanws": np.array([1e5])}}
    t0 = time.perf_counter()
    log = logger.Log(os.path.join(tmp, "run.log"), verbose=False)
    jm = classes.JetModel(p, log=log)
    pl = classes.Pipeline(jm, pp, log=log)
    t1 = time.perf_counter()
    pl.execute(simobserve=False, verbose=False, dryrun=False, resume=False, clobber=True)
    t2 = time.perf_counter()
    nfits = sum(len([f for f in fs if f.endswith(".fits")]) for _, _, fs in os.walk(tmp))
    key = "grid_%dx%dx%d" % (jm.nx, jm.ny, jm.nz)
    out[key] = {"runs": len(pl.runs), "fits_files": nfits, "construct_s": t1 - t0,
                "execute_s": t2 - t1, "ms_per_run": (t2 - t1) / len(pl.runs) * 1e3,
                "flux_first_last_jy": [pl.runs[0].results["flux"], pl.runs[-1].results["flux"]]}
    print(key, json.dumps(out[key]), flush=True)
print(json.dumps(out))
