#!/usr/bin/env python3
"""Where does the time of a fresh JetModel at 512x4096x512 go?  (VERDICT r04 item 7:
construct_ms <= 200 ms, <= 4 grid-sized arrays resident for a continuum pipeline.)
Phases: JetModel() on the host, K4 incl. its allocations, the occupied y-ranges; first in a
fresh process (device memory comes from the driver), then again with PyTorch's cached blocks."""
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from rajepy_amd import classes, engine as E, logger
    meta = json.loads(str(np.load(os.path.join(ROOT, "tests", "golden", "cfg1_example.npz"))["meta"]))
    eng = E.RTEngine(0)
    out = []
    for rnd in ("fresh process", "cached blocks"):
        par = json.loads(json.dumps(meta["params"]))
        for k in ("t_0", "hl", "chi", "which"):
            par["ejection"][k] = np.array(par["ejection"][k])
        par["geometry"].pop("mod_r_0", None)
        for k in ("q_n", "q_tau"):
            par["power_laws"].pop(k, None)
        par["properties"].pop("n_0", None)
        scale = 512.0 / par["grid"]["n_x"]
        par["grid"].update(n_x=512, n_y=4096, n_z=512, c_size=par["grid"]["c_size"] / scale)
        log = logger.Log(os.path.join(tempfile.mkdtemp(), "run.log"), verbose=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        jm = classes.JetModel(par, log=log, engine=eng)
        t1 = time.perf_counter()
        geom = classes.geometry_struct(jm.params, jm.nx, jm.ny, jm.nz)
        dev = classes.build_model_fields(jm, geom, want_wide=False, want_vy=False)
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        eng.compute_y_bounds(dev)
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        jm._dev = dev
        lc = jm.flux_vs_time(np.linspace(0., 5., 32) * 31536000.0, np.geomspace(1e9, 5e10, 64))
        torch.cuda.synchronize()
        t5 = time.perf_counter()
        resident = [k for k in ("nd", "xi", "temp", "pf", "ts", "vy", "em0", "a0")
                    if getattr(dev, k) is not None]
        out.append({"round": rnd, "host_JetModel_ms": (t1 - t0) * 1e3,
                    "k4_enqueue_incl_allocation_ms": (t2 - t1) * 1e3,
                    "k4_wait_ms": (t3 - t2) * 1e3, "y_bounds_ms": (t4 - t3) * 1e3,
                    "construct_ms": (t4 - t0) * 1e3, "first_light_curve_ms": (t5 - t4) * 1e3,
                    "grid_sized_arrays_resident": resident,
                    "allocated_GB": torch.cuda.memory_allocated() / 1e9,
                    "flux0": float(lc[0, 32])})
        del jm, dev
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
