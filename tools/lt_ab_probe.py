"""Same-buffer A/B of cfg5's 32-epoch sweep: epoch tiles / LDS moments / launch-time-ordered layout,
all on ONE allocation of a0 and ts in one process (the scan's time depends on where the driver
places the fields by more than some of these differences), plus the one-off costs: the layout
build and the cold first call of a (bursts, epochs) request (coefficient tables built and checked
on the device).  python tools/lt_ab_probe.py [K ...]"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from rajepy_amd import engine as E            # noqa: E402
import bench                                  # noqa: E402

YEAR = 31536000.0


def main():
    Ks = [int(a) for a in sys.argv[1:]] or [32]
    eng = E.RTEngine(0)
    eng.cache_moments = False      # every sweep here runs its own pass over the grid
    shape = (512, 4096, 512)
    f = eng.synth_fields(shape, bench.SEED, 0, E.RJP_F64, csize_au=0.5, wide=False,
                         tau_mode=E.RJP_GFF_SCALAR)
    ej = bench.EXAMPLE_BURSTS
    red, blue = [], []
    for t0, hl, chi, which in zip(ej["t_0"], ej["hl"], ej["chi"], ej["which"]):
        sig = hl * YEAR * 2. / (2. * np.sqrt(2. * np.log(2.)))
        for jet, lst in (("R", red), ("B", blue)):
            if jet in str(which):
                lst.append((t0 * YEAR, chi - 1., sig))
    bursts = E.make_bursts(red, blue)
    ep = [float(t) for t in np.linspace(0., 5., 32) * YEAR]
    out = {"shape": shape, "epochs": 32}

    def timed(reps=10):
        return eng.time_ff_scan(f, bursts, ep, E.RJP_GFF_SCALAR, reps=reps, want_em=False,
                                want_tavg=False)

    def cold_call(ep_shift):
        """Wall time of the FIRST scan of a new epochs request (synchronised), and of the same
        request again."""
        e2 = [t + ep_shift for t in ep]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.ff_scan(f, bursts, e2, E.RJP_GFF_SCALAR, want_em=False, want_tavg=False)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        eng.ff_scan(f, bursts, e2, E.RJP_GFF_SCALAR, want_em=False, want_tavg=False)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        return {"first_ms": (t1 - t0) * 1e3, "again_ms": (t2 - t1) * 1e3,
                "table_build_ms": eng.last_table_build_ms(), "path": eng.last_scan_path()[0],
                "shape_kn": eng.last_moment_shape}

    eng.use_moments = False
    timed(2)
    out["tiles_ms"] = timed()
    eng.use_moments, eng.use_lt = True, False
    timed(2)
    out["lds_moments_ms"] = timed()
    out["lds_moments_path"] = eng.last_scan_path()[0]
    out["lds_moments_shape"] = eng.last_moment_shape
    out["lds_moments_cold"] = cold_call(1234.5)
    eng.use_lt = True
    out["lt"] = {}
    for K in Ks:
        lt = eng.build_lt(f, K)
        timed(2)
        ms = timed()
        path = eng.last_scan_path()
        d = {"build_ms": lt["build_ms"], "rows": lt["rows"],
             "padding": lt["rows"] * 64 / float(f.ncells), "layout_GB": lt["bytes"] / 1e9,
             "sweep_ms": ms, "path": path[0], "err": path[1], "shape_kn": eng.last_moment_shape,
             "frac_of_8TBs_on_algorithmic_bytes": f.ncells * 16 / (ms * 1e-3) / 8e12,
             "cold": cold_call(777.0 + K)}
        out["lt"][str(K)] = d
        f.lt = None
        del lt
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
