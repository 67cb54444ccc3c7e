"""Same-buffer A/B of the single-epoch cfg4 scan: burst factor from the LDS table (one y-range)
against the Gaussians (eight y-ranges), alternating on ONE allocation of a0 and ts."""
import json
import sys

import numpy as np

sys.path.insert(0, ".")
from rajepy_amd import engine as E            # noqa: E402
import bench                                  # noqa: E402

YEAR = bench.YEAR
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
eng = E.RTEngine(0)
shape = bench.CONFIGS[cfg][0]
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
out = {"config": cfg, "rounds": []}
for yr in (1.0, 2.5):
    ep = [yr * YEAR]
    for rnd in range(3):
        row = {"epoch_yr": yr}
        for name, flag in (("table", True), ("gaussians", False)):
            eng.use_chi_table = flag
            eng.time_ff_scan(f, bursts, ep, 0, reps=2, want_em=False, want_tavg=False)
            row[name + "_ms"] = eng.time_ff_scan(f, bursts, ep, 0, reps=20, want_em=False,
                                                 want_tavg=False)
            row[name + "_path"] = eng.last_scan_path()[0]
            if flag:
                row["table_intervals"] = eng.last_moment_shape[0]
        row["table_over_gaussians"] = row["table_ms"] / row["gaussians_ms"]
        out["rounds"].append(row)
n = shape[0] * shape[1] * shape[2]
best_t = min(r["table_ms"] for r in out["rounds"])
best_g = min(r["gaussians_ms"] for r in out["rounds"])
out["best"] = {"table_ms": best_t, "gaussians_ms": best_g,
               "table_frac_of_8TBs": 16.0 * n / (best_t * 1e-3) / 8e12,
               "gaussians_frac_of_8TBs": 16.0 * n / (best_g * 1e-3) / 8e12}
print(json.dumps(out))
