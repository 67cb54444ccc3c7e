#!/usr/bin/env python3
"""Soak of the launch-time range guard against FALSE alarms (round 5): random models -- shapes
large enough for the single-epoch table scan, 1-12 bursts in one or both jets, NaN sprinkles in
the launch times and the weights, occupied y-ranges, random epochs -- through the table scan, the
LDS moments and the launch-time-ordered layout.  Every result must equal the Gaussian scan / the
epoch tiles (which know no range) and the guard must stay down.
    python tools/guard_soak.py [n_cases]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def main():
    import torch
    from rajepy_amd import engine as E
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    eng = E.RTEngine(0)
    eng.cache_moments = False
    YEAR = 31536000.0
    worst = {"table": 0.0, "moments": 0.0, "lt": 0.0}
    taken = {"table": 0, "moments": 0, "lt": 0, "tiles": 0}
    for case in range(n_cases):
        rng = np.random.default_rng(9000 + case)
        big = case % 2 == 0
        if big:
            shape = (int(rng.integers(64, 100)), int(rng.integers(64, 160)),
                     2 * int(rng.integers(256, 330)))
        else:
            shape = (int(rng.integers(2, 9)), int(rng.integers(64, 200)), 2 * int(rng.integers(8, 40)))
        mode = int(rng.integers(0, 2))
        f = eng.synth_fields(shape, 5000 + case, mode, E.RJP_F64, csize_au=0.5, wide=False,
                             tau_mode=mode)
        g = torch.Generator(device=eng.device)
        g.manual_seed(case)
        for t in (f.ts, f.a0):
            m = torch.rand(f.ncells, device=eng.device, generator=g) < 0.02
            t[m] = float("nan")
        if case % 3 == 0:
            eng.compute_y_bounds(f)
        nb = int(rng.integers(1, 13))
        jets = rng.choice(["R", "B", "RB"], size=nb) if case % 4 else np.array(["R"] * nb)
        red, blue = [], []
        for j in range(nb):
            b = (float(rng.uniform(-0.5, 5.5)) * YEAR, float(rng.uniform(0.3, 11.0)),
                 float(rng.uniform(0.12, 1.0)) * YEAR / 1.1774)
            if "R" in str(jets[j]):
                red.append(b)
            if "B" in str(jets[j]):
                blue.append(b)
        bursts = E.make_bursts(red, blue)
        # single epoch: table (big maps) against the Gaussians
        ep1 = [float(rng.uniform(-0.5, 6.0)) * YEAR]
        eng.use_chi_table = True
        a = eng.ff_scan(f, bursts, ep1, mode, want_em=False, want_tavg=False)[0].clone()
        p1 = eng.last_scan_path()[0]
        eng.use_chi_table = False
        b_ = eng.ff_scan(f, bursts, ep1, mode, want_em=False, want_tavg=False)[0]
        eng.use_chi_table = True
        assert not eng.range_guard(), ("guard raised", case, p1)
        taken[p1] += 1
        ok = (a == b_) | ((a - b_).abs() <= 3e-12 * b_.abs())
        assert bool(ok.all()), (case, p1, float(((a - b_).abs() / b_.abs()).nan_to_num().max()))
        if p1 == "table":
            worst["table"] = max(worst["table"], float(((a - b_).abs() / b_.abs()).nan_to_num().max()))
        # sweeps: LDS moments and the layout against the tiles
        ep = [float(x) * YEAR for x in np.linspace(rng.uniform(0., 1.), rng.uniform(3., 5.), 16)]
        eng.use_moments = False
        til = eng.ff_scan(f, bursts, ep, mode, want_em=False, want_tavg=False)[0].clone()
        eng.use_moments = True
        eng.force_moments = True
        for name in ("moments", "lt"):
            if name == "lt":
                eng.build_lt(f, int(rng.choice([16, 20, 32])))
            got = eng.ff_scan(f, bursts, ep, mode, want_em=False, want_tavg=False)[0]
            path = eng.last_scan_path()[0]
            assert not eng.range_guard(), ("guard raised", case, name, path)
            taken[path] += 1
            if path in ("moments", "lt"):
                rel = float(((got - til).abs() / til.abs()).nan_to_num().max())
                assert rel < 1e-10, (case, path, rel)
                worst[path] = max(worst[path], rel)
        f.lt = None
        eng.force_moments = False
        del f, a, b_, til, got
    print("cases", n_cases, "paths taken", taken, "worst rel. difference", worst,
          "-- guard never raised")


if __name__ == "__main__":
    main()
