#!/usr/bin/env python3
"""One rank's share of an N-way x-slab split, timed stand-alone on ONE GPU (first look; the
tracked figure is bench.py's `rank_share` leg).  usage: python tools/rank_share_probe.py [cfg ...]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import torch
    from rajepy_amd import engine as E
    # arguments: config names, optionally "cfg5:8" = only the 8-way share of cfg5 (for a run under
    # rocprofv3 --kernel-trace --stats: the per-kernel averages are then that share's)
    only = {a.split(":")[0]: int(a.split(":")[1]) for a in sys.argv[1:] if ":" in a}
    cfgs = [a.split(":")[0] for a in sys.argv[1:]] or ["cfg4", "cfg2", "cfg5", "cfg3"]
    args = bench.parse([])
    eng = E.RTEngine(0)
    eng.cache_moments = False
    out = {}
    for cfg in cfgs:
        rows = {}
        for n in ((only[cfg],) if cfg in only else (1, 2, 4, 8)):
            w = bench.Workload(eng, args, "xslab" if n > 1 else "none", 0, n, config=cfg, lt=False)
            steps, warm = (3, 1) if cfg == "cfg3" else (50, 10)
            for _ in range(warm):
                w.local_step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                w.local_step()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / steps * 1e3
            k1 = None
            if not w.rrl:
                k1 = eng.time_ff_scan(w.fields, w.bursts, w.my_epochs, w.gmode, reps=20,
                                      want_em=False, want_tavg=False)
            rows[n] = {"lshape": list(w.pl["lshape"]), "ms_per_step": ms, "k1_ms": k1,
                       "path": eng.last_scan_path()[0]}
            w.release()
            del w
        t1 = rows[1]["ms_per_step"] if 1 in rows else float("nan")
        for n, r in rows.items():
            r["projected_speedup"] = t1 / r["ms_per_step"]
            r["efficiency"] = r["projected_speedup"] / n
        out[cfg] = rows
        print(cfg, json.dumps(rows), flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
