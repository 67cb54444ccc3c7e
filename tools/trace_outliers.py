#!/usr/bin/env python3
"""Which launches of a kernel are the slow ones?  Reads a rocprofv3 --kernel-trace CSV
(<dir>/**/*kernel_trace.csv) and prints, per kernel name prefix, the launch-duration
distribution, the outliers (> 2 x the median) with their position in the launch sequence and
the kernels that ran right before each of them.
usage: tools/trace_outliers.py <trace-dir> [kernel-substring ...]"""
import csv
import glob
import json
import os
import sys


def main():
    d = sys.argv[1]
    subs = sys.argv[2:] or ["ff_scan_table_kernel", "ff_scan_table_wide_kernel"]
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        print("no kernel_trace.csv under", d)
        return 1
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    t0 = rows[0][0]
    out = {}
    for sub in subs:
        idx = [i for i, r in enumerate(rows) if sub in r[2]]
        if not idx:
            continue
        dur = sorted(rows[i][1] - rows[i][0] for i in idx)
        med = dur[len(dur) // 2]
        outl = []
        for k, i in enumerate(idx):
            dt = rows[i][1] - rows[i][0]
            if dt > 2 * med:
                prev = [(rows[j][2][:60], (rows[j][1] - rows[j][0]) / 1e6) for j in range(max(0, i - 3), i)]
                gap = (rows[i][0] - rows[i - 1][1]) / 1e6 if i else None
                outl.append({"launch_no": k, "of": len(idx), "ms": dt / 1e6,
                             "t_since_first_kernel_s": (rows[i][0] - t0) / 1e9,
                             "idle_gap_before_ms": gap, "previous_kernels": prev})
        keep = [x for x in dur if x <= 2 * med]
        out[sub] = {"launches": len(idx), "median_ms": med / 1e6, "mean_ms": sum(dur) / len(dur) / 1e6,
                    "mean_without_outliers_ms": sum(keep) / len(keep) / 1e6,
                    "max_ms": dur[-1] / 1e6, "n_outliers": len(outl), "outliers": outl[:40]}
    print(json.dumps(out, indent=1))
    return 0


if __name__ == "__main__":
    sys.exit(main())
