#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes into profiles/<tag>_pmc.json.

usage: pmc_summary.py <tag> <kernel-substring> <fetch_dir> <write_dir>

Each directory holds the csv output of one counter pass (FETCH_SIZE and WRITE_SIZE need
separate passes: they do not fit the TCC slots together, MI355X_MICROARCH.md "rocprofv3 PMC
slots").  Units and the gfx950 correction follow that guide's HBM section: both counters
are in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane)
coalesced streaming read, so it is doubled; WRITE_SIZE is exact for 16 B/lane stores.
"""
import csv
import glob
import json
import os
import sys


def per_launch(dirname, counter, kernel):
    vals = []
    for fn in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        with open(fn) as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") == counter and kernel in row.get("Kernel_Name", ""):
                    vals.append(float(row["Counter_Value"]))
    if not vals:
        raise SystemExit("no %s rows for %s under %s" % (counter, kernel, dirname))
    return sum(vals) / len(vals), len(vals)


def main():
    tag, kernel, fetch_dir, write_dir = sys.argv[1:5]
    fetch_kib, nf = per_launch(fetch_dir, "FETCH_SIZE", kernel)
    write_kib, nw = per_launch(write_dir, "WRITE_SIZE", kernel)
    rec = {
        "kernel": kernel,
        "FETCH_SIZE_KiB_raw": fetch_kib, "WRITE_SIZE_KiB_raw": write_kib,
        "launches_averaged": [nf, nw],
        "fetch_correction": "x2 (gfx950: FETCH_SIZE counts 128-B requests at 64 B for 16 B/lane "
                            "streaming reads)",
        "hbm_bytes_per_launch": (2.0 * fetch_kib + write_kib) * 1024.0,
    }
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles",
                       tag + "_pmc.json")
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
