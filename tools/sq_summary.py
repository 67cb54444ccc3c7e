#!/usr/bin/env python3
"""Summarise rocprofv3 SQ/GRBM counter passes of one kernel into profiles/<tag>_sq.json.

usage: sq_summary.py <tag> <kernel-substring> <work-items-per-launch> <pass_dir> [<pass_dir> ...]

Every pass directory holds the csv output of one `rocprofv3 --kernel-trace --pmc ...` run
(tools/prof_k3_k1.sh).  Counters are averaged over the launches of the named kernel.
Units (MI355X_MICROARCH.md, "rocprofv3 PMC slots" / "s_memtime tick vs SQ PMC units"):
SQ_INSTS_* count wave-instructions; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* / SQ_BUSY_*
count quad-cycles (x4 = shader cycles) summed over waves; GRBM_GUI_ACTIVE is summed over the
8 XCDs.  <work-items-per-launch> = (cell, channel) pairs for K3, (cell, epoch) pairs for K1:
the census is reported per work item (x64 lanes per wave-instruction).
"""
import csv
import glob
import json
import os
import sys


def collect(dirname, kernel):
    acc, dur = {}, []
    for fn in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        with open(fn) as f:
            for row in csv.DictReader(f):
                if kernel not in row.get("Kernel_Name", ""):
                    continue
                acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
                meta = {k: row[k] for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count",
                                            "LDS_Block_Size", "Scratch_Size", "Grid_Size",
                                            "Workgroup_Size")}
                acc["_meta"] = meta
    for fn in glob.glob(os.path.join(dirname, "**", "*kernel_trace.csv"), recursive=True):
        with open(fn) as f:
            for row in csv.DictReader(f):
                if kernel in row.get("Kernel_Name", ""):
                    dur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6)
    return acc, dur


def main():
    tag, kernel, items = sys.argv[1], sys.argv[2], float(sys.argv[3])
    rec = {"kernel_substring": kernel, "work_items_per_launch": items, "counters": {},
           "passes": []}
    for d in sys.argv[4:]:
        acc, dur = collect(d, kernel)
        meta = acc.pop("_meta", None)
        if meta:
            rec["kernel_resources"] = meta
        for k, v in acc.items():
            rec["counters"][k] = sum(v) / len(v)
        rec["passes"].append({"dir": os.path.basename(os.path.normpath(d)),
                              "launches": len(dur),
                              "ms_per_launch_under_pmc": sum(dur) / len(dur) if dur else None})
    c = rec["counters"]
    ms = [p["ms_per_launch_under_pmc"] for p in rec["passes"] if p["ms_per_launch_under_pmc"]]
    d = rec["derived"] = {}
    if "SQ_INSTS_VALU" in c:
        d["valu_wave_insts_per_launch"] = c["SQ_INSTS_VALU"]
        d["valu_lane_insts_per_work_item"] = c["SQ_INSTS_VALU"] * 64.0 / items
    if "SQ_INSTS_VALU" in c and "SQ_ACTIVE_INST_VALU" in c:
        d["cycles_per_valu_inst"] = 4.0 * c["SQ_ACTIVE_INST_VALU"] / c["SQ_INSTS_VALU"]
    if "SQ_WAVE_CYCLES" in c:
        for k in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
                  "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA"):
            if k in c:
                d[k + "_over_WAVE_CYCLES"] = c[k] / c["SQ_WAVE_CYCLES"]
    if "SQ_BUSY_CYCLES" in c and "SQ_ACTIVE_INST_VALU" in c:
        d["note_busy"] = "SQ_BUSY_CYCLES is per SE (quad-cycles the SQ had work)"
    for k in ("SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD",
              "SQ_INSTS_VMEM_WR", "SQ_INSTS_FLAT"):
        if k in c:
            d[k.lower() + "_wave_insts_per_work_item_x64"] = c[k] * 64.0 / items
    if "GRBM_GUI_ACTIVE" in c and ms:
        # guide, "DVFS give-back": effective clock = GRBM_GUI_ACTIVE / 8 / wall time
        grbm_ms = [p["ms_per_launch_under_pmc"] for p in rec["passes"] if "grbm" in p["dir"]]
        t = (grbm_ms or ms)[0] * 1e-3
        d["effective_clock_GHz"] = c["GRBM_GUI_ACTIVE"] / 8.0 / t / 1e9
    if "SQ_INSTS_VALU" in c and ms and "effective_clock_GHz" in d:
        # VALU issue capacity of the chip: 256 CUs x 4 SIMDs, one FP64 wave-instruction per
        # 4 cycles per SIMD (16 lanes/cycle)
        t = ms[0] * 1e-3
        cap = 256 * 4 * d["effective_clock_GHz"] * 1e9 / 4.0 * t
        d["valu_issue_utilisation_if_4cyc_each"] = c["SQ_INSTS_VALU"] / cap
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles",
                       tag + "_sq.json")
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec["derived"], indent=1))
    print(json.dumps(rec.get("kernel_resources")))
    print(json.dumps(rec["passes"]))


if __name__ == "__main__":
    main()
