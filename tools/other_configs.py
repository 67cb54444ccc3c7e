#!/usr/bin/env python3
"""Runs the secondary bench configurations one after another (one process each) and writes
gpurun_out/<round>_other_configs_table.md: python tools/other_configs.py r02 [--no-cpu-baseline]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUNS = [("cfg2", "f64", {}, []), ("cfg5", "f64", {}, []), ("cfg3", "f64", {}, []),
        ("cfg4", "f64", {}, ["--em"]),
        ("cfg4", "f64", {}, ["--gaunt", "powerlaw"]),
        ("cfg4", "f64", {}, ["--layout", "compact", "--em"]),
        ("cfg4", "f64", {}, ["--layout", "wide", "--em"]),
        ("cfg4x8", "f64", {}, []),
        ("cfg4", "f32", {}, ["--em"]), ("cfg5", "f32", {}, []),
        ("cfg4", "f32", {}, ["--layout", "wide", "--em"])]


def main():
    tag = sys.argv[1]
    extra = sys.argv[2:]
    rows, lines = [], []
    for cfg, storage, env, more in RUNS:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", cfg, "--storage",
               storage, "--steps", "5", "--warmup", "2", "--sustained-seconds", "0.5"] + more + extra
        out = subprocess.run(cmd, env=dict(os.environ, **env), capture_output=True, text=True,
                             timeout=900)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if out.returncode or not line:
            sys.stderr.write(out.stderr)
            raise SystemExit("bench %s %s failed" % (cfg, storage))
        r = json.loads(line[-1])
        lines.append(line[-1])
        rf, cb = r["roofline"], r.get("cpu_baseline", {})
        rows.append("| %s_%s%s | %s | %s | %.3f | %.3e | %s | %.3f | %.0f | %.3f | %s |" % (
            cfg, storage, "".join("_" + m.lstrip("-") for m in more if m not in ("--layout", "--gaunt")),
            r["config"]["workload"], r["config"]["layout"],
            r["ms_per_step"], r["value"], rf["kernel"], rf["ms_per_launch"], rf["achieved"],
            rf["frac"], ("%.2f" % cb["value"]) if cb else "-"))
        print(rows[-1], flush=True)
    # written under gpurun_out/ (the only directory a GPU box hands back); copy to profiles/
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", tag + "_other_configs_table.md"), "w") as f:
        f.write("| config | workload | layout | ms/step | Mvoxel-freq/s | dominant kernel | "
                "ms/launch | alg. GB/s | frac of 8 TB/s | CPU oracle (1 core) Mvoxel-freq/s |\n")
        f.write("|---|---|---|---|---|---|---|---|---|---|\n")
        f.write("\n".join(rows) + "\n\n")
        for l in lines:
            f.write("```json\n" + l + "\n```\n")


if __name__ == "__main__":
    main()
