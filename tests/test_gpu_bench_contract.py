"""The driver's bench.py contract, checked on the tiny configuration (GPU)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
            "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
            "roofline", "cpu_baseline"}


@pytest.mark.parametrize("config", ["tiny", "tiny_rrl"])
def test_bench_prints_one_contract_line(config):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1",
                          "--steps", "3", "--warmup", "1", "--config", config,
                          "--cpu-seconds", "0.5"] +
                         (["--no-cpu-all-cores"] if config == "tiny_rrl" else []),
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    r = json.loads(lines[0])
    assert REQUIRED <= set(r), REQUIRED - set(r)
    assert r["n_gpus"] == 1 and r["steps"] == 3 and r["warmup"] == 1
    assert r["metric"] == "Mvoxel-freq/s" and r["unit"] == "Mvoxel-freq/s"
    assert r["higher_is_better"] is True and r["vs_baseline"] is None
    assert r["dtype"] == "f64" and r["data"] == "synthetic" and "workload" in r["config"]
    assert "model" not in r["config"]
    assert r["value"] > 0 and r["ms_per_step"] > 0
    rf = r["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(rf)
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and rf["unit"] == "GB/s"
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    cb = r["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(cb)
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0
    if config == "tiny":          # the one-process-per-core leg
        ac = cb["all_cores"]
        assert "error" not in ac, ac
        assert ac["cores"] >= 1 and ac["value"] > 0 and ac["late_start"] is False
    else:
        assert "all_cores" not in cb


def test_bench_refuses_mismatched_world_size():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2",
                          "--config", "tiny"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 2 and "WORLD_SIZE" in out.stderr
