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
            "roofline", "cpu_baseline", "sustained", "ranks_seen", "backend", "devices",
            "distinct_devices"}


@pytest.mark.parametrize("config", ["tiny", "tiny_rrl"])
def test_bench_prints_one_contract_line(config):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1",
                          "--steps", "3", "--warmup", "1", "--config", config,
                          "--cpu-seconds", "0.5", "--sustained-seconds", "0.2"] +
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
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source",
            "algorithmic_bytes", "algorithmic_bytes_8d"} <= set(rf)
    if config == "tiny":
        # round 5: the top-level figures of `roofline` are SURVEY 8(d)'s five-field byte model on
        # the kernel that moves it (one epoch from the five model fields); the kernel of the timed
        # step (tau layout: a0 and ts per cell, no EM map) sits beside it, priced on its own bytes
        sk = rf["timed_step_kernel"]
        assert rf["fields_streamed_per_cell"] == 5 and rf["algorithmic_bytes"] == rf["algorithmic_bytes_8d"]
        assert abs(rf["frac"] - rf["frac_8d"]) < 1e-12 and rf["ms_per_launch"] == rf["wide_ms_per_launch"]
        assert sk["fields_streamed_per_cell"] == 2 and sk["timed_step_asks_for_em"] is False
        assert sk["with_em"]["fields_streamed_per_cell"] == 3 and sk["with_em"]["ms_per_launch"] > 0
        assert abs(sk["frac"] - sk["achieved"] / rf["peak"]) < 1e-12
        assert rf["algorithmic_bytes_8d"] > sk["algorithmic_bytes"] > 0
        assert rf["tavg"]["ms"] > 0 and "layout" in r["config"] and "tau" in r["config"]["layout"]
        assert rf["per_model_state_ms"]["a0_and_em0_from_the_model_fields"] > 0
        assert "per-MODEL" in r["config"]["workload"] and r["value_from_model_fields"] > 0
        assert r["value_from_model_fields"] < r["value"]
        assert r["api_level"]["ms_per_step"] >= r["ms_per_step"] * 0.5
    assert "rehearsal" not in r and r["backend"] is None
    assert r["distinct_devices"] == 1 and len(r["devices"]) == 1 and r["devices"][0]["id"]
    assert r["sustained"]["steps"] >= 3 and r["sustained"]["ms_per_step"] > 0
    assert r["ranks_seen"] == 1 and r["storage_dtype"] == "f64"
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and rf["unit"] == "GB/s"
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    cb = r["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(cb)
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0
    # the host the CPU figures ran on, and the reference-proper provenance
    assert cb["host"]["os_cpu_count"] >= 1 and "lscpu_model" in cb["host"]
    assert cb["host"]["affinity"] is None or cb["host"]["affinity"] >= 1
    assert cb["reference_proper"]["unit"] == "Mvoxel-freq/s"
    if config == "tiny":          # the one-process-per-core leg
        ac = cb["all_cores"]
        assert "error" not in ac, ac
        assert ac["cores"] >= 1 and ac["value"] > 0 and ac["late_start"] is False
        assert ac["cores_available"] >= ac["cores"] and (ac["cap"] is None) == (
            ac["cores"] == ac["cores_available"])
    else:
        assert "all_cores" not in cb


def test_bench_refuses_mismatched_world_size():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2",
                          "--config", "tiny"], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0"))
    assert out.returncode == 2 and "WORLD_SIZE" in out.stderr


@pytest.mark.parametrize("config,sharding,scaling", [("tiny", "xslab", "strong"),
                                                     ("tiny5", "xslab", "strong"),
                                                     ("tiny_rrl", "xslab", "strong")])
def test_bench_launches_its_own_ranks(config, sharding, scaling):
    """`python bench.py --gpus 2` with no WORLD_SIZE starts two child ranks itself and relays
    rank 0's line (here both ranks share the one GPU of the box and talk over gloo; on an
    8-GPU node the same path runs one rank per GPU over RCCL)."""
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2",
                          "--steps", "3", "--warmup", "1", "--config", config, "--backend",
                          "gloo", "--share-gpu", "--sustained-seconds", "0.05"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["ranks_seen"] == 2 and r["scaling"] == scaling
    # two ranks on ONE device over gloo: plumbing only -- the line says so and has no `value`
    assert r["rehearsal"] is True and r["value"] is None and r["rehearsal_value"] > 0
    assert r["backend"] == "gloo" and r["distinct_devices"] == 1 and len(r["devices"]) == 2
    assert r["devices"][0]["id"] == r["devices"][1]["id"] and "strong_value" not in r
    assert r["rehearsal_strong_value"] > 0
    assert r["config"]["sharding"] == sharding
    assert r["n1"]["value"] > 0
    if config == "tiny":
        # round 5: the timed region is BASELINE's workload itself in x-slabs (strong); the step
        # that ends with the map gather, the weak epoch leg and the channel-sharded leg beside it
        assert set(r["legs"]) == {"strong_xslab_gather_maps", "strong_xslab_rank_local_host_maps",
                                  "weak_epochs", "channel_sharded"}
        assert r["legs"]["strong_xslab_rank_local_host_maps"]["bytes_to_host_per_rank"] > 0
        for name, leg in r["legs"].items():
            assert leg["scaling"] == ("weak" if name == "weak_epochs" else "strong")
            assert leg["value"] > 0 and leg["speedup_vs_n1"] > 0
        g = r["legs"]["strong_xslab_gather_maps"]
        assert g["gather_only_ms"] > 0 and g["compute_ms"] > 0 and g["bytes_into_root"] > 0
        assert g["ms_per_step"] >= 0.5 * g["compute_ms"] and "gather" in g["gather"]
        assert g["overlapped_sweep"]["ms_per_step"] > 0 and g["overlapped_sweep"]["value"] > 0
    elif config == "tiny5":
        assert "32 epoch(s) per step" in r["config"]["workload"]
        assert r["roofline"]["epochs_per_launch"] == 32 and r["roofline"]["grid_passes_per_launch"] == 1
    else:
        assert r["roofline"]["kernel"] == "rrl_scan_kernel" and r["roofline"]["voigt_evals_per_s"] > 0
