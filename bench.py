#!/usr/bin/env python3
"""Throughput benchmark of the line-of-sight RT hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json, the configuration its metric is quoted on): a 512x4096x512 grid of
dense synthetic fields (generated on the device), 256 continuum channels
1-50 GHz, with the example model's four ejection bursts.  One "step" = one pass of the hot
path over one epoch: K1 (grid scan -> base maps) + K2 (tau and flux cubes for all 256
channels + per-channel total flux), fields already resident in HBM.
N > 1 ranks: burst-time epochs shard embarrassingly -- every rank holds the grid (generated
on its own GPU from the same hash) and processes a different epoch per step; the only
exchange is an all_gather of the per-channel flux vectors (flux-vs-time) over RCCL.
Per-GPU work is fixed as N grows -> "scaling": "weak".

Prints ONE JSON line (rank 0) with the driver's contract fields plus `roofline` and
`cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
YEAR = 31536000.0

CONFIGS = {
    # name: (shape, n_chan) -- BASELINE.json configs[1] and configs[3]
    "cfg4": ((512, 4096, 512), 256),
    "cfg2": ((256, 1024, 256), 32),
    "tiny": ((16, 64, 64), 8),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default=os.environ.get("RJP_BENCH_CONFIG", "cfg4"),
                    choices=sorted(CONFIGS))
    ap.add_argument("--storage", default="f64", choices=("f64", "f32"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0,
                    help="target CPU time of the bounded cpu_baseline sample")
    return ap.parse_args()


def cpu_baseline(shape, freqs, seed, target_s):
    """The CPU oracle (oracle/rt_oracle.py, a literal NumPy restatement of the reference's
    per-channel re-streaming path) timed on a y-truncated block with the same n_x, n_z and a
    subset of the same channels: what Pipeline.execute issues per run, optical_depth_ff +
    flux_ff (classes.py:2411, 2423).  Single process: NumPy elementwise work uses one core."""
    from oracle import rt_oracle as orc
    from tests import gpu_util as U
    nx, ny, nz = shape
    nyb = 8
    nch = 4
    # ~0.2-0.35 us per cell-channel for the tau+flux pair on one core
    while nx * nyb * 2 * nz * nch * 0.3e-6 < target_s and nyb * 2 <= ny:
        nyb *= 2
    sub = (nx, nyb, nz)
    g = U.synth_host(sub, seed, 0)
    p = U.load_golden("cfg1_example")[2]
    p["ejection"] = U.example_bursts_params()
    p["grid"].update(n_x=nx, n_y=nyb, n_z=nz)
    jet = orc.OracleJet.from_fields(p, g["nd"], g["xi"], g["temp"], g["ff"], g["areas"],
                                    g["ts"], g["rr"], g["vy"])
    jet.time = 1.0 * YEAR
    sel = np.asarray(freqs)[np.linspace(0, len(freqs) - 1, nch).astype(int)]
    t0 = time.perf_counter()
    jet.optical_depth_ff(sel)
    jet.flux_ff(sel)
    dt = time.perf_counter() - t0
    ncell = nx * nyb * nz
    return {"value": ncell * nch / dt / 1e6, "unit": "Mvoxel-freq/s", "cores": 1,
            "kind": "port",
            "sample": "oracle optical_depth_ff+flux_ff on a %dx%dx%d y-truncated block of the "
                      "same synthetic grid x %d of the channels (%.1f s)" % (nx, nyb, nz, nch, dt)}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)"
                  % (args.gpus, world), file=sys.stderr)
        sys.exit(2)

    import torch
    import torch.distributed as dist
    from rajepy_amd import engine as E
    from rajepy_amd.maths import physics as ph
    from rajepy_amd.parallel import EpochShards, gather_flux_vs_time
    from tests import gpu_util as U            # burst parameters of the example model

    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local))
    eng = E.RTEngine(local)
    shape, nchan = CONFIGS[args.config]
    dtype = E.RJP_F64 if args.storage == "f64" else E.RJP_F32
    seed = 20240504
    ncell = shape[0] * shape[1] * shape[2]
    P = shape[0] * shape[2]

    fields = eng.synth_fields(shape, seed, 0, dtype, csize_au=0.5)
    ej = U.example_bursts_params()
    ss = 1.0
    red, blue = [], []
    for t0, hl, chi, which in zip(ej["t_0"], ej["hl"], ej["chi"], ej["which"]):
        sig = hl * YEAR * 2. / (2. * np.sqrt(2. * np.log(2.)))
        for jet, lst in (("R", red), ("B", blue)):
            if jet in str(which):
                lst.append((t0 * YEAR, (ss * chi - ss) / ss, sig))
    bursts = E.make_bursts(red, blue)

    freqs = np.geomspace(1e9, 5e10, nchan)
    gv = [ph.gff(nu, 1e4) for nu in freqs]
    ctau, cflux = E.ff_channel_coeffs(freqs, 0.5, 120., E.RJP_GFF_SCALAR, gv)

    # epochs: one per rank per step (weak scaling over the burst-time sweep)
    epochs = np.linspace(0., 5., world) * YEAR if world > 1 else np.array([1.0 * YEAR])
    shards = EpochShards(epochs, world)
    my_epochs = shards.local(rank)
    E_loc = len(my_epochs)

    sumA = eng._f64(E_loc, P)
    em = eng._f64(E_loc, P)
    tavg = eng._f64(P)
    tau = eng._f64(E_loc, nchan, P)
    flux = eng._f64(E_loc, nchan, P)
    ftot = eng._f64(E_loc, nchan)

    def step():
        eng.ff_scan(fields, bursts, my_epochs, E.RJP_GFF_SCALAR, out=(sumA, em, tavg))
        eng.ff_maps(sumA, tavg, ctau, cflux, out=(tau, flux, ftot))
        return gather_flux_vs_time(ftot, shards, rank) if world > 1 else ftot

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=eng.device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    ms_step = dt / args.steps * 1e3
    total_epochs = shards.n_epochs
    value = ncell * nchan * total_epochs / (ms_step * 1e-3) / 1e6

    # dominant kernel: K1 (ff_scan).  Live HIP-event timing on the launch stream.
    k1_ms = eng.time_ff_scan(fields, bursts, my_epochs, E.RJP_GFF_SCALAR, reps=5)
    alg_bytes = 5 * ncell * int(dtype) * 1 + E_loc * P * 2 * 8      # fields in, base maps out
    achieved = alg_bytes / (k1_ms * 1e-3) / 1e9
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "r01_ff_scan_pmc.json")
    if os.path.exists(pmc):
        try:
            rec = json.load(open(pmc))
            if rec.get("config") == args.config and rec.get("storage") == args.storage:
                traffic = rec.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": "ff_scan_kernel", "achieved": achieved,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "ms_per_launch": k1_ms, "algorithmic_bytes": alg_bytes}

    result = {
        "metric": "Mvoxel-freq/s", "value": value, "unit": "Mvoxel-freq/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%s: %dx%dx%d grid x %d continuum channels 1-50 GHz, %d epoch(s) "
                               "per step (one per GPU), K1 scan + K2 tau/flux cubes"
                               % ((args.config,) + shape + (nchan, total_epochs)),
                   "storage": args.storage, "sharding": "epochs" if world > 1 else "none",
                   "gather": "all_gather of flux-vs-time [E,F]" if world > 1 else "none"},
        "roofline": roofline,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(shape, freqs, seed, args.cpu_seconds)
    if rank == 0:
        chk = float(out.sum().item())
        result["checksum_flux_total_jy"] = chk
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
