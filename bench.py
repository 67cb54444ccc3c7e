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
    # name: (shape, n_chan, n_epochs_total or None, kind) -- BASELINE.json configs[1..4]
    "cfg4": ((512, 4096, 512), 256, None, "continuum"),   # headline: metric's configuration
    "cfg2": ((256, 1024, 256), 32, None, "continuum"),
    "cfg5": ((512, 4096, 512), 64, 32, "continuum"),      # 64 ch x 32 epochs, epoch-fused passes
    "cfg3": ((512, 2048, 512), 256, None, "rrl"),         # H66a cube, LTE
    # capacity line: 8x the cells of cfg4 on ONE GPU, compact layout only (206 GB of the 288)
    "cfg4x8": ((1024, 8192, 1024), 256, None, "continuum"),
    "tiny": ((16, 64, 64), 8, None, "continuum"),
    "tiny_rrl": ((8, 64, 64), 40, None, "rrl"),
}


# the four ejection bursts of the reference's example model
# (files/example-model-params.py:51-54): peak time [yr], FWHM [yr], burst factor, jet(s)
EXAMPLE_BURSTS = {"t_0": [0.5, 0.75, 1., 2.], "hl": [0.15, 0.15, 0.45, 0.5],
                  "chi": [5., 5., 2.5, 10.], "which": ["R", "B", "B", "RB"]}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: long enough that clock ramp-up and first-touch effects of a fresh box stay in
    # the warm-up and the timed region averages over > 100 ms of device work
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default=os.environ.get("RJP_BENCH_CONFIG", "cfg4"),
                    choices=sorted(CONFIGS))
    ap.add_argument("--storage", default="f64", choices=("f64", "f32"))
    ap.add_argument("--gaunt", default="scalar", choices=("scalar", "powerlaw"),
                    help="scalar = the q_T == 0 branch (T = 1e4 K, one van Hoof Gaunt factor per "
                         "channel: the headline); powerlaw = the q_T != 0 branch "
                         "(T = 5e3 + 1.5e4 u, 11.95 T^0.15 nu^-0.1 per cell), SURVEY.md 8(d)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sharding", default="epochs", choices=("epochs", "xslab", "channels"),
                    help="N>1: epochs = one epoch of the full grid per rank (weak scaling, the "
                         "default); xslab = the ONE grid split into n_x/N row slabs (strong "
                         "scaling, per-channel fluxes all_reduced); channels = the north "
                         "star's frequency-sharded sweep: every rank scans the whole grid and "
                         "maps 1/N of the channels (continuum channels share the grid pass, so "
                         "this cannot scale the scan -- SURVEY finding 2)")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="gloo = rehearsal of the N>1 path (e.g. several ranks on one GPU)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal only: all ranks use cuda:0")
    ap.add_argument("--cpu-seconds", type=float, default=20.0,
                    help="target CPU time of the bounded cpu_baseline sample")
    ap.add_argument("--no-cpu-all-cores", action="store_true",
                    help="skip the one-process-per-core leg of cpu_baseline")
    return ap.parse_args()


def _oracle_jet(sub, seed, plaw=False):
    from oracle import rt_oracle as orc
    from tests import gpu_util as U
    g = U.synth_host(sub, seed, 1 if plaw else 0)
    p = U.load_golden("cfg1_example")[2]
    p["ejection"] = U.example_bursts_params()
    if plaw:
        p["power_laws"]["q_T"] = -0.5            # selects the power-law Gaunt branch
    p["grid"].update(n_x=sub[0], n_y=sub[1], n_z=sub[2])
    jet = orc.OracleJet.from_fields(p, g["nd"], g["xi"], g["temp"], g["ff"], g["areas"],
                                    g["ts"], g["rr"], g["vy"])
    jet.time = 1.0 * YEAR
    return jet


def cpu_baseline(shape, freqs, seed, target_s, rrl=None, all_cores=True, plaw=False):
    """The CPU oracle (oracle/rt_oracle.py, a literal NumPy restatement of the reference's
    per-channel re-streaming path) timed on a y-truncated block with the same n_x, n_z and a
    subset of the same channels: what Pipeline.execute issues per run -- optical_depth_ff +
    flux_ff (classes.py:2411, 2423), or optical_depth_rrl + flux_rrl(contsub=False)
    (classes.py:2437, 2450).  Single process: NumPy elementwise work uses one core.  The
    block is sized from a short calibration run so the sample costs about `target_s`."""
    nx, ny, nz = shape
    nch = 4
    sel = np.asarray(freqs)[np.linspace(0, len(freqs) - 1, nch).astype(int)]

    def run(jet):
        t0 = time.perf_counter()
        if rrl:
            jet.optical_depth_rrl(rrl, sel)
            jet.flux_rrl(rrl, sel, contsub=False)
        else:
            jet.optical_depth_ff(sel)
            jet.flux_ff(sel)
        return time.perf_counter() - t0

    nyb, dt = 2, 0.0
    for _ in range(4):                       # grow the block until it costs ~target_s
        dt = run(_oracle_jet((nx, nyb, nz), seed, plaw))
        if dt >= 0.6 * target_s or nyb >= ny:
            break
        nyb = int(min(ny, max(nyb + 1, nyb * min(target_s / dt, 64.0))))
    ncell = nx * nyb * nz
    what = "optical_depth_rrl+flux_rrl(contsub=False)" if rrl else "optical_depth_ff+flux_ff"
    out = {"value": ncell * nch / dt / 1e6, "unit": "Mvoxel-freq/s", "cores": 1,
           "kind": "port",
           "sample": "oracle %s on a %dx%dx%d y-truncated block of the same synthetic grid x "
                     "%d of the channels (%.1f s)" % (what, nx, nyb, nz, nch, dt)}
    if all_cores:
        out["all_cores"] = cpu_baseline_all_cores((nx, max(2, nyb // 4), nz), sel, seed, rrl,
                                                  plaw)
    return out


_CHILD = """
import sys, time, json
sys.path.insert(0, %(root)r)
import numpy as np
import bench
sub, seed, sel, rrl, t_start = %(sub)r, %(seed)r, np.array(%(sel)r), %(rrl)r, %(t_start)r
jet = bench._oracle_jet(sub, seed, %(plaw)r)
ready = time.time()
time.sleep(max(0.0, t_start - ready))
if rrl:
    jet.optical_depth_rrl(rrl, sel); jet.flux_rrl(rrl, sel, contsub=False)
else:
    jet.optical_depth_ff(sel); jet.flux_ff(sel)
print(json.dumps({"late": ready > t_start, "end": time.time() - t_start}))
"""


def cpu_baseline_all_cores(sub, sel, seed, rrl, plaw=False):
    """The same oracle calls in one process per host core of this job's CPU share, each on its
    own copy of a (smaller) block, started together: aggregate rate = what a channel-sharded
    process pool of the reference path would reach on this host."""
    import subprocess
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))           # one GPU's share of the box
    t_start = time.time() + 25.0             # children import, build their block, then wait
    code = _CHILD % {"root": ROOT, "sub": tuple(sub), "seed": seed, "sel": [float(x) for x in sel],
                     "rrl": rrl, "t_start": t_start, "plaw": bool(plaw)}
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE,
                              stderr=subprocess.DEVNULL, text=True, env=env)
             for _ in range(cores)]
    ends, late = [], False
    for pr in procs:
        try:
            o, _ = pr.communicate(timeout=300)
            r = json.loads(o.strip().splitlines()[-1])
            ends.append(r["end"])
            late |= r["late"]
        except Exception as exc:                      # a failed child voids the sample
            for q in procs:
                q.kill()
            return {"error": "%s: %s" % (type(exc).__name__, exc)}
    ncell = sub[0] * sub[1] * sub[2]
    return {"value": cores * ncell * len(sel) / max(ends) / 1e6, "unit": "Mvoxel-freq/s",
            "cores": cores, "late_start": late,
            "sample": "%d processes, each the same calls on its own %dx%dx%d block x %d "
                      "channels, started together (%.1f s)" % (cores, sub[0], sub[1], sub[2],
                                                               len(sel), max(ends))}


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
    from rajepy_amd import _lib, engine as E
    from rajepy_amd.maths import physics as ph, rrls
    from rajepy_amd.parallel import (ChannelShards, EpochShards, SlabShards,
                                     all_gather_blocks, gather_flux_vs_time)

    if args.share_gpu:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = E.RTEngine(local)
    shape, nchan, n_ep_cfg, kind = CONFIGS[args.config]
    dtype = E.RJP_F64 if args.storage == "f64" else E.RJP_F32
    seed = 20240504
    ncell = shape[0] * shape[1] * shape[2]          # cells of the whole grid
    rrl = kind == "rrl"
    xslab = args.sharding == "xslab" and world > 1
    chsh = args.sharding == "channels" and world > 1 and not rrl
    if xslab:
        x0, x1 = SlabShards(shape[0], world).bounds[rank]
        lshape = (x1 - x0, shape[1], shape[2])
        cell0 = x0 * shape[1] * shape[2]
    else:
        lshape, cell0 = shape, 0
    P = lshape[0] * lshape[2]
    ncell_loc = lshape[0] * lshape[1] * lshape[2]

    lean = args.config == "cfg4x8"            # generate em0, temp, ts only (24 B/cell)
    plaw = args.gaunt == "powerlaw"
    gmode = E.RJP_GFF_POWERLAW if plaw else E.RJP_GFF_SCALAR
    fields = eng.synth_fields(lshape, seed, 1 if plaw else 0, dtype, csize_au=0.5, with_vy=rrl,
                              cell0=cell0,
                              wide=not lean)
    ej = EXAMPLE_BURSTS
    red, blue = [], []
    for t0, hl, chi, which in zip(ej["t_0"], ej["hl"], ej["chi"], ej["which"]):
        sig = hl * YEAR * 2. / (2. * np.sqrt(2. * np.log(2.)))
        for jet, lst in (("R", red), ("B", blue)):
            if jet in str(which):
                lst.append((t0 * YEAR, chi - 1., sig))
    bursts = E.make_bursts(red, blue)

    if rrl:
        line_c = rrls.line_constants("H66a")
        line = _lib.Line(**line_c)
        freqs = line_c["nu_rest"] - nchan * 1e5 / 2. + 1e5 / 2. + np.arange(nchan) * 1e5
        cfl_rrl, hnu_k = E.rrl_channel_coeffs(freqs, 0.5, 120.)
    else:
        freqs = np.geomspace(1e9, 5e10, nchan)
    nchan_total = nchan
    if chsh:                      # this rank's slice of the channel list
        cshards = ChannelShards(freqs, world)
        freqs = cshards.local(rank)
        nchan = len(freqs)
    gv = None if plaw else [ph.gff(nu, 1e4) for nu in freqs]
    ctau, cflux = E.ff_channel_coeffs(freqs, 0.5, 120., gmode, gv)

    # epochs: cfg5 = its 32 epochs split over the ranks; otherwise one epoch per rank per
    # step (weak scaling over the burst-time sweep)
    if n_ep_cfg:
        epochs = np.linspace(0., 5., n_ep_cfg) * YEAR
    elif xslab or chsh:
        epochs = np.array([1.0 * YEAR])
    else:
        epochs = np.linspace(0., 5., world) * YEAR if world > 1 else np.array([1.0 * YEAR])
    shards = EpochShards(epochs, 1 if (xslab or chsh) else world)
    my_epochs = [float(t) for t in shards.local(0 if (xslab or chsh) else rank)]
    E_loc = len(my_epochs)

    sumA = eng._f64(E_loc, P)
    # flux-vs-time sweeps (cfg5) ask for no emission-measure maps, as parallel.sweep_flux_vs_time
    em = None if n_ep_cfg else eng._f64(E_loc, P)
    tavg = eng._f64(P)
    ftot = eng._f64(E_loc, nchan)
    if n_ep_cfg:
        tau = flux = None                   # flux-vs-time output: maps reduced on the device
    else:
        tau = eng._f64(E_loc, nchan, P)
        flux = eng._f64(E_loc, nchan, P)

    def step():
        eng.ff_scan(fields, bursts, my_epochs, gmode, out=(sumA, em, tavg))
        eng.ff_maps(sumA, tavg, ctau, cflux, out=(tau, flux, ftot))
        res = ftot
        if rrl:
            tau_rrl = eng.rrl_scan(fields, bursts, my_epochs[0], line, freqs)
            _, res = eng.rrl_maps(tau_rrl, tau.reshape(nchan, P), tavg, flux.reshape(nchan, P),
                                  cfl_rrl, hnu_k)
            res = res.reshape(1, nchan)
        if chsh:                  # [1, F/N] per rank -> [1, F]
            return all_gather_blocks(res, cshards, rank, axis=1)
        if xslab:                 # partial per-channel fluxes of this slab -> whole-map totals
            if args.backend == "nccl":
                dist.all_reduce(res)
            else:
                tmp = res.cpu()
                dist.all_reduce(tmp)
                res = tmp.to(res.device)
            return res
        return gather_flux_vs_time(res, shards, rank) if world > 1 else res

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
        tmax = torch.tensor([dt], dtype=torch.float64,
                            device=eng.device if args.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    ms_step = dt / args.steps * 1e3
    total_epochs = shards.n_epochs
    value = ncell * nchan_total * total_epochs / (ms_step * 1e-3) / 1e6

    # dominant kernel, timed live with HIP events on the launch stream
    if rrl:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(2):
            eng.rrl_scan(fields, bursts, my_epochs[0], line, freqs)
        ev1.record()
        torch.cuda.synchronize()
        k_ms = ev0.elapsed_time(ev1) / 2
        # K3 is vector-ALU bound by construction (SURVEY.md finding 3): report its HBM rate
        # against the HBM peak anyway and the Voigt-evaluation rate beside it
        alg_bytes = 6 * ncell_loc * int(dtype) + nchan * P * 8
        kname, extra = "rrl_scan_kernel", {"voigt_evals_per_s": ncell_loc * nchan / (k_ms * 1e-3)}
    else:
        k_ms = eng.time_ff_scan(fields, bursts, my_epochs, gmode, reps=5,
                                want_em=em is not None)
        # epoch tiles share a pass over the grid: 32 uniformly spaced epochs when no EM maps are
        # asked for, 16 with them, else 8 (f64
        # lanes) or 4 (f32 lanes)
        tile = (32 if (E_loc >= 32 and em is None) else 16 if E_loc >= 16 else
                (8 if args.storage == "f64" else 4))
        npass = -(-E_loc // tile) if E_loc > 1 else 1
        # fields K1 streams per cell: em0, temp, ts in the compact f64 layout (DESIGN.md
        # "Data layout"), else nd, xi, temp, pf, ts
        nfld = 3 if fields.em0 is not None else 5
        alg_bytes = npass * nfld * ncell_loc * int(dtype) + E_loc * P * 2 * 8
        kname, extra = "ff_scan_kernel", {"grid_passes_per_launch": npass,
                                          "fields_streamed_per_cell": nfld}
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "r01_%s_%s_pmc.json" % (args.config, args.storage))
    if os.path.exists(pmc):
        try:
            traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": kname, "achieved": achieved,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "ms_per_launch": k_ms, "algorithmic_bytes": alg_bytes}
    roofline.update(extra)

    result = {
        "metric": "Mvoxel-freq/s", "value": value, "unit": "Mvoxel-freq/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_step, "higher_is_better": True,
        "scaling": "strong" if (n_ep_cfg or xslab or chsh) else "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%s: %dx%dx%d grid x %d %s, %d epoch(s) per step, %s"
                               % ((args.config,) + shape + (
                                   nchan_total, "H66a channels of 100 kHz" if rrl else
                                   "continuum channels 1-50 GHz", total_epochs,
                                   "K3 RRL scan + K1/K2 continuum + line flux cube" if rrl else
                                   "K1 scan + K2 flux-vs-time" if n_ep_cfg else
                                   "K1 scan + K2 tau/flux cubes")),
                   "storage": args.storage, "gaunt": args.gaunt,
                   "layout": "compact (3 fields/cell)" if fields.em0 is not None else
                             "wide (5 fields/cell)",
                   "sharding": ("xslab" if xslab else "channels" if chsh else "epochs")
                   if world > 1 else "none",
                   "gather": ("all_reduce of per-channel fluxes [E,F]" if xslab else
                              "all_gather of per-channel fluxes along F" if chsh else
                              "all_gather of flux-vs-time [E,F]") if world > 1 else "none"},
        "roofline": roofline,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(shape, freqs, seed, args.cpu_seconds,
                                              rrl="H66a" if rrl else None,
                                              all_cores=not args.no_cpu_all_cores, plaw=plaw)
    if rank == 0:
        chk = float(out.sum().item())
        result["checksum_flux_total_jy"] = chk
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
