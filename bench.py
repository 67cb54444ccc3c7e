#!/usr/bin/env python3
"""Throughput benchmark of the line-of-sight RT hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1 without WORLD_SIZE in the environment: this process starts N fresh children itself
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...
bench.py <same flags>`), BEFORE anything touches the GPU, relays rank 0's single JSON line and
exits with the children's status.  Under torch.distributed.run (WORLD_SIZE set) it is one
rank of the job.

Workload (BASELINE.json, the configuration its metric is quoted on): a 512x4096x512 grid of
dense synthetic fields (generated on the device), 256 continuum channels
1-50 GHz, with the example model's four ejection bursts.  One "step" = one pass of the hot
path over one epoch: K1 (grid scan -> base maps) + K2 (tau and flux cubes for all 256
channels + per-channel total flux), fields already resident in HBM.  Default layout "tau": K1
streams the two fields that are left once everything independent of frequency and epoch has been
evaluated per model (a0 = (n x)^2 pf T^-1.5 and the launch time ts, 16 B/cell); the T_avg map
is per-model state as well (rjp_tavg, timed and reported, not part of a step).  `--em` adds the
emission-measure map to the step (a third field, em0: 24 B/cell); `--layout compact|wide` scan
3 / 5 fields per cell (the line always prices SURVEY 8(d)'s five-field byte model on the wide
kernel, live, as `frac_8d`).

N > 1 (no data-path collective, one gather/reduce per step over RCCL; "scaling": "strong"):
  every config timed region = BASELINE's workload itself, the ONE grid split into n_x/N row slabs
               (x-slabs: sightlines are independent), every rank runs the whole step on its
               rows, per-channel fluxes all_reduced.  For cfg4 / cfg2 the same line carries
               three more measured legs, each labelled: `strong_xslab_gather_maps` (the step
               ends with BASELINE config 4's "RCCL gather": the tau and flux slabs of every
               rank gathered into whole cubes on rank 0 -- compute, gather and total reported
               separately), `weak_epochs` (every rank holds the whole grid and scans another
               burst-time epoch per step, flux-vs-time vectors all_gathered) and
               `channel_sharded` (the north star's frequency-sharded sweep: every rank scans
               the whole grid and maps F/N channels -- continuum channels share the grid
               pass, SURVEY finding 2, so this cannot scale the scan).
N = 1, default config: `rank_share` prices rank 0's share of a 2- / 4- / 8-way split of cfg4, cfg5
and cfg3 as a stand-alone workload on this one GPU (projected strong-scaling speedup).

Prints ONE JSON line (rank 0) with the driver's contract fields plus `roofline`,
`cpu_baseline` (N = 1), `sustained`, `ranks_seen` and, for N > 1, `n1` and `legs`.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CPU_MEM_BUDGET = 96 * 2 ** 30    # host bytes the all-cores CPU leg may hold in total
MAX_CPU_PROCS = 96               # cap of the all-cores CPU leg (host memory, see there)
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
YEAR = 31536000.0
SEED = 20240504

CONFIGS = {
    # name: (shape, n_chan, n_epochs_total or None, kind) -- BASELINE.json configs[1..4]
    "cfg4": ((512, 4096, 512), 256, None, "continuum"),   # headline: metric's configuration
    "cfg2": ((256, 1024, 256), 32, None, "continuum"),
    "cfg5": ((512, 4096, 512), 64, 32, "continuum"),      # 64 ch x 32 epochs, epoch-fused passes
    "cfg3": ((512, 2048, 512), 256, None, "rrl"),         # H66a cube, LTE
    # capacity line: 8x the cells of cfg4 on ONE GPU, compact layout only (206 GB of the 288)
    "cfg4x8": ((1024, 8192, 1024), 256, None, "continuum"),
    "tiny": ((16, 64, 64), 8, None, "continuum"),
    "tiny5": ((16, 64, 64), 8, 32, "continuum"),
    "tiny_rrl": ((8, 64, 64), 40, None, "rrl"),
}


# the four ejection bursts of the reference's example model
# (files/example-model-params.py:51-54): peak time [yr], FWHM [yr], burst factor, jet(s)
EXAMPLE_BURSTS = {"t_0": [0.5, 0.75, 1., 2.], "hl": [0.15, 0.15, 0.45, 0.5],
                  "chi": [5., 5., 2.5, 10.], "which": ["R", "B", "B", "RB"]}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: long enough that clock ramp-up and first-touch effects of a fresh box stay in
    # the warm-up and the timed region averages over > 100 ms of device work
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default=os.environ.get("RJP_BENCH_CONFIG", "cfg4"),
                    choices=sorted(CONFIGS))
    ap.add_argument("--storage", default="f64", choices=("f64", "f32"))
    ap.add_argument("--layout", default="tau", choices=("tau", "compact", "wide"),
                    help="tau = K1 streams a0, ts (2 fields/cell, the default; em0 as a third "
                         "with --em); compact = em0, temp, ts (3 fields/cell); "
                         "wide = nd, xi, temp, pf, ts (5 fields/cell, SURVEY 8(d)'s byte model)")
    ap.add_argument("--em", action="store_true",
                    help="the step also produces the emission-measure map of its epoch "
                         "(JetModel.emission_measure); default: optical depths and fluxes only")
    ap.add_argument("--gaunt", default="scalar", choices=("scalar", "powerlaw"),
                    help="scalar = the q_T == 0 branch (T = 1e4 K, one van Hoof Gaunt factor per "
                         "channel: the headline); powerlaw = the q_T != 0 branch "
                         "(T = 5e3 + 1.5e4 u, 11.95 T^0.15 nu^-0.1 per cell), SURVEY.md 8(d)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sharding", default="auto",
                    choices=("auto", "epochs", "xslab", "channels"),
                    help="N>1.  auto = xslab (strong scaling of the ONE grid; single-epoch "
                         "continuum configs add the strong_xslab_gather_maps, weak_epochs and "
                         "channel_sharded legs to the same line).  epochs = one epoch of the "
                         "full grid per rank (weak scaling); "
                         "xslab = the ONE grid split into n_x/N row slabs (strong scaling, "
                         "per-channel fluxes all_reduced); channels = every rank scans the whole "
                         "grid and maps 1/N of the channels")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="N>1: skip the additional labelled legs and the N=1 reference")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="gloo = rehearsal of the N>1 path (e.g. several ranks on one GPU)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal only: all ranks use cuda:0")
    ap.add_argument("--rehearse-cpu", action="store_true",
                    help="no GPU, no kernels: drive launcher, shard planners and collectives "
                         "under gloo with placeholder per-rank vectors; the line says "
                         "\"rehearsal\": true and carries no throughput")
    ap.add_argument("--cpu-seconds", type=float, default=20.0,
                    help="target CPU time of the bounded cpu_baseline sample")
    ap.add_argument("--no-cpu-all-cores", action="store_true",
                    help="skip the one-process-per-core leg of cpu_baseline")
    ap.add_argument("--sustained-seconds", type=float, default=10.0,
                    help="length of the back-to-back leg after the contract timing (0 = skip): "
                         "long enough that clock / power droop would show and that a monitor "
                         "sampling the GPU every few seconds sees it busy")
    ap.add_argument("--lt", action="store_true",
                    help="cfg5-like sweeps: build the launch-time-ordered layout of (a0, ts) before "
                         "the timed region (per-model state like a0 itself, its build time is "
                         "reported): 12-32-epoch sweeps then keep their moments in registers")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="N = 1, default config only: skip the `configs` object (cfg2 / cfg3 / "
                         "cfg5 measured in the same process after the contract timing)")
    ap.add_argument("--no-api-level", action="store_true",
                    help="skip the PCIe-inclusive leg (step + copy of the cubes to pinned host)")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------
# self-launch (N > 1, plain `python bench.py --gpus N`)
# ---------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args, argv):
    """Start N ranks of this script under torch.distributed.run as CHILD processes (this
    process never touches a GPU), relay rank 0's JSON line.  Returns the exit status."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for raw in proc.stdout:
        if raw.startswith('{"metric"'):
            line = raw.strip()                 # the contract line: relayed once, at the end
        else:
            sys.stderr.write(raw)              # anything else a rank printed
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        print("bench.py: the ranks exited cleanly but printed no result line", file=sys.stderr)
        rc = 1
    return rc


# ---------------------------------------------------------------------------------------
# CPU baseline (the oracle, timed on the host; rank 0 at N = 1 only)
# ---------------------------------------------------------------------------------------
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
           "kind": "port", "host": host_description(),
           "sample": "oracle %s on a %dx%dx%d y-truncated block of the same synthetic grid x "
                     "%d of the channels (%.1f s)" % (what, nx, nyb, nz, nch, dt)}
    if all_cores:
        out["all_cores"] = cpu_baseline_all_cores((nx, max(2, nyb // 4), nz), sel, seed, rrl,
                                                  plaw)
    # provenance: the REFERENCE itself (unmodified, imported under its own pins) timed in the
    # build container for SURVEY.md section 6 -- another host, quoted, not measured here
    out["reference_proper"] = {
        "host": "build container, 8-core Intel Xeon @ 2.1 GHz, NumPy single-threaded",
        "optical_depth_ff+flux_ff, 4 channels, 128x512x128": 3.72,
        "optical_depth_rrl (9.26 s) + flux_rrl(contsub=False) (32.2 s), 4 channels, 128x512x128": 0.81,
        "unit": "Mvoxel-freq/s", "source": "BASELINE.md / SURVEY.md section 6"}
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
for _ in range(%(reps)d):
    if rrl:
        jet.optical_depth_rrl(rrl, sel); jet.flux_rrl(rrl, sel, contsub=False)
    else:
        jet.optical_depth_ff(sel); jet.flux_ff(sel)
print(json.dumps({"late": ready > t_start, "end": time.time() - t_start}))
"""


def host_description():
    """What the CPU figures ran on: logical CPUs of the box, the CPUs this job
    may use (affinity / cgroup quota) and the `lscpu` model string."""
    d = {"os_cpu_count": os.cpu_count()}
    try:
        d["affinity"] = len(os.sched_getaffinity(0))
    except AttributeError:
        d["affinity"] = None
    d["cgroup_cpus"] = None                # CPU quota of the container, if any (cgroup v2, v1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        d["cgroup_cpus"] = None if q == "max" else float(q) / float(per)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            d["cgroup_cpus"] = q / per if q > 0 else None
        except Exception:
            pass
    model = None
    try:
        txt = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        for line in txt.splitlines():
            if line.startswith("Model name"):
                model = line.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    if model is None:
        try:
            for line in open("/proc/cpuinfo"):
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
        except Exception:
            pass
    d["lscpu_model"] = model
    return d


def usable_cores():
    """Cores the all-cores leg may occupy: the affinity mask, cut to the cgroup quota when one
    is set."""
    h = host_description()
    n = h["affinity"] or h["os_cpu_count"] or 1
    if h["cgroup_cpus"]:
        n = min(n, max(1, int(h["cgroup_cpus"])))
    return max(1, n)


def cpu_baseline_all_cores(sub, sel, seed, rrl, plaw=False):
    """The same oracle calls in one process per host core of this job's CPU share, each on its
    own copy of a (smaller) block, started together: aggregate rate = what a channel-sharded
    process pool of the reference path would reach on this host."""
    avail = usable_cores()
    # every core the job may use (at most MAX_CPU_PROCS processes)
    cores = max(1, min(avail, MAX_CPU_PROCS))
    # host memory: every process holds its own block + the oracle's ~25 grid-sized arrays and
    # temporaries of it; the blocks shrink with the process count (CPU_MEM_BUDGET in total) and
    # the calls repeat so that a process still works for about as long
    rows = max(2, min(sub[1], int(CPU_MEM_BUDGET / (cores * sub[0] * sub[2] * 8 * 25))))
    reps = max(1, sub[1] // rows)
    sub = (sub[0], rows, sub[2])
    t_start = time.time() + 25.0 + 0.25 * cores   # children import, build their block, then wait
    code = _CHILD % {"root": ROOT, "sub": tuple(sub), "seed": seed, "sel": [float(x) for x in sel],
                     "rrl": rrl, "t_start": t_start, "plaw": bool(plaw), "reps": reps}
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
    return {"value": cores * reps * ncell * len(sel) / max(ends) / 1e6, "unit": "Mvoxel-freq/s",
            "cores": cores, "cores_available": avail,
            "cap": None if cores == avail else
            "%d processes (process count per job on the GPU boxes)" % MAX_CPU_PROCS,
            "late_start": late,
            "sample": "%d processes, each the same calls %d time(s) on its own %dx%dx%d block "
                      "x %d channels, started together (%.1f s)"
                      % (cores, reps, sub[0], sub[1], sub[2], len(sel), max(ends))}


# ---------------------------------------------------------------------------------------
# shard plan (pure: shared by the GPU path and the CPU rehearsal)
# ---------------------------------------------------------------------------------------
def resolve_sharding(args, world):
    """The sharding the TIMED region uses."""
    _, _, n_ep_cfg, kind = CONFIGS[args.config]
    if world == 1:
        return "none"
    if args.sharding != "auto":
        if args.sharding == "channels" and kind == "rrl":
            return "xslab"
        return args.sharding
    # BASELINE's workload itself on N GPUs: the ONE grid in n_x/N row slabs (strong scaling)
    return "xslab"


def plan(config, sharding, rank, world):
    """Local grid shape, first flat cell, local channel slice and local epochs of one rank."""
    from rajepy_amd.parallel import ChannelShards, EpochShards, SlabShards
    shape, nchan, n_ep_cfg, kind = CONFIGS[config]
    rrl = kind == "rrl"
    if rrl:
        from rajepy_amd.maths import rrls
        nu_rest = rrls.line_constants("H66a")["nu_rest"]
        freqs = nu_rest - nchan * 1e5 / 2. + 1e5 / 2. + np.arange(nchan) * 1e5
    else:
        freqs = np.geomspace(1e9, 5e10, nchan)
    lshape, cell0 = shape, 0
    if sharding in ("xslab", "xslab_gather_maps"):
        x0, x1 = SlabShards(shape[0], world).bounds[rank]
        lshape, cell0 = (x1 - x0, shape[1], shape[2]), x0 * shape[1] * shape[2]
    cshards = None
    my_freqs = freqs
    if sharding == "channels":
        cshards = ChannelShards(freqs, world)
        my_freqs = cshards.local(rank)
    # epochs: cfg5 = its 32 uniformly spaced epochs; otherwise one epoch per rank per step
    # under epoch sharding (weak scaling over the burst-time sweep), else the single epoch
    if n_ep_cfg:
        epochs = np.linspace(0., 5., n_ep_cfg) * YEAR
    elif sharding == "epochs":
        epochs = np.linspace(0., 5., world) * YEAR
    else:
        epochs = np.array([1.0 * YEAR])
    eshards = EpochShards(epochs, world if sharding == "epochs" else 1)
    my_epochs = [float(t) for t in eshards.local(rank if sharding == "epochs" else 0)]
    return {"shape": shape, "lshape": lshape, "cell0": cell0, "freqs": freqs,
            "my_freqs": my_freqs, "cshards": cshards, "eshards": eshards,
            "my_epochs": my_epochs, "rrl": rrl, "n_ep_cfg": n_ep_cfg, "sharding": sharding}


GATHER = {"none": "none", "epochs": "all_gather of flux-vs-time [E,F]",
          "xslab": "all_reduce of per-channel fluxes [E,F]",
          "xslab_gather_maps": "all_reduce of per-channel fluxes [E,F] + gather of the tau and "
                               "flux slabs into whole cubes [E,F,n_x,n_z] on rank 0",
          "channels": "all_gather of per-channel fluxes along F"}


def gather_cubes(tau, flux, pl, rank, world, out=None, async_op=False):
    """BASELINE config 4's "RCCL gather": this rank's tau / flux slabs [E, F, n_x/N, n_z] ->
    the whole cubes on rank 0 (None elsewhere).  `out`: rank 0's preallocated (tau, flux).
    `async_op`: returns the two handles of gathers in flight (`.wait()` -> cube or None)."""
    from rajepy_amd.parallel import SlabShards, gather_slabs_to_root
    slabs = SlabShards(pl["shape"][0], world)
    o_t, o_f = out if out is not None else (None, None)
    return (gather_slabs_to_root(tau, slabs, rank, 2, out=o_t, async_op=async_op),
            gather_slabs_to_root(flux, slabs, rank, 2, out=o_f, async_op=async_op))


def collect(res, pl, rank, world, backend, force=False):
    """The one collective of a step: per-rank flux vectors -> the whole job's [E, F].
    `force`: issue the collective in a one-rank group too (the RCCL test on a one-GPU box)."""
    import torch.distributed as dist
    from rajepy_amd.parallel import all_gather_blocks, gather_flux_vs_time
    sh = pl["sharding"]
    if (world == 1 and not force) or sh == "none":
        return res
    if sh == "xslab_gather_maps":
        sh = "xslab"
    if sh == "channels":                  # [E, F/N] per rank -> [E, F]
        return all_gather_blocks(res, pl["cshards"], rank, axis=1)
    if sh == "xslab":                     # partial per-channel fluxes of this slab -> totals
        if backend == "nccl" or not res.is_cuda:
            dist.all_reduce(res)
            return res
        tmp = res.cpu()
        dist.all_reduce(tmp)
        return tmp.to(res.device)
    return gather_flux_vs_time(res, pl["eshards"], rank)


# ---------------------------------------------------------------------------------------
# CPU rehearsal of the N > 1 plumbing (no GPU, no kernels, no throughput)
# ---------------------------------------------------------------------------------------
def rehearse_cpu(args, rank, world):
    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    legs = {}
    main_sh = resolve_sharding(args, world)
    order = [main_sh]
    _, _, n_ep_cfg, kind = CONFIGS[args.config]
    if world > 1 and main_sh == "xslab" and not (n_ep_cfg or kind == "rrl") and \
            not args.no_extra_legs:
        order += ["xslab_gather_maps", "epochs", "channels"]
    for sh in order:
        pl = plan(args.config, sh, rank, world)
        nx = pl["shape"][0]
        E_tot = pl["eshards"].n_epochs if sh == "epochs" else len(pl["my_epochs"])
        F_tot = len(pl["freqs"])
        # placeholder "flux" of epoch e, channel f, summed over the rows a rank owns
        full = (1.0 + np.arange(E_tot))[:, None] * 1e-3 + np.arange(F_tot)[None, :] * 1e-6
        if sh == "epochs":
            loc = full[pl["eshards"].slice(rank)]
        elif sh == "channels":
            loc = full[:, pl["cshards"].slice(rank)]
        elif sh in ("xslab", "xslab_gather_maps"):
            loc = full * (pl["lshape"][0] / nx)
        else:
            loc = full
        cube_ok = True
        for _ in range(args.warmup + args.steps):
            out = collect(torch.from_numpy(np.ascontiguousarray(loc)).clone(), pl, rank, world,
                          "gloo")
            if sh == "xslab_gather_maps":
                # placeholder cubes: every rank contributes its rows of arange-filled maps
                nz = pl["shape"][2]
                x0 = pl["cell0"] // (pl["shape"][1] * nz)
                whole = torch.arange(E_tot * F_tot * nx * nz, dtype=torch.float64).reshape(
                    E_tot, F_tot, nx, nz)
                mine = whole[:, :, x0:x0 + pl["lshape"][0]].contiguous()
                ct, cf = gather_cubes(mine, -mine, pl, rank, world)
                ha, hb = gather_cubes(mine, -mine, pl, rank, world, async_op=True)
                at, af = ha.wait(), hb.wait()
                if rank == 0:      # (the blocking gather and the asynchronous one)
                    cube_ok = cube_ok and bool(torch.equal(ct, whole) and torch.equal(cf, -whole) and
                                               torch.equal(at, whole) and torch.equal(af, -whole))
                else:
                    cube_ok = cube_ok and all(v is None for v in (ct, cf, at, af))
        ok = bool(np.allclose(out.numpy(), full, rtol=1e-12, atol=0)) and cube_ok
        legs[sh] = {"ok": ok, "shape": list(out.shape), "gather": GATHER[sh]}
    seen = torch.ones(1, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(seen)
        dist.barrier()
    result = {"metric": "Mvoxel-freq/s", "value": None, "unit": "Mvoxel-freq/s",
              "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None,
              "higher_is_better": True, "scaling": "weak" if main_sh == "epochs"
              else "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
              "rehearsal": True, "ranks_seen": int(seen.item()),
              "config": {"workload": args.config + " (CPU rehearsal of launcher, planners and "
                                                   "collectives; no kernels run)",
                         "sharding": main_sh, "gather": GATHER[main_sh]},
              "legs": legs}
    # every rank's verdict counts (rank 0 checks the gathered cubes, the others that they got none)
    allok = torch.tensor([1.0 if all(v["ok"] for v in legs.values()) else 0.0],
                         dtype=torch.float64)
    if world > 1:
        dist.all_reduce(allok, op=dist.ReduceOp.MIN)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0 if allok.item() == 1.0 and result["ranks_seen"] == world else 1


# ---------------------------------------------------------------------------------------
# the GPU workload of one rank under one sharding
# ---------------------------------------------------------------------------------------
class Workload:
    def __init__(self, eng, args, sharding, rank, world, config=None, lt=None):
        from rajepy_amd import _lib, engine as E
        from rajepy_amd.maths import physics as ph, rrls
        self.eng, self.args, self.rank, self.world = eng, args, rank, world
        config = config or args.config
        self.pl = pl = plan(config, sharding, rank, world)
        self.rrl = pl["rrl"]
        self.dtype = E.RJP_F64 if args.storage == "f64" else E.RJP_F32
        self.plaw = args.gaunt == "powerlaw"
        self.gmode = E.RJP_GFF_POWERLAW if self.plaw else E.RJP_GFF_SCALAR
        lshape = pl["lshape"]
        self.P = lshape[0] * lshape[2]
        self.ncell_loc = lshape[0] * lshape[1] * lshape[2]
        lean = config == "cfg4x8"                 # generate em0, temp, ts only (24 B/cell)
        tau = args.layout == "tau" and self.dtype == E.RJP_F64 and not (lean and args.em)
        self.fields = eng.synth_fields(lshape, SEED, 1 if self.plaw else 0, self.dtype,
                                       csize_au=0.5, with_vy=self.rrl, cell0=pl["cell0"],
                                       wide=not lean, tau_mode=self.gmode if tau else None,
                                       with_em0=not (lean and tau))
        self._em0 = self.fields.em0
        if args.layout == "wide":
            self.fields.em0 = None
        # the launch-time-ordered layout (epoch sweeps): per-model state, built once
        self.lt_info = None
        if (args.lt if lt is None else lt) and pl["n_ep_cfg"] and self.fields.a0 is not None:
            info = eng.build_lt(self.fields, 20)
            self.lt_info = {k: info[k] for k in ("K", "rows", "build_ms",
                                                 "build_with_allocation_ms", "bytes")}
            self.lt_info["padding"] = info["rows"] * 64 / float(self.ncell_loc)
        ej = EXAMPLE_BURSTS
        red, blue = [], []
        for t0, hl, chi, which in zip(ej["t_0"], ej["hl"], ej["chi"], ej["which"]):
            sig = hl * YEAR * 2. / (2. * np.sqrt(2. * np.log(2.)))
            for jet, lst in (("R", red), ("B", blue)):
                if jet in str(which):
                    lst.append((t0 * YEAR, chi - 1., sig))
        self.bursts = E.make_bursts(red, blue)
        freqs = pl["my_freqs"]
        self.nchan = len(freqs)
        if self.rrl:
            self.line = _lib.Line(**rrls.line_constants("H66a"))
            self.cfl_rrl, self.hnu_k = E.rrl_channel_coeffs(freqs, 0.5, 120.)
        gv = None if self.plaw else [ph.gff(nu, 1e4) for nu in freqs]
        self.ctau, self.cflux = E.ff_channel_coeffs(freqs, 0.5, 120., self.gmode, gv)
        # (the per-channel tables as ctypes arrays, converted once: a step is one library call)
        self.ctau, self.cflux = _lib.dbl_array(self.ctau), _lib.dbl_array(self.cflux)
        self.freqs = freqs
        self.my_epochs = pl["my_epochs"]
        E_loc = self.E_loc = len(self.my_epochs)
        self.sumA = eng._f64(E_loc, self.P)
        # the step produces optical depths and fluxes; the emission-measure map of the epoch
        # only with --em (flux-vs-time sweeps, cfg5, never ask for it, as
        # parallel.sweep_flux_vs_time)
        self.em = eng._f64(E_loc, self.P) if (args.em and not pl["n_ep_cfg"]) else None
        # T_avg depends on neither frequency nor epoch: per-model state, like the fields
        # (JetModel._model_tavg); its one-off pass is timed for the record
        import torch
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        eng.tavg(self.fields)
        ev0.record()
        self.tavg = eng.tavg(self.fields)
        ev1.record()
        torch.cuda.synchronize()
        self.tavg_ms = ev0.elapsed_time(ev1)
        self.ftot = eng._f64(E_loc, self.nchan)
        if pl["n_ep_cfg"]:
            self.tau = self.flux = None       # flux-vs-time output: maps reduced on the device
        else:
            self.tau = eng._f64(E_loc, self.nchan, self.P)
            self.flux = eng._f64(E_loc, self.nchan, self.P)
        # BASELINE config 4's gather: the step ends with the slabs of every rank laid into whole
        # cubes on rank 0 (preallocated there, like every other output of the step)
        self.gather_maps = sharding == "xslab_gather_maps"
        self.cubes = None
        if self.gather_maps and rank == 0:
            nx, _, nz = pl["shape"]
            self.cubes = (eng._f64(E_loc, self.nchan, nx, nz), eng._f64(E_loc, self.nchan, nx, nz))

    @property
    def total_epochs(self):
        pl = self.pl
        return pl["eshards"].n_epochs if pl["sharding"] == "epochs" else self.E_loc

    def local_step(self):
        """The hot path on this rank's shard; returns its per-channel fluxes [E_loc, F_loc]."""
        eng = self.eng
        # K1 + K2 from ONE call into the library (rjp_ff_step)
        eng.ff_step(self.fields, self.bursts, self.my_epochs, self.gmode, self.tavg, self.ctau,
                    self.cflux, out=(self.sumA, self.em, self.tau, self.flux, self.ftot))
        res = self.ftot
        if self.rrl:
            tau_rrl = eng.rrl_scan(self.fields, self.bursts, self.my_epochs[0], self.line,
                                   self.freqs)
            _, res = eng.rrl_maps(tau_rrl, self.tau.reshape(self.nchan, self.P), self.tavg,
                                  self.flux.reshape(self.nchan, self.P), self.cfl_rrl,
                                  self.hnu_k)
            res = res.reshape(1, self.nchan)
        return res

    def step(self):
        res = collect(self.local_step(), self.pl, self.rank, self.world, self.args.backend)
        if self.gather_maps:
            self.gather_only()
        return res

    def gather_only(self):
        """The map gather of a `xslab_gather_maps` step alone (timed on its own for the leg)."""
        lx, _, nz = self.pl["lshape"]
        shp = (self.E_loc, self.nchan, lx, nz)
        return gather_cubes(self.tau.view(shp), self.flux.view(shp), self.pl, self.rank,
                            self.world, out=self.cubes)

    def step_overlapped(self):
        """A sweep over epochs with the map gather of epoch i running beside the scan of epoch
        i + 1: the slabs are packed into send buffers at the end of a step, the collective is
        asynchronous, and the NEXT step waits for it (and lets the root lay the slabs into the
        cubes) only after its own compute has been enqueued."""
        res = collect(self.local_step(), self.pl, self.rank, self.world, self.args.backend)
        self.drain()
        lx, _, nz = self.pl["lshape"]
        shp = (self.E_loc, self.nchan, lx, nz)
        self._pending = gather_cubes(self.tau.view(shp), self.flux.view(shp), self.pl, self.rank,
                                     self.world, out=self.cubes, async_op=True)
        return res

    def drain(self):
        for h in getattr(self, "_pending", None) or ():
            h.wait()
        self._pending = None

    def release(self, to_driver=True):
        """Drop this workload's device buffers; `to_driver=False` leaves them in PyTorch's
        allocator cache (the next allocations reuse them instead of waiting for the driver)."""
        if self.fields is not None:
            self.fields.lt = None
            self.fields.mom_cache = None
        self.fields = self._em0 = self.sumA = self.em = self.tau = self.flux = self.cubes = None
        if to_driver:
            import torch
            torch.cuda.empty_cache()


# ---------------------------------------------------------------------------------------
# the other BASELINE configurations, measured in the same process (N = 1, default config)
# ---------------------------------------------------------------------------------------
def _latest_profile(pattern):
    """Newest committed profiles/<round>_<pattern> (rounds sort r04 > r03e > r03 ...)."""
    import glob
    hits = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_" + pattern)))
    return hits[-1] if hits else None


def measure_other_configs(eng, args, torch):
    """cfg2, cfg5 (warm and cold, LDS moments and launch-time-ordered layout) and cfg3 on this
    GPU, each with freshly generated fields, after the contract timing of the default config.
    Wall times are synchronised on both sides; kernel times come from HIP events on the launch
    stream.  Returns the `configs` object of the line."""
    from rajepy_amd import engine as E
    out = {}

    def wall(fn, steps, warm):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    def rate(cfg, ms, epochs):
        shp, nch = CONFIGS[cfg][0], CONFIGS[cfg][1]
        return shp[0] * shp[1] * shp[2] * nch * epochs / (ms * 1e-3) / 1e6

    # ---- cfg2: 256x1024x256 x 32 channels, one epoch -------------------------------------
    try:
        w = Workload(eng, args, "none", 0, 1, config="cfg2", lt=False)
        ms = wall(w.local_step, 300, 30)
        k1 = eng.time_ff_scan(w.fields, w.bursts, w.my_epochs, w.gmode, reps=50, want_em=False,
                              want_tavg=False)
        nf = w.fields.scan_fields(w.gmode, False)
        b = nf * w.ncell_loc * int(w.dtype) + w.P * 8
        out["cfg2"] = {"workload": "256x1024x256 x 32 continuum channels, one epoch per step "
                                   "(K1 scan + K2 tau/flux cubes)",
                       "ms_per_step": ms, "value": rate("cfg2", ms, 1),
                       "k1_ms_per_launch": k1, "k1_algorithmic_bytes": b,
                       "k1_frac": b / (k1 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                       "fields_streamed_per_cell": nf}
        w.release()
        del w
    except Exception as exc:                                   # a leg must never void the line
        out["cfg2"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}

    # ---- cfg5: 512x4096x512 x 64 channels x 32 epochs, flux-vs-time ------------------------
    try:
        w = Workload(eng, args, "none", 0, 1, config="cfg5", lt=False)
        E5 = len(w.my_epochs)
        alg = 2 * w.ncell_loc * 8 + E5 * w.P * 8          # a0 + ts once, E base maps
        alg_unfused = E5 * 5 * w.ncell_loc * 8 + E5 * w.P * 2 * 8
        orig = list(w.my_epochs)

        def leg():
            """first call of a NEW (bursts, epochs) request -- the coefficient tables are built
            and checked on the device inside it -- then the warm figure and the kernel time"""
            w.my_epochs = [t + 1.0 for t in orig]          # same sweep, one second later
            w.local_step()
            torch.cuda.synchronize()
            w.my_epochs = orig
            t0 = time.perf_counter()
            w.local_step()                                 # a new request again
            torch.cuda.synchronize()
            cold = (time.perf_counter() - t0) * 1e3
            tab = eng.last_table_build_ms()
            warm = wall(w.local_step, 20, 2)
            k1 = eng.time_ff_scan(w.fields, w.bursts, w.my_epochs, w.gmode, reps=10,
                                  want_em=False, want_tavg=False)
            path, err = eng.last_scan_path()
            return {"scan_path": path, "shape_bins_order": list(eng.last_moment_shape),
                    "worst_rel_err_of_the_expansion": err,
                    "ms_per_step": warm, "value": rate("cfg5", warm, E5),
                    "cold_new_request_ms": cold, "cold_over_warm": cold / warm,
                    "table_build_ms": tab, "k1_stage_ms": k1,
                    "frac": alg / (k1 * 1e-3) / 1e9 / HBM_PEAK_GBS}

        torch.cuda.synchronize()
        t0 = time.perf_counter()
        w.local_step()
        torch.cuda.synchronize()
        first_ever = (time.perf_counter() - t0) * 1e3
        lds = leg()
        lds["first_call_in_this_process_ms"] = first_ever  # + the kernels' code objects
        info = eng.build_lt(w.fields, 20)
        lt = leg()
        lt.update({"layout_build_ms": info["build_ms"],
                   "layout_build_with_allocation_ms": info["build_with_allocation_ms"],
                   "layout_bytes": info["bytes"],
                   "padding": info["rows"] * 64 / float(w.ncell_loc),
                   "sweeps_to_amortise_the_build": info["build_ms"] /
                   max(1e-9, lds["ms_per_step"] - lt["ms_per_step"])})
        # a model swept again and again (other epochs, other burst parameters): the moment maps of
        # a0 are model state -- with the engine's cache on, the second sweep onwards is the
        # contraction + the light-curve kernel alone (NOT a pass over the grid: never `value`)
        w.fields.lt = None
        eng.cache_moments = True
        for _ in range(2):
            w.local_step()
        cached_path = eng.last_scan_path()[0]
        cached_ms = wall(w.local_step, 20, 0)
        eng.cache_moments = False
        w.fields.mom_cache = None
        w.release(to_driver=False)
        # (no empty_cache() here: giving ~100 GB back to the driver makes the NEXT allocation
        # wait seconds for the frees -- tools/alloc_probe.py: ten 8.6 GB buffers take 0.25 s in
        # a fresh process and 2.7 s right after an empty_cache(); the model below reuses the
        # blocks PyTorch's allocator kept)
        # a FRESH model to its first light curve: the reference's example jet on a
        # 512x4096x512 grid of the same physical box, JetModel() -- K4 builds the fields on the
        # GPU -- then flux_vs_time(32 epochs, 64 channels).  A real jet fills ~1 % of its grid:
        # the library keeps the epoch tiles there (its cost model), the scans honour the occupied
        # y-ranges.
        fresh = None
        try:
            import tempfile
            from rajepy_amd import classes, logger
            meta = json.loads(str(np.load(os.path.join(ROOT, "tests", "golden",
                                                       "cfg1_example.npz"))["meta"]))
            par = meta["params"]
            for k in ("t_0", "hl", "chi", "which"):
                par["ejection"][k] = np.array(par["ejection"][k])
            par["geometry"].pop("mod_r_0", None)
            for k in ("q_n", "q_tau"):
                par["power_laws"].pop(k, None)
            par["properties"].pop("n_0", None)
            scale = 512.0 / par["grid"]["n_x"]
            par["grid"].update(n_x=512, n_y=4096, n_z=512, c_size=par["grid"]["c_size"] / scale)
            log = logger.Log(os.path.join(tempfile.mkdtemp(), "run.log"), verbose=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            jm = classes.JetModel(par, log=log, engine=eng)
            _ = jm.device_fields
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            lc = jm.flux_vs_time(np.asarray(orig), w.freqs)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            jm.flux_vs_time(np.asarray(orig), w.freqs)
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            fresh = {"model": "files/example-model-params.py jet on 512x4096x512 (cell size / %.2f)"
                              % scale,
                     "construct_ms": (t1 - t0) * 1e3,
                     "construct_note": "JetModel() to resident fields: FOUR 8.6 GB arrays -- a0, em0, "
                                       "temp, ts; nd / xi / pf / vy are built by the first call "
                                       "that needs them -- taken from PyTorch's cached blocks here "
                                       "(~1.1 s from the driver in a fresh process: "
                                       "profiles/r05_fresh_model_probe.json), K4, occupied y-ranges",
                     "grid_sized_arrays_resident": [
                         k for k in ("nd", "xi", "temp", "pf", "ts", "vy", "em0", "a0")
                         if getattr(jm.device_fields, k) is not None],
                     "first_light_curve_ms": (t2 - t1) * 1e3,
                     "second_light_curve_ms": (t3 - t2) * 1e3,
                     "scan_path": eng.last_scan_path()[0],
                     "occupied_fraction": jm.device_fields.occupied_cells /
                     float(jm.nx * jm.ny * jm.nz),
                     "flux_5GHz_like_first_epoch_jy": float(lc[0, len(w.freqs) // 2])}
            del jm
        except Exception as exc:
            fresh = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}
        torch.cuda.empty_cache()
        c5 = {"workload": "512x4096x512 x 64 continuum channels x 32 uniformly spaced epochs per "
                          "step, flux-vs-time output (K1 epoch sweep + light-curve kernel)",
              "algorithmic_bytes": alg, "algorithmic_bytes_8d_unfused": alg_unfused,
              "frac_is": "algorithmic_bytes (a0 + ts once + 32 base maps) / k1_stage_ms / 8 TB/s",
              "lds_moments": lds, "lt_layout": lt, "fresh_model_to_first_light_curve": fresh,
              "repeat_sweep_with_cached_moment_maps": {
                  "scan_path": cached_path, "ms_per_sweep": cached_ms,
                  "what": "second and later sweeps of ONE model with RTEngine.cache_moments (the "
                          "default of the Python layer): contraction of the kept moment maps "
                          "(2.7 GB) + light-curve kernel; no pass over the grid, so this is a "
                          "property of the workload, not a kernel figure"},
              "ms_per_step": lds["ms_per_step"], "value": lds["value"],
              "ms_per_step_on_prepared_layout": lt["ms_per_step"],
              "value_on_prepared_layout": lt["value"],
              "note": "lds_moments = what a model's FIRST sweep runs (no per-model preparation "
                      "beyond a0); lt_layout = sweeps of a model whose launch-time-ordered layout "
                      "was built once (layout_build_ms, like a0 itself)"}
        for tag, key in (("cfg5_f64_pmc.json", "traffic"), ("cfg5_f64_lt_pmc.json", "traffic_lt")):
            f = _latest_profile(tag)
            if f:
                try:
                    c5[key] = json.load(open(f)).get("hbm_bytes_per_launch")
                    c5[key + "_source"] = os.path.relpath(f, ROOT) + \
                        " (rocprofv3 --pmc passes of an earlier run, not measured in this run)"
                except Exception:
                    pass
        out["cfg5"] = c5
        w.release()
        del w, info
    except Exception as exc:
        out["cfg5"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}
    torch.cuda.empty_cache()

    # ---- cfg3: 512x2048x512 x 256 H66a channels --------------------------------------------
    try:
        w = Workload(eng, args, "none", 0, 1, config="cfg3", lt=False)
        ms = wall(w.local_step, 2, 1)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(2):
            eng.rrl_scan(w.fields, w.bursts, w.my_epochs[0], w.line, w.freqs)
        ev1.record()
        torch.cuda.synchronize()
        k3 = ev0.elapsed_time(ev1) / 2
        evals = w.ncell_loc * w.nchan / (k3 * 1e-3)
        c3 = {"workload": "512x2048x512 x 256 H66a channels of 100 kHz (K3 RRL scan + K1/K2 "
                          "continuum + line flux cube)",
              "ms_per_step": ms, "value": rate("cfg3", ms, 1), "k3_ms_per_launch": k3,
              "voigt_evals_per_s": evals,
              "fp64_vector_peak_lane_ops_per_s": 256 * 4 * 16 * 2.4e9,
              "hbm_frac": (6 * w.ncell_loc * 8 + w.nchan * w.P * 8) / (k3 * 1e-3) / 1e9 /
              HBM_PEAK_GBS}
        f = _latest_profile("cfg3_k3_sq.json")
        if f:
            try:
                ipe = json.load(open(f))["derived"]["valu_lane_insts_per_work_item"]
                c3["valu_insts_per_eval"] = ipe
                c3["valu_insts_per_eval_source"] = os.path.relpath(f, ROOT) + \
                    " (SQ counters of an earlier run)"
                c3["valu_issue_frac_of_256x4x16_lanes_at_2.4GHz"] = evals * ipe / (256 * 4 * 16 * 2.4e9)
            except Exception:
                pass
        f = _latest_profile("cfg3_f64_pmc.json")
        if f:
            try:
                pm = json.load(open(f))
                c3["traffic"] = pm["hbm_bytes_per_launch"]
                c3["traffic_raw_fetch"] = pm.get("hbm_bytes_per_launch_raw_fetch")
                c3["algorithmic_bytes"] = 6 * w.ncell_loc * 8 + w.nchan * w.P * 8
                c3["traffic_source"] = os.path.relpath(f, ROOT) + \
                    " (rocprofv3 --pmc passes of an earlier run; 8 B/lane reads in 64-byte " \
                    "runs are uncalibrated for the FETCH_SIZE rule: between raw and doubled; " \
                    "<= 5 % of the HBM peak either way -- this kernel is FP64-vector-bound)"
            except Exception:
                pass
        out["cfg3"] = c3
        w.release()
        del w
    except Exception as exc:
        out["cfg3"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}
    torch.cuda.empty_cache()
    return out


def measure_rank_share(eng, args, torch, base):
    """what ONE rank of an N-way split has to do, run stand-alone on this one
    GPU -- rank 0's plan of a 2- / 4- / 8-way x-slab split of cfg4, cfg5 (both sweep paths) and
    cfg3 (x-slabs and channel shards) -- with `projected_speedup` = the N = 1 step of the same
    config in this process / the share's step, and `efficiency` = that / N.  Compute only: the
    per-step collective (an all_reduce of [E, F] doubles, latency-bound) and, where maps are
    gathered, the link time of the gather are not in it -- `gather_bytes_into_root` says what the
    map gather of the split would move.  `base`: ms per N = 1 step per variant."""
    out = {"what": "rank 0's share of an N-way split as a stand-alone workload on ONE GPU "
                   "(compute only; the collectives of the real N-rank step are not in it)",
           "projected_speedup_is": "ms_per_step of the N = 1 workload in this process / "
                                   "ms_per_step of the share"}

    def wall(fn, steps, warm):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    variants = [("cfg4", "xslab", "xslab", False, (100, 10)),
                ("cfg5", "xslab_lds_moments", "xslab", False, (30, 3)),
                ("cfg5", "xslab_lt_layout", "xslab", True, (30, 3)),
                ("cfg3", "xslab", "xslab", False, (3, 1)),
                ("cfg3", "channels", "channels", False, (3, 1))]
    for cfg, tag, sh, lt, (steps, warm) in variants:
        t1 = base.get((cfg, tag))
        rows = {}
        try:
            for n in (2, 4, 8):
                w = Workload(eng, args, sh, 0, n, config=cfg, lt=lt)
                ms = wall(w.local_step, steps, warm)
                row = {"local_shape": list(w.pl["lshape"]), "local_channels": w.nchan,
                       "ms_per_step": ms}
                if w.rrl:
                    ev0, ev1 = (torch.cuda.Event(enable_timing=True),
                                torch.cuda.Event(enable_timing=True))
                    ev0.record()
                    eng.rrl_scan(w.fields, w.bursts, w.my_epochs[0], w.line, w.freqs)
                    ev1.record()
                    torch.cuda.synchronize()
                    row["k3_ms"] = ev0.elapsed_time(ev1)
                else:
                    row["k1_stage_ms"] = eng.time_ff_scan(w.fields, w.bursts, w.my_epochs, w.gmode,
                                                          reps=10, want_em=False, want_tavg=False)
                    row["scan_path"] = eng.last_scan_path()[0]
                if t1:
                    row["projected_speedup"] = t1 / ms
                    row["efficiency"] = t1 / ms / n
                if w.tau is not None and sh == "xslab":
                    row["gather_bytes_into_root"] = int((n - 1) * 2 * w.tau.numel() * 8)
                rows[str(n)] = row
                w.release(to_driver=False)
                del w
            out.setdefault(cfg, {})[tag] = {"n1_ms_per_step": t1, "split": rows}
        except Exception as exc:                              # a leg must never void the line
            out.setdefault(cfg, {})[tag] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}
        torch.cuda.empty_cache()
    return out


def device_identity(torch, index):
    """What physically distinguishes the GPU behind cuda:<index>: UUID and PCI address (plus the
    host name, so that the ids of a multi-node job could not collide)."""
    pr = torch.cuda.get_device_properties(index)
    uuid = str(getattr(pr, "uuid", "")) or None
    pci = None
    if hasattr(pr, "pci_bus_id"):
        pci = "%04x:%02x:%02x" % (int(getattr(pr, "pci_domain_id", 0)), int(pr.pci_bus_id),
                                  int(getattr(pr, "pci_device_id", 0)))
    host = socket.gethostname()
    ident = "%s/%s" % (host, uuid or pci or "cuda:%d" % index)
    return {"id": ident, "uuid": uuid, "pci": pci, "host": host, "index": int(index),
            "name": pr.name, "arch": getattr(pr, "gcnArchName", None)}


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and world_env is None:
        # plain `python bench.py --gpus N`: become the launcher.  Nothing above imported torch
        # or touched a GPU, and the ranks are children, not a re-exec of this process.
        sys.exit(self_launch(args, argv))
    rank = int(os.environ.get("RANK", "0"))
    world = int(world_env or "1")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    if args.rehearse_cpu:
        sys.exit(rehearse_cpu(args, rank, world))

    import torch
    import torch.distributed as dist
    from rajepy_amd import engine as E

    if args.share_gpu:
        if args.backend == "nccl" and world > 1:
            print("bench.py: --share-gpu puts several ranks on ONE device; RCCL needs one device "
                  "per rank (use --backend gloo for that rehearsal)", file=sys.stderr)
            sys.exit(2)
        local = 0
    ndev = torch.cuda.device_count()
    if local >= ndev:
        print("bench.py: rank %d needs cuda:%d but %d device(s) are visible (one process per "
              "GPU; --share-gpu --backend gloo rehearses several ranks on one)"
              % (rank, local, ndev), file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = E.RTEngine(local)
    # (every timed step streams the grid: the engine's cache of the launch-time moment maps,
    # which would turn cfg5's repeated sweeps into contractions, is reported on its own)
    eng.cache_moments = False
    shape, nchan_total, n_ep_cfg, kind = CONFIGS[args.config]
    ncell = shape[0] * shape[1] * shape[2]          # cells of the whole grid
    rrl = kind == "rrl"

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(dt):
        if world == 1:
            return dt
        t = torch.tensor([dt], dtype=torch.float64,
                         device=eng.device if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def timed(step, steps, warmup):
        out = None
        for _ in range(warmup):
            out = step()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = step()
        fence()
        return max_over_ranks(time.perf_counter() - t0), out

    def rate(ms_step, epochs):
        return ncell * nchan_total * epochs / (ms_step * 1e-3) / 1e6

    # ---- the contract's timed region ---------------------------------------------------
    sharding = resolve_sharding(args, world)
    wl = Workload(eng, args, sharding, rank, world)
    dt, out = timed(wl.step, args.steps, args.warmup)
    ms_step = dt / args.steps * 1e3
    total_epochs = wl.total_epochs
    value = rate(ms_step, total_epochs)
    chk = float(out.sum().item())

    # every rank really took part: an all_reduce of ones over the job's process group
    ranks_seen = 1
    if world > 1:
        ones = torch.ones(1, dtype=torch.float64,
                          device=eng.device if args.backend == "nccl" else "cpu")
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())
    # ... each on a GPU of its own: the physical identity of every rank's device, gathered over
    # the same group.  A line whose ranks shared devices is a rehearsal of the plumbing, not a
    # multi-GPU result: it says so and carries no `value`.
    me = device_identity(torch, local)
    devices = [me]
    if world > 1:
        devices = [None] * world
        dist.all_gather_object(devices, me)
    distinct_devices = len({d["id"] for d in devices})
    rehearsal = bool(args.share_gpu or distinct_devices < world or
                     (world > 1 and args.backend != "nccl"))
    if world > 1 and args.backend == "nccl" and distinct_devices < world:
        if rank == 0:
            print("bench.py: %d ranks over RCCL but only %d distinct device(s): %s"
                  % (world, distinct_devices, json.dumps(devices)), file=sys.stderr)
        dist.destroy_process_group()
        sys.exit(3)

    # ---- sustained leg: ~10 s of back-to-back steps (clock / power droop would show) ----
    sustained = None
    if args.sustained_seconds > 0:
        n_sus = int(min(50000, max(3, np.ceil(args.sustained_seconds * 1e3 / ms_step))))
        dts, _ = timed(wl.step, n_sus, 0)
        sustained = {"steps": n_sus, "seconds": dts, "ms_per_step": dts / n_sus * 1e3,
                     "value": rate(dts / n_sus * 1e3, total_epochs)}

    # ---- dominant kernel, timed live with HIP events on the launch stream ---------------
    fields = wl.fields
    dsz = int(wl.dtype)
    E_loc, P, ncell_loc = wl.E_loc, wl.P, wl.ncell_loc
    roof_extra = {}
    if rrl:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(2):
            eng.rrl_scan(fields, wl.bursts, wl.my_epochs[0], wl.line, wl.freqs)
        ev1.record()
        torch.cuda.synchronize()
        k_ms = ev0.elapsed_time(ev1) / 2
        # K3 is vector-ALU bound by construction (SURVEY.md finding 3): report its HBM rate
        # against the HBM peak anyway and the Voigt-evaluation rate beside it
        alg_bytes = 6 * ncell_loc * dsz + wl.nchan * P * 8
        alg_8d = alg_bytes
        kname = "rrl_scan_kernel"
        roof_extra = {"voigt_evals_per_s": ncell_loc * wl.nchan / (k_ms * 1e-3),
                      "fp64_vector_peak_lane_ops_per_s": 256 * 4 * 16 * 2.4e9,
                      "bound_note": "K3 is FP64-vector-bound by construction (no contraction: no "
                                    "MFMA): `frac` against HBM is low by design; the figure that "
                                    "prices it is valu_issue_frac below"}
        f = _latest_profile("cfg3_k3_sq.json")
        if f:
            try:
                ipe = json.load(open(f))["derived"]["valu_lane_insts_per_work_item"]
                roof_extra["valu_insts_per_eval"] = ipe
                roof_extra["valu_insts_per_eval_source"] = os.path.relpath(f, ROOT) + \
                    " (SQ counters of an earlier run)"
                roof_extra["valu_issue_frac_of_256x4x16_lanes_at_2.4GHz"] = \
                    roof_extra["voigt_evals_per_s"] * ipe / (256 * 4 * 16 * 2.4e9)
            except Exception:
                pass
    else:
        want_em = wl.em is not None
        k_ms = eng.time_ff_scan(fields, wl.bursts, wl.my_epochs, wl.gmode, reps=5 if n_ep_cfg else 30,
                                want_em=want_em, want_tavg=False)
        # epoch tiles share a pass over the grid: 32 uniformly spaced epochs, 16 below that,
        # else 8 (f64 lanes) or 4 (f32 lanes)
        # fields K1 streams per cell (DESIGN.md "Data layout"): a0, ts on the tau layout (+ em0
        # with EM maps), em0, temp, ts on the compact one, else nd, xi, temp, pf, ts
        nfld = fields.scan_fields(wl.gmode, want_em)
        # (tiles of >= 16 epochs: f64 storage on the tau / compact layouts)
        long_tiles = args.storage == "f64" and nfld != 5
        tile = ((32 if E_loc >= 32 else 16) if (E_loc >= 16 and long_tiles) else
                (8 if args.storage == "f64" else 4))
        npass = -(-E_loc // tile) if E_loc > 1 else 1
        base_maps = E_loc * P * (2 if want_em else 1) * 8
        alg_bytes = npass * nfld * ncell_loc * dsz + base_maps
        # SURVEY 8(d)'s byte model: 5 fields per cell (per grid pass of this launch), tau and
        # EM base maps
        alg_8d = npass * 5 * ncell_loc * dsz + E_loc * P * 2 * 8
        # tiles of >= 16 epochs on f64 fields run the LDS-DMA variant of the scan
        dma = E_loc >= 16 and long_tiles and fields.shape[2] % 2 == 0
        kname = "ff_scan_tile_kernel" if dma else "ff_scan_kernel"
        scan_path, mom_err = eng.last_scan_path()
        if scan_path == "table":
            # single epoch, tau layout: burst factor from a table in LDS, ONE y-range, the sums
            # written straight to the map (ff_scan_tab.hip)
            kname = "ff_scan_table_wide_kernel" if nfld == 5 else "ff_scan_table_kernel"
            roof_extra_table = {"intervals_per_jet": eng.last_moment_shape[0],
                                "degree": eng.last_moment_shape[1] - 1,
                                "bound_on_chi2_rel_err": mom_err}
        if scan_path in ("moments", "lt"):
            # the sweep went through the launch-time moments (ff_moments.hip / ff_lt.hip): ONE pass
            # over the grid for all epochs of the launch (+ a contraction over the moment maps
            # on the LDS path; fused on the launch-time-ordered layout)
            kname, npass = ("moments_kernel" if scan_path == "moments" else "lt_moments_kernel"), 1
        if scan_path in ("moments", "lt"):
            alg_bytes = npass * nfld * ncell_loc * dsz + base_maps
            alg_8d = npass * 5 * ncell_loc * dsz + E_loc * P * 2 * 8
        roof_extra = {"grid_passes_per_launch": npass, "fields_streamed_per_cell": nfld,
                      "epochs_per_launch": E_loc, "timed_step_asks_for_em": want_em,
                      "scan_path": scan_path,
                      "tavg": {"ms": wl.tavg_ms, "bytes": ncell_loc * dsz,
                               "what": "T_avg = nanmean_y(T > 0) is independent of frequency "
                                       "and epoch: one rjp_tavg pass per MODEL, not part of "
                                       "a step"}}
        if scan_path == "table":
            roof_extra["chi_table"] = roof_extra_table
        if scan_path == "lt":
            roof_extra["lt"] = dict(wl.lt_info or {}, **{
                "what": "launch-time-ordered layout (per-model state): every group of 64 "
                        "sightlines bucketed by (jet, launch-time bin); the sweep keeps a bin's "
                        "Chebyshev moments in registers and contracts them at the bin's end -- no "
                        "LDS atomics, no moment maps in HBM; `frac` is priced on the ALGORITHMIC "
                        "bytes (a0 + ts), the kernel reads `padding` x as many",
                "shape_bins_order": list(eng.last_moment_shape),
                "worst_rel_err_of_the_expansion": mom_err,
                "padded_bytes_per_launch": (wl.lt_info or {}).get("bytes")})
        if scan_path == "moments":
            roof_extra["moments"] = {
                "what": "sum_y a0 chi(t_e - ts)^2 as a convolution over launch time: one pass "
                        "accumulates 2 jets x K bins x N Chebyshev moments of a0 per sightline "
                        "(LDS atomics: they, not HBM, bound it), every epoch is a contraction "
                        "with host-built coefficient tables; the host picks the cheapest "
                        "(K, N) shape that passes its accuracy check",
                "shape_bins_order": list(eng.last_moment_shape),
                "worst_rel_err_of_the_expansion": mom_err,
                "moment_maps_bytes": 2 * eng.last_moment_shape[0] * eng.last_moment_shape[1]
                                     * P * 8}
        if n_ep_cfg:
            # 8(d) prices one grid pass PER EPOCH; the fused tiles make `npass` passes serve
            # E_loc epochs -- both figures, as 8(d) asks
            roof_extra["algorithmic_bytes_8d_unfused"] = (E_loc * 5 * ncell_loc * dsz +
                                                          E_loc * P * 2 * 8)
        if nfld == 2 and fields.em0 is not None:
            # the same launch WITH the emission-measure map of the epoch (a third field, em0)
            # (one untimed launch first: a kernel's code object is loaded on its first launch)
            eng.time_ff_scan(fields, wl.bursts, wl.my_epochs, wl.gmode, reps=1, want_em=True,
                             want_tavg=False)
            em_ms = eng.time_ff_scan(fields, wl.bursts, wl.my_epochs, wl.gmode, reps=3,
                                     want_em=True, want_tavg=False)
            em_path = eng.last_scan_path()[0]
            if em_path == "moments":
                # two moment passes: (a0, ts) for the optical-depth sums, (em0, ts) for the EM
                b3 = 2 * 2 * ncell_loc * dsz + E_loc * P * 2 * 8
                nf3 = "2 + 2 (two passes)"
            else:
                b3 = npass * 3 * ncell_loc * dsz + E_loc * P * 2 * 8
                nf3 = 3
            roof_extra["with_em"] = {"ms_per_launch": em_ms, "fields_streamed_per_cell": nf3,
                                     "scan_path": em_path, "algorithmic_bytes": b3,
                                     "frac": b3 / (em_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        if nfld < 5 and fields.nd is not None:
            # the 8(d) bytes are what the WIDE layout moves: time that kernel on the same
            # fields (with the EM and T_avg sums 8(d)'s model includes), and the one-off
            # passes that derive the scan fields from the wide ones
            em0, a0 = fields.em0, fields.a0
            fields.em0 = fields.a0 = None
            eng.time_ff_scan(fields, wl.bursts, wl.my_epochs, wl.gmode, reps=1, want_em=True)
            wide_ms = eng.time_ff_scan(fields, wl.bursts, wl.my_epochs, wl.gmode,
                                       reps=3 if n_ep_cfg else 20, want_em=True)
            wide_path = eng.last_scan_path()[0]
            # PLACEMENT: where the driver puts 43 GB of fields moves this kernel's time by a few
            # per cent between allocations -- more than anything else that
            # differs between two runs.  The same launch is therefore timed on three more FRESH
            # allocations of the five model fields (one at a time), and the headline figure of
            # `roofline` is the MEDIAN of the four placements, with all four beside it.
            placements = [wide_ms]
            if world == 1 and not n_ep_cfg:
                try:
                    for k in range(3):
                        f2 = eng.synth_fields(wl.pl["lshape"], SEED, 1 if wl.plaw else 0, wl.dtype,
                                              csize_au=0.5, cell0=wl.pl["cell0"], wide=True)
                        f2.em0 = f2.a0 = None
                        eng.time_ff_scan(f2, wl.bursts, wl.my_epochs, wl.gmode, reps=1, want_em=True)
                        placements.append(eng.time_ff_scan(f2, wl.bursts, wl.my_epochs, wl.gmode,
                                                           reps=10, want_em=True))
                        del f2
                except RuntimeError:                          # (no room for a second field set)
                    pass
            wide_ms = float(np.median(placements))
            lt_keep = fields.lt                  # (rebuilding em0 / a0 below drops derived state)

            def ev_ms(fn):
                ev0, ev1 = (torch.cuda.Event(enable_timing=True),
                            torch.cuda.Event(enable_timing=True))
                fn()
                torch.cuda.synchronize()
                ev0.record()
                fn()
                ev1.record()
                torch.cuda.synchronize()
                return ev0.elapsed_time(ev1)
            build_ms = ev_ms(lambda: eng.compact(fields))
            if a0 is not None:
                build_ms += ev_ms(lambda: eng.tau_layout(fields, wl.gmode))
            fields.em0, fields.a0 = em0, a0
            fields.lt = lt_keep
            roof_extra.update({
                "wide_ms_per_launch": wide_ms, "layout_build_ms": build_ms,
                "wide_scan_path": wide_path,
                "wide_placements_ms": placements,
                "wide_placements_what": "the same launch on %d allocations of the five model fields "
                                        "(the first: the timed step's own fields, 20 launches; the "
                                        "others fresh, 10 launches each); wide_ms_per_launch is "
                                        "their median" % len(placements),
                "frac_8d": alg_8d / (wide_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "frac_8d_kernel": "the wide-layout scan (5 fields/cell, tau + EM + T_avg sums; "
                                  "path: %s), timed live on the same fields" % wide_path,
                "first_epoch_from_wide_fields_ms": {"scan_layout": build_ms + wl.tavg_ms + k_ms,
                                                    "wide": wide_ms}})
        elif nfld == 5:
            roof_extra["frac_8d"] = alg_8d / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9
    traffic, traffic_source = None, None
    has_tau = wl.fields.a0 is not None
    nf = roof_extra.get("fields_streamed_per_cell")
    lay_tag = {2: "_tau2", 3: "_tau3" if has_tau else "", 5: "_wide5"}.get(nf, "")
    k1_layout = {2: "tau (a0, ts: 2 fields/cell)", 5: "wide (5 fields/cell)",
                 3: "tau + EM (a0, em0, ts: 3 fields/cell)" if has_tau
                 else "compact (3 fields/cell)"}.get(nf)
    layout_desc = k1_layout if not rrl else \
        "K3 reads the 6 wide fields; K1 %s" % ("tau" if has_tau else "compact")
    if roof_extra.get("scan_path") == "lt":
        lay_tag = "_lt"
    elif roof_extra.get("scan_path") == "moments":
        lay_tag = ""
    pf = _latest_profile("%s_%s%s_pmc.json" % (args.config, args.storage, lay_tag))
    if pf:
        try:
            traffic = json.load(open(pf)).get("hbm_bytes_per_launch")
            traffic_source = os.path.relpath(pf, ROOT) + \
                " (rocprofv3 --pmc passes of an earlier run of this command, not measured in " \
                "this run)"
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": kname, "achieved": achieved,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_source": traffic_source,
                "ms_per_launch": k_ms, "algorithmic_bytes": alg_bytes,
                "algorithmic_bytes_8d": alg_8d}
    roofline.update(roof_extra)
    if not rrl and not n_ep_cfg and "wide_ms_per_launch" in roofline:
        # the top-level figures of `roofline` are SURVEY 8(d)'s single figure --
        # the five-field byte model (n, x, T, path factor, launch time: 40 B/cell + the tau and EM
        # base maps) on the kernel that moves exactly those bytes, one epoch from the five MODEL
        # fields (tau + EM + T_avg sums in one pass), timed live in this run on the same fields.
        # The kernel of the TIMED step reads two derived per-model fields instead (a0 = (n x)^2 pf
        # T^-1.5 and ts: 16 B/cell; their one-off build and the T_avg pass are
        # `per_model_state_ms`) and sits beside it as `timed_step_kernel`, priced on ITS bytes.
        step_keys = ("kernel", "achieved", "frac", "ms_per_launch", "algorithmic_bytes",
                     "traffic", "traffic_source", "fields_streamed_per_cell", "scan_path",
                     "chi_table", "grid_passes_per_launch", "epochs_per_launch",
                     "timed_step_asks_for_em", "with_em")
        roofline["timed_step_kernel"] = {k: roofline.pop(k) for k in step_keys if k in roofline}
        w_ms = roofline["wide_ms_per_launch"]
        w_kernel = ("ff_scan_table_wide_kernel" if roofline["wide_scan_path"] == "table"
                    else "ff_scan_kernel (wide layout)")
        w_traffic, w_src = None, None
        pfw = _latest_profile("%s_%s_wide5_pmc.json" % (args.config, args.storage))
        if pfw:
            try:
                w_traffic = json.load(open(pfw)).get("hbm_bytes_per_launch")
                w_src = os.path.relpath(pfw, ROOT) + \
                    " (rocprofv3 --pmc passes of `bench.py --layout wide --em`, an earlier run)"
            except Exception:
                pass
        roofline.update({
            "kernel": w_kernel, "achieved": alg_8d / (w_ms * 1e-3) / 1e9,
            "frac": alg_8d / (w_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "ms_per_launch": w_ms,
            "algorithmic_bytes": alg_8d, "fields_streamed_per_cell": 5,
            "traffic": w_traffic, "traffic_source": w_src,
            "what": "SURVEY 8(d)'s byte model (5 fields x 8 B per cell + tau and EM base maps) / "
                    "the launch time of the kernel that moves those bytes -- one epoch from the "
                    "five model fields: optical-depth sums, emission measure and T_avg in one "
                    "pass -- timed live with HIP events on the launch stream in this run",
            "per_model_state_ms": {"a0_and_em0_from_the_model_fields": roofline["layout_build_ms"],
                                   "tavg_map": wl.tavg_ms,
                                   "what": "one-off passes per MODEL that the timed step's "
                                           "kernel relies on (K4 / the synthetic generator "
                                           "write a0 and em0 in their own pass instead)"}})
    if not rrl:
        # whole step = K1 + K2: the bytes both really move against the step's wall time
        k2_bytes = 0 if wl.tau is None else E_loc * wl.nchan * P * 16
        roofline["step"] = {"bytes": alg_bytes + k2_bytes, "per": "rank",
                            "achieved": (alg_bytes + k2_bytes) / (ms_step * 1e-3) / 1e9}
        roofline["step"]["frac"] = roofline["step"]["achieved"] / HBM_PEAK_GBS

    # ---- PCIe-inclusive figure: what a caller who wants NumPy cubes waits for ------------
    api_level = None
    if world == 1 and wl.tau is not None and not args.no_api_level:
        try:
            ht = torch.empty(wl.tau.shape, dtype=torch.float64, pin_memory=True)
            hf = torch.empty(wl.flux.shape, dtype=torch.float64, pin_memory=True)

            def api_step():
                wl.local_step()
                ht.copy_(wl.tau, non_blocking=True)
                hf.copy_(wl.flux, non_blocking=True)
                return wl.ftot
            n_api = 3
            dta, _ = timed(api_step, n_api, 1)
            nbytes = (ht.numel() + hf.numel()) * 8
            api_level = {"ms_per_step": dta / n_api * 1e3,
                         "value": rate(dta / n_api * 1e3, total_epochs),
                         "what": "step + device->pinned-host copy of the tau and flux cubes "
                                 "(%.2f GB), one after the other; never `value`" % (nbytes / 1e9)}
            # the copy alone = the PCIe-bound floor of any caller that wants both cubes on the host
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            ht.copy_(wl.tau, non_blocking=True)
            hf.copy_(wl.flux, non_blocking=True)
            ev1.record()
            torch.cuda.synchronize()
            api_level["copy_only_ms"] = ev0.elapsed_time(ev1)
            api_level["pcie_GBs"] = nbytes / (ev0.elapsed_time(ev1) * 1e-3) / 1e9
            # a sweep over epochs with the copy hidden behind the next epoch's compute: two sets of
            # device cubes and pinned buffers, the copy on its own stream (events both ways)
            ht2, hf2 = torch.empty_like(ht).pin_memory(), torch.empty_like(hf).pin_memory()
            dev = [(wl.tau, wl.flux), (torch.empty_like(wl.tau), torch.empty_like(wl.flux))]
            host = [(ht, hf), (ht2, hf2)]
            cstream = torch.cuda.Stream()
            done = [torch.cuda.Event(), torch.cuda.Event()]
            ready = [torch.cuda.Event(), torch.cuda.Event()]
            count = [0]

            def piped_step():
                k = count[0] % 2
                count[0] += 1
                torch.cuda.current_stream().wait_event(done[k])    # set k's last copy is out
                wl.tau, wl.flux = dev[k]
                wl.local_step()
                ready[k].record()
                with torch.cuda.stream(cstream):
                    cstream.wait_event(ready[k])
                    host[k][0].copy_(dev[k][0], non_blocking=True)
                    host[k][1].copy_(dev[k][1], non_blocking=True)
                    done[k].record(cstream)
                return wl.ftot
            n_p = 8
            dtp, _ = timed(piped_step, n_p, 2)
            wl.tau, wl.flux = dev[0]
            api_level["pipelined"] = {
                "ms_per_step": dtp / n_p * 1e3, "value": rate(dtp / n_p * 1e3, total_epochs),
                "what": "the same products per step in a sweep over epochs: epoch i+1 is computed "
                        "into a second set of device cubes while epoch i's cubes travel to pinned "
                        "host memory on a copy stream -- copy-bound (copy_only_ms is the floor)"}
            del ht, hf, ht2, hf2, dev, host
        except RuntimeError as exc:
            api_level = {"error": str(exc)[:200]}

    # ---- the other BASELINE configs under the same clock (N = 1, default config) ---------
    other = None
    if world == 1 and args.config == "cfg4" and not args.no_other_configs and not rehearsal:
        wl.release()
        other = measure_other_configs(eng, args, torch)

    # ---- N = 1: one rank's share of a 2- / 4- / 8-way split, priced on this GPU --------------
    rank_share = None
    if other is not None:
        base = {("cfg4", "xslab"): ms_step}
        c5, c3 = other.get("cfg5", {}), other.get("cfg3", {})
        if "lds_moments" in c5:
            base[("cfg5", "xslab_lds_moments")] = c5["lds_moments"]["ms_per_step"]
            base[("cfg5", "xslab_lt_layout")] = c5["lt_layout"]["ms_per_step"]
        if "ms_per_step" in c3:
            base[("cfg3", "xslab")] = base[("cfg3", "channels")] = c3["ms_per_step"]
        rank_share = measure_rank_share(eng, args, torch, base)

    # ---- N > 1: the other labelled legs and the N = 1 reference --------------------------
    legs, n1 = {}, None
    if world > 1 and not args.no_extra_legs:
        wl.release()
        if sharding == "xslab" and not (n_ep_cfg or rrl):
            # (1) BASELINE config 4 to the letter: the step ends with the gather of every rank's
            # tau and flux slabs into whole cubes on rank 0; the gather is also timed alone
            wg = Workload(eng, args, "xslab_gather_maps", rank, world)
            dtg, _ = timed(wg.step, args.steps, args.warmup)
            dto, _ = timed(wg.gather_only, args.steps, 1)
            msg, mso = dtg / args.steps * 1e3, dto / args.steps * 1e3
            legs["strong_xslab_gather_maps"] = {
                "sharding": "xslab", "scaling": "strong", "ms_per_step": msg,
                "value": rate(msg, wg.total_epochs), "gather": GATHER["xslab_gather_maps"],
                "compute_ms": ms_step, "gather_only_ms": mso,
                "bytes_into_root": int((world - 1) * 2 * wg.tau.numel() * 8),
                "epochs_per_step": wg.total_epochs,
                "what": "compute_ms = the timed region of this line (the same step without the "
                        "map gather); gather_only_ms = the two gathers + the root's copy of the "
                        "slabs into the cubes, back to back, nothing else in flight; "
                        "ms_per_step = both in one step"}
            # ... and as a SWEEP over epochs: the gather of epoch i overlaps the scan of epoch
            # i + 1 (per-step time -> max(compute, gather) instead of their sum)
            for _ in range(args.warmup):
                wg.step_overlapped()
            wg.drain()
            fence()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                wg.step_overlapped()
            wg.drain()
            fence()
            mso = max_over_ranks(time.perf_counter() - t0) / args.steps * 1e3
            legs["strong_xslab_gather_maps"]["overlapped_sweep"] = {
                "ms_per_step": mso, "value": rate(mso, wg.total_epochs),
                "what": "the same products per step in a sweep over epochs: the slabs of epoch i "
                        "are packed and gathered asynchronously while epoch i + 1 is scanned; the "
                        "next step waits for the gather (and the root lays the slabs into the "
                        "cubes) after enqueueing its own compute"}
            wg.release()
            # (1b) the alternative the sweep API is built around: the map slabs stay RANK-LOCAL
            # and every rank hands its own slab cubes to the host over its own PCIe link (as
            # Pipeline does per run: each rank writes the FITS files of its products) -- nothing
            # crosses xGMI but the [E, F] all_reduce.  PCIe-inclusive, never `value`.
            try:
                wh = Workload(eng, args, "xslab", rank, world)
                ht = torch.empty(wh.tau.shape, dtype=torch.float64, pin_memory=True)
                hf = torch.empty(wh.flux.shape, dtype=torch.float64, pin_memory=True)

                def host_step():
                    r = wh.step()
                    ht.copy_(wh.tau, non_blocking=True)
                    hf.copy_(wh.flux, non_blocking=True)
                    return r
                dth, _ = timed(host_step, args.steps, args.warmup)
                msh = dth / args.steps * 1e3
                legs["strong_xslab_rank_local_host_maps"] = {
                    "sharding": "xslab", "scaling": "strong", "ms_per_step": msh,
                    "value": rate(msh, wh.total_epochs),
                    "gather": GATHER["xslab"] + "; every rank copies its own tau / flux slabs to "
                              "pinned host memory (its own PCIe link)",
                    "bytes_to_host_per_rank": int(2 * wh.tau.numel() * 8),
                    "epochs_per_step": wh.total_epochs}
                del ht, hf
                wh.release()
            except RuntimeError as exc:                      # (pinned allocation refused)
                legs["strong_xslab_rank_local_host_maps"] = {"error": str(exc)[:200]}
            # (2) weak scaling over burst-time epochs, (3) the frequency-sharded sweep as named
            for name, sh, scal in (("weak_epochs", "epochs", "weak"),
                                   ("channel_sharded", "channels", "strong")):
                w2 = Workload(eng, args, sh, rank, world)
                dt2, _ = timed(w2.step, args.steps, args.warmup)
                ms2 = dt2 / args.steps * 1e3
                legs[name] = {"sharding": sh, "scaling": scal, "ms_per_step": ms2,
                              "value": rate(ms2, w2.total_epochs), "gather": GATHER[sh],
                              "epochs_per_step": w2.total_epochs}
                w2.release()
            legs["weak_epochs"]["note"] = (
                "every rank holds the whole grid (generated on its own GPU from the same hash) "
                "and scans another epoch per step: N epochs' worth of work per step")
            legs["channel_sharded"]["note"] = (
                "every rank scans the whole grid (continuum channels share the grid pass, "
                "SURVEY finding 2): only the map stage is divided")
        # the one-GPU workload on rank 0 alone, no collective: the N = 1 value of this box
        if rank == 0:
            w1 = Workload(eng, args, "none", 0, 1)
            for _ in range(args.warmup):
                w1.local_step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                w1.local_step()
            torch.cuda.synchronize()
            ms1 = (time.perf_counter() - t0) / args.steps * 1e3
            n1 = {"ms_per_step": ms1, "value": rate(ms1, w1.total_epochs),
                  "what": "the N=1 workload on rank 0 alone while the other ranks wait"}
            w1.release()
        dist.barrier()
        for leg in legs.values():
            # (a weak leg does N epochs per step: its ratio to the N = 1 value is its scaling)
            if n1 and "value" in leg:
                leg["speedup_vs_n1"] = leg["value"] / n1["value"]

    result = {
        "metric": "Mvoxel-freq/s", "value": value, "unit": "Mvoxel-freq/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_step, "higher_is_better": True,
        # (the same word at every N: the driver's N = 1, 2, 4, 8 values are one curve)
        "scaling": "weak" if sharding == "epochs" else "strong",
        "vs_baseline": None, "dtype": "f64", "storage_dtype": args.storage,
        "data": "synthetic",
        "config": {"workload": "%s: %dx%dx%d grid x %d %s, %d epoch(s) per step, %s"
                               % ((args.config,) + shape + (
                                   nchan_total, "H66a channels of 100 kHz" if rrl else
                                   "continuum channels 1-50 GHz", total_epochs,
                                   "K3 RRL scan + K1/K2 continuum + line flux cube" if rrl else
                                   "K1 scan + K2 flux-vs-time" if n_ep_cfg else
                                   "K1 scan + K2 tau/flux cubes" +
                                   (" + EM map" if args.em else ""))) + (
                           "; the scan reads the per-MODEL scan fields a0 = (n x)^2 pf T^-1.5|-1.35 "
                           "and ts (16 B/cell), T_avg is a per-model map: their one-off costs are "
                           "roofline.per_model_state_ms, and value_from_model_fields is the same "
                           "step from the five model fields (SURVEY 8(d))"
                           if (not rrl and not n_ep_cfg and roofline.get("timed_step_kernel")) else ""),
                   "storage": args.storage, "gaunt": args.gaunt,
                   "arithmetic": "f64 accumulation and transcendental functions; storage "
                                 "dtype of the 3-D fields as given",
                   "layout": layout_desc,
                   "sharding": sharding, "gather": GATHER[sharding]},
        "ranks_seen": ranks_seen,
        "backend": (args.backend if world > 1 else None),
        "devices": devices, "distinct_devices": distinct_devices,
        "roofline": roofline,
    }
    if world > 1 and result["scaling"] == "strong":
        # "how much faster is ONE model on N GPUs": `value` itself (x-slabs of the one grid)
        result["strong_value"] = value
        result["strong_speedup_vs_n1"] = value / n1["value"] if n1 else None
    if rehearsal:
        # several ranks on one device and / or gloo instead of RCCL: plumbing only
        result["rehearsal"] = True
        result["rehearsal_value"] = result["value"]
        result["value"] = None
        if "strong_value" in result:
            result["rehearsal_strong_value"] = result.pop("strong_value")
        result["rehearsal_why"] = ("--share-gpu" if args.share_gpu else
                                   "%d distinct device(s) for %d ranks" % (distinct_devices, world)
                                   if distinct_devices < world else
                                   "backend %s, not RCCL" % args.backend)
    if not rrl and "wide_ms_per_launch" in roofline:
        # SURVEY 8(d)'s workload literally: ONE epoch from the five model fields (nd, xi, temp,
        # pf, ts) -- the wide kernel's scan (tau + EM + T_avg sums) + this step's map stage
        ms_wide_step = roofline["wide_ms_per_launch"] + max(0.0, ms_step - k_ms)
        result["value_from_model_fields"] = rate(ms_wide_step, total_epochs)
        result["value_from_model_fields_what"] = (
            "%.3f ms = the wide-layout scan of the five model fields (%.3f ms, frac_8d) + the map "
            "stage of this step (%.3f ms): what one epoch costs without the per-model scan "
            "fields a0 / em0 and the per-model T_avg map" %
            (ms_wide_step, roofline["wide_ms_per_launch"], max(0.0, ms_step - k_ms)))
    tsk = roofline.get("timed_step_kernel", roofline)
    if not rrl and "with_em" in tsk and not n_ep_cfg:
        # the step as rounds 1-2 defined it (EM map of the epoch and T_avg inside the step)
        ms_r02 = tsk["with_em"]["ms_per_launch"] + wl.tavg_ms + max(0.0, ms_step - k_ms)
        result["value_r02_workload"] = rate(ms_r02, total_epochs)
        result["config"]["workload_version"] = (
            "r03+: the timed step produces tau and flux cubes; the emission-measure map (--em) and "
            "T_avg (per-model rjp_tavg) are outside it.  value_r02_workload re-prices the step "
            "with both inside (%.3f ms), comparable with BENCH_r02" % ms_r02)
    if other:
        result["configs"] = other
    if rank_share:
        result["rank_share"] = rank_share
    if sustained:
        result["sustained"] = sustained
    if api_level:
        result["api_level"] = api_level
    if legs:
        result["legs"] = legs
    if n1:
        result["n1"] = n1
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(shape, wl.pl["freqs"], SEED, args.cpu_seconds,
                                              rrl="H66a" if rrl else None,
                                              all_cores=not args.no_cpu_all_cores,
                                              plaw=wl.plaw)
    if rank == 0:
        result["checksum_flux_total_jy"] = chk
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
