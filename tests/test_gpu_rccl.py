"""RCCL under test on a one-GPU box: a fresh child process initialises the `nccl` backend
(= RCCL on ROCm) with world_size 1 on cuda:0 and pushes EVERY collective the product issues
through it on device tensors -- bench.collect for the epoch / x-slab / channel shardings,
parallel.all_gather_blocks / gather_to_root / gather_flux_vs_time, the x-slab sweep's all_reduce,
JetModel.flux_vs_time and a whole Pipeline.execute (all_gather_object + barrier) inside the
group.  (Two RCCL ranks cannot share one device, so rank counts above 1 are rehearsed over gloo
elsewhere; what this test pins is that the code the ranks run meets RCCL at all.)  The reference
has no counterpart: it is single-process."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent('''
    import json, os, sys
    sys.path.insert(0, %(root)r)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=%(port)r, RANK="0", WORLD_SIZE="1",
                      LOCAL_RANK="0")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    out = {"backend": dist.get_backend()}

    import bench
    from rajepy_amd import classes, logger, parallel
    from rajepy_amd.parallel import ChannelShards, EpochShards, SlabShards
    from tests.test_host_logic import example_params, pline_params
    dev = torch.device("cuda", 0)

    # 1. the bench's one collective per step, every sharding, on device tensors
    for sh in ("epochs", "xslab", "channels"):
        pl = bench.plan("tiny", sh, 0, 1)
        E_loc, F_loc = len(pl["my_epochs"]), len(pl["my_freqs"])
        res = torch.arange(E_loc * F_loc, dtype=torch.float64, device=dev).reshape(E_loc, F_loc) + 1.
        got = bench.collect(res.clone(), pl, 0, 1, "nccl", force=True)
        assert got.is_cuda and torch.equal(got, res), sh
        out["collect_" + sh] = list(got.shape)
    # ... and the map gather that ends a `strong_xslab_gather_maps` step (BASELINE config 4's
    # "RCCL gather"): dist.gather of the slab buffers on device tensors, the root's cubes
    pl = bench.plan("tiny", "xslab_gather_maps", 0, 1)
    nx, _, nz = pl["shape"]
    tau = torch.rand(1, len(pl["my_freqs"]), nx, nz, dtype=torch.float64, device=dev)
    pre = (torch.empty_like(tau), torch.empty_like(tau))
    ct, cf = bench.gather_cubes(tau, -tau, pl, 0, 1, out=pre)
    assert ct is pre[0] and cf is pre[1] and torch.equal(ct, tau) and torch.equal(cf, -tau)
    ha, hb = bench.gather_cubes(tau, -tau, pl, 0, 1, async_op=True)      # ... and asynchronously
    assert torch.equal(ha.wait(), tau) and torch.equal(hb.wait(), -tau)
    out["gather_cubes"] = list(ct.shape)

    # 2. the gathers of rajepy_amd.parallel
    blk = torch.rand(3, 5, 7, dtype=torch.float64, device=dev)
    assert torch.equal(parallel.all_gather_blocks(blk, SlabShards(5, 1), 0, axis=1), blk)
    assert torch.equal(parallel.gather_to_root(blk, ChannelShards(np.arange(3.), 1), 0, axis=0), blk)
    assert torch.equal(parallel.gather_slabs_to_root(blk, SlabShards(5, 1), 0, 1), blk)
    fl = torch.rand(4, 6, dtype=torch.float64, device=dev)
    assert torch.equal(parallel.gather_flux_vs_time(fl, EpochShards(np.arange(4.), 1), 0), fl)
    t = torch.ones(2, 3, dtype=torch.float64, device=dev)
    dist.all_reduce(t)
    dist.barrier()
    assert float(t.sum()) == 6.0
    out["gathers"] = "ok"

    # 3. the sweeps through the JetModel API inside the group
    log = logger.Log(%(tmp)r + "/m.log", verbose=False)
    jm = classes.JetModel(example_params(), log=log)
    times = np.array([0.0, 0.5, 1.0, 2.0]) * 31536000.0
    freqs = np.array([1e9, 5e9, 2.2e10])
    lc = jm.flux_vs_time(times, freqs)                      # epoch shards + all_gather
    ftot, tau, flux = parallel.sweep_xslab(jm, times[:2], freqs, rank=0, world=1,
                                           gather_maps=True)      # all_reduce + all_gather
    np.testing.assert_allclose(ftot, lc[:2], rtol=1e-12)
    f2, tau_r, flux_r = parallel.sweep_xslab(jm, times[:2], freqs, rank=0, world=1, gather_maps=True,
                                             maps_on_root_only=True)     # all_reduce + gather
    np.testing.assert_array_equal(tau_r, tau)
    np.testing.assert_array_equal(flux_r, flux)
    cube = parallel.sweep_channel_sharded(jm, freqs, rank=0, world=1)   # gather to root
    jm.time = 0.0
    np.testing.assert_array_equal(cube, jm.flux_ff(freqs))
    np.testing.assert_allclose(np.nansum(flux[0], axis=(1, 2)), lc[0], rtol=1e-12)
    out["lightcurve_5GHz"] = [float(v) for v in lc[:, 1]]

    # 4. Pipeline.execute inside the group: all_gather_object of the run results + barrier
    dcy = %(tmp)r + "/run"
    pp = pline_params(dcy)
    pl = classes.Pipeline(classes.JetModel(example_params(), log=log), pp, log=log)
    pl.execute(simobserve=False, verbose=False, resume=False)
    assert all(r.completed for r in pl.runs)
    out["pipeline_flux"] = [float(np.sum(r.results["flux"])) for r in pl.runs]
    assert os.path.exists(os.path.join(dcy, "pipeline.save"))
    dist.barrier()
    dist.destroy_process_group()
    print("RCCL_CHILD " + json.dumps(out), flush=True)
''')


def test_every_collective_of_the_product_runs_over_rccl(tmp_path):
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = str(s.getsockname()[1])
    s.close()
    code = CHILD % {"root": ROOT, "port": port, "tmp": str(tmp_path)}
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True,
                         timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    line = [l for l in out.stdout.splitlines() if l.startswith("RCCL_CHILD ")]
    assert len(line) == 1, out.stdout[-2000:]
    r = json.loads(line[0][len("RCCL_CHILD "):])
    assert r["backend"] == "nccl"
    assert r["collect_epochs"] and r["collect_xslab"] and r["collect_channels"]
    assert len(r["gather_cubes"]) == 4
    # the example jet's light curve at 5 GHz (reference anchors): the numbers
    # that went through RCCL are the model's
    ref = [1.158223515e-3, 1.279591671e-3, 1.379008153e-3, 1.429076011e-3]
    for got, want in zip(r["lightcurve_5GHz"], ref):
        assert abs(got - want) <= 2e-9 * want
    assert len(r["pipeline_flux"]) == 3 and all(v > 0 for v in r["pipeline_flux"])


def test_bench_refuses_rccl_ranks_on_a_shared_device():
    """`--backend nccl --share-gpu` cannot be a result and must not look like one."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2",
                          "--config", "tiny", "--backend", "nccl", "--share-gpu"],
                         capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0",
                                  MASTER_ADDR="127.0.0.1", MASTER_PORT="29999"))
    assert out.returncode == 2 and "RCCL needs one device per rank" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith('{"metric"')]
