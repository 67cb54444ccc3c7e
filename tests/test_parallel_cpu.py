"""Multi-rank logic without GPUs: shard planners and the gathers under `gloo`, world_size 2
(and 3 for the ragged case), on CPU tensors."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rajepy_amd import parallel as par


def test_planners_cover_everything_once():
    for n, w in ((32, 8), (5, 2), (3, 4), (256, 8), (1, 1), (0, 2)):
        sh = par.Shards(np.arange(n), w)
        got = np.concatenate([sh.local(r) for r in range(w)]) if n else np.array([])
        assert np.array_equal(got, np.arange(n))
        c = sh.counts()
        assert sum(c) == n and max(c) - min(c) <= 1
    assert par.SlabShards(512, 8).counts() == [64] * 8
    assert par.EpochShards(np.linspace(0, 5, 32), 8).n_epochs == 32
    with pytest.raises(ValueError):
        par.Shards([1, 2], 0)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_epochs, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        F = 5
        full = torch.arange(n_epochs * F, dtype=torch.float64).reshape(n_epochs, F)
        sh = par.EpochShards(np.arange(n_epochs), world)
        mine = full[sh.slice(rank)]
        got = par.gather_flux_vs_time(mine, sh, rank)
        ok = torch.equal(got, full)
        # channel blocks of maps gathered on the root only
        maps = torch.arange(n_epochs * 6, dtype=torch.float64).reshape(n_epochs, 2, 3)
        root = par.gather_to_root(maps[sh.slice(rank)], sh, rank, axis=0, root=0)
        ok = ok and ((rank == 0 and torch.equal(root, maps)) or (rank != 0 and root is None))
        # x-slabs gathered along a middle axis
        cube = torch.arange(4 * n_epochs * 3, dtype=torch.float64).reshape(4, n_epochs, 3)
        got2 = par.all_gather_blocks(cube[:, sh.slice(rank), :], sh, rank, axis=1)
        ok = ok and torch.equal(got2, cube)
        # round 5: the map gather of BASELINE config 4 -- slabs along a middle axis of a 4-D cube
        # laid into the whole cube on the root (into a preallocated buffer and without one);
        # ragged and EMPTY slabs (more ranks than rows) included
        c4 = torch.arange(2 * 3 * n_epochs * 5, dtype=torch.float64).reshape(2, 3, n_epochs, 5)
        mine4 = c4[:, :, sh.slice(rank), :].contiguous()
        g1 = par.gather_slabs_to_root(mine4, sh, rank, 2, root=0)
        pre = torch.full_like(c4, -1.0) if rank == 0 else None
        g2 = par.gather_slabs_to_root(mine4, sh, rank, 2, root=0, out=pre)
        if rank == 0:
            ok = ok and torch.equal(g1, c4) and g2 is pre and torch.equal(pre, c4)
        else:
            ok = ok and g1 is None and g2 is None
        # ... asynchronously: the slab is packed at once (the source may be overwritten right
        # away), wait() completes the gather and assembles the cube on the root
        src = mine4.clone()
        h = par.gather_slabs_to_root(src, sh, rank, 2, root=0, async_op=True)
        src.fill_(-99.0)
        g4 = h.wait()
        ok = ok and ((rank == 0 and torch.equal(g4, c4) and h.wait() is g4) or
                     (rank != 0 and g4 is None))
        few = par.SlabShards(1, world)                  # one row, `world` ranks
        row = torch.full((2, few.counts()[rank], 3), float(rank + 1), dtype=torch.float64)
        g3 = par.gather_slabs_to_root(row, few, rank, 1, root=0)
        ok = ok and ((rank == 0 and tuple(g3.shape) == (2, 1, 3) and bool((g3 == 1.0).all())) or
                     (rank != 0 and g3 is None))
        bad = False
        try:
            par.all_gather_blocks(full, sh, rank)       # wrong local size must be refused
        except ValueError:
            bad = True
        ret[rank] = bool(ok and (bad or world == 1))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_epochs", [(2, 8), (2, 5), (3, 4)])
def test_gathers_gloo(world, n_epochs):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n_epochs, ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world)), dict(ret)


def _local_in_group_worker(rank, world, port, ret):
    """The world = 1 entry points inside a 2-rank group: purely local, no collective -- only
    rank 0 calls them here, so a collective would hang (and an all_reduce would double the
    fluxes); a sharding planned for 3 ranks inside the 2-rank group is refused."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ok = True
        if rank == 0:
            one = par.Shards(np.arange(4), 1)
            x = torch.arange(12, dtype=torch.float64).reshape(4, 3)
            ok = ok and par.all_gather_blocks(x, one, 0) is x
            ok = ok and par.gather_to_root(x, one, 0) is x
            ok = ok and par.gather_slabs_to_root(x, one, 0, 0) is x
            ok = ok and par.gather_slabs_to_root(x, one, 0, 0, async_op=True).wait() is x
            ok = ok and par.gather_flux_vs_time(x, one, 0) is x
            ok = ok and par._talks(1) is False and par._talks(2) is True
        try:
            par._talks(3)
            ok = False
        except ValueError:
            pass
        dist.barrier()
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_world1_entry_points_stay_local_inside_a_larger_group():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_local_in_group_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert ret[0] is True and ret[1] is True, dict(ret)
    # and without any group: world = 1 is local, world > 1 cannot be served
    assert par._talks(1) is False
    with pytest.raises(RuntimeError):
        par._talks(2)


def _pipeline_worker(rank, world, port, dcy, ret):
    """Pipeline.execute under a 2-rank gloo group, dry run (no GPU here): epochs are dealt to
    the ranks, results/completion flags are gathered, rank 0 writes the state files."""
    import pickle
    from rajepy_amd import classes, logger
    from tests.test_host_logic import example_params, pline_params
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        log = logger.Log(os.path.join(dcy, "rank%d.log" % rank), verbose=False)
        pp = pline_params(dcy)
        pp["continuum"]["times"] = np.array([0., 0.5, 1., 1.5, 2.])
        pl = classes.Pipeline(classes.JetModel(example_params(), log=log), pp, log=log)
        pl.execute(simobserve=False, verbose=False, dryrun=True, resume=False, clobber=True)
        text = open(log.filename).read()
        years = sorted({float(r.year) for r in pl.runs})
        mine = {y for i, y in enumerate(years) if i % world == rank}
        executed = {float(r.year) for i, r in enumerate(pl.runs)
                    if "Executing run #%d " % (i + 1) in text}
        ok = executed == mine and all(r.completed for r in pl.runs)
        dist.barrier()
        if rank == 0:
            saved = pickle.load(open(os.path.join(dcy, "pipeline.save"), "rb"))
            ok = ok and len(saved["runs"]) == len(pl.runs) and all(r.completed for r in saved["runs"])
            ok = ok and os.path.exists(os.path.join(dcy, "jetmodel.save"))
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_pipeline_epoch_sharding_gloo(tmp_path):
    dcy = str(tmp_path / "out")
    os.makedirs(dcy)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_pipeline_worker, args=(2, _free_port(), dcy, ret), nprocs=2, join=True)
    assert all(ret[r] for r in range(2)), dict(ret)


@pytest.mark.parametrize("config,world,legs", [("tiny", 2, {"xslab", "xslab_gather_maps", "epochs",
                                                           "channels"}),
                                               ("tiny", 3, {"xslab", "xslab_gather_maps", "epochs",
                                                           "channels"}),
                                               ("tiny5", 3, {"xslab"})])
def test_bench_self_launch_rehearsal(config, world, legs):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment must start its own N
    ranks (children under torch.distributed.run) and relay rank 0's single JSON line.  Driven
    here end to end without a GPU: --rehearse-cpu runs launcher, shard planners and the
    step's collective under gloo on placeholder vectors (no kernels, no throughput)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world),
                          "--steps", "2", "--warmup", "1", "--config", config,
                          "--rehearse-cpu"], capture_output=True, text=True, timeout=600,
                         env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    r = json.loads(lines[0])
    assert r["rehearsal"] is True and r["value"] is None
    assert r["n_gpus"] == world and r["ranks_seen"] == world
    assert set(r["legs"]) == legs and all(v["ok"] for v in r["legs"].values())
    assert r["scaling"] == "strong" and r["config"]["sharding"] == "xslab"


def test_bench_refuses_mismatched_world_size_cpu():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2",
                          "--config", "tiny", "--rehearse-cpu"], capture_output=True, text=True,
                         timeout=300, env=dict(os.environ, WORLD_SIZE="3", RANK="0"))
    assert out.returncode == 2 and "WORLD_SIZE" in out.stderr


def _failing_pipeline_worker(rank, world, port, dcy, ret):
    """One rank's run raises: every rank must still reach the result gather and the barrier
    and then raise -- not leave its peers waiting in a collective."""
    from rajepy_amd import classes, logger
    from tests.test_host_logic import example_params, pline_params
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        log = logger.Log(os.path.join(dcy, "rank%d.log" % rank), verbose=False)
        pp = pline_params(dcy)
        pp["continuum"]["times"] = np.array([0., 0.5, 1., 1.5])

        class P(classes.Pipeline):
            def _radiative_transfer(self, idx, run, clobber):
                if rank == 1:
                    raise ValueError("boom in run %d" % (idx + 1))
                self.runs[idx].results["flux"] = 1.0

        model = classes.JetModel(example_params(), log=log)
        model.prefetch_epochs = lambda times: None          # no GPU here
        pl = P(model, pp, log=log)
        try:
            pl.execute(simobserve=False, verbose=False, dryrun=False, resume=False, clobber=True)
            ret[rank] = "no exception"
        except RuntimeError as exc:
            ok = "rank 1" in str(exc) and "boom" in str(exc) and "ValueError" in str(exc)
            if rank == 0:
                # rank 0 finished its own runs and wrote the state before re-raising
                done = [r.completed for r in pl.runs]
                ok = ok and any(done) and not all(done)
                ok = ok and os.path.exists(os.path.join(dcy, "pipeline.save"))
            ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_pipeline_failure_on_one_rank_does_not_hang_the_others(tmp_path):
    dcy = str(tmp_path / "out")
    os.makedirs(dcy)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_failing_pipeline_worker, args=(2, _free_port(), dcy, ret), nprocs=2, join=True)
    assert ret[0] is True and ret[1] is True, dict(ret)
