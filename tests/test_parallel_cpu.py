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
