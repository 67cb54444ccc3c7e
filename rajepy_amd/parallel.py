"""Multi-GPU sharding of the RT sweep: one process per GPU, `torch.distributed` over RCCL
(backend "nccl" on ROCm) for the single gather at the end; no collective on the data path.

The path shards along three independent axes with no halo:
  * epochs    -- every epoch is a genuine pass over the grid (chi(t) enters as n^2); the grid
                 is replicated (built per rank on its own GPU), only flux-vs-time vectors or
                 map blocks are gathered.  This is the axis that carries real work.
  * channels  -- the north star's "frequency-sharded sweep".  Continuum channels share one
                 grid pass, so channel sharding splits only the map stage (K2); RRL channels
                 are real per-channel work (K3) and do scale.
  * x-slabs   -- sightlines are independent: each rank scans n_x/N of the grid.

The planners are pure Python; the gathers work with any backend (tests run them under
`gloo` on CPU tensors with world_size 2).
"""
import numpy as np


def _split(n, world):
    """Contiguous near-equal split of range(n): list of (start, stop)."""
    base, extra = divmod(n, world)
    out, s = [], 0
    for r in range(world):
        e = s + base + (1 if r < extra else 0)
        out.append((s, e))
        s = e
    return out


class Shards:
    """Contiguous block partition of a 1-D list of work items over `world` ranks."""

    def __init__(self, items, world):
        self.items = np.asarray(items)
        self.world = int(world)
        if self.world < 1:
            raise ValueError("world must be >= 1")
        self.bounds = _split(len(self.items), self.world)

    def __len__(self):
        return len(self.items)

    def slice(self, rank):
        s, e = self.bounds[rank]
        return slice(s, e)

    def local(self, rank):
        return self.items[self.slice(rank)]

    def counts(self):
        return [e - s for s, e in self.bounds]


class EpochShards(Shards):
    """Model times [s] split over ranks."""

    @property
    def n_epochs(self):
        return len(self.items)


class ChannelShards(Shards):
    """Channel frequencies [Hz] split over ranks."""


class SlabShards(Shards):
    """x-rows of the grid split over ranks (each rank scans rows [x0, x1))."""

    def __init__(self, nx, world):
        super().__init__(np.arange(nx), world)


def _dist():
    import torch.distributed as dist
    return dist


def _in_group():
    dist = _dist()
    return dist.is_available() and dist.is_initialized()


def _talks(world, group=None):
    """Does a sharding over `world` ranks issue its collective on `group`?  Only when the
    group's size IS the sharding's world: a one-rank RCCL group still runs the collective (the
    path an eight-rank job takes), but a caller that keeps the default rank=0 / world=1 inside
    an N-rank job stays purely local -- its result is complete on its own, and a collective
    there would multiply an all_reduce by N or hang a gather only some ranks entered.  Any
    other mismatch is a planning error and raises on every rank before anything is sent."""
    world = int(world)
    if not _in_group():
        if world > 1:
            raise RuntimeError("a sharding over %d ranks needs an initialised "
                               "torch.distributed process group" % world)
        return False
    size = _dist().get_world_size(group)
    if size == world:
        return True
    if world == 1:
        return False
    raise ValueError("sharding planned for %d ranks inside a process group of %d"
                     % (world, size))


def all_gather_blocks(local, shards, rank, axis=0, group=None):
    """Gather per-rank blocks of unequal leading size along `axis` to every rank.

    `local` has size shards.counts()[rank] along `axis`.  Blocks are padded to the largest
    count so a single fixed-size all_gather suffices (one collective per sweep)."""
    import torch
    dist = _dist()
    counts = shards.counts()
    if local.shape[axis] != counts[rank]:
        raise ValueError("local block has %d items on axis %d, plan says %d"
                         % (local.shape[axis], axis, counts[rank]))
    if not _talks(shards.world, group):
        return local                         # world = 1 outside a one-rank group: local
    loc = local.movedim(axis, 0).contiguous()
    cmax = max(counts)
    if loc.shape[0] < cmax:
        pad = torch.zeros((cmax - loc.shape[0],) + tuple(loc.shape[1:]), dtype=loc.dtype,
                          device=loc.device)
        loc = torch.cat([loc, pad], dim=0)
    dev = loc.device
    if loc.is_cuda and dist.get_backend(group) == "gloo":
        loc = loc.cpu()                      # rehearsal backend: gloo gathers host tensors
    out = [torch.empty_like(loc) for _ in range(shards.world)]
    dist.all_gather(out, loc, group=group)
    full = torch.cat([o[:c] for o, c in zip(out, counts)], dim=0).to(dev)
    return full.movedim(0, axis)


def gather_flux_vs_time(ftot_local, shards, rank, group=None):
    """[E_local, F] per-channel total fluxes of this rank's epochs -> [E_total, F] on every
    rank (2 KB-scale messages: latency-bound, one all_gather)."""
    return all_gather_blocks(ftot_local, shards, rank, axis=0, group=group)


def gather_to_root(local, shards, rank, axis=0, root=0, group=None):
    """Gather map blocks (channel blocks or x-slabs) onto `root` only; other ranks get None.
    The root ingests (world-1)/world of the product over its xGMI links."""
    import torch
    dist = _dist()
    counts = shards.counts()
    if not _talks(shards.world, group):
        return local
    loc = local.movedim(axis, 0).contiguous()
    cmax = max(counts)
    if loc.shape[0] < cmax:
        pad = torch.zeros((cmax - loc.shape[0],) + tuple(loc.shape[1:]), dtype=loc.dtype,
                          device=loc.device)
        loc = torch.cat([loc, pad], dim=0)
    dev = loc.device
    if loc.is_cuda and dist.get_backend(group) == "gloo":
        loc = loc.cpu()
    bufs = [torch.empty_like(loc) for _ in range(shards.world)] if rank == root else None
    dist.gather(loc, bufs, dst=root, group=group)
    if rank != root:
        return None
    full = torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0).to(dev)
    return full.movedim(0, axis)


class _SlabGather:
    """A map gather in flight (`gather_slabs_to_root(..., async_op=True)`): `wait()` completes the
    collective and, on the root, lays the slabs into the cube; returns the cube (None elsewhere).
    The packed send buffer and the root's receive buffers live as long as this object."""

    def __init__(self, work, finish, keep):
        self._work, self._finish, self._keep, self._out, self._done = work, finish, keep, None, False

    def wait(self):
        if not self._done:
            if self._work is not None:
                self._work.wait()
            self._out = self._finish()
            self._keep = None
            self._done = True
        return self._out


def gather_slabs_to_root(local, shards, rank, axis, root=0, group=None, out=None, async_op=False):
    """Slabs of a map cube (this rank's rows along `axis`, e.g. [E, F, n_x/N, n_z] with axis 2)
    -> the whole cube on `root`, None on the other ranks: BASELINE config 4's "RCCL gather".
    ONE `gather` of flat, equally padded buffers (no transposes on the senders), then the root
    lays the slabs into the cube (`out`, if given, is the preallocated destination).  The root
    ingests (world - 1) / world of the product over its xGMI links: 0.94 GB for the tau and flux
    cubes of a 512 x 512 map x 256 channels -- link-bound, several times a slab's compute.
    `async_op`: the slab is packed at once (the caller may overwrite `local` as soon as this
    returns), the collective runs beside whatever the caller enqueues next -- the next epoch's
    scan -- and the returned handle's `wait()` finishes it (see `_SlabGather`)."""
    import torch
    dist = _dist()
    counts = shards.counts()
    if local.shape[axis] != counts[rank]:
        raise ValueError("local block has %d rows on axis %d, plan says %d"
                         % (local.shape[axis], axis, counts[rank]))
    if not _talks(shards.world, group):
        if out is not None:
            out.copy_(local)
            res = out
        else:
            res = local
        return _SlabGather(None, lambda: res, None) if async_op else res
    shape = list(local.shape)
    per = 1
    for i, n in enumerate(shape):
        if i != axis:
            per *= n
    dev = local.device
    buf = torch.zeros(max(counts) * per, dtype=local.dtype, device=dev)
    buf[:local.numel()] = local.reshape(-1)
    on_host = buf.is_cuda and dist.get_backend(group) == "gloo"
    if on_host:
        buf = buf.cpu()                      # rehearsal backend: gloo gathers host tensors
    bufs = [torch.empty_like(buf) for _ in range(shards.world)] if rank == root else None
    work = dist.gather(buf, bufs, dst=root, group=group, async_op=async_op)

    def finish():
        if rank != root:
            return None
        dst = out
        if dst is None:
            shp = list(shape)
            shp[axis] = len(shards)
            dst = torch.empty(shp, dtype=local.dtype, device=dev)
        for (s, e), b in zip(shards.bounds, bufs):
            if e > s:
                shp = list(dst.shape)
                shp[axis] = e - s
                dst.narrow(axis, s, e - s).copy_(b[:(e - s) * per].view(shp))
        return dst
    if async_op:
        return _SlabGather(work, finish, (buf, bufs))
    return finish()


def sweep_flux_vs_time(model, epochs_s, freqs, rank=0, world=1, group=None):
    """Epoch-sharded continuum sweep through the JetModel API: every rank scans its epochs
    (8-32 per pass over HBM), reduces each (epoch, channel) map to its total flux on the
    device and the [E, F] light curves are all_gathered.  Returns a host array [E, F] [Jy]."""
    from . import engine as E
    from .maths import physics as mphys
    shards = EpochShards(epochs_s, world)
    mine = [float(t) for t in shards.local(rank)]
    freqs = np.atleast_1d(np.asarray(freqs, dtype=np.float64))
    eng = model.engine
    dev = model.device_fields
    gv = None
    if model.gff_mode == E.RJP_GFF_SCALAR:
        gv = [mphys.gff(nu, model.params['properties']['T_0']) for nu in freqs]
    ctau, cflux = E.ff_channel_coeffs(freqs, model.csize, model.params["target"]["dist"],
                                      model.gff_mode, gv)
    if mine:
        # one call into the library per sweep: scan + light-curve stage (rjp_ff_step)
        ftot = eng._f64(len(mine), len(freqs))
        eng.ff_step(dev, model._rjp_bursts(), mine, model.gff_mode, model._model_tavg(), ctau,
                    cflux, out=(eng._f64(len(mine), dev.npix), None, None, None, ftot))
    else:
        ftot = eng._f64(0, len(freqs))
    full = gather_flux_vs_time(ftot, shards, rank, group=group)
    return full.cpu().numpy()


def sweep_channel_sharded(model, freqs, rank=0, world=1, root=0, group=None, kind="flux"):
    """The north star's frequency-sharded continuum sweep at the model's current time: every
    rank performs the (shared) grid scan and the map stage for ITS channels; channel blocks
    are gathered on `root`.  Returns (F, n_x, n_z) on root, None elsewhere."""
    shards = ChannelShards(np.atleast_1d(np.asarray(freqs, dtype=np.float64)), world)
    mine = shards.local(rank)
    import torch
    if len(mine):
        block = model._ff_products(mine, tau=(kind == "tau"), flux=(kind == "flux"),
                                   device=True)
    else:
        block = torch.empty((0, model.nx, model.nz), dtype=torch.float64,
                            device=model.engine.device)
    full = gather_to_root(block, shards, rank, axis=0, root=root, group=group)
    return None if full is None else full.cpu().numpy()


def xslab_local(model, epochs_s, freqs, rank, world, want_maps=True):
    """This rank's share of an x-slab sharded continuum sweep: build rows [x0, x1) of the
    model grid on the local GPU (K4 with an x offset -- the grid is never replicated), scan
    them and run the map stage for every channel.  Returns (x-range, tau[E,F,nx_loc,n_z] or
    None, flux[...] or None, ftot_partial[E,F]) as device tensors."""
    from . import engine as E
    from .classes import build_model_fields, geometry_struct
    from .maths import physics as mphys
    x0, x1 = SlabShards(model.nx, world).bounds[rank]
    eng = model.engine
    freqs = np.atleast_1d(np.asarray(freqs, dtype=np.float64))
    epochs = [float(t) for t in epochs_s]
    F, Ep = len(freqs), len(epochs)
    if x1 == x0:
        # more ranks than rows: an EMPTY slab, but well-formed -- zero-length maps and zero
        # partial fluxes, so that this rank enters every collective of the sweep like its peers
        empty = eng._f64(Ep, F, 0, model.nz) if want_maps else None
        return (x0, x1), empty, (empty.clone() if want_maps else None), eng._f64(Ep, F).zero_()
    geom = geometry_struct(model.params, x1 - x0, model.ny, model.nz, ix0=x0,
                           nx_total=model.nx)
    # same degenerate-2F1 fallback as JetModel.device_fields (host launch times of the slab);
    # lean like it: the continuum sweep reads a0, em0, temp, ts only
    lean = model._dtype == E.RJP_F64 and eng.use_compact and eng.use_tau
    dev = build_model_fields(model, geom, want_vy=False, want_raw=False, want_wide=not lean)
    gv = None
    if model.gff_mode == E.RJP_GFF_SCALAR:
        gv = [mphys.gff(nu, model.params['properties']['T_0']) for nu in freqs]
    ctau, cflux = E.ff_channel_coeffs(freqs, model.csize, model.params["target"]["dist"],
                                      model.gff_mode, gv)
    P = dev.npix
    tau = eng._f64(Ep, F, P) if want_maps else None
    flux = eng._f64(Ep, F, P) if want_maps else None
    ftot = eng._f64(Ep, F)
    eng.ff_step(dev, model._rjp_bursts(), epochs, model.gff_mode, eng.tavg(dev), ctau, cflux,
                out=(eng._f64(Ep, P), None, tau, flux, ftot))
    shp = (Ep, F, x1 - x0, model.nz)
    return ((x0, x1), tau.reshape(shp) if want_maps else None,
            flux.reshape(shp) if want_maps else None, ftot)


def sweep_xslab(model, epochs_s, freqs, rank=0, world=1, gather_maps=False, group=None,
                maps_on_root_only=False, root=0):
    """x-slab sharded sweep (strong scaling of one model): every rank scans n_x/world rows.
    The per-channel total fluxes are summed over ranks (one all_reduce of [E,F]); with
    `gather_maps` the tau / flux slabs are all_gathered along x as well (`maps_on_root_only`:
    gathered onto `root` alone -- the other ranks return None for them).
    Returns (ftot[E,F] host array, tau or None, flux or None)."""
    dist = _dist()
    _, tau, flux, ftot = xslab_local(model, epochs_s, freqs, rank, world, want_maps=gather_maps)
    if _talks(world, group):
        if ftot.is_cuda and dist.get_backend(group) == "gloo":
            t = ftot.cpu()
            dist.all_reduce(t, group=group)
            ftot = t
        else:
            dist.all_reduce(ftot, group=group)
    out_t = out_f = None
    if gather_maps and maps_on_root_only:
        slabs = SlabShards(model.nx, world)
        out_t = gather_slabs_to_root(tau, slabs, rank, 2, root=root, group=group)
        out_f = gather_slabs_to_root(flux, slabs, rank, 2, root=root, group=group)
        out_t = None if out_t is None else out_t.cpu().numpy()
        out_f = None if out_f is None else out_f.cpu().numpy()
    elif gather_maps:
        slabs = SlabShards(model.nx, world)
        out_t = all_gather_blocks(tau, slabs, rank, axis=2, group=group).cpu().numpy()
        out_f = all_gather_blocks(flux, slabs, rank, axis=2, group=group).cpu().numpy()
    return ftot.cpu().numpy(), out_t, out_f
