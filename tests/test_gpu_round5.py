"""Round-5 GPU checks through the C-ABI: workspace bounds of the single-epoch table scan and of
the launch-time-ordered sweep on the map shapes ADVICE r04 names (exact-size workspaces with a
guard band behind them), the launch-time range guard (a wrong fields.ts_lo / ts_hi is refused by
the call that passes it; launch times that leave the range later poison their sightlines with
NaN and raise the context's flag), rjp_ff_step against rjp_ff_scan + rjp_ff_maps, and the lazily
built wide fields of JetModel."""
import ctypes as C

import numpy as np
import pytest

from oracle import rt_oracle as orc
from tests import gpu_util as U

pytestmark = pytest.mark.gpu
SEED = 20240511
SENTINEL = 0xA5


@pytest.fixture(scope="module")
def eng():
    from rajepy_amd.engine import RTEngine
    e = RTEngine(0)
    e.cache_moments = False
    yield e
    e.close()


def _bursts():
    from rajepy_amd.engine import make_bursts
    ej = U.example_bursts_params()
    red, blue = [], []
    for t0, hl, chi, which in zip(ej["t_0"], ej["hl"], ej["chi"], ej["which"]):
        sig = hl * orc.YEAR * 2. / (2. * np.sqrt(2. * np.log(2.)))
        for jet, lst in (("R", red), ("B", blue)):
            if jet in str(which):
                lst.append((t0 * orc.YEAR, chi - 1., sig))
    return make_bursts(red, blue)


def _exact_workspace(eng, nbytes, band=1 << 20):
    """An engine workspace of EXACTLY `nbytes` with `band` sentinel bytes behind it."""
    import torch
    nbytes = (int(nbytes) + 15) // 16 * 16
    big = torch.full((nbytes + band,), SENTINEL, dtype=torch.uint8, device=eng.device)
    eng._work = big[:nbytes]
    return big, nbytes


@pytest.mark.parametrize("shape", [(256, 128, 256), (256, 256, 256), (100, 400, 700),
                                   (300, 256, 400)])
@pytest.mark.parametrize("want_em", [False, True])
def test_table_scan_stays_inside_an_exact_size_workspace(eng, shape, want_em):
    """the table scan chose its own y-ranges (up to 16) while the workspace was
    sized from another rule -- on these mid-size maps it wrote behind the caller's buffer.  Now
    rjp_ff_scan_workspace() covers the table path's rule and the scan never takes more ranges than
    the buffer it was given holds: exact-size buffer, sentinel band behind it untouched, map
    equal to the Gaussian scan's."""
    import torch
    from rajepy_amd import engine as E
    nx, ny, nz = shape
    fields = eng.synth_fields(shape, SEED, 0, 8, csize_au=0.5, wide=False,
                              tau_mode=E.RJP_GFF_SCALAR)
    b = _bursts()
    t = [1.0 * orc.YEAR]
    eng.use_chi_table = False
    ref = eng.ff_scan(fields, b, t, E.RJP_GFF_SCALAR, want_em=want_em, want_tavg=False)
    ref = [None if r is None else r.clone() for r in ref]
    eng.use_chi_table = True
    old = eng._work
    try:
        big, n = _exact_workspace(eng, eng.lib.rjp_ff_scan_workspace(nx, ny, nz, 1))
        got = eng.ff_scan(fields, b, t, E.RJP_GFF_SCALAR, want_em=want_em, want_tavg=False)
        assert eng.last_scan_path()[0] == "table"
        eng.synchronize()
        assert bool((big[n:] == SENTINEL).all()), "the scan wrote behind its workspace"
        assert ((got[0] - ref[0]).abs() / ref[0]).max().item() < 3e-12
        if want_em:
            assert ((got[1] - ref[1]).abs() / ref[1]).max().item() < 3e-12
        # a workspace that holds ONE y-range only (what round 4's rule handed out for the first
        # of these shapes): the scan takes fewer ranges instead of writing past the end
        small = (4 * nx * nz * 8 + 15) // 16 * 16
        big2, n2 = _exact_workspace(eng, small)
        fs = eng._scan_struct(fields, b, 1)
        sumA = eng._f64(1, fields.npix)
        em = eng._f64(1, fields.npix) if want_em else None
        rc = eng.lib.rjp_ff_scan(eng.ctx, C.byref(fs), C.byref(b), (C.c_double * 1)(t[0]), 1,
                                 E.RJP_GFF_SCALAR, sumA.data_ptr(),
                                 em.data_ptr() if want_em else None, None, eng._work.data_ptr(),
                                 n2, eng._stream())
        eng.synchronize()
        assert bool((big2[n2:] == SENTINEL).all())
        # (smaller than rjp_ff_scan_workspace(): refused, not overrun)
        from rajepy_amd import _lib
        assert rc == _lib.RJP_ERR_WORKSPACE
    finally:
        eng._work = old


@pytest.mark.parametrize("shape,K", [((1, 64, 16), 20), ((1, 64, 16), 32), ((3, 96, 16), 32),
                                     ((1, 128, 80), 32), ((2, 64, 68), 32)])
def test_lt_sweep_stays_inside_an_exact_size_workspace(eng, shape, K):
    """on tiny maps the sweep on the launch-time-ordered layout split its key
    range into up to 32 ranges of 32 planes of 64 G doubles, while the workspace holds 1280 planes
    of ceil16(npix): an overrun for npix <= 48 and a few more sizes.  Exact-size buffer with a
    guard band; result against the epoch tiles."""
    import torch
    from rajepy_amd import engine as E
    nx, ny, nz = shape
    fields = eng.synth_fields(shape, SEED + 1, 0, 8, csize_au=0.5, wide=False,
                              tau_mode=E.RJP_GFF_SCALAR)
    b = _bursts()
    ep = [float(x) for x in np.linspace(0., 4., 16) * orc.YEAR]
    eng.use_moments = False
    ref = eng.ff_scan(fields, b, ep, E.RJP_GFF_SCALAR, want_em=False, want_tavg=False)[0].clone()
    eng.use_moments = True
    eng.build_lt(fields, K)
    old, force = eng._work, eng.force_moments
    eng.force_moments = True
    try:
        big, n = _exact_workspace(eng, eng.lib.rjp_ff_scan_workspace(nx, ny, nz, len(ep)))
        got = eng.ff_scan(fields, b, ep, E.RJP_GFF_SCALAR, want_em=False, want_tavg=False)[0]
        assert eng.last_scan_path()[0] == "lt"
        eng.synchronize()
        assert bool((big[n:] == SENTINEL).all()), "the sweep wrote behind its workspace"
        assert ((got - ref).abs() / ref).max().item() < 5e-11
    finally:
        eng._work, eng.force_moments = old, force
        fields.lt = None


def _narrow(fields, frac=0.9):
    lo, hi = fields.ts_range
    return (lo, lo + frac * (hi - lo))


@pytest.mark.parametrize("path", ["table", "moments"])
def test_a_wrong_launch_time_range_is_refused_by_the_call_that_passes_it(eng, path):
    """fields.ts_lo / ts_hi narrower than the launch times of d_ts.  The
    reference evaluates every cell (classes.py:844-845) and cannot go wrong this way; here the
    scan clamps into its table / bins, so the first call that uses an unseen range checks it and
    returns RJP_ERR_ARG before anything is enqueued."""
    from rajepy_amd import _lib, engine as E
    shape = (64, 72, 512) if path == "table" else (16, 96, 64)
    fields = eng.synth_fields(shape, SEED + 2, 0, 8, csize_au=0.5, wide=False,
                              tau_mode=E.RJP_GFF_SCALAR)
    b = _bursts()
    ep = [1.0 * orc.YEAR] if path == "table" else \
        [float(x) for x in np.linspace(0., 4., 16) * orc.YEAR]
    eng.force_moments = path == "moments"
    try:
        good = eng.ff_scan(fields, b, ep, E.RJP_GFF_SCALAR, want_em=False, want_tavg=False)[0].clone()
        assert eng.last_scan_path()[0] == path
        true_range = fields.ts_range
        fields.ts_range = _narrow(fields)
        out = eng._f64(len(ep), fields.npix).fill_(-7.0)
        with pytest.raises(_lib.RjprtError) as ei:
            eng.ff_scan(fields, b, ep, E.RJP_GFF_SCALAR, want_em=False, out=(out, None, None))
        assert ei.value.status == _lib.RJP_ERR_ARG and "ts_lo" in str(ei.value)
        eng.synchronize()
        assert bool((out == -7.0).all()), "the refused call enqueued a scan"
        assert not eng.range_guard()                    # refused up front: no flag left behind
        # the true range again: accepted, same maps
        fields.ts_range = true_range
        again = eng.ff_scan(fields, b, ep, E.RJP_GFF_SCALAR, want_em=False, want_tavg=False)[0]
        eng.synchronize()
        assert bool((again == good).all()) if path == "table" else \
            ((again - good).abs() / good).max().item() < 1e-12
        # rjp_lt_count checks inside its counting pass
        fields.ts_range = _narrow(fields)
        with pytest.raises(_lib.RjprtError) as ei:
            eng.build_lt(fields, 16)
        assert ei.value.status == _lib.RJP_ERR_ARG and "rjp_lt_count" in str(ei.value)
        fields.ts_range = true_range
    finally:
        eng.force_moments = False


@pytest.mark.parametrize("path", ["table", "wide_table", "moments"])
def test_launch_times_that_leave_the_range_poison_their_sightlines(eng, path):
    """... and the kernels keep watching: a launch time edited IN PLACE (same pointer, same
    declared range -- nothing the up-front check can see) to a value outside the range gives NaN
    sums on exactly the sightlines concerned, never a clamped value; the context's flag is
    raised, the next entry point reports it once, and the one after runs."""
    import torch
    from rajepy_amd import _lib, engine as E
    shape = (64, 72, 512) if path != "moments" else (16, 96, 64)
    nx, ny, nz = shape
    wide = path == "wide_table"
    fields = eng.synth_fields(shape, SEED + 3, 0, 8, csize_au=0.5, wide=wide,
                              tau_mode=None if wide else E.RJP_GFF_SCALAR)
    if wide:
        fields.em0 = None                               # scan the five model fields
    b = _bursts()
    ep = [1.0 * orc.YEAR] if path != "moments" else \
        [float(x) for x in np.linspace(0., 4., 16) * orc.YEAR]
    eng.force_moments = path == "moments"
    try:
        good = eng.ff_scan(fields, b, ep, E.RJP_GFF_SCALAR, want_em=False, want_tavg=False)[0].clone()
        assert eng.last_scan_path()[0] == ("moments" if path == "moments" else "table")
        lo, hi = fields.ts_range
        ts3 = fields.ts.view(nx, ny, nz)
        hit = [(3, 5, 7), (nx - 1, ny - 1, nz - 1), (10, 0, nz // 2)]
        saved = [ts3[i].item() for i in hit]
        ts3[hit[0]] = hi + 0.25 * (hi - lo)             # above, below, far above
        ts3[hit[1]] = lo - 1.0
        ts3[hit[2]] = hi * 10.0
        bad = eng.ff_scan(fields, b, ep, E.RJP_GFF_SCALAR, want_em=False, want_tavg=False)[0].clone()
        eng.synchronize()
        pix = sorted({x * nz + z for (x, _, z) in hit})
        # (two sightlines per lane in the table kernels: the lane's neighbour goes with it)
        touched = set(pix) | ({p ^ 1 for p in pix} if path != "moments" else set())
        nanpix = set(torch.nonzero(torch.isnan(bad[0])).flatten().tolist())
        assert set(pix) <= nanpix <= touched, (pix, sorted(nanpix))
        keep = torch.ones(fields.npix, dtype=torch.bool, device=eng.device)
        keep[sorted(touched)] = False
        assert bool((bad[:, keep] == good[:, keep]).all()) if path != "moments" else \
            ((bad[:, keep] - good[:, keep]).abs() / good[:, keep]).max().item() < 1e-12
        # the flag: reported ONCE by the next entry point, which enqueues nothing
        out = eng._f64(len(ep), fields.npix).fill_(-7.0)
        with pytest.raises(_lib.RjprtError) as ei:
            eng.ff_scan(fields, b, ep, E.RJP_GFF_SCALAR, want_em=False, out=(out, None, None))
        assert ei.value.status == _lib.RJP_ERR_ARG and "earlier scan" in str(ei.value)
        eng.synchronize()
        assert bool((out == -7.0).all())
        for i, v in zip(hit, saved):
            ts3[i] = v
        again = eng.ff_scan(fields, b, ep, E.RJP_GFF_SCALAR, want_em=False, want_tavg=False)[0]
        eng.synchronize()
        assert not eng.range_guard()
        assert bool((again == good).all()) if path != "moments" else \
            ((again - good).abs() / good).max().item() < 1e-12
        # NaN and infinite launch times are not breaches (NaN: outside the jet; inf: chi == 1)
        ts3[hit[0]] = float("nan")
        ts3[hit[1]] = float("inf")
        eng.ff_scan(fields, b, ep, E.RJP_GFF_SCALAR, want_em=False, want_tavg=False)
        assert not eng.range_guard()
    finally:
        eng.force_moments = False
        eng.range_guard()


@pytest.mark.parametrize("case", ["table", "tiles8", "moments", "small"])
def test_ff_step_is_scan_plus_maps(eng, case):
    """rjp_ff_step == rjp_ff_scan followed by rjp_ff_maps, bit for bit (the same kernels in the
    same order from one call), for the single-epoch table scan, an 8-epoch tile, a sweep through
    the launch-time moments and a map too small for the table; sampled against the oracle."""
    import torch
    from rajepy_amd import engine as E
    from rajepy_amd.maths import physics as ph
    shape = {"table": (64, 72, 512), "tiles8": (8, 64, 128), "moments": (16, 96, 64),
             "small": (4, 64, 128)}[case]
    fields = eng.synth_fields(shape, SEED + 4, 0, 8, csize_au=0.5, wide=False,
                              tau_mode=E.RJP_GFF_SCALAR)
    b = _bursts()
    ep = {"table": [1.0], "tiles8": list(np.linspace(0.2, 3., 8)),
          "moments": list(np.linspace(0., 4., 16)), "small": [0.7]}[case]
    ep = [float(x) * orc.YEAR for x in ep]
    freqs = np.geomspace(1e9, 5e10, 5)
    ctau, cflux = E.ff_channel_coeffs(freqs, 0.5, 120., E.RJP_GFF_SCALAR,
                                      [ph.gff(nu, 1e4) for nu in freqs])
    tavg = eng.tavg(fields)
    eng.force_moments = case == "moments"
    try:
        sumA, em, _ = eng.ff_scan(fields, b, ep, E.RJP_GFF_SCALAR, want_em=True, want_tavg=False)
        path = eng.last_scan_path()[0]
        tau, flux, ftot = eng.ff_maps(sumA, tavg, ctau, cflux)
        Ep, P, F = len(ep), fields.npix, len(freqs)
        out = (eng._f64(Ep, P), eng._f64(Ep, P), eng._f64(Ep, F, P), eng._f64(Ep, F, P),
               eng._f64(Ep, F))
        eng.ff_step(fields, b, ep, E.RJP_GFF_SCALAR, tavg, ctau, cflux, out)
        assert eng.last_scan_path()[0] == path
        assert path == {"table": "table", "tiles8": "tiles", "moments": "moments",
                        "small": "tiles"}[case]
        eng.synchronize()
        if case == "moments":      # LDS atomics: reproducible to rounding
            for a, c in zip(out, (sumA, em, tau, flux, ftot)):
                assert ((a - c).abs() / c.abs()).max().item() < 1e-12
        else:
            for a, c in zip(out, (sumA, em, tau, flux, ftot)):
                assert torch.equal(a.view(torch.int64), c.view(torch.int64))
        # light-curve form: no maps, no EM
        ft = eng._f64(Ep, F)
        eng.ff_step(fields, b, ep, E.RJP_GFF_SCALAR, tavg, ctau, cflux,
                    (eng._f64(Ep, P), None, None, None, ft))
        eng.synchronize()
        np.testing.assert_allclose(ft.cpu().numpy(), ftot.cpu().numpy(), rtol=1e-12)
        # the oracle on the whole (small) grid at the first epoch
        g = U.synth_host(shape, SEED + 4, 0)
        p = U.load_golden("cfg1_example")[2]
        p["ejection"] = U.example_bursts_params()
        p["grid"].update(n_x=shape[0], n_y=shape[1], n_z=shape[2])
        jet = orc.OracleJet.from_fields(p, g["nd"], g["xi"], g["temp"], g["ff"], g["areas"],
                                        g["ts"], g["rr"], g["vy"])
        jet.time = ep[0]
        shp = (F, shape[0], shape[2])
        np.testing.assert_allclose(out[2][0].cpu().numpy().reshape(shp),
                                   jet.optical_depth_ff(freqs), rtol=1e-10)
        np.testing.assert_allclose(out[3][0].cpu().numpy().reshape(shp), jet.flux_ff(freqs),
                                   rtol=1e-9)
    finally:
        eng.force_moments = False


def test_ff_step_validates_both_stages_before_enqueueing(eng):
    from rajepy_amd import _lib, engine as E
    shape = (4, 64, 128)
    fields = eng.synth_fields(shape, SEED + 5, 0, 8, csize_au=0.5, wide=False,
                              tau_mode=E.RJP_GFF_SCALAR)
    b = _bursts()
    tavg = eng.tavg(fields)
    P = fields.npix
    sumA = eng._f64(1, P).fill_(-7.0)
    ft = eng._f64(1, 3)
    fs = eng._scan_struct(fields, b, 1)
    work = eng._workspace(eng.lib.rjp_ff_scan_workspace(*shape, 1))
    t = (C.c_double * 1)(1.0 * orc.YEAR)
    ct = (C.c_double * 3)(1e-20, 2e-20, 3e-20)
    # a light curve is asked for but the map stage has no workspace: nothing may run
    rc = eng.lib.rjp_ff_step(eng.ctx, C.byref(fs), C.byref(b), t, 1, E.RJP_GFF_SCALAR,
                             tavg.data_ptr(), ct, ct, 3, sumA.data_ptr(), None, None, None,
                             ft.data_ptr(), work.data_ptr(), work.numel(), None, 0, eng._stream())
    assert rc == _lib.RJP_ERR_WORKSPACE
    rc = eng.lib.rjp_ff_step(eng.ctx, C.byref(fs), C.byref(b), t, 1, E.RJP_GFF_SCALAR,
                             None, ct, ct, 3, sumA.data_ptr(), None, None, None, None,
                             work.data_ptr(), work.numel(), None, 0, eng._stream())
    assert rc == _lib.RJP_ERR_ARG
    eng.synchronize()
    assert bool((sumA == -7.0).all())


def test_jetmodel_builds_the_wide_fields_lazily(tmp_path):
    """a continuum pipeline keeps FOUR grid-sized arrays resident (a0, em0,
    temp, ts); nd / xi / pf / vy appear with the first call that needs them -- an RRL product, a
    grid accessor, the ion_fraction / vel setters -- and every product still equals the
    reference's (classes.py:571-1000: the reference's own fields are lazy properties)."""
    from rajepy_amd import classes, logger
    z, meta, p = U.load_golden("cfg1_example")
    for k in ("mod_r_0",):
        p["geometry"].pop(k, None)
    for k in ("q_n", "q_tau"):
        p["power_laws"].pop(k, None)
    p["properties"].pop("n_0", None)
    log = logger.Log(str(tmp_path / "m.log"), verbose=False)
    jm = classes.JetModel(p, log=log)
    dev = jm.device_fields
    grid_arrays = lambda: [k for k in ("nd", "xi", "temp", "pf", "ts", "vy", "em0", "a0")
                           if getattr(dev, k) is not None]
    assert sorted(grid_arrays()) == ["a0", "em0", "temp", "ts"]
    jm.time = 0.
    nu = 5e9
    tau = jm.optical_depth_ff(nu)
    flux = jm.flux_ff(nu)
    em = jm.emission_measure()
    lc = jm.flux_vs_time(np.array([0., 0.5, 1.0]) * orc.YEAR, [nu])
    assert sorted(grid_arrays()) == ["a0", "em0", "temp", "ts"], "a continuum call built wide fields"
    assert np.isfinite(tau).all() and np.nansum(flux) > 0 and em.max() > 0 and lc.shape == (3, 1)
    # first RRL call: the wide fields + vy appear, the cube equals the oracle's
    rf = orc.chan_freqs(22364174326.22781, 8e5, 1e5)
    trrl = jm.optical_depth_rrl("H66a", rf)
    assert sorted(grid_arrays()) == ["a0", "em0", "nd", "pf", "temp", "ts", "vy", "xi"]
    _, _, _, g, jet = U.golden_dense("cfg1_example")
    jet.time = 0.
    np.testing.assert_allclose(trrl, jet.optical_depth_rrl("H66a", rf), rtol=1e-9, atol=1e-300)
    np.testing.assert_allclose(tau, jet.optical_depth_ff(nu), rtol=1e-10, atol=1e-300)
    # a lean model's accessor and setter
    jm2 = classes.JetModel(p, log=log, engine=jm.engine)
    xi = jm2.ion_fraction
    assert jm2.device_fields.xi is not None and np.nanmax(xi) > 0
    jm3 = classes.JetModel(p, log=log, engine=jm.engine)
    jm3.time = 0.
    before = jm3.optical_depth_ff(nu)
    jm3.ion_fraction = xi * 0.5                          # (n x)^2: a quarter of the optical depth
    after = jm3.optical_depth_ff(nu)
    np.testing.assert_allclose(after, 0.25 * before, rtol=1e-12)


@pytest.mark.parametrize("nchan", [65, 128, 129])
def test_rrl_channel_shards_of_65_to_128_channels(eng, nchan):
    """A channel shard of a 256-channel cube on two ranks is 128 channels: such blocks run as two
    64-lane blocks since round 5 (one 256-lane block would leave half its lanes idle); 129 keeps
    the 256-lane layout.  Against the oracle's wofz cube (rrls.py:350-389) at K3's bound."""
    from rajepy_amd import _lib, engine as E
    from rajepy_amd.maths import rrls
    shape = (4, 48, 32)
    fields = eng.synth_fields(shape, SEED + 6, 0, 8, csize_au=0.5, with_vy=True)
    g = U.synth_host(shape, SEED + 6, 0)
    p = U.load_golden("cfg1_example")[2]
    p["ejection"] = U.example_bursts_params()
    p["grid"].update(n_x=shape[0], n_y=shape[1], n_z=shape[2])
    jet = orc.OracleJet.from_fields(p, g["nd"], g["xi"], g["temp"], g["ff"], g["areas"],
                                    g["ts"], g["rr"], g["vy"])
    jet.time = 1.0 * orc.YEAR
    line = _lib.Line(**rrls.line_constants("H66a"))
    rf = orc.chan_freqs(line.nu_rest, nchan * 1e5, 1e5)
    assert len(rf) == nchan
    got = eng.rrl_scan(fields, U.bursts_from_oracle(jet), jet.time, line, rf)
    eng.synchronize()
    np.testing.assert_allclose(got.cpu().numpy().reshape(nchan, shape[0], shape[2]),
                               jet.optical_depth_rrl("H66a", rf), rtol=U.K3_RTOL_WAVE)


def test_occupied_cells_is_the_sum_of_the_y_ranges(eng):
    """rjp_occupied_cells == sum_p max(0, yhi[p] - ylo[p]) of rjp_y_bounds (the hint the
    tiles-or-moments cost model reads), on a sparse model with empty sightlines and on a dense one."""
    import torch
    from rajepy_amd import engine as E
    z, meta, p, g, jet = U.golden_dense("cfg1_example")
    fields = eng.upload_fields(g["nd"], g["xi"], g["temp"], g["ff"], g["areas"], g["ts"],
                               g["rr"] < 0, vy=g["vy"], csize_au=jet.csize)
    lo, hi = eng.compute_y_bounds(fields)
    want = int((hi - lo).clamp(min=0).sum(dtype=torch.int64).item())
    assert fields.occupied_cells == want and 0 < want < fields.ncells
    assert int((hi <= lo).sum().item()) > 0                  # empty sightlines: [ny, 0)
    dense = eng.synth_fields((8, 40, 64), SEED + 7, 0, 8, csize_au=0.5)
    eng.compute_y_bounds(dense)
    assert dense.occupied_cells == dense.ncells
