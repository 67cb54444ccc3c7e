"""The tau scan layout (rjp_fields.d_a0 = em0 T^-1.5|-1.35 + ts, 16 B/cell) and the once-per-model
T_avg pass (rjp_tavg), through the C-ABI: bit-identical to the compact (3-field) and wide
(5-field) layouts for every tile size, with and without emission-measure maps, with NumPy's NaN
semantics, on fields from every producer.  What is epoch- and frequency-independent in the
reference: classes.py:1395-1397 (T^-1.5 and the Gaunt power law multiply n^2 per cell),
classes.py:1471-1472 (nanmean of T > 0)."""
import copy

import numpy as np
import pytest

from oracle import rt_oracle as orc
from tests import gpu_util as U

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    """This file pins the epoch TILES on the tau layout (bit-identity with the other layouts):
    sweeps of >= 12 epochs would otherwise take the launch-time moments, which agree with the
    tiles to 1e-11, not bit for bit (tests/test_gpu_moments.py)."""
    from rajepy_amd.engine import RTEngine
    e = RTEngine(0)
    e.use_moments = False
    yield e
    e.close()


def _example_bursts():
    p = copy.deepcopy(U.load_golden("cfg1_example")[2])
    p["ejection"] = U.example_bursts_params()
    g = U.synth_host((1, 2, 2), 1, 0)
    jet = orc.OracleJet.from_fields(dict(p, grid=dict(p["grid"], n_x=1, n_y=2, n_z=2)), g["nd"],
                                    g["xi"], g["temp"], g["ff"], g["areas"], g["ts"], g["rr"],
                                    g["vy"])
    return U.bursts_from_oracle(jet)


def _eq(a, b):
    return np.array_equal(a.cpu().numpy(), b.cpu().numpy(), equal_nan=True)


def _close(a, b, rtol=1e-11):
    """Same NaN / zero pattern, values to `rtol`: the wide layout has no tiles of >= 16 epochs
    (it evaluates every epoch directly, in tiles of 8), the tau and compact layouts run the
    uniform-epoch recurrence there -- the same numbers to rounding, not bit for bit."""
    x, y = a.cpu().numpy(), b.cpu().numpy()
    return (np.array_equal(np.isnan(x), np.isnan(y)) and np.array_equal(x == 0, y == 0) and
            np.allclose(x, y, rtol=rtol, atol=0, equal_nan=True))


@pytest.mark.parametrize("temp_mode", [0, 1])
@pytest.mark.parametrize("n_ep", [1, 2, 3, 8, 11, 16, 32, 37])
@pytest.mark.parametrize("want_em", [True, False])
def test_tau_layout_is_bit_identical_to_compact_and_wide(eng, temp_mode, n_ep, want_em):
    """K1 streaming a0, ts (+ em0 with EM maps) against the 3-field and the 5-field scans of
    the same model: every tile size (direct 1 / 2 / 4 / 8, uniform 16 / 32 incl. the LDS-DMA
    kernels, ragged tails), sumA and EM bit for bit; T_avg from rjp_tavg == the scan's."""
    from rajepy_amd import engine as E
    shape = (8, 96, 64)
    mode = E.RJP_GFF_SCALAR if temp_mode == 0 else E.RJP_GFF_POWERLAW
    f = eng.synth_fields(shape, 20240509, temp_mode, 8, csize_au=0.5, tau_mode=mode)
    assert f.a0 is not None and f.a0_mode == mode and f.em0 is not None
    assert f.scan_fields(mode, want_em) == (3 if want_em else 2)
    bursts = _example_bursts()
    ep = list(np.linspace(0.2, 3.4, n_ep) * orc.YEAR) if n_ep > 1 else [1.3 * orc.YEAR]
    a2, e2, t2 = (None if x is None else x.clone()
                  for x in eng.ff_scan(f, bursts, ep, mode, want_em=want_em))
    tav = eng.tavg(f).clone()
    a0, f.a0 = f.a0, None                                   # compact
    a1, e1, t1 = (None if x is None else x.clone()
                  for x in eng.ff_scan(f, bursts, ep, mode, want_em=want_em))
    em0, f.em0 = f.em0, None                                # wide
    aw, ew, tw = eng.ff_scan(f, bursts, ep, mode, want_em=want_em)
    t_single = eng.ff_scan(f, bursts, ep[:1], mode)[2]
    eng.synchronize()
    same = _eq if n_ep < 16 else _close          # (wide: direct tiles only)
    assert _eq(a2, a1) and same(a2, aw)
    if want_em:
        assert _eq(e2, e1) and same(e2, ew)
    else:
        assert e2 is None
    # T_avg: the one-off pass == what a single-epoch scan of either other layout derives ==
    # what the scan call itself returns on the tau layout
    assert _eq(tav, t_single) and _eq(t2, tav)
    f.em0, f.a0 = em0, a0


@pytest.mark.parametrize("producer", ["synth0", "synth1", "cfg1_example", "tilted"])
def test_producers_write_the_tau_field_themselves(eng, producer):
    """The synthetic generator and K4 emit a0 in their own pass, bit-identical to what
    rjp_tau_field derives from em0 and temp; a model holding ONLY a0 and ts (16 B/cell
    resident) scans to the same tau sums and refuses what it cannot serve."""
    import torch
    from rajepy_amd import engine as E
    from rajepy_amd._lib import RjprtError
    from rajepy_amd.classes import geometry_struct
    if producer.startswith("synth"):
        shape, tm = (6, 40, 32), int(producer[-1])
        mode = E.RJP_GFF_SCALAR if tm == 0 else E.RJP_GFF_POWERLAW
        f = eng.synth_fields(shape, 4711, tm, 8, csize_au=0.5, tau_mode=mode)
    else:
        z, meta, p = U.load_golden(producer)
        jet = orc.OracleJet(p)
        geom = geometry_struct(jet.params, jet.nx, jet.ny, jet.nz)
        mode = E.RJP_GFF_SCALAR if p["power_laws"]["q_T"] == 0. else E.RJP_GFF_POWERLAW
        f = eng.build_fields(geom, 8, want_ts=True, want_vy=False, want_raw=False, tau_mode=mode)
    direct = f.a0.clone()
    eng.tau_layout(f, mode)
    eng.synchronize()
    assert bool((direct.view(torch.int64) == f.a0.view(torch.int64)).all())
    # the flag rides in the sign bit, the magnitude is em0 * T^p
    assert bool((torch.signbit(f.a0) == torch.signbit(f.em0)).all())
    ep = [0.4 * orc.YEAR, 1.1 * orc.YEAR, 2.0 * orc.YEAR]
    bursts = E.make_bursts([(0.5 * orc.YEAR, 4., 2e6)], [(1.0 * orc.YEAR, 1.5, 6e6)])
    eng.compute_y_bounds(f)
    ref = eng.ff_scan(f, bursts, ep, mode, want_em=False, want_tavg=False)[0].clone()
    lean = E.DeviceFields(f.shape, 8, f.csize_au, None, None, None, None, ts=f.ts)
    lean.a0, lean.a0_mode = f.a0, mode
    got = eng.ff_scan(lean, bursts, ep, mode, want_em=False, want_tavg=False)[0]
    eng.synchronize()
    assert _eq(got, ref)
    with pytest.raises(RjprtError, match="tau layout only"):
        eng.ff_scan(lean, bursts, ep, mode, want_em=True, want_tavg=False)
    with pytest.raises(RjprtError, match="tau layout only"):
        eng.ff_scan(lean, bursts, ep, mode, want_em=False, want_tavg=True)
    # built for the other Gaunt branch: the field is ignored, and a lean model has nothing else
    other = E.RJP_GFF_POWERLAW if mode == E.RJP_GFF_SCALAR else E.RJP_GFF_SCALAR
    with pytest.raises(RjprtError, match="nd/xi/temp/pf"):
        eng.ff_scan(lean, bursts, ep, other, want_em=False, want_tavg=False)


def test_tau_field_for_the_other_gaunt_branch_is_not_used(eng):
    """a0 carries ONE temperature power: a scan in the other Gaunt mode must fall back to the
    compact layout (and give that layout's numbers), not reuse the field."""
    from rajepy_amd import engine as E
    f = eng.synth_fields((4, 48, 32), 99, 1, 8, csize_au=0.5, tau_mode=E.RJP_GFF_SCALAR)
    bursts = _example_bursts()
    ep = [0.7 * orc.YEAR]
    assert f.scan_fields(E.RJP_GFF_POWERLAW, False) == 3
    got = eng.ff_scan(f, bursts, ep, E.RJP_GFF_POWERLAW)[0].clone()
    right = eng.ff_scan(f, bursts, ep, E.RJP_GFF_SCALAR)[0].clone()
    f.a0 = None
    ref_p = eng.ff_scan(f, bursts, ep, E.RJP_GFF_POWERLAW)[0]
    ref_s = eng.ff_scan(f, bursts, ep, E.RJP_GFF_SCALAR)[0]
    eng.synchronize()
    assert _eq(got, ref_p) and _eq(right, ref_s) and not _eq(ref_p, ref_s)


@pytest.mark.parametrize("n_ep", [1, 5, 16, 32])
@pytest.mark.parametrize("bursts_on", [True, False])
def test_tau_layout_keeps_numpys_nan_semantics(eng, n_ep, bursts_on):
    """NaN / zero / negative / infinite entries sprinkled over every field, odd n_y tails, an
    odd n_z (one sightline per lane): the tau layout masks exactly what the wide arithmetic
    masks (nansum per product, classes.py:1116-1120, 1395-1432), bit for bit, with and without
    bursts, and the T_avg pass counts exactly the cells with T > 0."""
    from rajepy_amd import engine as E
    rng = np.random.default_rng(7 + n_ep)
    for shape in ((5, 37, 24), (3, 21, 9)):
        g = U.synth_host(shape, 31, 1)
        for k, vals in (("nd", [np.nan, 0.0]), ("xi", [np.nan, 0.0]),
                        ("temp", [np.nan, 0.0, -3.0, np.inf, 1e-40, 1e35]),
                        ("ff", [np.nan, 0.0]), ("ts", [np.nan])):
            m = rng.random(shape) < 0.08
            g[k] = np.where(m, rng.choice(vals, size=shape), g[k])
        g["temp"][0, :, 1] = np.nan                    # an empty sightline for T_avg
        f = eng.upload_fields(g["nd"], g["xi"], g["temp"], g["ff"], g["areas"], g["ts"],
                              g["rr"] < 0, csize_au=0.7, dtype=8)
        assert f.em0 is not None
        bursts = _example_bursts() if bursts_on else None
        ep = list(np.linspace(0.1, 3.0, n_ep) * orc.YEAR) if n_ep > 1 else [0.9 * orc.YEAR]
        for mode in (E.RJP_GFF_SCALAR, E.RJP_GFF_POWERLAW):
            eng.tau_layout(f, mode)
            assert f.a0 is not None
            a2, e2, t2 = (x.clone() for x in eng.ff_scan(f, bursts, ep, mode))
            a2n = eng.ff_scan(f, bursts, ep, mode, want_em=False, want_tavg=False)[0].clone()
            a0 = f.a0
            f.a0 = None
            a1, e1, t1 = (x.clone() for x in eng.ff_scan(f, bursts, ep, mode))
            em0, f.em0 = f.em0, None
            aw, ew, tw = eng.ff_scan(f, bursts, ep, mode)
            eng.synchronize()
            f.em0, f.a0 = em0, a0
            same = _eq if (n_ep < 16 or not bursts_on) else _close
            assert _eq(a2, a1) and same(a2, aw) and _eq(a2n, a2)
            assert _eq(e2, e1) and same(e2, ew)
            assert _eq(t2, t1) and _eq(t2, tw)
        tav = eng.tavg(f).cpu().numpy().reshape(shape[0], shape[2])
        with np.errstate(all="ignore"):
            tpos = np.where(g["temp"] > 0, g["temp"], np.nan)
            cnt = np.sum(g["temp"] > 0, axis=1)
            want = np.where(cnt > 0, np.nansum(np.where(np.isnan(tpos), 0., tpos), axis=1) /
                            np.maximum(cnt, 1), np.nan)
        assert np.array_equal(np.isnan(tav), np.isnan(want))
        ok = np.isfinite(want)
        np.testing.assert_allclose(tav[ok], want[ok], rtol=1e-13)
        assert np.array_equal(np.isinf(tav), np.isinf(want))


def test_tau_layout_with_occupied_y_ranges_and_many_bursts(eng):
    """The example jet built by K4 (0.4 % of the grid occupied, y-ranges attached) with twelve
    red and thirteen blue bursts (overflow table, step tables of the two-operation recurrence):
    tau layout == compact layout bit for bit for direct, 16- and 32-epoch tiles, and both
    follow the oracle's chained closures at 1e-11."""
    from rajepy_amd import engine as E
    from rajepy_amd.classes import geometry_struct
    z, meta, p = U.load_golden("cfg1_example")
    rng = np.random.default_rng(5)
    nb = 25
    p = copy.deepcopy(p)
    p["ejection"] = {"t_0": np.sort(rng.uniform(0.1, 3.0, nb)), "hl": rng.uniform(0.1, 0.6, nb),
                     "chi": rng.uniform(1.5, 8., nb),
                     "which": np.array(["R"] * 12 + ["B"] * 13)}
    jet = orc.OracleJet(p)
    geom = geometry_struct(jet.params, jet.nx, jet.ny, jet.nz)
    f = eng.build_fields(geom, 8, want_ts=True, want_vy=False, want_raw=False,
                         tau_mode=E.RJP_GFF_SCALAR)
    eng.compute_y_bounds(f)
    bursts = U.bursts_from_oracle(jet)
    for years in ([0.0, 0.33, 0.9, 1.7, 2.95], list(np.linspace(0., 3., 16)),
                  list(np.linspace(0., 3.1, 32))):
        ep = [y * orc.YEAR for y in years]
        got = eng.ff_scan(f, bursts, ep, E.RJP_GFF_SCALAR, want_em=False, want_tavg=False)[0].clone()
        gem = eng.ff_scan(f, bursts, ep, E.RJP_GFF_SCALAR, want_tavg=False)
        a0, f.a0 = f.a0, None
        ref = eng.ff_scan(f, bursts, ep, E.RJP_GFF_SCALAR, want_em=False, want_tavg=False)[0].clone()
        rem = eng.ff_scan(f, bursts, ep, E.RJP_GFF_SCALAR, want_tavg=False)
        eng.synchronize()
        f.a0 = a0
        assert _eq(got, ref) and _eq(gem[0], rem[0]) and _eq(gem[1], rem[1])
    # against the oracle at two of the epochs (tau = ctau * sumA)
    from rajepy_amd.maths import physics as ph
    gv = [ph.gff(5e9, p["properties"]["T_0"])]
    ctau, cflux = E.ff_channel_coeffs([5e9], jet.csize, p["target"]["dist"], E.RJP_GFF_SCALAR, gv)
    ep = [y * orc.YEAR for y in np.linspace(0., 3.1, 32)]
    sumA = eng.ff_scan(f, bursts, ep, E.RJP_GFF_SCALAR, want_em=False, want_tavg=False)[0]
    tau, _, _ = eng.ff_maps(sumA, eng.tavg(f), ctau, cflux, want_flux=False, want_ftot=False)
    tau = tau.cpu().numpy().reshape(32, jet.nx, jet.nz)
    for k in (3, 20):
        jet.time = ep[k]
        np.testing.assert_allclose(tau[k], jet.optical_depth_ff(5e9), rtol=1e-11, atol=0)


@pytest.mark.parametrize("which", ["B", "R"])
def test_nan_launch_times_only_mask_cells_of_a_jet_that_has_bursts(eng, which):
    """The reference's Gaussians propagate a NaN launch time into the cell's density (dropped
    by nansum, classes.py:442-448, 866-875) -- but the mass-loss rate of a jet WITHOUT any
    registered burst is the steady-state constant, whatever the launch time
    (classes.py:232-233): its cells keep contributing with chi = 1.  Bursts in one jet only,
    NaN launch times sprinkled over both (densities finite): K1 on every layout and tile kind,
    the collapse=False cells and K3, all against the oracle."""
    from rajepy_amd import _lib, engine as E
    from rajepy_amd.maths import physics as ph, rrls
    shape = (4, 45, 16)
    g = U.synth_host(shape, 777, 0)
    rng = np.random.default_rng(3)
    g["ts"] = np.where(rng.random(shape) < 0.15, np.nan, g["ts"])
    p = copy.deepcopy(U.load_golden("cfg1_example")[2])
    p["ejection"] = {"t_0": np.array([0.6, 1.4]), "hl": np.array([0.3, 0.5]),
                     "chi": np.array([4., 7.]), "which": np.array([which, which])}
    p["grid"].update(n_x=shape[0], n_y=shape[1], n_z=shape[2])
    jet = orc.OracleJet.from_fields(p, g["nd"], g["xi"], g["temp"], g["ff"], g["areas"],
                                    g["ts"], g["rr"], g["vy"])
    bursts = U.bursts_from_oracle(jet)
    f = eng.upload_fields(g["nd"], g["xi"], g["temp"], g["ff"], g["areas"], g["ts"],
                          g["rr"] < 0, vy=g["vy"], csize_au=jet.csize, dtype=8)
    eng.tau_layout(f, E.RJP_GFF_SCALAR)
    freqs = np.array([5e9])
    gv = [ph.gff(5e9, p["properties"]["T_0"])]
    ctau, cflux = E.ff_channel_coeffs(freqs, jet.csize, p["target"]["dist"], E.RJP_GFF_SCALAR, gv)
    layouts = {"tau": (f.a0, f.em0), "compact": (None, f.em0), "wide": (None, None)}
    for years in ([0.9], [0.2, 0.7, 1.5], list(np.linspace(0., 3., 16)),
                  list(np.linspace(0., 3.1, 32))):
        ep = [y * orc.YEAR for y in years]
        ref_tau, ref_em = [], []
        for t in (ep[0], ep[-1]):
            jet.time = t
            ref_tau.append(jet.optical_depth_ff(5e9))
            ref_em.append(jet.emission_measure())
        for name, (a0, em0) in layouts.items():
            f.a0, f.em0 = a0, em0
            sumA, em, _ = eng.ff_scan(f, bursts, ep, E.RJP_GFF_SCALAR, want_tavg=False)
            eng.synchronize()
            tau = (ctau[0] * sumA.cpu().numpy()).reshape(len(ep), shape[0], shape[2])
            emh = em.cpu().numpy().reshape(len(ep), shape[0], shape[2])
            for k, e in enumerate((0, len(ep) - 1)):
                np.testing.assert_allclose(tau[e], ref_tau[k], rtol=1e-11, err_msg=name)
                np.testing.assert_allclose(emh[e], ref_em[k], rtol=1e-11, err_msg=name)
    f.a0, f.em0 = layouts["tau"]
    # the masked cells really were in play: dropping the NaNs changes the maps
    jet.time = 0.9 * orc.YEAR
    with_nan = jet.optical_depth_ff(5e9)
    jet2 = orc.OracleJet.from_fields(p, g["nd"], g["xi"], g["temp"], g["ff"], g["areas"],
                                     np.nan_to_num(g["ts"], nan=0.0), g["rr"], g["vy"])
    jet2.time = jet.time
    assert not np.allclose(with_nan, jet2.optical_depth_ff(5e9), rtol=1e-6)
    # collapse=False cells and K3
    cells = eng.ff_cells(f, bursts, jet.time, E.RJP_GFF_SCALAR, ctau)
    line = _lib.Line(**rrls.line_constants("H66a"))
    nu0 = rrls.rrl_nu_0("H", 66, 1)
    eng.synchronize()
    np.testing.assert_allclose(cells.cpu().numpy().reshape(shape),
                               jet.optical_depth_ff(5e9, collapse=False), rtol=1e-11)
    for nchan in (8, 40):
        rf = orc.chan_freqs(nu0, nchan * 2e5, 2e5)
        trrl = eng.rrl_scan(f, bursts, jet.time, line, rf)
        eng.synchronize()
        ref = jet.optical_depth_rrl("H66a", np.asarray(rf))
        np.testing.assert_allclose(trrl.cpu().numpy().reshape(ref.shape), ref,
                                   rtol=U.k3_rtol(nchan))
