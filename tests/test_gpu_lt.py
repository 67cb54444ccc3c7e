"""Epoch sweeps on the launch-time-ordered layout (rajepy_amd/csrc/ff_lt.hip) through the C-ABI.

The layout buckets every group of 64 sightlines by (jet, launch-time bin) once per model
(rjp_lt_count + rjp_lt_fill); a sweep of 12-32 epochs then accumulates the Chebyshev moments of
sum_y a0 chi(t_e - ts)^2 (classes.py:861-875, 1395-1432) in registers and contracts them per bin.
Same expansion, same 1e-11 acceptance as the LDS moment path, so the maps must agree with the
epoch tiles at 5e-11 and with the oracle's chained closures at 1e-10; what the layout cannot
serve (EM maps, more than 32 epochs, a stale layout, bursts too narrow for any order) must fall
back to the other paths."""
import copy

import numpy as np
import pytest

from oracle import rt_oracle as orc
from tests import gpu_util as U

pytestmark = pytest.mark.gpu
RTOL = 5e-11


@pytest.fixture(scope="module")
def eng():
    from rajepy_amd.engine import RTEngine
    e = RTEngine(0)
    e.force_moments = True
    e.cache_moments = False        # (paths are asserted sweep by sweep; the cache has its own test)          # (the fallbacks below: the LDS moments, not the cost model)
    yield e
    e.close()


def _jet(shape, seed, ejection=None, temp_mode=0):
    g = U.synth_host(shape, seed, temp_mode)
    p = copy.deepcopy(U.load_golden("cfg1_example")[2])
    p["ejection"] = ejection if ejection is not None else U.example_bursts_params()
    if temp_mode:
        p["power_laws"]["q_T"] = -0.5
    p["grid"].update(n_x=shape[0], n_y=shape[1], n_z=shape[2])
    return g, p


def _oracle(p, g):
    return orc.OracleJet.from_fields(p, g["nd"], g["xi"], g["temp"], g["ff"], g["areas"],
                                     g["ts"], g["rr"], g["vy"])


def _upload(eng, g, csize, mode):
    f = eng.upload_fields(g["nd"], g["xi"], g["temp"], g["ff"], g["areas"], g["ts"],
                          g["rr"] < 0, csize_au=csize, dtype=8)
    eng.tau_layout(f, mode)
    return f


def _three(eng, f, bursts, ep, mode):
    """The sweep by the layout, by the LDS moments and by the tiles."""
    eng.use_lt, eng.use_moments = True, True
    a = eng.ff_scan(f, bursts, ep, mode, want_em=False, want_tavg=False)[0].clone()
    pa = eng.last_scan_path()
    shape = eng.last_moment_shape
    eng.use_lt = False
    b = eng.ff_scan(f, bursts, ep, mode, want_em=False, want_tavg=False)[0].clone()
    pb = eng.last_scan_path()[0]
    eng.use_moments = False
    c = eng.ff_scan(f, bursts, ep, mode, want_em=False, want_tavg=False)[0].clone()
    assert eng.last_scan_path()[0] == "tiles"
    eng.use_lt, eng.use_moments = True, True
    eng.synchronize()
    return a.cpu().numpy(), b.cpu().numpy(), c.cpu().numpy(), pa, pb, shape


@pytest.mark.parametrize("temp_mode", [0, 1])
@pytest.mark.parametrize("shape,K", [((6, 200, 48), 32), ((3, 130, 37), 20), ((1, 90, 64), 53),
                                     ((5, 64, 13), 16)])
@pytest.mark.parametrize("years", [list(np.linspace(0., 5., 32)), list(np.linspace(0.3, 4.1, 12)),
                                   sorted([0.0, 0.11, 0.5, 0.52, 0.9, 1.0, 1.3, 1.31, 1.9, 2.2, 2.25,
                                           2.8, 3.3, 3.9, 4.4, 4.95, 5.0])])
def test_lt_sweeps_agree_with_the_tiles_the_lds_moments_and_the_oracle(eng, temp_mode, shape, K,
                                                                       years):
    """Map widths that are and are not multiples of 64 sightlines (the last group is partly
    empty), both Gaunt branches, uniform and irregular epoch lists, several bin counts."""
    from rajepy_amd import engine as E
    from rajepy_amd.maths import physics as ph
    g, p = _jet(shape, 20240777 + K, temp_mode=temp_mode)
    jet = _oracle(p, g)
    mode = E.RJP_GFF_SCALAR if temp_mode == 0 else E.RJP_GFF_POWERLAW
    f = _upload(eng, g, jet.csize, mode)
    lt = eng.build_lt(f, K)
    assert lt["rows"] % 4 == 0 and lt["rows"] * 64 >= np.isfinite(g["nd"]).sum()
    bursts = U.bursts_from_oracle(jet)
    ep = [y * orc.YEAR for y in years]
    a, b, c, (path, err), pb, (k_used, n_used) = _three(eng, f, bursts, ep, mode)
    assert path == "lt" and pb == "moments" and 0 < err <= 1e-11
    assert k_used == K and n_used % 4 == 0 and 8 <= n_used <= 32
    np.testing.assert_allclose(a, c, rtol=RTOL)
    np.testing.assert_allclose(a, b, rtol=RTOL)
    gv = [ph.gff(5e9, p["properties"]["T_0"])] if temp_mode == 0 else None
    ctau, _ = E.ff_channel_coeffs([5e9], jet.csize, p["target"]["dist"], mode, gv)
    for e in (0, len(ep) // 2, len(ep) - 1):
        jet.time = ep[e]
        np.testing.assert_allclose(ctau[0] * a[e].reshape(shape[0], shape[2]),
                                   jet.optical_depth_ff(5e9), rtol=1e-10)


def test_lt_layout_holds_every_contributing_cell_once_and_is_reproducible(eng):
    """Per sightline the layout's weights sum to the sightline's |a0| over the cells it keeps;
    rows come in chunks of 4 per (group, bin); padding has zero weight and an in-bin launch time;
    two builds are identical and two sweeps of one layout are bit-identical (fixed order)."""
    import torch
    from rajepy_amd import engine as E
    shape, K = (4, 150, 80), 16
    g, p = _jet(shape, 424242)
    jet = _oracle(p, g)
    f = _upload(eng, g, jet.csize, E.RJP_GFF_SCALAR)
    lt1 = eng.build_lt(f, K)
    cells1, off1 = lt1["cells"].clone(), lt1["rowoff"].clone()
    lt2 = eng.build_lt(f, K)
    assert torch.equal(cells1, lt2["cells"]) and torch.equal(off1, lt2["rowoff"])
    npix = shape[0] * shape[2]
    G = (npix + 63) // 64
    off = off1.cpu().numpy()
    assert off.shape == (G * 2 * K + 1,) and off[0] == 0 and off[-1] == lt1["rows"]
    assert np.all(np.diff(off) % 4 == 0) and np.all(np.diff(off) >= 0)
    cells = cells1.cpu().numpy().reshape(-1, 64, 2)
    a0 = np.abs(f.a0.cpu().numpy().reshape(shape))
    want = np.nansum(a0, axis=1).reshape(-1)                     # per sightline
    per = np.zeros(G * 64)
    for gi in range(G):
        r0, r1 = off[gi * 2 * K], off[(gi + 1) * 2 * K]
        per[gi * 64:(gi + 1) * 64] = cells[r0:r1, :, 0].sum(axis=0)
    np.testing.assert_allclose(per[:npix], want, rtol=1e-13)
    assert np.all(per[npix:] == 0.0)
    # launch times of every slot lie inside the slot's bin (padding: the bin centre)
    lo, hi = f.ts_range
    h = (hi - lo) / K
    for gi in range(G):
        for q in range(2 * K):
            r0, r1 = off[gi * 2 * K + q], off[gi * 2 * K + q + 1]
            if r1 > r0:
                t = cells[r0:r1, :, 1]
                k = q % K
                assert t.min() >= lo + k * h * (1 - 1e-12) - 1e-6 * h
                assert t.max() <= lo + (k + 1) * h * (1 + 1e-12) + 1e-6 * h
    bursts = U.bursts_from_oracle(jet)
    ep = [y * orc.YEAR for y in np.linspace(0., 4., 24)]
    r1 = eng.ff_scan(f, bursts, ep, E.RJP_GFF_SCALAR, want_em=False, want_tavg=False)[0].clone()
    assert eng.last_scan_path()[0] == "lt"
    r2 = eng.ff_scan(f, bursts, ep, E.RJP_GFF_SCALAR, want_em=False, want_tavg=False)[0].clone()
    assert torch.equal(r1.view(torch.int64), r2.view(torch.int64))


@pytest.mark.parametrize("which", ["RB", "R", "B"])
def test_lt_numpy_nan_semantics(eng, which):
    """NaN / zero weights are dropped, a NaN launch time drops the cell when its jet has bursts
    and counts with chi = 1 when it has none (bursts in one jet only), an infinite weight (T = 0)
    makes its sightline +inf: everything as the epoch tiles do it."""
    from rajepy_amd import engine as E
    shape = (3, 120, 70)
    ej = U.example_bursts_params()
    keep = [i for i, w in enumerate(ej["which"]) if any(c in str(w) for c in which)]
    ej = {k: np.asarray(v)[keep] for k, v in ej.items()}
    ej["which"] = np.array([w if which == "RB" else which for w in ej["which"]])
    g, p = _jet(shape, 99, ejection=ej)
    rng = np.random.default_rng(7)
    for name, frac in (("nd", 0.06), ("ts", 0.05), ("xi", 0.03)):
        m = rng.random(shape) < frac
        g[name] = np.where(m, np.nan, g[name])
    g["ff"] = np.where(rng.random(shape) < 0.04, 0.0, g["ff"])        # exact zeros
    g["temp"][1, 17, 5] = 0.0                                          # T^-1.5 = inf
    g["nd"][1, 17, 5], g["xi"][1, 17, 5], g["ff"][1, 17, 5] = 1e6, 0.2, 1.0
    g["ts"][1, 17, 5] = 1.0 * orc.YEAR
    jet = _oracle(p, g)
    f = _upload(eng, g, jet.csize, E.RJP_GFF_SCALAR)
    eng.build_lt(f, 24)
    bursts = U.bursts_from_oracle(jet)
    ep = [y * orc.YEAR for y in np.linspace(0.1, 4.7, 20)]
    a, b, c, (path, _), pb, _ = _three(eng, f, bursts, ep, E.RJP_GFF_SCALAR)
    assert path == "lt"
    assert np.isinf(c[:, 1 * shape[2] + 5]).all()
    assert np.array_equal(np.isinf(a), np.isinf(c)) and np.array_equal(np.isnan(a), np.isnan(c))
    fin = np.isfinite(c)
    np.testing.assert_allclose(a[fin], c[fin], rtol=RTOL)
    np.testing.assert_array_equal(a == 0.0, c == 0.0)


def test_what_the_layout_cannot_serve_takes_the_other_paths(eng):
    """EM maps and sweeps of more than 32 epochs run the LDS moments, fewer than 12 epochs the
    tiles; a layout built from launch times that were replaced since is ignored; bursts too
    narrow for every order at the layout's bin width keep the tiles (deterministic sigma guard:
    a burst narrower than the node spacing could hide between all nodes and test points)."""
    from rajepy_amd import engine as E
    shape = (2, 140, 64)
    g, p = _jet(shape, 5150)
    jet = _oracle(p, g)
    f = _upload(eng, g, jet.csize, E.RJP_GFF_SCALAR)
    eng.build_lt(f, 32)
    bursts = U.bursts_from_oracle(jet)
    yr = orc.YEAR
    scan = lambda ep, em=False: eng.ff_scan(f, bursts, ep, 0, want_em=em, want_tavg=False)
    scan([y * yr for y in np.linspace(0, 5, 32)])
    assert eng.last_scan_path()[0] == "lt"
    scan([y * yr for y in np.linspace(0, 5, 32)], em=True)
    assert eng.last_scan_path()[0] == "moments"
    scan([y * yr for y in np.linspace(0, 5, 33)])
    assert eng.last_scan_path()[0] == "moments"
    scan([y * yr for y in np.linspace(0, 5, 11)])
    assert eng.last_scan_path()[0] == "tiles"
    # bursts far too narrow: sigma = 1e-4 of the bin width
    lo, hi = f.ts_range
    h = (hi - lo) / 32
    narrow = E.make_bursts([(0.4 * (lo + hi), 3.0, 1e-4 * h)], [(0.6 * (lo + hi), 2.0, 1e-4 * h)])
    ep = [y * yr for y in np.linspace(0, 5, 16)]
    got = eng.ff_scan(f, narrow, ep, 0, want_em=False, want_tavg=False)[0].clone()
    assert eng.last_scan_path()[0] == "tiles"
    eng.use_moments = False
    ref = eng.ff_scan(f, narrow, ep, 0, want_em=False, want_tavg=False)[0]
    eng.use_moments = True
    assert bool((got == ref).all())
    # the launch times are replaced: the old layout no longer belongs to the fields
    eng.replace_field(f, "ts", np.where(np.isnan(g["ts"]), np.nan, 0.5 * g["ts"]))
    assert f.struct().d_lt_cells is None
    scan([y * yr for y in np.linspace(0, 5, 32)])
    assert eng.last_scan_path()[0] == "moments"
    eng.build_lt(f, 32)
    scan([y * yr for y in np.linspace(0, 5, 32)])
    assert eng.last_scan_path()[0] == "lt"


def test_lt_abi_refusals(eng):
    import ctypes as C
    from rajepy_amd import _lib
    lib = eng.lib
    f = eng.synth_fields((2, 16, 64), 1, 0, 8, tau_mode=0)
    eng.launch_time_range(f)
    fs = f.struct()
    tot = C.c_int64()
    off = eng._f64(1024)
    st = eng._stream()
    assert lib.rjp_lt_rowoff_entries(2, 64, 0) == 0 and lib.rjp_lt_rowoff_entries(2, 64, 81) == 0
    assert lib.rjp_lt_rowoff_entries(2, 64, 5) == 2 * 2 * 5 + 1
    assert lib.rjp_lt_count(eng.ctx, C.byref(fs), 0, off.data_ptr(), C.byref(tot), st) == -1
    assert lib.rjp_lt_count(eng.ctx, C.byref(fs), 5, None, C.byref(tot), st) == -1
    no_range = f.struct()
    no_range.ts_lo = no_range.ts_hi = 0.0
    assert lib.rjp_lt_count(eng.ctx, C.byref(no_range), 5, off.data_ptr(), C.byref(tot), st) == -1
    assert b"ts_lo" in lib.rjp_last_error(eng.ctx)
    no_a0 = f.struct()
    no_a0.d_a0 = None
    assert lib.rjp_lt_count(eng.ctx, C.byref(no_a0), 5, off.data_ptr(), C.byref(tot), st) == -1
    assert lib.rjp_lt_fill(eng.ctx, C.byref(fs), 5, off.data_ptr(), None, None, st) == -1
    with pytest.raises(ValueError):
        eng.build_lt(eng.synth_fields((2, 16, 64), 1, 0, 4), 8)       # f32 storage
    eng.synchronize()


def test_jetmodel_light_curves_on_the_prepared_layout(eng, tmp_path):
    """JetModel.prepare_epoch_sweeps + flux_vs_time on the example jet (K4-built fields with
    occupied y-ranges, most sightlines empty): 30 epochs through the layout reproduce the
    reference's anchors (total flux at 5 GHz at t = 0 / 0.5 / 1 / 2 / 3 yr) and
    the tiles' light curve; replacing the launch times drops the layout."""
    from rajepy_amd import classes, logger
    from tests.test_host_logic import example_params
    jm = classes.JetModel(example_params(), log=logger.Log(str(tmp_path / "a.log"), verbose=False),
                          engine=eng)
    info = jm.prepare_epoch_sweeps(24)
    assert info["K"] == 24 and info["rows"] > 0
    times = np.array(sorted(set(np.linspace(0., 3., 28)) | {0.5, 1.0, 2.0})) * orc.YEAR
    assert 12 <= len(times) <= 32
    lc = jm.flux_vs_time(times, [5e9])[:, 0]
    assert eng.last_scan_path()[0] == "lt"
    ref = {0.0: 1.158223515e-3, 0.5: 1.279591671e-3, 1.0: 1.379008153e-3, 2.0: 1.429076011e-3,
           3.0: 1.418864143e-3}
    for yr, want in ref.items():
        k = int(np.argmin(np.abs(times - yr * orc.YEAR)))
        assert abs(times[k] - yr * orc.YEAR) < 1.0
        assert abs(lc[k] - want) <= 2e-9 * want
    eng.use_moments = False
    jm2 = classes.JetModel(example_params(), log=jm.log, engine=eng)
    lc2 = jm2.flux_vs_time(times, [5e9])[:, 0]
    eng.use_moments = True
    np.testing.assert_allclose(lc, lc2, rtol=RTOL)
    jm.ts = jm.ts * 1.0 + 0.0                      # the setter: a new launch-time tensor
    assert jm.device_fields.struct().d_lt_cells is None


def test_mixed_jet_sightlines_on_every_sweep_path(eng):
    """Sightlines that cross BOTH lobes (a real, inclined jet: the red / blue flag changes along
    y; the synthetic set has one jet per sightline): the layout keys its buckets by (jet, bin),
    the LDS moments index their table by jet, the table scan picks the jet's half per cell --
    every path against the oracle, whose chi_xyz selects the lobe per cell (classes.py:866-875)."""
    from rajepy_amd import engine as E
    from rajepy_amd.maths import physics as ph
    shape = (3, 150, 40)
    g, p = _jet(shape, 2718)
    rng = np.random.default_rng(5)
    g["rr"] = np.where(rng.random(shape) < 0.45, -1.0, 1.0)        # lobe per CELL
    jet = _oracle(p, g)
    f = _upload(eng, g, jet.csize, E.RJP_GFF_SCALAR)
    eng.build_lt(f, 24)
    bursts = U.bursts_from_oracle(jet)
    ep = [y * orc.YEAR for y in np.linspace(0.2, 4.6, 20)]
    a, b, c, (path, _), pb, _ = _three(eng, f, bursts, ep, E.RJP_GFF_SCALAR)
    assert path == "lt" and pb == "moments"
    np.testing.assert_allclose(a, c, rtol=RTOL)
    np.testing.assert_allclose(b, c, rtol=RTOL)
    ctau, _ = E.ff_channel_coeffs([5e9], jet.csize, p["target"]["dist"], E.RJP_GFF_SCALAR,
                                  [ph.gff(5e9, p["properties"]["T_0"])])
    for e in (0, 9, 19):
        jet.time = ep[e]
        np.testing.assert_allclose(ctau[0] * a[e].reshape(shape[0], shape[2]),
                                   jet.optical_depth_ff(5e9), rtol=1e-10)
    # the single-epoch table scan on a map large enough for it, lobes mixed the same way
    big = (64, 80, 512)
    f2 = eng.synth_fields(big, 99, 0, 8, csize_au=0.5, tau_mode=0)
    import torch
    gen = torch.Generator(device=eng.device)
    gen.manual_seed(3)
    flip = torch.rand(f2.ncells, device=eng.device, generator=gen) < 0.5
    for t in (f2.a0, f2.em0, f2.nd):
        t[flip] = -t[flip]                                           # the flag lives in the sign bit
    for ep1 in ([1.0 * orc.YEAR], [2.4 * orc.YEAR]):
        eng.use_chi_table = True
        tab = eng.ff_scan(f2, bursts, ep1, 0, want_em=True, want_tavg=False)
        assert eng.last_scan_path()[0] == "table"
        tab = [t.clone() for t in tab[:2]]
        eng.use_chi_table = False
        ref = eng.ff_scan(f2, bursts, ep1, 0, want_em=True, want_tavg=False)
        eng.use_chi_table = True
        eng.synchronize()
        for got, want in zip(tab, ref[:2]):
            assert ((got - want).abs() / want).max().item() < 3e-12
