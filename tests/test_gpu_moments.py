"""Epoch sweeps by launch-time moments (rajepy_amd/csrc/ff_moments.hip) through the C-ABI.

sum_y a0 chi(t_e - ts)^2 (classes.py:861-875, 1395-1432) is a convolution of the sightline's
launch-time distribution with chi^2: one pass accumulates Chebyshev moments of a0 over launch-time
bins, any number of epochs is a contraction with host-computed coefficient tables.  The host
accepts the expansion only when it is good to 1e-11 for the call's bursts and epochs, so the maps
must agree with the epoch tiles (themselves at 1e-11 of the oracle) and with the oracle; sweeps
the expansion cannot serve (narrow bursts over a wide launch-time range, EM maps, few epochs, no
range given) must run the tiles, bit for bit as before."""
import copy

import numpy as np
import pytest

from oracle import rt_oracle as orc
from tests import gpu_util as U

pytestmark = pytest.mark.gpu
RTOL = 5e-11          # 1e-11 expansion + the tiles' own 1e-11 + summation order


@pytest.fixture(scope="module")
def eng():
    """`force_moments`: the library's cost model would keep the epoch tiles on grids this small
    (the moment path pays up to 10 KiB per sightline, so it wins on long, densely filled sightlines:
    cfg5); the last test checks the model's own decisions."""
    from rajepy_amd.engine import RTEngine
    e = RTEngine(0)
    e.force_moments = True
    e.cache_moments = False        # (paths are asserted sweep by sweep; the cache has its own test)
    yield e
    e.close()


def _jet(shape, seed, ejection=None, temp_mode=0):
    g = U.synth_host(shape, seed, temp_mode)
    p = copy.deepcopy(U.load_golden("cfg1_example")[2])
    p["ejection"] = ejection if ejection is not None else U.example_bursts_params()
    if temp_mode:
        p["power_laws"]["q_T"] = -0.5
    p["grid"].update(n_x=shape[0], n_y=shape[1], n_z=shape[2])
    jet = orc.OracleJet.from_fields(p, g["nd"], g["xi"], g["temp"], g["ff"], g["areas"],
                                    g["ts"], g["rr"], g["vy"])
    return g, p, jet


def _both(eng, f, bursts, ep, mode, **kw):
    eng.use_moments = True
    a = eng.ff_scan(f, bursts, ep, mode, want_em=False, want_tavg=False, **kw)[0].clone()
    path = eng.last_scan_path()
    eng.use_moments = False
    b = eng.ff_scan(f, bursts, ep, mode, want_em=False, want_tavg=False, **kw)[0].clone()
    assert eng.last_scan_path()[0] == "tiles"
    eng.use_moments = True
    eng.synchronize()
    return a.cpu().numpy(), b.cpu().numpy(), path


def _both_em(eng, f, bursts, ep, mode):
    """(sums, EM maps) by the moment path and by the tiles, EM maps asked for."""
    eng.use_moments = True
    a, ea, _ = eng.ff_scan(f, bursts, ep, mode, want_em=True, want_tavg=False)
    a, ea = a.clone(), ea.clone()
    path = eng.last_scan_path()
    eng.use_moments = False
    b, eb, _ = eng.ff_scan(f, bursts, ep, mode, want_em=True, want_tavg=False)
    assert eng.last_scan_path()[0] == "tiles"
    eng.use_moments = True
    eng.synchronize()
    return (a.cpu().numpy(), ea.cpu().numpy()), (b.cpu().numpy(), eb.cpu().numpy()), path


@pytest.mark.parametrize("temp_mode", [0, 1])
@pytest.mark.parametrize("years", [list(np.linspace(0., 5., 32)), list(np.linspace(0.3, 4.1, 12)),
                                   list(np.linspace(0., 5., 37)), list(np.linspace(0., 5., 100)),
                                   sorted([0.0, 0.11, 0.5, 0.52, 0.9, 1.0, 1.3, 1.31, 1.9, 2.2, 2.25,
                                           2.8, 3.3, 3.9, 4.4, 4.95, 5.0])])
def test_moments_agree_with_the_epoch_tiles_and_the_oracle(eng, temp_mode, years):
    """Uniform sweeps of 32 / 12 / 37 / 100 epochs (one to four contraction passes) and an irregular
    list of 17 (which the tiles evaluate directly, 8 epochs at a time): the moment path is
    taken, agrees with the tiles at 5e-11 and with the oracle's chained closures at 1e-10."""
    from rajepy_amd import engine as E
    from rajepy_amd.maths import physics as ph
    shape = (6, 200, 48)
    g, p, jet = _jet(shape, 20240777, temp_mode=temp_mode)
    mode = E.RJP_GFF_SCALAR if temp_mode == 0 else E.RJP_GFF_POWERLAW
    f = eng.upload_fields(g["nd"], g["xi"], g["temp"], g["ff"], g["areas"], g["ts"],
                          g["rr"] < 0, csize_au=jet.csize, dtype=8)
    eng.tau_layout(f, mode)
    lo, hi = eng.launch_time_range(f)
    assert lo == np.nanmin(g["ts"]) and hi == np.nanmax(g["ts"])
    bursts = U.bursts_from_oracle(jet)
    ep = [y * orc.YEAR for y in years]
    mom, til, (path, err) = _both(eng, f, bursts, ep, mode)
    assert path == "moments" and 0 < err <= 1e-11
    np.testing.assert_allclose(mom, til, rtol=RTOL)
    # with the emission-measure maps of every epoch: a second pass takes the moments of em0
    (mom2, em_m), (til2, em_t), (path2, _) = _both_em(eng, f, bursts, ep, mode)
    assert path2 == "moments"
    np.testing.assert_allclose(mom2, til2, rtol=RTOL)
    np.testing.assert_allclose(em_m, em_t, rtol=RTOL)
    gv = [ph.gff(5e9, p["properties"]["T_0"])] if temp_mode == 0 else None
    ctau, _ = E.ff_channel_coeffs([5e9], jet.csize, p["target"]["dist"], mode, gv)
    for e in (0, len(ep) // 2, len(ep) - 1):
        jet.time = ep[e]
        np.testing.assert_allclose(ctau[0] * mom[e].reshape(shape[0], shape[2]),
                                   jet.optical_depth_ff(5e9), rtol=1e-10)
        np.testing.assert_allclose(em_m[e].reshape(shape[0], shape[2]), jet.emission_measure(),
                                   rtol=1e-10)


@pytest.mark.parametrize("hl_scale,shape_kn", [(2.5, (80, 8)), (1.0, (53, 12)), (0.8, (39, 16))])
def test_the_cheapest_shape_that_passes_the_accuracy_check_is_taken(eng, hl_scale, shape_kn):
    """The moment pass exists for three (bins, order) shapes inside its LDS budget; the host
    tries them in order of cost (atomics per cell) and keeps the first whose expansion is
    good to 1e-11 for the call: broad bursts (the example's half lives x 2.5) take (80, 8),
    the example's (53, 12), narrower ones (x 0.8) need (39, 16).  Each against the tiles."""
    from rajepy_amd import engine as E
    shape = (3, 160, 40)
    ej = U.example_bursts_params()
    ej["hl"] = np.asarray(ej["hl"], float) * hl_scale
    g, p, jet = _jet(shape, 4242, ejection=ej)
    f = eng.upload_fields(g["nd"], g["xi"], g["temp"], g["ff"], g["areas"], g["ts"],
                          g["rr"] < 0, csize_au=jet.csize, dtype=8)
    eng.tau_layout(f, E.RJP_GFF_SCALAR)
    ep = [y * orc.YEAR for y in np.linspace(0., 5., 32)]
    mom, til, (path, err) = _both(eng, f, U.bursts_from_oracle(jet), ep, E.RJP_GFF_SCALAR)
    assert path == "moments" and err <= 1e-11
    eng.use_moments = True
    eng.ff_scan(f, U.bursts_from_oracle(jet), ep, E.RJP_GFF_SCALAR, want_em=False, want_tavg=False)
    eng.last_scan_path()
    assert eng.last_moment_shape == shape_kn
    np.testing.assert_allclose(mom, til, rtol=RTOL)


def test_moments_keep_nan_semantics_y_ranges_and_the_burstless_jet(eng):
    """NaN / zero entries in every field, occupied y-ranges attached, an odd n_z (sightline
    groups that straddle rows and a ragged last group), bursts in ONE jet only with NaN launch
    times in both: the moment path drops exactly the cells the reference drops
    (classes.py:232-233, 442-448, 1395-1432) -- against the tiles and the oracle."""
    from rajepy_amd import engine as E
    from rajepy_amd.maths import physics as ph
    shape = (5, 90, 23)
    ej = {"t_0": np.array([0.6, 1.4, 2.6]), "hl": np.array([0.4, 0.5, 0.45]),
          "chi": np.array([4., 7., 3.]), "which": np.array(["B", "B", "B"])}
    g, p, jet = _jet(shape, 99, ejection=ej)
    rng = np.random.default_rng(8)
    for k, vals in (("nd", [np.nan, 0.0]), ("xi", [np.nan]), ("temp", [np.nan, 0.0, -1.0]),
                    ("ff", [np.nan, 0.0]), ("ts", [np.nan])):
        m = rng.random(shape) < 0.1
        g[k] = np.where(m, rng.choice(vals, size=shape), g[k])
    g["nd"][:, :7, :] = np.nan                      # empty rows at both ends: y-ranges matter
    g["nd"][:, -9:, :] = np.nan
    jet = orc.OracleJet.from_fields(p, g["nd"], g["xi"], g["temp"], g["ff"], g["areas"],
                                    g["ts"], g["rr"], g["vy"])
    f = eng.upload_fields(g["nd"], g["xi"], g["temp"], g["ff"], g["areas"], g["ts"],
                          g["rr"] < 0, csize_au=jet.csize, dtype=8)
    eng.tau_layout(f, E.RJP_GFF_SCALAR)
    eng.compute_y_bounds(f)
    bursts = U.bursts_from_oracle(jet)
    ep = [y * orc.YEAR for y in np.linspace(0., 4., 20)]
    mom, til, (path, err) = _both(eng, f, bursts, ep, E.RJP_GFF_SCALAR)
    assert path == "moments"
    assert np.array_equal(mom == 0, til == 0)
    np.testing.assert_allclose(mom, til, rtol=RTOL)
    ctau, _ = E.ff_channel_coeffs([5e9], jet.csize, p["target"]["dist"], E.RJP_GFF_SCALAR,
                                  [ph.gff(5e9, p["properties"]["T_0"])])
    for e in (0, 9, 19):
        jet.time = ep[e]
        with np.errstate(all="ignore"):
            ref = jet.optical_depth_ff(5e9)
        np.testing.assert_allclose(ctau[0] * mom[e].reshape(shape[0], shape[2]), ref, rtol=1e-10)


def test_sweeps_the_expansion_cannot_serve_run_the_tiles(eng):
    """(a) a burst far narrower than a launch-time bin: the host's accuracy check refuses the
    tables and the tiles run -- the result is the tiles' bit for bit; (b) EM maps asked for
    without the em0 field, (c) fewer than 12 epochs, (d) no launch-time range in the struct: tiles as well."""
    from rajepy_amd import engine as E
    shape = (4, 64, 32)
    ej = {"t_0": np.array([1.0, 2.0]), "hl": np.array([0.004, 0.3]), "chi": np.array([6., 3.]),
          "which": np.array(["RB", "RB"])}
    g, p, jet = _jet(shape, 5, ejection=ej)
    f = eng.upload_fields(g["nd"], g["xi"], g["temp"], g["ff"], g["areas"], g["ts"],
                          g["rr"] < 0, csize_au=jet.csize, dtype=8)
    eng.tau_layout(f, E.RJP_GFF_SCALAR)
    ep = [y * orc.YEAR for y in np.linspace(0., 4., 16)]
    narrow = U.bursts_from_oracle(jet)
    mom, til, (path, _) = _both(eng, f, narrow, ep, E.RJP_GFF_SCALAR)
    assert path == "tiles" and np.array_equal(mom, til)
    wide = E.make_bursts([(1.0 * orc.YEAR, 4., 0.2 * orc.YEAR)], [(2.0 * orc.YEAR, 2., 0.3 * orc.YEAR)])
    assert _both(eng, f, wide, ep, E.RJP_GFF_SCALAR)[2][0] == "moments"
    eng.ff_scan(f, wide, ep, E.RJP_GFF_SCALAR, want_em=True, want_tavg=False)
    assert eng.last_scan_path()[0] == "moments"       # em0 is attached: two moment passes
    em0, f.em0 = f.em0, None
    eng.ff_scan(f, wide, ep, E.RJP_GFF_SCALAR, want_em=True, want_tavg=False)
    assert eng.last_scan_path()[0] == "tiles"         # EM maps without em0: the wide layout
    f.em0 = em0
    eng.ff_scan(f, wide, ep[:11], E.RJP_GFF_SCALAR, want_em=False, want_tavg=False)
    assert eng.last_scan_path()[0] == "tiles"
    f.ts_range = None
    f._ts_range_of = f.ts.data_ptr()          # "measured": no finite launch time known
    eng.use_moments = False
    eng.ff_scan(f, wide, ep, E.RJP_GFF_SCALAR, want_em=False, want_tavg=False)
    eng.use_moments = True
    assert eng.last_scan_path()[0] == "tiles"


def test_jetmodel_light_curves_through_the_moment_path(eng, tmp_path):
    """JetModel.flux_vs_time on the example jet (K4-built fields, occupied y-ranges): 38 epochs
    through the (forced) moment path reproduce the light curve of the reference's anchors
    (total flux at 5 GHz at t = 0 / 0.5 / 1 / 2 / 3 yr) and the tiles'."""
    from rajepy_amd import classes, logger
    from tests.test_host_logic import example_params
    jm = classes.JetModel(example_params(), log=logger.Log(str(tmp_path / "a.log"), verbose=False),
                          engine=eng)
    times = np.array(sorted(set(np.linspace(0., 3., 36)) | {0.5, 1.0, 2.0})) * orc.YEAR
    lc = jm.flux_vs_time(times, [5e9])[:, 0]
    assert eng.last_scan_path()[0] == "moments"
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


def test_cost_model_keeps_the_tiles_where_moments_do_not_pay(eng, tmp_path):
    """The library's own decision (no `force_moments`): the example jet -- 400 rows, 0.4 % of the
    grid occupied -- keeps the tiles for a 38-epoch light curve; a dense 2000-row grid takes the
    moment path for 32 uniformly spaced epochs and for 17 irregular ones, not for 12 uniform ones
    (the recurrence tiles are cheaper there)."""
    from rajepy_amd import classes, engine as E, logger
    from tests.test_host_logic import example_params
    eng.force_moments = False
    try:
        jm = classes.JetModel(example_params(), log=logger.Log(str(tmp_path / "a.log"), verbose=False),
                              engine=eng)
        jm.flux_vs_time(np.linspace(0., 3., 38) * orc.YEAR, [5e9])
        assert eng.last_scan_path()[0] == "tiles"
        f = eng.synth_fields((2, 2000, 32), 11, 0, 8, csize_au=0.5, tau_mode=E.RJP_GFF_SCALAR)
        bursts = E.make_bursts([(1.0 * orc.YEAR, 4., 0.2 * orc.YEAR)], [(2.0 * orc.YEAR, 2., 0.3 * orc.YEAR)])
        for years, want in ((np.linspace(0., 5., 32), "moments"), (np.linspace(0., 5., 12), "tiles"),
                            (np.sort(np.random.default_rng(1).uniform(0., 5., 17)), "moments")):
            eng.ff_scan(f, bursts, [y * orc.YEAR for y in years], E.RJP_GFF_SCALAR, want_em=False,
                        want_tavg=False)
            assert eng.last_scan_path()[0] == want, (len(years), want)
    finally:
        eng.force_moments = True


def test_replacing_the_launch_times_remeasures_their_range(eng):
    """The `ts` setter path (engine.replace_field): launch times stretched to twice their range --
    the moment path must bin them over the NEW range (a stale range would clamp half of the
    cells into the last bin); against the tiles on the same fields."""
    from rajepy_amd import engine as E
    shape = (4, 120, 32)
    ej = {"t_0": np.array([1.0, 3.0]), "hl": np.array([0.8, 1.0]), "chi": np.array([5., 3.]),
          "which": np.array(["RB", "RB"])}
    g, p, jet = _jet(shape, 321, ejection=ej)
    f = eng.upload_fields(g["nd"], g["xi"], g["temp"], g["ff"], g["areas"], g["ts"],
                          g["rr"] < 0, csize_au=jet.csize, dtype=8)
    eng.tau_layout(f, E.RJP_GFF_SCALAR)
    bursts = U.bursts_from_oracle(jet)
    ep = [y * orc.YEAR for y in np.linspace(0., 8., 24)]
    mom, til, (path, _) = _both(eng, f, bursts, ep, E.RJP_GFF_SCALAR)
    assert path == "moments"
    np.testing.assert_allclose(mom, til, rtol=RTOL)
    lo0, hi0 = f.ts_range
    eng.replace_field(f, "ts", 2.0 * g["ts"])
    mom2, til2, (path2, _) = _both(eng, f, bursts, ep, E.RJP_GFF_SCALAR)
    assert path2 == "moments" and f.ts_range == (2.0 * lo0, 2.0 * hi0)
    np.testing.assert_allclose(mom2, til2, rtol=RTOL)
    assert not np.allclose(mom2, mom, rtol=1e-3)


def test_cached_moment_maps_serve_further_sweeps_without_a_pass_over_the_grid(eng):
    """rjp_fields.d_mom_cache: the moment maps of a0 depend on the fields, the launch-time range,
    the shape and on which jets have bursts -- not on epochs or burst parameters.  The first long
    sweep of a densely filled model fills the caller-kept buffer, every later one --
    other epochs, other burst parameters -- is the contraction alone and equals the uncached
    result to rounding (same maps, same tables); a different SET of jets with bursts, replaced
    launch times or a sweep that needs another shape refill it."""
    from rajepy_amd import engine as E
    shape = (4, 180, 48)
    g, p, jet = _jet(shape, 777)
    f = eng.upload_fields(g["nd"], g["xi"], g["temp"], g["ff"], g["areas"], g["ts"],
                          g["rr"] < 0, csize_au=jet.csize, dtype=8)
    eng.tau_layout(f, E.RJP_GFF_SCALAR)
    bursts = U.bursts_from_oracle(jet)
    yr = orc.YEAR
    sweep = lambda b, ep: eng.ff_scan(f, b, ep, 0, want_em=False, want_tavg=False)[0].clone()
    ep1 = [y * yr for y in np.linspace(0., 5., 32)]
    ep2 = [y * yr for y in np.linspace(0.2, 4.4, 17)]
    eng.cache_moments = True
    try:
        eng.cache_moments = False
        a1 = sweep(bursts, ep1)                                   # reference: no cache anywhere
        eng.cache_moments = True
        a2 = sweep(bursts, ep1)                                   # a dense model: fills at once
        assert eng.last_scan_path()[0] == "moments" and (f.mom_cache["K"], f.mom_cache["N"]) == (53, 12)
        a3 = sweep(bursts, ep1)
        assert eng.last_scan_path()[0] == "cached"
        b3 = sweep(bursts, ep2)                                   # other epochs: still cached
        assert eng.last_scan_path()[0] == "cached"
        # other burst parameters (same jets): cached as well
        other = E.make_bursts([(0.6 * yr, 3.0, 0.2 * yr), (2.2 * yr, 6.0, 0.5 * yr)],
                              [(1.1 * yr, 2.0, 0.4 * yr)])
        c3 = sweep(other, ep2)
        assert eng.last_scan_path()[0] == "cached"
        eng.cache_moments = False
        b0, c0 = sweep(bursts, ep2), sweep(other, ep2)
        assert eng.last_scan_path()[0] == "moments"
        eng.cache_moments = True
        eng.synchronize()
        for got, ref in ((a2, a1), (a3, a1), (b3, b0), (c3, c0)):
            np.testing.assert_allclose(got.cpu().numpy(), ref.cpu().numpy(), rtol=1e-12)
        # bursts in ONE jet only: the other jet's NaN launch times count now -> refilled
        one = E.make_bursts([(0.6 * yr, 3.0, 0.3 * yr)], [])
        sweep(one, ep2)
        assert eng.last_scan_path()[0] == "moments"
        sweep(one, ep2)
        assert eng.last_scan_path()[0] == "cached"
        # a sparse model (occupied y-ranges, < half of the grid): the buffer comes only after a
        # sweep has taken the moment path
        g3 = {k: v.copy() for k, v in g.items()}
        g3["nd"][:, 40:, :] = np.nan
        g3["temp"][:, 40:, :] = np.nan
        f3 = eng.upload_fields(g3["nd"], g3["xi"], g3["temp"], g3["ff"], g3["areas"], g3["ts"],
                               g3["rr"] < 0, csize_au=jet.csize, dtype=8)
        eng.tau_layout(f3, E.RJP_GFF_SCALAR)
        eng.compute_y_bounds(f3)
        eng.ff_scan(f3, bursts, ep1, 0, want_em=False, want_tavg=False)
        assert eng.last_scan_path()[0] == "moments" and f3.mom_cache["K"] == 0     # reserved
        eng.ff_scan(f3, bursts, ep1, 0, want_em=False, want_tavg=False)
        assert f3.mom_cache["K"] == 53
        eng.ff_scan(f3, bursts, ep1, 0, want_em=False, want_tavg=False)
        assert eng.last_scan_path()[0] == "cached"
        # narrower bursts need another shape: the pass runs again into the buffer
        ej = U.example_bursts_params()
        ej["hl"] = np.asarray(ej["hl"], float) * 0.8
        g2, p2, jet2 = _jet(shape, 777, ejection=ej)
        nb = U.bursts_from_oracle(jet2)
        n1 = sweep(nb, ep1)
        assert eng.last_scan_path()[0] == "moments" and eng.last_moment_shape == (39, 16)
        assert (f.mom_cache["K"], f.mom_cache["N"]) == (39, 16)
        n2 = sweep(nb, ep1)
        assert eng.last_scan_path()[0] == "cached"
        np.testing.assert_allclose(n2.cpu().numpy(), n1.cpu().numpy(), rtol=1e-12)
        # replaced launch times drop the cache
        eng.replace_field(f, "ts", np.where(np.isnan(g["ts"]), np.nan, 0.9 * g["ts"]))
        assert f.mom_cache is None
        sweep(bursts, ep1)
        assert eng.last_scan_path()[0] == "moments"
    finally:
        eng.cache_moments = False
