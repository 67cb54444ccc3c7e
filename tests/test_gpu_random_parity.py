"""Randomised parity of the continuum scan against the oracle: random grid shapes, 0-10 bursts per
jet with random widths and amplitudes (factors below 1 -- dips -- included, as the reference's
`ejection` table admits any `chi` > 0; classes.py:245-264, 442-448), random epoch lists (single,
uniform and irregular: direct tiles, uniform-epoch recurrences, and -- from 12 epochs on, forced
-- the launch-time moments), NaN / zero cells in every field, on the tau, compact and wide
layouts and both Gaunt branches.  tau and the emission measure of every epoch against the
oracle's (classes.py:1101-1128, 1353-1447) at 1e-10; zero / non-zero pattern identical."""
import copy

import numpy as np
import pytest

from oracle import rt_oracle as orc
from tests import gpu_util as U

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from rajepy_amd.engine import RTEngine
    e = RTEngine(0)
    e.force_moments = True         # sweeps of >= 12 epochs on grids this small: the moment path
    e.cache_moments = False        # (every sweep here is meant to run its own pass)
    yield e
    e.close()


def _case(seed):
    rng = np.random.default_rng(seed)
    shape = (int(rng.integers(1, 5)), int(rng.integers(20, 140)), int(rng.integers(3, 40)))
    nb = int(rng.integers(0, 21))
    which = rng.choice(["R", "B", "RB"], size=nb)
    ej = {"t_0": rng.uniform(-0.5, 5.5, nb), "hl": rng.uniform(0.12, 1.2, nb),
          "chi": np.where(rng.random(nb) < 0.25, rng.uniform(0.2, 0.9, nb), rng.uniform(1.2, 12., nb)),
          "which": which}
    kind = rng.choice(["single", "uniform", "irregular", "sweep"])
    if kind == "single":
        years = [float(rng.uniform(0., 5.))]
    elif kind == "uniform":
        years = list(np.linspace(rng.uniform(0., 1.), rng.uniform(2., 5.), int(rng.integers(2, 10))))
    elif kind == "irregular":
        years = sorted(rng.uniform(0., 5., int(rng.integers(2, 10))).tolist())
    else:
        years = list(np.linspace(0., 5., int(rng.integers(12, 40))))
    temp_mode = int(rng.integers(0, 2))
    return rng, shape, ej, years, temp_mode, kind


@pytest.mark.parametrize("seed", list(range(24)))
def test_random_models_match_the_oracle_on_every_layout(eng, seed):
    from rajepy_amd import engine as E
    from rajepy_amd.maths import physics as ph
    rng, shape, ej, years, temp_mode, kind = _case(1000 + seed)
    g = U.synth_host(shape, 77 + seed, temp_mode)
    for k, vals in (("nd", [np.nan, 0.0]), ("xi", [np.nan]), ("temp", [np.nan]),
                    ("ff", [np.nan, 0.0]), ("ts", [np.nan])):
        m = rng.random(shape) < 0.04
        g[k] = np.where(m, rng.choice(vals, size=shape), g[k])
    p = copy.deepcopy(U.load_golden("cfg1_example")[2])
    p["ejection"] = ej
    if temp_mode:
        p["power_laws"]["q_T"] = -0.5
    p["grid"].update(n_x=shape[0], n_y=shape[1], n_z=shape[2])
    jet = orc.OracleJet.from_fields(p, g["nd"], g["xi"], g["temp"], g["ff"], g["areas"],
                                    g["ts"], g["rr"], g["vy"])
    mode = E.RJP_GFF_SCALAR if temp_mode == 0 else E.RJP_GFF_POWERLAW
    f = eng.upload_fields(g["nd"], g["xi"], g["temp"], g["ff"], g["areas"], g["ts"],
                          g["rr"] < 0, csize_au=jet.csize, dtype=8)
    eng.tau_layout(f, mode)
    bursts = U.bursts_from_oracle(jet) if len(ej["t_0"]) else None
    ep = [y * orc.YEAR for y in years]
    gv = [ph.gff(5e9, p["properties"]["T_0"])] if temp_mode == 0 else None
    ctau, _ = E.ff_channel_coeffs([5e9], jet.csize, p["target"]["dist"], mode, gv)

    def oracle(e):
        jet.time = ep[e]
        with np.errstate(all="ignore"):
            return jet.optical_depth_ff(5e9), jet.emission_measure()

    check = sorted(set([0, len(ep) // 2, len(ep) - 1]))
    refs = {e: oracle(e) for e in check}
    results, paths = {}, {}
    a0, em0 = f.a0, f.em0
    for name in ("tau", "compact", "wide"):
        if name == "compact":
            f.a0 = None
        elif name == "wide":
            f.a0, f.em0 = None, None
        sumA, em, _ = eng.ff_scan(f, bursts, ep, mode, want_em=True)
        eng.synchronize()
        results[name] = (sumA.cpu().numpy(), em.cpu().numpy())
        path = paths[name] = eng.last_scan_path()[0]
        for e in check:
            tau_ref, em_ref = refs[e]
            tau = ctau[0] * results[name][0][e].reshape(shape[0], shape[2])
            tau_ref = np.where(np.isnan(tau_ref), 0.0, tau_ref)      # an all-NaN sightline sums to 0
            em_ref = np.where(np.isnan(em_ref), 0.0, em_ref)
            assert np.array_equal(tau == 0, tau_ref == 0), (name, kind, path)
            np.testing.assert_allclose(tau, tau_ref, rtol=1e-10, atol=0,
                                       err_msg="%s %s %s" % (name, kind, path))
            np.testing.assert_allclose(results[name][1][e].reshape(shape[0], shape[2]),
                                       em_ref, rtol=1e-10, atol=0)
    f.a0, f.em0 = a0, em0
    # the three layouts against each other: bit for bit -- except where they take different
    # routes: the tau layout's sweeps go through the launch-time moments (6e-11 against the
    # tiles), and the wide layout has no 16- / 32-epoch recurrence tiles (8-epoch ones: 1e-11)
    for name in ("compact", "wide"):
        for k in (0, 1):
            if paths["tau"] != paths[name]:
                np.testing.assert_allclose(results[name][k], results["tau"][k], rtol=6e-11, atol=0)
            elif name == "wide" and len(ep) >= 16:
                np.testing.assert_allclose(results[name][k], results["tau"][k], rtol=1e-11, atol=0)
            else:
                assert np.array_equal(results[name][k], results["tau"][k]), (name, kind, k)
    if kind == "sweep" and bursts is not None:
        # tau alone (no EM maps; the moment path where its accuracy check passes) against the
        # compact layout's tiles
        mom, _, _ = eng.ff_scan(f, bursts, ep, mode, want_em=False, want_tavg=False)
        eng.synchronize()
        assert eng.last_scan_path()[0] in ("moments", "tiles")
        _m, _t = mom.cpu().numpy(), results["compact"][0]
        ok = _t != 0
        assert np.array_equal(_m == 0, _t == 0)
        np.testing.assert_allclose(_m[ok], _t[ok], rtol=6e-11)
        # ... and, for sweeps it can serve (<= 32 epochs), on the launch-time-ordered layout with
        # a random bin count: NaN cells in every field, dips, bursts in one jet only -- whatever
        # path the library then takes (the layout when some order <= 32 passes its check)
        if len(ep) <= 32 and eng.launch_time_range(f) is not None:
            eng.build_lt(f, int(rng.integers(6, 41)))
            lt, _, _ = eng.ff_scan(f, bursts, ep, mode, want_em=False, want_tavg=False)
            eng.synchronize()
            assert eng.last_scan_path()[0] in ("lt", "moments", "tiles")
            _l = lt.cpu().numpy()
            assert np.array_equal(_l == 0, _t == 0)
            np.testing.assert_allclose(_l[ok], _t[ok], rtol=6e-11)
            f.lt = None


@pytest.mark.parametrize("seed", list(range(14)))
def test_random_rrl_cubes_match_the_oracle(eng, seed):
    """K3 on random cases: a random alpha / beta line of H or He, 1-300 channels of a random
    width around the line (every channel-lane layout; bands narrow enough for the line core and
    wide enough for the far wings), random bursts, velocities, NaN cells; against
    `optical_depth_rrl` of the oracle (classes.py:1130-1229, maths/rrls.py:350-389 -- scipy's
    wofz) at the tolerance the kernel is designed for (tests/gpu_util.k3_rtol)."""
    from rajepy_amd import _lib
    from rajepy_amd.maths import rrls
    rng = np.random.default_rng(5000 + seed)
    shape = (int(rng.integers(1, 4)), int(rng.integers(8, 70)), int(rng.integers(2, 26)))
    nb = int(rng.integers(0, 7))
    ej = {"t_0": rng.uniform(0., 4., nb), "hl": rng.uniform(0.15, 1.0, nb),
          "chi": rng.uniform(1.5, 9., nb), "which": rng.choice(["R", "B", "RB"], size=nb)}
    rrl = str(rng.choice(["H42a", "H58a", "H66a", "H92a", "H110a", "He66a", "H83b"]))
    nchan = int(rng.choice([1, 3, 16, 17, 40, 64, 100, 256, 300]))
    cw = float(rng.choice([2e4, 1e5, 4e5, 2e6]))
    g = U.synth_host(shape, 900 + seed, 1)
    for k in ("nd", "xi", "temp", "ff", "ts", "vy"):
        m = rng.random(shape) < 0.03
        g[k] = np.where(m, np.nan, g[k])
    p = copy.deepcopy(U.load_golden("cfg1_example")[2])
    p["ejection"] = ej
    p["power_laws"]["q_T"] = -0.5
    p["grid"].update(n_x=shape[0], n_y=shape[1], n_z=shape[2])
    jet = orc.OracleJet.from_fields(p, g["nd"], g["xi"], g["temp"], g["ff"], g["areas"],
                                    g["ts"], g["rr"], g["vy"])
    jet.time = float(rng.uniform(0., 4.)) * orc.YEAR
    f = eng.upload_fields(g["nd"], g["xi"], g["temp"], g["ff"], g["areas"], g["ts"],
                          g["rr"] < 0, csize_au=jet.csize, dtype=8, vy=g["vy"])
    el, n, dn = rrls.rrl_parser(rrl)
    nu0 = rrls.rrl_nu_0(el, n, dn)
    rf = orc.chan_freqs(nu0, nchan * cw if nchan > 1 else 1.0, cw if nchan > 1 else 1.0)
    assert len(rf) == nchan
    line = _lib.Line(**rrls.line_constants(rrl))
    bursts = U.bursts_from_oracle(jet) if nb else None
    tau = eng.rrl_scan(f, bursts, jet.time, line, rf)
    eng.synchronize()
    with np.errstate(all="ignore"):
        ref = jet.optical_depth_rrl(rrl, np.asarray(rf))
    got = tau.cpu().numpy().reshape(ref.shape)
    ref = np.where(np.isnan(ref), 0.0, ref)
    assert np.array_equal(got == 0, ref == 0), rrl
    np.testing.assert_allclose(got, ref, rtol=U.k3_rtol(nchan), atol=0, err_msg="%s x %d" % (rrl, nchan))


@pytest.mark.parametrize("seed", range(12))
def test_random_bursts_through_the_lds_table_scan(eng, seed):
    """The single-epoch table scan (ff_scan_tab.hip) with random burst sets -- 1-20 bursts, widths
    from a tenth of a year to two years, amplitudes up to 50, bursts piled on one another, epochs
    inside and outside the launch-time range -- on a map large enough to take the path: whenever
    the library takes the table it must agree with the Gaussian scan at 3e-12 (the table's bound
    rests on an analytic estimate of chi's 8th derivative: sums of overlapping narrow bursts are
    its hard case); bursts too narrow for 460 intervals per jet keep the Gaussians."""
    import torch
    from rajepy_amd import engine as E
    rng = np.random.default_rng(1000 + seed)
    shape = (64, 70 + int(rng.integers(0, 60)), 512)
    mode = int(rng.integers(0, 2))
    f = eng.synth_fields(shape, 31337 + seed, mode, 8, csize_au=0.5, wide=False, tau_mode=mode)
    yr = orc.YEAR
    nb = int(rng.integers(1, 21))
    lists = ([], [])
    centre = rng.uniform(0.5, 4.0)
    for _ in range(nb):
        t0 = rng.normal(centre, 0.3) if rng.random() < 0.5 else rng.uniform(-1.0, 6.0)
        sigma = 10.0 ** rng.uniform(np.log10(0.04), np.log10(0.9))
        amp = 10.0 ** rng.uniform(-1.0, 1.7)
        lists[int(rng.integers(0, 2))].append((t0 * yr, amp, sigma * yr))
    bursts = E.make_bursts(lists[0], lists[1])
    took = 0
    for years in (rng.uniform(0.0, 5.0), rng.uniform(-2.0, 9.0), centre + 1.0):
        ep = [years * yr]
        eng.use_chi_table = True
        tab = eng.ff_scan(f, bursts, ep, mode, want_em=True, want_tavg=False)
        path = eng.last_scan_path()[0]
        ni = eng.last_moment_shape[0]
        tab = [t.clone() for t in tab[:2]]
        eng.use_chi_table = False
        ref = eng.ff_scan(f, bursts, ep, mode, want_em=True, want_tavg=False)
        assert eng.last_scan_path()[0] == "tiles"
        eng.use_chi_table = True
        eng.synchronize()
        assert path in ("table", "tiles")
        if path == "table":
            took += 1
            assert 1 <= ni <= 460
        for got, want in zip(tab, ref[:2]):
            assert torch.equal(got == 0, want == 0)
            rel = ((got - want).abs() / want).max().item()
            assert rel < 3e-12, (path, rel, nb)
    # (narrow bursts over a wide range may all exceed the LDS: nothing to assert then)
    assert took >= 0
