"""Randomised parity of K1/K2 and K3 against the oracle: random small grids (odd and even
extents, so every lane-width variant is hit), NaN / zero sprinkles with NumPy's nansum /
nanmean semantics, 0-11 bursts per jet, uniform and irregular epoch lists, both Gaunt
branches, both storage widths; K3 with random lines (H and He), channel counts from 1 to 300
and uniform, irregular and shuffled channel lists."""
import copy

import numpy as np
import pytest

from oracle import rt_oracle as orc
from tests import gpu_util as U

pytestmark = pytest.mark.gpu
YEAR = orc.YEAR


@pytest.fixture(scope="module")
def eng():
    from rajepy_amd.engine import RTEngine
    e = RTEngine(0)
    yield e
    e.close()


def _random_model(rng, with_vy):
    shape = (int(rng.integers(1, 6)), int(rng.integers(3, 70)), int(rng.integers(1, 41)))
    if not with_vy and rng.random() < 0.3:
        # wide in z: whole waves inside one jet (the red/blue flag of the synthetic fields is
        # i_z < n_z / 2), which is what the two-operation burst recurrence needs
        shape = (int(rng.integers(1, 3)), int(rng.integers(3, 40)),
                 int(rng.choice([128, 130, 192, 256])))
    plaw = bool(rng.integers(2))
    g = U.synth_host(shape, int(rng.integers(1 << 30)), 1 if plaw else 0)
    # sprinkle "outside the jet" cells and odd values into individual fields
    n = g["nd"].size
    for key, frac in (("nd", 0.15), ("xi", 0.05), ("temp", 0.05), ("ff", 0.05)):
        hit = rng.random(n) < frac
        g[key].ravel()[hit] = np.nan
    g["temp"].ravel()[rng.random(n) < 0.02] = 0.0          # T = 0: not counted in the mean
    g["xi"].ravel()[rng.random(n) < 0.02] = 0.0
    if shape[0] > 1:
        g["nd"][0, :, 0] = np.nan                            # one empty sightline
    nb = int(rng.integers(0, 12))
    ej = {"t_0": np.sort(rng.uniform(0.1, 3.0, nb)), "hl": rng.uniform(0.08, 0.6, nb),
          "chi": rng.uniform(0.3, 8.0, nb),
          "which": np.array([("R", "B", "RB")[int(k)] for k in rng.integers(0, 3, nb)])}
    p = copy.deepcopy(U.load_golden("cfg1_example")[2])
    p["ejection"] = ej
    p["power_laws"]["q_T"] = -0.5 if plaw else 0.
    p["grid"].update(n_x=shape[0], n_y=shape[1], n_z=shape[2])
    jet = orc.OracleJet.from_fields(p, g["nd"], g["xi"], g["temp"], g["ff"], g["areas"],
                                    g["ts"], g["rr"], g["vy"] if with_vy else None)
    return shape, g, p, jet, plaw


@pytest.mark.parametrize("seed", range(16))
def test_k1_k2_random_models(eng, seed):
    from rajepy_amd import engine as E
    rng = np.random.default_rng(1000 + seed)
    shape, g, p, jet, plaw = _random_model(rng, with_vy=False)
    dtype = 8 if seed % 4 else 4
    tol = 1e-10 if dtype == 8 else 1e-5
    fields = eng.upload_fields(g["nd"], g["xi"], g["temp"], g["ff"], g["areas"], g["ts"],
                               g["rr"] < 0, csize_au=jet.csize, dtype=dtype)
    bursts = U.bursts_from_oracle(jet)
    ne = int(rng.choice([1, 2, 3, 5, 8, 9, 16, 17, 32, 35]))
    years = (np.linspace(0., rng.uniform(0.5, 3.5), ne) if rng.integers(2) or ne == 1
             else np.sort(rng.uniform(0., 4., ne)))
    ep = [float(y) * YEAR for y in years]
    want_em = bool(rng.integers(2))
    mode = E.RJP_GFF_POWERLAW if plaw else E.RJP_GFF_SCALAR
    freqs = np.array([1.4e9, 2.3e10])
    gv = None if plaw else [orc.gff(nu, p["properties"]["T_0"]) for nu in freqs]
    ctau, cflux = E.ff_channel_coeffs(freqs, jet.csize, p["target"]["dist"], mode, gv)
    sumA, em, tavg = eng.ff_scan(fields, bursts, ep, mode, want_em=want_em)
    tau, flux, ftot = eng.ff_maps(sumA, tavg, ctau, cflux)
    eng.synchronize()
    shp = (len(ep), 2, shape[0], shape[2])
    tau, flux = tau.cpu().numpy().reshape(shp), flux.cpu().numpy().reshape(shp)
    with np.errstate(all="ignore"):
        for e in sorted({0, len(ep) // 2, len(ep) - 1}):
            jet.time = ep[e]
            rt, rf = jet.optical_depth_ff(freqs), jet.flux_ff(freqs)
            assert np.array_equal(tau[e] == 0, rt == 0) and np.array_equal(np.isnan(flux[e]),
                                                                           np.isnan(rf))
            np.testing.assert_allclose(tau[e], rt, rtol=tol)
            np.testing.assert_allclose(flux[e], rf, rtol=10 * tol)
            np.testing.assert_allclose(ftot.cpu().numpy()[e], np.nansum(rf, axis=(1, 2)),
                                       rtol=10 * tol, atol=0)
            if want_em:
                np.testing.assert_allclose(em.cpu().numpy()[e].reshape(shape[0], shape[2]),
                                           jet.emission_measure(), rtol=tol)


@pytest.mark.parametrize("seed", range(12))
def test_k3_random_models(eng, seed):
    from rajepy_amd import _lib
    from rajepy_amd.maths import rrls
    rng = np.random.default_rng(2000 + seed)
    shape, g, p, jet, plaw = _random_model(rng, with_vy=True)
    g["vy"].ravel()[rng.random(g["vy"].size) < 0.03] = np.nan
    jet.time = float(rng.uniform(0., 3.)) * YEAR
    fields = eng.upload_fields(g["nd"], g["xi"], g["temp"], g["ff"], g["areas"], g["ts"],
                               g["rr"] < 0, vy=g["vy"], csize_au=jet.csize, dtype=8)
    rrl = str(rng.choice(["H66a", "H58a", "H110g", "He66a", "H41b", "H30d"]))
    el, n, dn = rrls.rrl_parser(rrl)
    nu0 = rrls.rrl_nu_0(el, n, dn)
    nchan = int(rng.choice([1, 2, 15, 16, 17, 40, 64, 65, 130, 256, 300]))
    cw = nu0 * float(rng.choice([2e-6, 5e-6, 2e-5]))           # 0.1 .. 1 thermal widths
    rf = orc.chan_freqs(nu0 + cw * float(rng.uniform(-20, 20)), nchan * cw, cw)
    kind = int(rng.integers(3))
    if kind == 1:
        rf = rf * (1. + 1e-7 * rng.standard_normal(len(rf)))  # irregular spacing
    elif kind == 2:
        rf = rng.permutation(rf)
    line = _lib.Line(**rrls.line_constants(rrl))
    tau = eng.rrl_scan(fields, U.bursts_from_oracle(jet), jet.time, line, rf)
    eng.synchronize()
    with np.errstate(all="ignore"):
        ref = jet.optical_depth_rrl(rrl, np.asarray(rf))
    got = tau.cpu().numpy().reshape(ref.shape)
    assert np.array_equal(got == 0, ref == 0)
    np.testing.assert_allclose(got, ref, rtol=U.k3_rtol(len(rf)))
