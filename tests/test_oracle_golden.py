"""Pins the CPU oracle (oracle/rt_oracle.py) to golden vectors produced by importing the
unmodified reference (tests/golden/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest

from oracle import rt_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_model(tag):
    z = np.load(os.path.join(GOLDEN, tag + ".npz"))
    meta = json.loads(str(z["meta"]))
    return z, meta


def params_from_meta(meta):
    p = meta["params"]
    for k in ("t_0", "hl", "chi", "which"):
        p["ejection"][k] = np.array(p["ejection"][k])
    return p


def injected(tag):
    z, meta = load_model(tag)
    shape = tuple(meta["shape"])
    idx = z["f_idx"]
    d = lambda k, fill=np.nan: orc.dense_from_sparse(shape, idx, z["f_" + k], fill)
    jet = orc.OracleJet.from_fields(params_from_meta(meta), d("nd"), d("xi"), d("temp"),
                                    d("ff"), d("areas"), d("ts0", 0.), d("rr", 1.),
                                    d("vy"))
    return jet, z, meta


def test_gff_table():
    g = np.load(os.path.join(GOLDEN, "gff.npz"))
    for i, nu in enumerate(g["nus"]):
        for j, t in enumerate(g["temps"]):
            assert orc.gff(nu, t) == pytest.approx(g["gff"][i, j], rel=1e-13)


def test_scalars():
    s = json.load(open(os.path.join(GOLDEN, "scalars.json")))
    assert orc.mod_r_0(25., 7. / 9., 1.) == pytest.approx(s["mod_r_0"], rel=1e-15)
    pairs = ((0., 0.), (-0.5, -0.5), (-1., 0.), (-1.5, -0.5), (0.5, -1.), (-0.25, 0.75))
    for (qn, qv), ref in zip(pairs, s["n_0_from_mlr"]):
        assert orc.n_0_from_mlr(1e-7, 150., 1., 1.3, qn, qv, .25, 2.5) == \
            pytest.approx(ref, rel=1e-14)
    for line, ref in s["rrl_nu_0"].items():
        assert orc.rrl_nu_0(*orc.rrl_parser(line)) == pytest.approx(ref, rel=1e-15)
    assert orc.doppler_shift(2.2364174326e10, 12.5) == pytest.approx(s["doppler_shift"], rel=1e-15)
    assert orc.blackbody_nu(2.2e10, 9e3) == pytest.approx(s["blackbody_nu"], rel=1e-14)
    assert orc.deltanu_g(2.2364e10, 1e4, "H") == pytest.approx(s["deltanu_g"], rel=1e-14)
    assert orc.deltanu_l(1e6, 66, 1) == pytest.approx(s["deltanu_l"], rel=1e-14)
    phi = orc.phi_voigt_nu(2.2364e10, 3e5, 1.6e6)
    for x, ref in zip(s["phi_voigt_x_mhz"], s["phi_voigt"]):
        assert phi(2.2364e10 + x * 1e6) == pytest.approx(ref, rel=1e-12)


@pytest.mark.parametrize("tag", ["cfg1_example", "tilted"])
def test_rt_from_injected_fields(tag):
    """RT stage alone: reference fields in, reference maps out."""
    jet, z, meta = injected(tag)
    freqs = z["freqs"]
    for e, yr in enumerate(z["years"]):
        jet.time = yr * orc.YEAR
        np.testing.assert_allclose(jet.emission_measure(), z["em"][e], rtol=1e-13)
        np.testing.assert_allclose(jet.optical_depth_ff(freqs), z["tau_ff"][e], rtol=1e-13)
        got = jet.flux_ff(freqs)
        assert np.array_equal(np.isnan(got), np.isnan(z["flux_ff"][e]))
        np.testing.assert_allclose(got, z["flux_ff"][e], rtol=1e-12)
        np.testing.assert_allclose(jet.intensity_ff(freqs), z["intensity_ff"][e], rtol=1e-12)
    jet.time = z["years"][0] * orc.YEAR
    rrl = meta["rrl"]
    rf = z["rrl_freqs"]
    np.testing.assert_allclose(jet.optical_depth_rrl(rrl, rf), z["tau_rrl"], rtol=1e-11)
    np.testing.assert_allclose(jet.flux_rrl(rrl, rf, contsub=True), z["flux_rrl_contsub"],
                               rtol=1e-10)
    np.testing.assert_allclose(jet.flux_rrl(rrl, rf, contsub=False), z["flux_rrl_total"],
                               rtol=1e-11)
    np.testing.assert_allclose(jet.intensity_rrl(rrl, float(rf[0])), z["intensity_rrl0"],
                               rtol=1e-10)


@pytest.mark.parametrize("tag", ["cfg1_example", "tilted"])
def test_field_builder(tag):
    """Geometry -> fields stage (classes.py:465-1099) against the reference's grids."""
    z, meta = load_model(tag)
    jet = orc.OracleJet(params_from_meta(meta))
    assert [jet.nx, jet.ny, jet.nz] == meta["shape"]
    assert jet.params["properties"]["n_0"] == pytest.approx(meta["params"]["properties"]["n_0"], rel=1e-14)
    idx = z["f_idx"]
    ff = jet.fill_factor
    assert np.array_equal(np.flatnonzero(np.isfinite(ff).ravel()), idx)
    take = lambda a: a.ravel()[idx]
    assert np.array_equal(take(ff), z["f_ff"])
    assert np.array_equal(take(jet.areas), z["f_areas"])
    np.testing.assert_allclose(take(jet.rr), z["f_rr"], rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(take(jet.ww), z["f_ww"], rtol=1e-13, atol=1e-13)
    for got, key in ((jet.nd0, "nd"), (jet.ion_fraction, "xi"), (jet.temperature, "temp"),
                     (jet.vy, "vy"), (jet.ts0, "ts0")):
        assert not np.isfinite(got.ravel()[np.setdiff1d(np.arange(got.size), idx)]).any() \
            or key == "ts0"
        np.testing.assert_allclose(take(got), z["f_" + key], rtol=1e-11, atol=1e-9,
                                   err_msg=key)
    # burst factor over the WHOLE grid (chi is evaluated outside the jet too, classes.py:866)
    for e, yr in enumerate(z["years"]):
        jet.time = yr * orc.YEAR
        with np.errstate(all="ignore"):
            assert np.nanmax(jet.chi_xyz) == pytest.approx(z["chi_max"][e], rel=1e-9)


def test_end_to_end_cfg1_anchor():
    """Builder + RT: the survey's known-answer anchors for config 1."""
    z, meta = load_model("cfg1_example")
    jet = orc.OracleJet(params_from_meta(meta))
    f = jet.flux_ff(5e9)
    assert np.nansum(f) == pytest.approx(0.0011582235145955692, rel=1e-11)
    assert jet.optical_depth_ff(5e9).max() == pytest.approx(z["tau_ff"][0, 0].max(), rel=1e-12)
    assert int(np.isnan(f).sum()) == 2044


def test_pipeline_flux_results():
    rec = json.load(open(os.path.join(GOLDEN, "pipeline_cfg1.json")))
    arr = np.load(os.path.join(GOLDEN, "pipeline_cfg1.npz"))
    for i, run in enumerate(rec["runs"]):
        flux_cube = arr["run%d_flux" % i]          # FITS order (F, n_z, n_x)
        got = np.atleast_1d(orc.pipeline_flux_result(flux_cube, run["obs_type"]))
        np.testing.assert_allclose(got, run["flux"], rtol=1e-13)
        if run["obs_type"] == "continuum":
            np.testing.assert_allclose(orc.chan_freqs(run["freq"], 4e8, 2e8), run["chan_freqs"])


def test_oracle_on_the_example_as_shipped():
    """Third pin: the reference's example file as shipped (l_z = 2 -> 108 x 110 x 588, 7 M
    cells): oracle builder + RT vs the reference's maps (continuum and one RRL channel)."""
    z = np.load(os.path.join(GOLDEN, "example_as_shipped.npz"))
    meta = json.loads(str(z["meta"]))
    p = params_from_meta(meta)
    p["grid"]["l_z"] = None                    # the oracle restates the explicit-size branch;
    assert [p["grid"][k] for k in ("n_x", "n_y", "n_z")] == meta["shape"]   # same grid
    jet = orc.OracleJet(p)
    jet.time = float(z["year"]) * orc.YEAR
    assert int(np.isfinite(jet.fill_factor).sum()) == int(z["n_jet_cells"])
    np.testing.assert_allclose(jet.emission_measure(), z["em"], rtol=1e-11)
    np.testing.assert_allclose(jet.optical_depth_ff(float(z["freqs"][0])), z["tau_ff"][0], rtol=1e-11)
    # thin columns: 1 - exp(-tau) amplifies last-bit differences of exp by 1/tau
    np.testing.assert_allclose(jet.flux_ff(float(z["freqs"][1])), z["flux_ff"][1], rtol=1e-9)
    np.testing.assert_allclose(jet.optical_depth_rrl("H66a", float(z["rrl_freqs"][1])),
                               z["tau_rrl"][1], rtol=1e-10)
