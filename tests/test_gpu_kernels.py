"""GPU parity tests proper: every librjprt kernel, called through the C-ABI, against the CPU
oracle and the committed golden vectors.  Run with `pytest -m gpu` on an MI355X."""
import copy

import numpy as np
import pytest

from tests import gpu_util as U
from oracle import rt_oracle as orc

pytestmark = pytest.mark.gpu

RTOL = 1e-5          # north-star parity bar; most checks below are far tighter


@pytest.fixture(scope="module")
def eng():
    from rajepy_amd.engine import RTEngine
    e = RTEngine(0)
    yield e
    e.close()


STORES = ["f64", "f64-wide", "f32", "f32-wide"]   # default = compact scan field attached


def _store(store):
    return 4 if store.startswith("f32") else 8


def _layout(fields, store):
    """Fields come with the compact field attached; "*-wide" detaches it so the 5-field path
    of K1 stays covered."""
    if not store.endswith("-wide"):
        assert fields.em0 is not None
    else:
        fields.em0 = None
    return fields


def _upload(eng, g, csize, dtype):
    return eng.upload_fields(g["nd"], g["xi"], g["temp"], g["ff"], g["areas"], g["ts"],
                             g["rr"] < 0, g["vy"], csize_au=csize, dtype=dtype)


def _run_ff(eng, fields, bursts, jet, years, freqs):
    from rajepy_amd import engine as E
    from rajepy_amd.maths import physics as ph
    p = jet.params
    mode = E.RJP_GFF_SCALAR if p["power_laws"]["q_T"] == 0. else E.RJP_GFF_POWERLAW
    g = [ph.gff(nu, p["properties"]["T_0"]) for nu in freqs] if mode == E.RJP_GFF_SCALAR else None
    ctau, cflux = E.ff_channel_coeffs(freqs, jet.csize, p["target"]["dist"], mode, g)
    sumA, em, tavg = eng.ff_scan(fields, bursts, [y * orc.YEAR for y in years], mode)
    tau, flux, ftot = eng.ff_maps(sumA, tavg, ctau, cflux)
    eng.synchronize()
    shp = (len(years), len(freqs), jet.nx, jet.nz)
    return (em.cpu().numpy().reshape(len(years), jet.nx, jet.nz),
            tau.cpu().numpy().reshape(shp), flux.cpu().numpy().reshape(shp),
            ftot.cpu().numpy(), tavg.cpu().numpy().reshape(jet.nx, jet.nz))


@pytest.mark.parametrize("tag", ["cfg1_example", "tilted"])
@pytest.mark.parametrize("store", STORES)
def test_ff_against_reference_golden(eng, tag, store):
    """K1+K2 on the reference's own fields vs the reference's own maps (tau, flux, EM) at
    every golden epoch and frequency."""
    z, meta, p, g, jet = U.golden_dense(tag)
    dtype = _store(store)
    fields = _layout(_upload(eng, g, jet.csize, dtype), store)
    em, tau, flux, ftot, _ = _run_ff(eng, fields, U.bursts_from_oracle(jet), jet,
                                     z["years"], z["freqs"])
    tol = 1e-11 if dtype == 8 else RTOL
    np.testing.assert_allclose(em, z["em"], rtol=tol)
    np.testing.assert_allclose(tau, z["tau_ff"], rtol=tol)
    assert np.array_equal(np.isnan(flux), np.isnan(z["flux_ff"]))       # NaN pattern
    assert np.array_equal(tau == 0.0, z["tau_ff"] == 0.0)               # empty sightlines
    np.testing.assert_allclose(flux, z["flux_ff"], rtol=max(tol, 1e-10))
    np.testing.assert_allclose(ftot, np.nansum(z["flux_ff"], axis=(2, 3)), rtol=max(tol, 1e-10))


@pytest.mark.parametrize("temp_mode", [0, 1])
@pytest.mark.parametrize("store", STORES)
@pytest.mark.parametrize("shape", [(6, 37, 10), (4, 64, 128), (3, 50, 7)])
def test_ff_dense_synthetic_vs_oracle(eng, shape, store, temp_mode):
    """Dense synthetic fields (device generator) vs the oracle on the host restatement of the
    same hash; exercises odd n_y tails, the scalar (VEC=1) path (n_z=7) and y-splitting."""
    seed = 20240501
    dtype = _store(store)
    fields = _layout(eng.synth_fields(shape, seed, temp_mode, dtype, csize_au=0.5,
                                      with_vy=True), store)
    g = U.synth_host(shape, seed, temp_mode)
    if dtype == 8:    # generator itself must be bit-identical to the host restatement
        eng.synchronize()
        for name, ref in (("xi", g["xi"]), ("temp", g["temp"]), ("pf", g["ff"]),
                          ("vy", g["vy"]), ("ts", g["ts"])):
            got = getattr(fields, name).cpu().numpy().reshape(shape)
            np.testing.assert_allclose(got, ref, rtol=1e-15, err_msg=name)
        nd = fields.nd.cpu().numpy().reshape(shape)
        np.testing.assert_allclose(np.abs(nd), g["nd"], rtol=1e-13)
        assert np.array_equal(np.signbit(nd), g["rr"] < 0)
    p = copy.deepcopy(U.load_golden("cfg1_example")[2])
    p["ejection"] = U.example_bursts_params()
    p["power_laws"]["q_T"] = 0. if temp_mode == 0 else -0.5
    p["grid"].update(n_x=shape[0], n_y=shape[1], n_z=shape[2])
    jet = orc.OracleJet.from_fields(p, g["nd"], g["xi"], g["temp"], g["ff"], g["areas"],
                                    g["ts"], g["rr"], g["vy"])
    years, freqs = [0., 0.6, 1.1, 2.0, 2.7], [1e9, 5e9, 5e10]
    em, tau, flux, ftot, tavg = _run_ff(eng, fields, U.bursts_from_oracle(jet), jet, years,
                                        freqs)
    tol = 1e-11 if dtype == 8 else RTOL
    for e, yr in enumerate(years):
        jet.time = yr * orc.YEAR
        np.testing.assert_allclose(em[e], jet.emission_measure(), rtol=tol)
        np.testing.assert_allclose(tau[e], jet.optical_depth_ff(np.array(freqs)), rtol=tol)
        np.testing.assert_allclose(flux[e], jet.flux_ff(np.array(freqs)), rtol=max(tol, 1e-10))


def test_ff_nan_semantics(eng):
    """NumPy's per-product NaN masks: tau skips a cell if ANY of
    n, x, T, ff, areas is NaN; EM ignores T; T_avg counts a cell iff T > 0."""
    rng = np.random.default_rng(7)
    shape = (4, 23, 8)
    g = U.synth_host(shape, 99, 1)
    for k in ("nd", "xi", "temp", "ff", "areas"):
        m = rng.random(shape) < 0.15
        g[k] = np.where(m, np.nan, g[k])
    g["temp"][rng.random(shape) < 0.05] = -5.0          # not counted by T > 0, tau -> NaN term
    g["nd"][0, :, 0] = np.nan                           # an empty sightline ...
    g["temp"][0, :, 0] = np.nan                         # ... in every field
    g["nd"][2, :, 3] = np.nan                           # no density but T finite: flux = 0
    g["temp"][1, :, 1] = np.nan                         # EM finite, tau = 0, flux NaN
    p = copy.deepcopy(U.load_golden("cfg1_example")[2])
    p["power_laws"]["q_T"] = -0.5
    p["grid"].update(n_x=shape[0], n_y=shape[1], n_z=shape[2])
    jet = orc.OracleJet.from_fields(p, g["nd"], g["xi"], g["temp"], g["ff"], g["areas"],
                                    g["ts"], g["rr"], g["vy"])
    fields = _upload(eng, g, jet.csize, 8)
    em, tau, flux, ftot, tavg = _run_ff(eng, fields, U.bursts_from_oracle(jet), jet, [0.7],
                                        [5e9])
    jet.time = 0.7 * orc.YEAR
    np.testing.assert_allclose(em[0], jet.emission_measure(), rtol=1e-11)
    ref_tau = jet.optical_depth_ff(5e9)
    np.testing.assert_allclose(tau[0, 0], ref_tau, rtol=1e-11)
    ref_flux = jet.flux_ff(5e9)
    assert np.array_equal(np.isnan(flux[0, 0]), np.isnan(ref_flux))
    np.testing.assert_allclose(flux[0, 0], ref_flux, rtol=1e-10)
    assert tau[0, 0][0, 0] == 0.0 and em[0][0, 0] == 0.0 and np.isnan(flux[0, 0][0, 0])
    assert tau[0, 0][1, 1] == 0.0 and em[0][1, 1] > 0.0 and np.isnan(flux[0, 0][1, 1])
    assert tau[0, 0][2, 3] == 0.0 and em[0][2, 3] == 0.0 and flux[0, 0][2, 3] == 0.0


@pytest.mark.parametrize("tag", ["cfg1_example", "tilted"])
@pytest.mark.parametrize("dtype", [8, 4])
def test_rrl_against_reference_golden(eng, tag, dtype):
    """K3 + map stage vs the reference's RRL optical depths and fluxes (scipy wofz)."""
    from rajepy_amd import _lib, engine as E
    from rajepy_amd.maths import physics as ph, rrls
    z, meta, p, g, jet = U.golden_dense(tag)
    fields = _upload(eng, g, jet.csize, dtype)
    bursts = U.bursts_from_oracle(jet)
    rf = z["rrl_freqs"]
    line = _lib.Line(**rrls.line_constants(meta["rrl"]))
    t0 = z["years"][0] * orc.YEAR
    tau_rrl = eng.rrl_scan(fields, bursts, t0, line, rf)
    mode = E.RJP_GFF_SCALAR if p["power_laws"]["q_T"] == 0. else E.RJP_GFF_POWERLAW
    gv = [ph.gff(nu, p["properties"]["T_0"]) for nu in rf] if mode == E.RJP_GFF_SCALAR else None
    ctau, cflux = E.ff_channel_coeffs(rf, jet.csize, p["target"]["dist"], mode, gv)
    sumA, em, tavg = eng.ff_scan(fields, bursts, [t0], mode)
    tau_ff, flux_ff, _ = eng.ff_maps(sumA, tavg, ctau, cflux)
    cfl, hnu = E.rrl_channel_coeffs(rf, jet.csize, p["target"]["dist"])
    F = len(rf)
    P = jet.nx * jet.nz
    f_sub, _ = eng.rrl_maps(tau_rrl, tau_ff.reshape(F, P), tavg, None, cfl, hnu)
    f_tot, ftot = eng.rrl_maps(tau_rrl, tau_ff.reshape(F, P), tavg, flux_ff.reshape(F, P), cfl, hnu)
    eng.synchronize()
    shp = (F, jet.nx, jet.nz)
    tol = 1e-9 if dtype == 8 else RTOL
    np.testing.assert_allclose(tau_rrl.cpu().numpy().reshape(shp), z["tau_rrl"], rtol=tol)
    got = f_sub.cpu().numpy().reshape(shp)
    assert np.array_equal(np.isnan(got), np.isnan(z["flux_rrl_contsub"]))
    np.testing.assert_allclose(got, z["flux_rrl_contsub"], rtol=tol)
    np.testing.assert_allclose(f_tot.cpu().numpy().reshape(shp), z["flux_rrl_total"], rtol=tol)
    np.testing.assert_allclose(ftot.cpu().numpy(), np.nansum(z["flux_rrl_total"], axis=(1, 2)),
                               rtol=tol)


@pytest.mark.parametrize("nchan", [1, 5, 40, 300])
def test_rrl_dense_synthetic_vs_oracle(eng, nchan):
    """Dense synthetic cells, every channel-lane layout (16/64/256 lanes, >1 channel block),
    wide band so the far-field continued fraction, the core and the pole term are all hit."""
    from rajepy_amd import _lib
    from rajepy_amd.maths import rrls
    shape = (2, 19, 21)
    seed = 20240503
    fields = eng.synth_fields(shape, seed, 1, 8, csize_au=0.5, with_vy=True)
    g = U.synth_host(shape, seed, 1)
    p = copy.deepcopy(U.load_golden("cfg1_example")[2])
    p["ejection"] = U.example_bursts_params()
    p["power_laws"]["q_T"] = -0.5
    p["grid"].update(n_x=shape[0], n_y=shape[1], n_z=shape[2])
    jet = orc.OracleJet.from_fields(p, g["nd"], g["xi"], g["temp"], g["ff"], g["areas"],
                                    g["ts"], g["rr"], g["vy"])
    jet.time = 1.3 * orc.YEAR
    nu0 = rrls.rrl_nu_0("H", 66, 1)
    rf = orc.chan_freqs(nu0, nchan * 1.5e5 if nchan > 1 else 1.0, 1.5e5 if nchan > 1 else 1.0)
    assert len(rf) == nchan
    line = _lib.Line(**rrls.line_constants("H66a"))
    tau = eng.rrl_scan(fields, U.bursts_from_oracle(jet), jet.time, line, rf)
    eng.synchronize()
    ref = jet.optical_depth_rrl("H66a", np.asarray(rf))
    np.testing.assert_allclose(tau.cpu().numpy().reshape(ref.shape), ref, rtol=U.k3_rtol(nchan))


@pytest.mark.parametrize("tag", ["cfg1_example", "tilted"])
def test_field_builder_vs_reference(eng, tag):
    """K4: geometry -> fields on the device vs the reference's own grids."""
    from rajepy_amd.classes import geometry_struct
    z, meta, p = U.load_golden(tag)
    jet = orc.OracleJet(p)                      # derived params (mod_r_0, q_n, n_0 ...)
    geom = geometry_struct(jet.params, jet.nx, jet.ny, jet.nz)
    f = eng.build_fields(geom, 8, want_ts=True, want_vxz=True)   # tilted: device 2F1
    eng.synchronize()
    idx = z["f_idx"]
    ff = f.ff_raw.cpu().numpy()
    assert np.array_equal(np.flatnonzero(np.isfinite(ff)), idx)          # identical jet mask
    assert np.array_equal(ff[idx], z["f_ff"])
    assert np.array_equal(f.areas_raw.cpu().numpy()[idx], z["f_areas"])
    nd = f.nd.cpu().numpy()
    np.testing.assert_allclose(np.abs(nd[idx]), z["f_nd"], rtol=1e-12)
    assert np.array_equal(np.signbit(nd[idx]), z["f_rr"] < 0)
    assert not np.isfinite(np.delete(nd, idx)).any()
    for name, key in (("xi", "xi"), ("temp", "temp"), ("vy", "vy")):
        got = getattr(f, name).cpu().numpy()
        np.testing.assert_allclose(got[idx], z["f_" + key], rtol=1e-11, atol=1e-12, err_msg=name)
        assert not np.isfinite(np.delete(got, idx)).any()
    np.testing.assert_allclose(f.pf.cpu().numpy()[idx], z["f_ff"] / z["f_areas"], rtol=0)
    np.testing.assert_allclose(f.vx_raw.cpu().numpy()[idx], z["f_vx"], rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(f.vz_raw.cpu().numpy()[idx], z["f_vz"], rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(f.ts.cpu().numpy()[idx], z["f_ts0"], rtol=1e-10, atol=1e-3)


def test_field_builder_refuses_degenerate_2f1(eng):
    """a - b a non-positive integer needs the logarithmic 2F1 cases: the library says so
    instead of returning garbage (JetModel then falls back to the host integral)."""
    from rajepy_amd import _lib
    from rajepy_amd.classes import geometry_struct
    z, meta, p = U.load_golden("tilted")
    p["power_laws"]["q_v"] = 1. - p["geometry"]["epsilon"]               # -> b = a + 1
    jet = orc.OracleJet(p)
    geom = geometry_struct(jet.params, jet.nx, jet.ny, jet.nz)
    with pytest.raises(_lib.RjprtError, match="degenerate"):
        eng.build_fields(geom, 8, want_ts=True)
    eng.build_fields(geom, 8, want_ts=False)


def test_abi_error_reporting(eng):
    """Every entry point returns a negative status + message instead of crashing on bad
    arguments (include/rjprt.h conventions)."""
    import ctypes as C
    from rajepy_amd import _lib, engine as E
    lib = eng.lib
    fields = eng.synth_fields((2, 8, 8), 1, 0, 8)
    fs = fields.struct()
    ep = _lib.dbl_array([0.0])
    out = eng._f64(1, 16)
    tavg = eng._f64(16)
    work = eng._workspace(lib.rjp_ff_scan_workspace(2, 8, 8, 1))
    st = eng._stream()
    call = lambda **kw: None

    def scan(f=fs, e=ep, ne=1, mode=0, o=out, w=work, wb=None):
        return lib.rjp_ff_scan(eng.ctx, C.byref(f) if f is not None else None, None, e, ne, mode,
                               o.data_ptr() if o is not None else None, None, tavg.data_ptr(),
                               w.data_ptr() if w is not None else None,
                               w.numel() if wb is None else wb, st)
    assert scan() == 0
    assert scan(f=None) == -1 and b"fields" in lib.rjp_last_error(eng.ctx)
    assert scan(ne=0) == -1
    assert scan(mode=7) == -1 and b"gff_mode" in lib.rjp_last_error(eng.ctx)
    assert scan(o=None) == -1
    assert scan(wb=16) == -4 and b"workspace" in lib.rjp_last_error(eng.ctx)
    bad = fields.struct()
    bad.dtype = 3
    assert scan(f=bad) == -1 and b"dtype" in lib.rjp_last_error(eng.ctx)
    bad = fields.struct()
    bad.ny = 0
    assert scan(f=bad) == -1
    # bursts without launch times
    nots = fields.struct()
    nots.d_ts = None
    b = E.make_bursts([(1.0, 2.0, 3.0)], [])
    assert lib.rjp_ff_scan(eng.ctx, C.byref(nots), C.byref(b), ep, 1, 0, out.data_ptr(), None,
                           tavg.data_ptr(), work.data_ptr(), work.numel(), st) == -1
    # RRL needs vy
    line = _lib.Line(1e10, 1e-6, 1.0, 1e-12, -30.0, 4.8e-11)
    assert lib.rjp_rrl_scan(eng.ctx, C.byref(fs), None, 0.0, C.byref(line),
                            _lib.dbl_array([1e10]), 1, out.data_ptr(), st) == -1
    assert b"vy" in lib.rjp_last_error(eng.ctx)
    # any number of bursts is legal (test_gpu_round2.py); a jet with n > 0 needs its arrays
    nine = E.make_bursts([(0., 1., 1.)] * 9, [])
    assert nine.n[0] == 9 and nine.n[1] == 0
    nine.t0[0] = None
    assert lib.rjp_ff_scan(eng.ctx, C.byref(fs), C.byref(nine), ep, 1, 0, out.data_ptr(), None,
                           tavg.data_ptr(), work.data_ptr(), work.numel(), st) == -1
    assert b"bursts" in lib.rjp_last_error(eng.ctx)
    with pytest.raises(_lib.RjprtError):
        _lib.check(-2, eng.ctx, "demo")
    # a tau-only field set (a0, temp, ts) asked for EM maps: the scan would move to the wide
    # kernels, whose fields are not there -- refused before anything is enqueued
    lean = eng.synth_fields((2, 8, 8), 1, 0, 8, wide=False, tau_mode=0, with_em0=False)
    assert lean.em0 is None and lean.nd is None and lean.a0 is not None
    em = eng._f64(1, 16)
    ls = lean.struct()
    assert lib.rjp_ff_scan(eng.ctx, C.byref(ls), None, ep, 1, 0, out.data_ptr(), None, None,
                           work.data_ptr(), work.numel(), st) == 0
    assert lib.rjp_ff_scan(eng.ctx, C.byref(ls), None, ep, 1, 0, out.data_ptr(), em.data_ptr(),
                           None, work.data_ptr(), work.numel(), st) == -1
    assert b"d_em" in lib.rjp_last_error(eng.ctx)
    with pytest.raises(_lib.RjprtError, match="d_em"):
        eng.ff_scan(lean, None, [0.0], 0, want_em=True, want_tavg=False)
    # ... and the other Gaunt mode (the a0 field is ignored): nothing left to scan from
    assert lib.rjp_ff_scan(eng.ctx, C.byref(ls), None, ep, 1, 1, out.data_ptr(), None, None,
                           work.data_ptr(), work.numel(), st) == -1
    # out-of-range device index
    ctx = C.c_void_p()
    assert lib.rjp_ctx_create(10 ** 6, C.byref(ctx)) == -1
    eng.synchronize()


@pytest.mark.parametrize("dtype", [8, 4])
def test_no_bursts_paths(eng, dtype):
    """A model without ejection events (chi == 1): the BURSTS=false kernels, `ts` absent,
    several epochs requested (all identical), continuum and RRL."""
    from rajepy_amd import _lib, engine as E
    from rajepy_amd.maths import physics as ph, rrls
    shape = (3, 41, 12)
    g = U.synth_host(shape, 77, 1)
    p = copy.deepcopy(U.load_golden("cfg1_example")[2])
    p["ejection"] = {k: np.array([]) for k in ("t_0", "hl", "chi", "which")}
    p["power_laws"]["q_T"] = -0.5
    p["grid"].update(n_x=shape[0], n_y=shape[1], n_z=shape[2])
    jet = orc.OracleJet.from_fields(p, g["nd"], g["xi"], g["temp"], g["ff"], g["areas"],
                                    g["ts"], g["rr"], g["vy"])
    fields = eng.upload_fields(g["nd"], g["xi"], g["temp"], g["ff"], g["areas"], None,
                               g["rr"] < 0, g["vy"], csize_au=jet.csize, dtype=dtype)
    assert fields.ts is None
    years, freqs = [0., 1.5, 3.0], [2e9, 3e10]
    for bursts in (None, E.make_bursts([], [])):
        ctau, cflux = E.ff_channel_coeffs(freqs, jet.csize, p["target"]["dist"], E.RJP_GFF_POWERLAW)
        sumA, em, tavg = eng.ff_scan(fields, bursts, [y * orc.YEAR for y in years],
                                     E.RJP_GFF_POWERLAW)
        tau, flux, _ = eng.ff_maps(sumA, tavg, ctau, cflux)
        eng.synchronize()
        tol = 1e-11 if dtype == 8 else RTOL
        ref_tau, ref_flux, ref_em = (jet.optical_depth_ff(np.array(freqs)),
                                     jet.flux_ff(np.array(freqs)), jet.emission_measure())
        for e in range(len(years)):          # every epoch equals the steady state
            np.testing.assert_allclose(tau[e].cpu().numpy().reshape(ref_tau.shape), ref_tau, rtol=tol)
            np.testing.assert_allclose(flux[e].cpu().numpy().reshape(ref_tau.shape), ref_flux,
                                       rtol=max(tol, 1e-10))
            np.testing.assert_allclose(em[e].cpu().numpy().reshape(ref_em.shape), ref_em, rtol=tol)
    rf = orc.chan_freqs(rrls.rrl_nu_0("H", 66, 1), 20 * 2e5, 2e5)
    line = _lib.Line(**rrls.line_constants("H66a"))
    t = eng.rrl_scan(fields, None, 0.0, line, rf)
    eng.synchronize()
    ref = jet.optical_depth_rrl("H66a", np.asarray(rf))
    np.testing.assert_allclose(t.cpu().numpy().reshape(ref.shape), ref,
                               rtol=U.k3_rtol(len(rf)) if dtype == 8 else RTOL)


@pytest.mark.parametrize("dtype", [8, 4])
def test_y_bounds_skip_is_exact(eng, dtype):
    """The sparse-model shortcut: occupied y-ranges equal the host definition and scans with
    them are bit-identical to scans without (continuum base maps and RRL cube)."""
    from rajepy_amd import _lib, engine as E
    from rajepy_amd.maths import rrls
    z, meta, p, g, jet = U.golden_dense("tilted")
    # a stray finite temperature outside the jet must widen the range (it counts in T_avg)
    g["temp"][5, 60, 7] = 123.0
    fields = eng.upload_fields(g["nd"], g["xi"], g["temp"], g["ff"], g["areas"], g["ts"],
                               g["rr"] < 0, g["vy"], csize_au=jet.csize, dtype=dtype)
    bursts = U.bursts_from_oracle(jet)
    ep = [y * orc.YEAR for y in (0., 0.4, 0.9)]
    ref = [t.clone() for t in eng.ff_scan(fields, bursts, ep, E.RJP_GFF_POWERLAW)]
    line = _lib.Line(**rrls.line_constants(meta["rrl"]))
    ref_rrl = eng.rrl_scan(fields, bursts, ep[1], line, z["rrl_freqs"]).clone()
    lo, hi = eng.compute_y_bounds(fields)
    eng.synchronize()
    matters = (g["temp"] > 0) | (np.isfinite(g["nd"]) & np.isfinite(g["xi"]) &
                                 np.isfinite(g["ff"] / g["areas"]))
    any_ = matters.any(axis=1)
    exp_lo = np.where(any_, matters.argmax(axis=1), jet.ny)
    exp_hi = np.where(any_, jet.ny - matters[:, ::-1, :].argmax(axis=1), 0)
    assert np.array_equal(lo.cpu().numpy().reshape(jet.nx, jet.nz), exp_lo)
    assert np.array_equal(hi.cpu().numpy().reshape(jet.nx, jet.nz), exp_hi)
    assert exp_hi[5, 7] >= 61
    got = eng.ff_scan(fields, bursts, ep, E.RJP_GFF_POWERLAW)
    got_rrl = eng.rrl_scan(fields, bursts, ep[1], line, z["rrl_freqs"])
    eng.synchronize()
    import torch
    for a, b in zip(got, ref):
        assert torch.equal(torch.nan_to_num(a, nan=-1.0), torch.nan_to_num(b, nan=-1.0))
    assert torch.equal(got_rrl, ref_rrl)


@pytest.mark.parametrize("shape", [(1, 1, 2), (2, 3, 4), (1, 17, 6), (5, 2, 2), (3, 16, 18),
                                   (2, 33, 64), (7, 5, 130), (4, 257, 8), (2, 1000, 3)])
def test_shape_edge_cases_with_and_without_bounds(eng, shape):
    """Degenerate and ragged shapes (single row/column, n_y below the unroll and y-split sizes,
    n_z odd / not a multiple of the lane width, tile tails), sparse random masks, both field
    widths, scans with and without the occupied-range shortcut -- continuum and RRL."""
    from rajepy_amd import _lib, engine as E
    from rajepy_amd.maths import rrls
    rng = np.random.default_rng(sum(shape))
    g = U.synth_host(shape, 4242, 1)
    hole = rng.random(shape) < 0.6                      # 60 % of the cells outside the "jet"
    for k in ("nd", "xi", "temp", "ff", "areas", "vy"):
        g[k] = np.where(hole, np.nan, g[k])
    p = copy.deepcopy(U.load_golden("cfg1_example")[2])
    p["ejection"] = U.example_bursts_params()
    p["power_laws"]["q_T"] = -0.5
    p["grid"].update(n_x=shape[0], n_y=shape[1], n_z=shape[2])
    jet = orc.OracleJet.from_fields(p, g["nd"], g["xi"], g["temp"], g["ff"], g["areas"],
                                    g["ts"], g["rr"], g["vy"])
    jet.time = 1.1 * orc.YEAR
    freqs = np.array([3e9, 2e10])
    rf = orc.chan_freqs(rrls.rrl_nu_0("H", 66, 1), 5 * 3e5, 3e5)
    line = _lib.Line(**rrls.line_constants("H66a"))
    ref_tau, ref_em = jet.optical_depth_ff(freqs), jet.emission_measure()
    ref_flux, ref_rrl = jet.flux_ff(freqs), jet.optical_depth_rrl("H66a", np.asarray(rf))
    bursts = U.bursts_from_oracle(jet)
    ctau, cflux = E.ff_channel_coeffs(freqs, jet.csize, p["target"]["dist"], E.RJP_GFF_POWERLAW)
    for dtype in (8, 4):
        tol = 1e-10 if dtype == 8 else RTOL
        fields = _upload(eng, g, jet.csize, dtype)
        for use_bounds in (False, True):
            if use_bounds:
                eng.compute_y_bounds(fields)
            sumA, em, tavg = eng.ff_scan(fields, bursts, [jet.time], E.RJP_GFF_POWERLAW)
            tau, flux, _ = eng.ff_maps(sumA, tavg, ctau, cflux)
            trrl = eng.rrl_scan(fields, bursts, jet.time, line, rf)
            eng.synchronize()
            np.testing.assert_allclose(tau.cpu().numpy().reshape(ref_tau.shape), ref_tau, rtol=tol)
            np.testing.assert_allclose(em.cpu().numpy().reshape(ref_em.shape), ref_em, rtol=tol)
            got = flux.cpu().numpy().reshape(ref_flux.shape)
            assert np.array_equal(np.isnan(got), np.isnan(ref_flux))
            # 1 - exp(-tau) amplifies the last-bit difference between libm's and the
            # device's exp by 1/tau; these thin columns reach tau ~ 1e-6
            np.testing.assert_allclose(got, ref_flux, rtol=max(tol, 1e-9))
            np.testing.assert_allclose(trrl.cpu().numpy().reshape(ref_rrl.shape), ref_rrl,
                                       rtol=max(tol, U.k3_rtol(len(rf))))


@pytest.mark.parametrize("dtype", [8, 4])
@pytest.mark.parametrize("n_ep,t1", [(8, 3.5), (16, 5.0), (12, 0.55), (4, 2.0)])
def test_uniform_epoch_sweeps_use_the_recurrence_correctly(eng, dtype, n_ep, t1):
    """Uniformly spaced epochs take the two-exp-per-burst recurrence (tiles of 8 and 4, and a
    ragged tail); results must equal the oracle at every epoch, including epochs far from any
    burst (anchor Gaussian underflows) and bursts that peak inside a tile."""
    from rajepy_amd import engine as E
    shape = (4, 37, 16)
    seed = 20240507
    fields = eng.synth_fields(shape, seed, 0, dtype, csize_au=0.5)
    g = U.synth_host(shape, seed, 0)
    p = copy.deepcopy(U.load_golden("cfg1_example")[2])
    p["ejection"] = U.example_bursts_params()
    p["ejection"]["hl"] = np.array([0.02, 0.15, 0.45, 0.5])      # one very narrow burst
    p["grid"].update(n_x=shape[0], n_y=shape[1], n_z=shape[2])
    jet = orc.OracleJet.from_fields(p, g["nd"], g["xi"], g["temp"], g["ff"], g["areas"],
                                    g["ts"], g["rr"], g["vy"])
    years = np.linspace(0., t1, n_ep)
    sumA, em, tavg = eng.ff_scan(fields, U.bursts_from_oracle(jet), years * orc.YEAR,
                                 E.RJP_GFF_SCALAR)
    eng.synchronize()
    tol = 1e-11 if dtype == 8 else RTOL
    em_h = em.cpu().numpy().reshape(n_ep, shape[0], shape[2])
    for e, yr in enumerate(years):
        jet.time = yr * orc.YEAR
        np.testing.assert_allclose(em_h[e], jet.emission_measure(), rtol=tol, err_msg=str(e))


@pytest.mark.parametrize("seed", range(24))
def test_field_builder_random_geometries_vs_oracle(eng, seed):
    """K4 fuzz: random inclinations / position angles / opening angles / power laws / rotation
    sense on a small grid.  The device jet mask must equal the oracle builder's exactly (the
    oracle builder is itself pinned to the reference on three models) and every field agree."""
    from rajepy_amd.classes import geometry_struct
    rng = np.random.default_rng(1000 + seed)
    p = copy.deepcopy(U.load_golden("tilted")[2])
    g, pl, pr = p["geometry"], p["power_laws"], p["properties"]
    g.update(inc=float(rng.uniform(20, 90)), pa=float(rng.uniform(-180, 180)),
             opang=float(rng.uniform(10, 60)), epsilon=float(rng.uniform(0.4, 1.0)),
             w_0=float(rng.uniform(0.8, 2.0)), r_0=float(rng.uniform(0.5, 3.0)),
             rotation="CCW" if rng.random() < 0.5 else "CW")
    pl.update({"q_v": float(rng.uniform(-0.3, 0.3)), "q_T": float(rng.uniform(-0.2, 0.1)),
               "q_x": float(rng.uniform(-0.4, 0.2)), "q^d_n": float(rng.uniform(-1.2, 0.3)),
               "q^d_T": float(rng.uniform(-0.3, 0.3)), "q^d_x": float(rng.uniform(-0.3, 0.3)),
               "q^d_v": float(rng.choice([0.0, rng.uniform(-0.9, 0.6)]))})
    for k in ("mod_r_0",):
        g.pop(k, None)
    pl.pop("q_n", None), pl.pop("q_tau", None), pr.pop("n_0", None)
    p["grid"].update(n_x=20, n_y=26, n_z=30, c_size=float(rng.uniform(0.6, 1.5)))
    jet = orc.OracleJet(p)
    geom = geometry_struct(jet.params, jet.nx, jet.ny, jet.nz)
    f = eng.build_fields(geom, 8, want_ts=True, want_vxz=True)
    eng.synchronize()
    ff = f.ff_raw.cpu().numpy().reshape(jet.nx, jet.ny, jet.nz)
    assert np.array_equal(np.isnan(ff), np.isnan(jet.fill_factor))
    assert np.array_equal(np.nan_to_num(ff), np.nan_to_num(jet.fill_factor))
    jetm = np.isfinite(jet.fill_factor)
    assert jetm.sum() > 50, "degenerate draw"
    nd = f.nd.cpu().numpy().reshape(ff.shape)
    with np.errstate(all="ignore"):
        ref = dict(nd=jet.nd0, xi=jet.ion_fraction, temp=jet.temperature, vy=jet.vy, ts=jet.ts0)
    np.testing.assert_allclose(np.abs(nd[jetm]), ref["nd"][jetm], rtol=1e-11)
    assert np.array_equal(np.signbit(nd[jetm]), jet.rr[jetm] < 0)
    for name in ("xi", "temp", "vy", "ts"):
        got = getattr(f, name).cpu().numpy().reshape(ff.shape)
        ok = jetm & np.isfinite(ref[name])
        assert np.array_equal(np.isfinite(got[jetm]), np.isfinite(ref[name][jetm])), name
        np.testing.assert_allclose(got[ok], ref[name][ok], rtol=1e-9, atol=1e-6, err_msg=name)


def test_compact_field_is_the_steady_state_em_density_with_the_jet_flag(eng):
    """rjp_compact_fields: em0 = (|nd| xi)^2 pf in exactly that order (IEEE-exact against
    NumPy), NaN / inf / 0 / denormal / overflow behave as the wide kernel's arithmetic does,
    the sign bit carries the red-jet flag also on NaN and zero cells."""
    nd = np.array([1e5, 3.3e7, np.nan, 2e6, np.inf, 0.0, 5e-324, 1.7976931348623157e308,
                   4.4e6, 7.7e5, 1.0, 1.0 + 2 ** -52, 1e-170, 2.5e8])
    xi = np.array([0.3, np.nan, 0.2, 1.0, 0.5, 0.4, 1.0, 1.0, 0.123456789, 0.9, 1.0, 1.0, 1.0,
                   0.37])
    ff = np.array([1.0, 1.0, 0.5, np.nan, 1.0, 0.5, 1.0, 0.5, 0.0, -0.0, 0.5, 1.0, 1.0, 0.37])
    red = np.arange(nd.size) % 2 == 0
    shape = (1, nd.size // 2, 2)
    r = lambda a: np.asarray(a).reshape(shape)
    f = eng.upload_fields(r(nd), r(xi), r(np.full(nd.size, 1e4)), r(ff), r(np.ones(nd.size)),
                          r(np.zeros(nd.size)), r(red), csize_au=1.0, dtype=8)
    assert f.em0 is not None
    got = f.em0.cpu().numpy()
    with np.errstate(all="ignore"):
        n0 = np.abs(nd) * xi
        want = n0 * n0 * ff
    assert np.array_equal(np.abs(got), np.abs(want), equal_nan=True)
    assert np.array_equal(np.signbit(got), red)


@pytest.mark.parametrize("temp_mode", [0, 1])
@pytest.mark.parametrize("n_ep", [1, 3, 8, 16])
@pytest.mark.parametrize("dtype", [8, 4])
def test_compact_layout_is_bit_identical_to_the_wide_one(eng, temp_mode, n_ep, dtype):
    """K1 from the 3-field compact layout vs the 5-field wide layout of the same model:
    bit-identical for f64 storage; f32 storage rounds the product once more (6e-8)."""
    from rajepy_amd import engine as E
    shape = (8, 96, 64)
    f = eng.synth_fields(shape, 20240509, temp_mode, dtype, csize_au=0.5)
    assert f.em0 is not None
    p = copy.deepcopy(U.load_golden("cfg1_example")[2])
    p["ejection"] = U.example_bursts_params()
    g = U.synth_host((1, 2, 2), 1, 0)
    jet = orc.OracleJet.from_fields(dict(p, grid=dict(p["grid"], n_x=1, n_y=2, n_z=2)), g["nd"],
                                    g["xi"], g["temp"], g["ff"], g["areas"], g["ts"], g["rr"],
                                    g["vy"])
    bursts = U.bursts_from_oracle(jet)
    ep = list(np.linspace(0.2, 3.4, n_ep) * orc.YEAR) if n_ep > 1 else [1.3 * orc.YEAR]
    mode = E.RJP_GFF_SCALAR if temp_mode == 0 else E.RJP_GFF_POWERLAW
    a1, e1, t1 = (x.clone() for x in eng.ff_scan(f, bursts, ep, mode))
    lo1, hi1 = (x.clone() for x in eng.compute_y_bounds(f))
    em0 = f.em0
    f.em0 = f.ylo = f.yhi = None
    a0, e0, t0 = eng.ff_scan(f, bursts, ep, mode)
    lo0, hi0 = eng.compute_y_bounds(f)
    eng.synchronize()
    for got, ref in ((a1, a0), (e1, e0), (t1, t0), (lo1, lo0), (hi1, hi0)):
        if n_ep >= 16 and dtype == 8 and (got is a1 or got is e1):
            # the wide layout has no 16-epoch tile: it evaluates these epochs directly, the
            # compact one by the uniform-epoch recurrence (same numbers to rounding)
            np.testing.assert_allclose(got.cpu().numpy(), ref.cpu().numpy(), rtol=1e-11)
        elif dtype == 8 or got is t1 or got.dtype != a1.dtype:
            assert np.array_equal(got.cpu().numpy(), ref.cpu().numpy())
        else:
            np.testing.assert_allclose(got.cpu().numpy(), ref.cpu().numpy(), rtol=2e-7)
    # continuum-only sweeps may free the wide fields; calls that need them refuse loudly
    f.em0 = em0
    f.ylo = f.yhi = None
    f.drop_wide()
    a2, _, _ = eng.ff_scan(f, bursts, ep, mode)
    assert np.array_equal(a2.cpu().numpy(), a1.cpu().numpy())
    from rajepy_amd._lib import RjprtError
    with pytest.raises(RjprtError, match="nd/xi/temp/pf"):
        eng.ff_cells(f, bursts, ep[0], mode, [1.0])


def test_f32_compact_field_and_its_range_guard(eng):
    """f32 storage: em0 is the float rounding of the f64 product of the float-rounded fields;
    a product beyond the float range keeps the model on the wide layout."""
    shape = (2, 6, 4)
    g = U.synth_host(shape, 77, 1)
    up = lambda gg: eng.upload_fields(gg["nd"], gg["xi"], gg["temp"], gg["ff"], gg["areas"],
                                      gg["ts"], gg["rr"] < 0, csize_au=1.0, dtype=4)
    f = up(g)
    assert f.em0 is not None and f.em0.dtype == f.temp.dtype
    f32 = lambda a: a.astype(np.float32).astype(np.float64)
    n0 = f32(g["nd"]) * f32(g["xi"])
    want = (n0 * n0 * f32(g["ff"] / g["areas"])).astype(np.float32)
    got = f.em0.cpu().numpy().reshape(shape)
    assert np.array_equal(np.abs(got), want)
    assert np.array_equal(np.signbit(got), g["rr"] < 0)
    big = dict(g, nd=g["nd"] * 1e17)                    # (n x)^2 ~ 1e43 > FLT_MAX
    assert up(big).em0 is None
    tiny = dict(g, nd=g["nd"] * 1e-27)                  # (n x)^2 ~ 1e-45: would flush to zero
    assert up(tiny).em0 is None


def test_negative_path_factors_keep_the_wide_layout(eng):
    """A negative path factor would collide with the jet flag in the sign bit of em0: the
    engine keeps the wide layout and the maps still follow the oracle."""
    shape = (3, 20, 8)
    g = U.synth_host(shape, 5, 0)
    g["ff"] = np.where(g["ff"] == 0.5, -0.37, 1.0)
    p = copy.deepcopy(U.load_golden("cfg1_example")[2])
    p["grid"].update(n_x=shape[0], n_y=shape[1], n_z=shape[2])
    jet = orc.OracleJet.from_fields(p, g["nd"], g["xi"], g["temp"], g["ff"], g["areas"],
                                    g["ts"], g["rr"], g["vy"])
    f = _upload(eng, g, jet.csize, 8)
    assert f.em0 is None
    em, tau, flux, _, _ = _run_ff(eng, f, U.bursts_from_oracle(jet), jet, [0.9], [5e9])
    jet.time = 0.9 * orc.YEAR
    np.testing.assert_allclose(em[0], jet.emission_measure(), rtol=1e-11)
    np.testing.assert_allclose(tau[0, 0], jet.optical_depth_ff(5e9), rtol=1e-11)


@pytest.mark.parametrize("store", STORES)
@pytest.mark.parametrize("epochs", [[0.3, 0.9, 1.0, 2.2], list(np.linspace(0.1, 3.1, 8)),
                                    list(np.linspace(0., 5., 16)),
                                    [0.2, 0.25, 0.4, 0.8, 1.3, 1.35, 2.0, 2.9, 3.3]])
def test_scans_without_emission_measure_give_the_same_tau_sums(eng, store, epochs):
    """Flux-vs-time sweeps pass d_em = NULL and get kernels without the EM accumulators (tiles
    of 4, 8 and 16 epochs, uniform and not): sumA and T_avg must not change by a bit.
    (From 32 uniformly spaced epochs on they also get a larger tile: next test.)"""
    from rajepy_amd import engine as E
    shape = (5, 70, 32)
    f = _layout(eng.synth_fields(shape, 31337, 1, _store(store), csize_au=0.5), store)
    p = copy.deepcopy(U.load_golden("cfg1_example")[2])
    p["ejection"] = U.example_bursts_params()
    g = U.synth_host((1, 2, 2), 1, 0)
    jet = orc.OracleJet.from_fields(dict(p, grid=dict(p["grid"], n_x=1, n_y=2, n_z=2)), g["nd"],
                                    g["xi"], g["temp"], g["ff"], g["areas"], g["ts"], g["rr"],
                                    g["vy"])
    bursts = U.bursts_from_oracle(jet)
    ep = [y * orc.YEAR for y in epochs]
    a1, e1, t1 = (x.clone() for x in eng.ff_scan(f, bursts, ep, E.RJP_GFF_POWERLAW))
    a0, e0, t0 = eng.ff_scan(f, bursts, ep, E.RJP_GFF_POWERLAW, want_em=False)
    eng.synchronize()
    assert e0 is None and e1 is not None
    assert np.array_equal(a0.cpu().numpy(), a1.cpu().numpy())
    assert np.array_equal(t0.cpu().numpy(), t1.cpu().numpy())


@pytest.mark.parametrize("producer", ["synth0", "synth1", "cfg1_example", "tilted"])
def test_producers_write_the_compact_field_themselves(eng, producer):
    """f64 producers (the synthetic generator and K4) emit em0 in their own pass: it must be
    bit-identical to what rjp_compact_fields derives from their nd, xi, pf, and a model
    generated WITHOUT the wide fields (24 B/cell resident) must scan to the same maps."""
    import torch
    from rajepy_amd import engine as E
    from rajepy_amd._lib import RjprtError
    from rajepy_amd.classes import geometry_struct
    if producer.startswith("synth"):
        shape, tm = (6, 40, 32), int(producer[-1])
        make = lambda wide: eng.synth_fields(shape, 4711, tm, 8, csize_au=0.5, wide=wide)
        mode = E.RJP_GFF_SCALAR if tm == 0 else E.RJP_GFF_POWERLAW
    else:
        z, meta, p = U.load_golden(producer)
        jet = orc.OracleJet(p)
        geom = geometry_struct(jet.params, jet.nx, jet.ny, jet.nz)
        make = lambda wide: eng.build_fields(geom, 8, want_ts=True, want_vy=False,
                                             want_raw=False, want_wide=wide)
        mode = E.RJP_GFF_SCALAR if p["power_laws"]["q_T"] == 0. else E.RJP_GFF_POWERLAW
    f = make(True)
    direct = f.em0.clone()
    eng.compact(f)
    eng.synchronize()
    same = (direct.view(torch.int64) == f.em0.view(torch.int64))
    assert bool(same.all())
    ep = [0.4 * orc.YEAR, 1.1 * orc.YEAR]
    bursts = E.make_bursts([(0.5 * orc.YEAR, 4., 2e6)], [(1.0 * orc.YEAR, 1.5, 6e6)])
    ref = [t.clone() for t in eng.ff_scan(f, bursts, ep, mode)]
    lean = make(False)
    assert lean.nd is None and lean.xi is None and lean.pf is None and lean.em0 is not None
    eng.compute_y_bounds(lean)
    got = eng.ff_scan(lean, bursts, ep, mode)
    eng.synchronize()
    for a, b in zip(got, ref):
        assert torch.equal(torch.nan_to_num(a, nan=-1.0), torch.nan_to_num(b, nan=-1.0))
    with pytest.raises(RjprtError, match="nd/xi/temp/pf"):
        eng.ff_cells(lean, bursts, ep[0], mode, [1.0])
    with pytest.raises(ValueError, match="compact layout"):
        eng.synth_fields((2, 4, 4), 1, 0, 4, wide=False)


@pytest.mark.parametrize("nchan,cw", [(64, 1.5e5), (256, 1.5e5), (300, 1.0e5), (16, 4e5)])
@pytest.mark.parametrize("thin", [1e-2, 1e-4, 1e-8])
def test_rrl_small_voigt_y_every_lane_layout(eng, nchan, cw, thin):
    """Cells with a tiny Lorentzian width (Voigt y from ~3e-2 down to ~1e-11): the lattice
    centred on x (kernels with >= 64 channel lanes: one cell per wave) and the half-shifted
    lattice (16 lanes) against scipy's wofz through the oracle, over a band wide enough for
    the core, the pole term, the far field and |x| beyond the centred table (|x| > 16)."""
    from rajepy_amd import _lib
    from rajepy_amd.maths import rrls
    shape = (2, 23, 9)
    g = U.synth_host(shape, 20240507, 1)
    g["nd"] = g["nd"] * thin
    g["vy"][0, :, 0] = 0.0                               # a sightline exactly on the line centre
    p = copy.deepcopy(U.load_golden("cfg1_example")[2])
    p["power_laws"]["q_T"] = -0.5
    p["grid"].update(n_x=shape[0], n_y=shape[1], n_z=shape[2])
    jet = orc.OracleJet.from_fields(p, g["nd"], g["xi"], g["temp"], g["ff"], g["areas"],
                                    g["ts"], g["rr"], g["vy"])
    jet.time = 0.4 * orc.YEAR
    fields = _upload(eng, g, jet.csize, 8)
    nu0 = rrls.rrl_nu_0("H", 66, 1)
    rf = orc.chan_freqs(nu0, nchan * cw, cw)
    line = _lib.Line(**rrls.line_constants("H66a"))
    tau = eng.rrl_scan(fields, U.bursts_from_oracle(jet), jet.time, line, rf)
    eng.synchronize()
    ref = jet.optical_depth_rrl("H66a", np.asarray(rf))
    assert np.isfinite(ref).all() and (ref > 0).all()
    np.testing.assert_allclose(tau.cpu().numpy().reshape(ref.shape), ref, rtol=U.k3_rtol(nchan))


@pytest.mark.parametrize("store", ["f64", "f32"])
@pytest.mark.parametrize("n_ep,t1", [(32, 5.0), (45, 4.0), (64, 2.0), (33, 0.6),
                                     (32, 8.0)])      # last: too wide for 32, falls back to 16
def test_32_epoch_tiles_follow_the_oracle(eng, store, n_ep, t1):
    """>= 32 uniformly spaced epochs: one pass serves 32 epochs (recurrence anchored at the
    tile's middle epoch), with the emission-measure accumulators (d_em given) or without
    (d_em = NULL) -- two kernels.  Both against the oracle, and against each other."""
    from rajepy_amd import engine as E
    shape = (4, 37, 16)
    dtype = _store(store)
    seed = 20240511
    fields = eng.synth_fields(shape, seed, 1, dtype, csize_au=0.5)
    g = U.synth_host(shape, seed, 1)
    p = copy.deepcopy(U.load_golden("cfg1_example")[2])
    p["ejection"] = U.example_bursts_params()
    p["power_laws"]["q_T"] = -0.5
    p["grid"].update(n_x=shape[0], n_y=shape[1], n_z=shape[2])
    jet = orc.OracleJet.from_fields(p, g["nd"], g["xi"], g["temp"], g["ff"], g["areas"],
                                    g["ts"], g["rr"], g["vy"])
    years = np.linspace(0.0, t1, n_ep)
    ep = [y * orc.YEAR for y in years]
    bursts = U.bursts_from_oracle(jet)
    a32, none, _ = eng.ff_scan(fields, bursts, ep, E.RJP_GFF_POWERLAW, want_em=False)
    a32 = a32.clone()
    a16, em16, _ = eng.ff_scan(fields, bursts, ep, E.RJP_GFF_POWERLAW, want_em=True)
    eng.synchronize()
    assert none is None
    tol = 1e-11 if dtype == 8 else RTOL
    np.testing.assert_allclose(a32.cpu().numpy(), a16.cpu().numpy(), rtol=tol)
    ctau, _ = E.ff_channel_coeffs([5e9], jet.csize, p["target"]["dist"], E.RJP_GFF_POWERLAW)
    got = a32.cpu().numpy().reshape(n_ep, shape[0], shape[2]) * ctau[0]
    em_h = em16.cpu().numpy().reshape(n_ep, shape[0], shape[2])
    for e in (0, 1, n_ep // 2, 31, n_ep - 1):
        jet.time = ep[e]
        np.testing.assert_allclose(got[e], jet.optical_depth_ff(5e9), rtol=tol)
        np.testing.assert_allclose(em_h[e], jet.emission_measure(), rtol=tol, err_msg=str(e))


def test_per_call_tables_are_reused_only_when_equal(eng):
    """The map stage keeps recent coefficient tables on the device and reuses a copy when a
    call brings the same content: alternate and cycle through more tables than the ring
    holds (same length, different values) and check every result against its own table."""
    import torch
    P = 700
    A = torch.linspace(0.5, 3.0, P, dtype=torch.float64, device=eng.device).reshape(1, P)
    tavg = torch.full((P,), 8e3, dtype=torch.float64, device=eng.device)
    tables = [(np.array([1.0, 2.0, 3.0]) * (k + 1), np.array([0.5, 0.25, 0.125]) * (k + 2))
              for k in range(11)]
    order = [0, 1, 0, 0, 2, 3, 4, 5, 6, 7, 8, 9, 10, 0, 1, 10, 0]
    outs = []
    for k in order:                                   # no synchronisation between the calls
        ctau, cflux = tables[k]
        tau, flux, ftot = eng.ff_maps(A, tavg, ctau, cflux)
        outs.append((k, tau, flux, ftot))
    eng.synchronize()
    a = A.cpu().numpy()[0]
    for k, tau, flux, ftot in outs:
        ctau, cflux = tables[k]
        want_tau = ctau[:, None] * a[None, :]
        want_flux = cflux[:, None] * (8e3 * (1.0 - np.exp(-want_tau)))
        np.testing.assert_allclose(tau.cpu().numpy()[0], want_tau, rtol=1e-15)
        np.testing.assert_allclose(flux.cpu().numpy()[0], want_flux, rtol=1e-13)
        np.testing.assert_allclose(ftot.cpu().numpy()[0], want_flux.sum(axis=1), rtol=1e-12)


@pytest.mark.parametrize("npix,nchan,n_ep", [(65536, 128, 1), (262144, 70, 1), (65536, 33, 4),
                                             (65535, 128, 1), (4096, 16, 1)])
def test_k2_per_lane_flux_accumulators_every_launch_shape(eng, npix, nchan, n_ep):
    """K2's three launch shapes -- 4 pixel groups x 16 channels per lane with the per-channel
    flux totals accumulated in registers (large maps), 1 pixel group x 16 channels, and the
    small-map kernel -- against the map formulas (classes.py:1395-1397, 1473-1475, 1519-1521)
    in NumPy, NaN pixels included (nansum), odd pixel counts (scalar lanes), ragged last
    channel slice."""
    import torch
    rng = np.random.default_rng(npix + nchan)
    A = rng.uniform(1e-3, 3e3, (n_ep, npix)) * 10.0 ** rng.uniform(-8, 2, (n_ep, npix))
    T = rng.uniform(5e3, 2e4, npix)
    T[rng.random(npix) < 0.01] = np.nan                       # empty sightlines
    A[:, np.isnan(T)] = 0.0
    ctau = 10.0 ** rng.uniform(-6, 1, nchan)
    cflux = 10.0 ** rng.uniform(-12, -8, nchan)
    dA = torch.from_numpy(A).to(eng.device)
    dT = torch.from_numpy(T).to(eng.device)
    tau, flux, ftot = eng.ff_maps(dA, dT, ctau, cflux)
    t2, f2, _ = eng.ff_maps(dA, dT, ctau, cflux, want_ftot=False)
    eng.synchronize()
    want_tau = ctau[None, :, None] * A[:, None, :]
    want_flux = cflux[None, :, None] * (T[None, None, :] * -np.expm1(-want_tau))
    np.testing.assert_allclose(tau.cpu().numpy(), want_tau, rtol=1e-14)
    got_f = flux.cpu().numpy()
    assert np.array_equal(np.isnan(got_f), np.isnan(want_flux))
    np.testing.assert_allclose(got_f, want_flux, rtol=1e-13)
    np.testing.assert_allclose(ftot.cpu().numpy(), np.nansum(want_flux, axis=2), rtol=1e-12)
    # the same cubes bit for bit when no totals are asked for
    assert torch.equal(tau.view(torch.int64), t2.view(torch.int64))
    assert bool(((flux == f2) | (flux.isnan() & f2.isnan())).all())
