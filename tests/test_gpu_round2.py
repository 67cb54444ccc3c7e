"""GPU parity tests added in round 2: any number of bursts per jet (the reference registers
every row of params["ejection"], classes.py:245-264), helium lines through K3, the dedicated
degenerate-2F1 status, a bounded base-map cache, and x-slab sweeps with more ranks than rows."""
import copy
import os
import socket

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


def _many_bursts(n_rb=12, extra_blue=1, seed=3):
    """n_rb bursts hitting both jets + `extra_blue` blue-only ones: 12 red / 13 blue."""
    rng = np.random.default_rng(seed)
    n = n_rb + extra_blue
    return {"t_0": np.sort(rng.uniform(0.2, 3.0, n)), "hl": rng.uniform(0.1, 0.5, n),
            "chi": rng.uniform(0.4, 6.0, n),          # some dips (chi < 1) among the bursts
            "which": np.array(["RB"] * n_rb + ["B"] * extra_blue)}


def _jet(shape, seed, temp_mode, q_T, ejection):
    g = U.synth_host(shape, seed, temp_mode)
    p = copy.deepcopy(U.load_golden("cfg1_example")[2])
    p["ejection"] = ejection
    p["power_laws"]["q_T"] = q_T
    p["grid"].update(n_x=shape[0], n_y=shape[1], n_z=shape[2])
    jet = orc.OracleJet.from_fields(p, g["nd"], g["xi"], g["temp"], g["ff"], g["areas"],
                                    g["ts"], g["rr"], g["vy"])
    return jet, p


@pytest.mark.parametrize("store,tol", [(8, 1e-11), (4, 1e-5)])
@pytest.mark.parametrize("shape", [(4, 41, 16), (2, 41, 256)])
def test_more_than_eight_bursts_per_jet_k1(eng, store, tol, shape):
    """12 red + 13 blue bursts: the first eight of a jet travel as scalar kernel arguments,
    the rest in the staged device table.  Direct evaluation (irregular epochs), the
    uniform-epoch recurrence with EM maps (16-epoch tiles) and without (32-epoch tile), all
    against the oracle's chained closures."""
    from rajepy_amd import engine as E
    # (4, 41, 16): z = 8 splits the jets inside one wave (mixed lanes, three-operation
    # recurrence); (2, 41, 256): whole waves inside one jet (two-operation recurrence, the
    # overflow bursts' step tables and parameters by scalar loads)
    seed = 20240521
    ej = _many_bursts()
    jet, p = _jet(shape, seed, 0, 0., ej)
    assert len(jet.bursts["R"]) == 12 and len(jet.bursts["B"]) == 13
    fields = eng.synth_fields(shape, seed, 0, store, csize_au=0.5)
    bursts = U.bursts_from_oracle(jet)
    assert bursts.n[0] == 12 and bursts.n[1] == 13
    ctau, _ = E.ff_channel_coeffs([5e9], jet.csize, p["target"]["dist"], E.RJP_GFF_SCALAR,
                                  [orc.gff(5e9, 1e4)])

    def check(years, want_em):
        ep = [y * YEAR for y in years]
        sumA, em, _ = eng.ff_scan(fields, bursts, ep, E.RJP_GFF_SCALAR, want_em=want_em)
        eng.synchronize()
        got = sumA.cpu().numpy().reshape(len(ep), shape[0], shape[2]) * ctau[0]
        pick = sorted({0, 1, len(ep) // 2, len(ep) - 1})
        for e in pick:
            jet.time = ep[e]
            np.testing.assert_allclose(got[e], jet.optical_depth_ff(5e9), rtol=tol)
            if want_em:
                np.testing.assert_allclose(em.cpu().numpy()[e].reshape(shape[0], shape[2]),
                                           jet.emission_measure(), rtol=tol)

    check([0.0, 0.33, 0.9, 1.7, 2.95], True)              # direct: tiles of 4 + 1
    check(list(np.linspace(0., 3., 16)), True)            # uniform recurrence, 16-epoch tile
    check(list(np.linspace(0., 3.1, 32)), False)          # uniform recurrence, 32-epoch tile
    check(list(np.linspace(0., 3.1, 37)), False)          # 32 + a tail of direct tiles


def test_more_than_eight_bursts_per_jet_k3_and_cells(eng):
    from rajepy_amd import _lib, engine as E
    from rajepy_amd.maths import rrls
    shape = (3, 29, 16)
    seed = 20240522
    jet, p = _jet(shape, seed, 1, -0.5, _many_bursts(seed=5))
    jet.time = 1.3 * YEAR
    fields = eng.synth_fields(shape, seed, 1, 8, csize_au=0.5, with_vy=True)
    bursts = U.bursts_from_oracle(jet)
    rf = orc.chan_freqs(rrls.rrl_nu_0("H", 66, 1), 20 * 4e5, 4e5)
    line = _lib.Line(**rrls.line_constants("H66a"))
    tau = eng.rrl_scan(fields, bursts, jet.time, line, rf)
    cells = eng.rrl_cells(fields, bursts, jet.time, line, rf[:3])
    ctau, _ = E.ff_channel_coeffs([5e9], jet.csize, p["target"]["dist"], E.RJP_GFF_POWERLAW)
    ffc = eng.ff_cells(fields, bursts, jet.time, E.RJP_GFF_POWERLAW, ctau)
    eng.synchronize()
    ref = jet.optical_depth_rrl("H66a", np.asarray(rf))
    np.testing.assert_allclose(tau.cpu().numpy().reshape(ref.shape), ref, rtol=U.k3_rtol(len(rf)))
    np.testing.assert_allclose(cells.cpu().numpy().reshape((3,) + shape),
                               jet.optical_depth_rrl("H66a", np.asarray(rf[:3]), collapse=False),
                               rtol=1e-9)
    np.testing.assert_allclose(ffc.cpu().numpy().reshape(shape),
                               jet.optical_depth_ff(5e9, collapse=False), rtol=1e-11)


@pytest.mark.parametrize("rrl,nchan,cw", [("He42b", 24, 4e6), ("He66a", 40, 3e5),
                                          ("He58a", 300, 2e5)])
def test_helium_lines_against_the_oracle(eng, rrl, nchan, cw):
    """maths/rrls.py handles helium (Z = 2 rest frequencies and level energies, the He mass
    in the thermal width, the He ion fraction; rrls.py:14-41, 62-83, 104-118).  K3 gets all of
    that through `rjp_line`: 16-, 64- and 256-lane layouts against scipy's wofz."""
    from rajepy_amd import _lib
    from rajepy_amd.maths import rrls
    shape = (3, 33, 16)
    seed = 20240523
    jet, p = _jet(shape, seed, 1, -0.5, U.example_bursts_params())
    jet.time = 0.8 * YEAR
    fields = eng.synth_fields(shape, seed, 1, 8, csize_au=0.5, with_vy=True)
    el, n, dn = rrls.rrl_parser(rrl)
    assert el == "He"
    rf = orc.chan_freqs(rrls.rrl_nu_0(el, n, dn), nchan * cw, cw)
    line = _lib.Line(**rrls.line_constants(rrl))
    tau = eng.rrl_scan(fields, U.bursts_from_oracle(jet), jet.time, line, rf)
    eng.synchronize()
    ref = jet.optical_depth_rrl(rrl, np.asarray(rf))
    assert np.isfinite(ref).all() and (ref > 0).all()
    np.testing.assert_allclose(tau.cpu().numpy().reshape(ref.shape), ref, rtol=U.k3_rtol(nchan))


def test_degenerate_2f1_has_its_own_status_and_only_it_falls_back(eng, tmp_path, monkeypatch):
    from rajepy_amd import _lib, classes, logger
    from rajepy_amd.classes import geometry_struct
    z, meta, p = U.load_golden("tilted")
    p["power_laws"]["q_v"] = 1. - p["geometry"]["epsilon"]               # -> b = a + 1
    jet = orc.OracleJet(copy.deepcopy(p))
    geom = geometry_struct(jet.params, jet.nx, jet.ny, jet.nz)
    with pytest.raises(_lib.RjprtError) as ei:
        eng.build_fields(geom, 8, want_ts=True)
    assert ei.value.status == _lib.RJP_ERR_DEGENERATE
    # JetModel answers exactly that status with the reference's own host integral ...
    q = copy.deepcopy(p)
    for k in ("q_n", "q_tau"):
        q["power_laws"].pop(k, None)
    q["geometry"].pop("mod_r_0", None)
    q["properties"].pop("n_0", None)
    log = logger.Log(str(tmp_path / "a.log"), verbose=False)
    jm = classes.JetModel(q, log=log, engine=eng)
    ts_dev = jm.device_fields.ts.cpu().numpy().reshape(jet.nx, jet.ny, jet.nz)
    inside = np.isfinite(jet.nd0)
    np.testing.assert_allclose(ts_dev[inside], jet.ts0[inside], rtol=1e-10, atol=1e-3)
    # ... and the x-slab path takes the same fallback
    from rajepy_amd import parallel as par
    (x0, x1), tau, _, _ = par.xslab_local(jm, [0.5 * YEAR], [5e9], 1, 3)
    jm.time = 0.5 * YEAR
    np.testing.assert_array_equal(tau[0, 0].cpu().numpy(), jm.optical_depth_ff(5e9)[x0:x1])
    # ... while any other failure of the builder surfaces unchanged
    jm2 = classes.JetModel(copy.deepcopy(q), log=log, engine=eng)

    def broken(*a, **k):
        raise _lib.RjprtError("rjp_build_fields failed (status -2): hipErrorLaunchFailure",
                              status=_lib.RJP_ERR_HIP)
    monkeypatch.setattr(eng, "build_fields", broken)
    with pytest.raises(_lib.RjprtError, match="hipErrorLaunchFailure"):
        jm2.device_fields


def test_base_map_cache_is_bounded(eng, tmp_path):
    from rajepy_amd import classes, logger
    from tests.test_host_logic import example_params
    jm = classes.JetModel(example_params(), log=logger.Log(str(tmp_path / "a.log"),
                                                           verbose=False), engine=eng)
    jm.SCAN_CACHE_EPOCHS = 4
    times = np.linspace(0., 3., 10) * YEAR
    jm.prefetch_epochs(times)
    assert len(jm._scan_cache) == 4 and list(jm._scan_cache) == [float(t) for t in times[:4]]
    for t in times[[0, 5, 9, 2]]:
        jm.time = t
        jm.flux_ff(5e9)
        assert len(jm._scan_cache) <= 4
    # entries own their maps (a slice would pin the whole [E, P] result of its scan)
    a, e = jm._scan_cache[float(times[2])]
    assert a.numel() == jm.nx * jm.nz and a.untyped_storage().nbytes() == a.numel() * 8
    # (`ref` comes from the prefetched tile -- its uniform-epoch recurrence -- the second from a
    # single-epoch scan: the same maps to the 1e-11 the two paths agree to)
    ref = jm.flux_ff(5e9)
    jm._invalidate()
    new = jm.flux_ff(5e9)
    assert np.array_equal(np.isnan(new), np.isnan(ref))
    np.testing.assert_allclose(np.nan_to_num(new), np.nan_to_num(ref), rtol=1e-11, atol=0)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _xslab_worker(rank, world, port, ret):
    import torch.distributed as dist
    from rajepy_amd import classes, logger, parallel as par
    from tests.test_host_logic import example_params
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RJP_DEVICE="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p = example_params()
        p["grid"].update(n_x=2, n_y=40, n_z=16)
        jm = classes.JetModel(p, log=logger.Log(os.devnull, verbose=False))
        epochs, freqs = np.array([0., 0.7]) * YEAR, np.array([2e9, 2e10])
        ft, tau, flux = par.sweep_xslab(jm, epochs, freqs, rank, world, gather_maps=True)
        ok = tau.shape == (2, 2, 2, 16) and flux.shape == tau.shape
        for e, t in enumerate(epochs):
            jm.time = t
            ok = ok and np.array_equal(tau[e], jm.optical_depth_ff(freqs))
            f = jm.flux_ff(freqs)
            ok = ok and np.array_equal(np.nan_to_num(flux[e]), np.nan_to_num(f))
            ok = ok and np.allclose(ft[e], np.nansum(f, axis=(1, 2)), rtol=1e-12)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_xslab_sweep_with_more_ranks_than_rows():
    """3 ranks, 2 rows: the third rank's slab is empty; it must still enter the all_reduce
    and both all_gathers (zero-length blocks) instead of raising alone and leaving its peers
    in the collective.  Three processes share the one GPU of the box and talk over gloo."""
    import torch.multiprocessing as mp
    ret = mp.Manager().dict()
    mp.spawn(_xslab_worker, args=(3, _free_port(), ret), nprocs=3, join=True)
    assert all(ret[r] for r in range(3)), dict(ret)


@pytest.mark.parametrize("temp_k", [6e3, 1.2e4])
def test_k3_channels_concentrated_at_the_path_boundaries(eng, temp_k):
    """The wave-uniform Faddeeva paths switch at |z|^2 = 64 (lattice -> 6-term far series) and
    |z|^2 = 196 (-> 4-term series), where each series is at its least accurate: 256 channels
    laid out so that, of the four waves of a channel block, one sits just outside and one just
    inside each boundary (|x| within +-2 % of 8 and of 14), over cells whose Voigt y runs from
    1e-3 to ~10 -- the truncation errors there are one-signed along a sightline, so this is the
    case that would eat the margin if a coefficient or a threshold moved.  Against the oracle
    (scipy.special.wofz) at the bound the design guarantees (tests/gpu_util.K3_RTOL_WAVE)."""
    from rajepy_amd import _lib
    from rajepy_amd.maths import rrls
    shape = (2, 40, 8)
    g = U.synth_host(shape, 424242, 0)
    g["temp"][:] = temp_k
    g["vy"][:] = 6.2                                   # one Doppler shift: x is the same map of nu
    p = copy.deepcopy(U.load_golden("cfg1_example")[2])
    p["ejection"] = U.example_bursts_params()
    p["grid"].update(n_x=shape[0], n_y=shape[1], n_z=shape[2])
    jet = orc.OracleJet.from_fields(p, g["nd"], g["xi"], g["temp"], g["ff"], g["areas"],
                                    g["ts"], g["rr"], g["vy"])
    jet.time = 1.0 * YEAR
    fields = eng.upload_fields(g["nd"], g["xi"], g["temp"], g["ff"], g["areas"], g["ts"],
                               g["rr"] < 0, vy=g["vy"], csize_au=jet.csize, dtype=8)
    lc = rrls.line_constants("H66a")
    nu_c = lc["nu_rest"] * (1.0 - 6.2 * 1000.0 / 299792458.0)
    sig2 = lc["kG"] * np.sqrt(temp_k) * nu_c / 2.0 / 1.1774100225154747 * np.sqrt(2.0)  # sigma sqrt 2
    # |x| ranges of the four waves of the block (lane l of a block takes channel l/2 or 255 - l/2)
    xr = [(8.001, 8.16), (7.84, 7.999), (14.001, 14.28), (13.72, 13.999)]
    nu = np.empty(256)
    for w, (lo, hi) in enumerate(xr):
        xs = np.linspace(lo, hi, 64)
        sign = np.where(np.arange(64) % 2 == 0, 1.0, -1.0)      # both wings
        vals = nu_c + sign * xs * sig2
        nu[32 * w:32 * w + 32] = vals[:32]
        nu[255 - 32 * w - 31:255 - 32 * w + 1] = vals[32:]
    line = _lib.Line(**lc)
    tau = eng.rrl_scan(fields, U.bursts_from_oracle(jet), jet.time, line, list(nu))
    eng.synchronize()
    ref = jet.optical_depth_rrl("H66a", nu)
    assert np.isfinite(ref).all() and (ref > 0).all()
    np.testing.assert_allclose(tau.cpu().numpy().reshape(ref.shape), ref, rtol=U.K3_RTOL_WAVE)


def test_k3_cells_pinned_at_the_y_boundaries_of_the_paths(eng):
    """The wave-uniform paths also switch on the cell's Voigt y: centred lattice below 0.03, the
    plain lattice with its full pole term up to 1.3, the pole term to leading order in q from
    there to pi/h = 4.654, none above.  Every cell here sits within 0.5 % of one of those three
    values, on either side, and the 256 channels cover the line core (|x| <= 6, where the pole
    term is needed) on both wings.  Against the oracle (scipy.special.wofz) at the design bound."""
    from rajepy_amd import _lib
    from rajepy_amd.maths import rrls
    shape = (2, 48, 8)
    g = U.synth_host(shape, 515151, 0)
    temp_k = 9.0e3
    g["temp"][:] = temp_k
    g["vy"][:] = 6.2
    g["xi"][:] = 0.2
    lc = rrls.line_constants("H66a")
    nu_c = lc["nu_rest"] * (1.0 - 6.2 * 1000.0 / 299792458.0)
    sig2 = lc["kG"] * np.sqrt(temp_k) * nu_c / 2.0 / 1.1774100225154747 * np.sqrt(2.0)  # sigma sqrt 2
    rng = np.random.default_rng(7)
    ys = np.array([0.03, 1.3, np.pi / 0.675])
    y = ys[rng.integers(0, 3, size=shape)] * (1.0 + rng.uniform(-5e-3, 5e-3, size=shape))
    ne = 2.0 * y * sig2 / lc["kL"]                      # y = 0.5 kL ne / (sigma sqrt 2)
    g["nd"] = np.where(np.isnan(g["nd"]), np.nan, ne / 0.2)
    p = copy.deepcopy(U.load_golden("cfg1_example")[2])
    p["ejection"] = U.example_bursts_params()
    p["grid"].update(n_x=shape[0], n_y=shape[1], n_z=shape[2])
    jet = orc.OracleJet.from_fields(p, g["nd"], g["xi"], g["temp"], g["ff"], g["areas"],
                                    g["ts"], g["rr"], g["vy"])
    jet.time = 40.0 * YEAR                              # long after the last burst: chi = 1
    fields = eng.upload_fields(g["nd"], g["xi"], g["temp"], g["ff"], g["areas"], g["ts"],
                               g["rr"] < 0, vy=g["vy"], csize_au=jet.csize, dtype=8)
    xs = np.linspace(0.0, 6.0, 128)
    nu = np.concatenate([nu_c - xs[::-1] * sig2, nu_c + xs[1:] * sig2, [nu_c + 6.05 * sig2]])
    assert nu.size == 256
    line = _lib.Line(**lc)
    tau = eng.rrl_scan(fields, U.bursts_from_oracle(jet), jet.time, line, list(nu))
    eng.synchronize()
    ref = jet.optical_depth_rrl("H66a", nu)
    assert np.isfinite(ref).all() and (ref > 0).all()
    np.testing.assert_allclose(tau.cpu().numpy().reshape(ref.shape), ref, rtol=U.K3_RTOL_WAVE)


def test_k3_generic_path_and_odd_channel_lists(eng):
    """The wave-uniform K3 paths are chosen from the frequency range of a wave's channels, so
    the list may come in any order; a wave whose channels sit beside the line AND absurdly
    far from it (|x| > 1e6), and cells with infinite or zero fields, take the generic per-lane
    path with NumPy's NaN filter.  All against the oracle."""
    from rajepy_amd import _lib
    from rajepy_amd.maths import rrls
    shape = (2, 24, 16)
    seed = 20240524
    g = U.synth_host(shape, seed, 1)
    g["nd"][0, 3, 2] = np.inf           # n_e = inf
    g["temp"][0, 5, 3] = np.inf         # zero line width ...
    g["temp"][1, 7, 4] = 0.0            # ... and infinite
    g["xi"][1, 2, 5] = 0.0              # no electrons: y = 0, contributes exactly nothing
    g["ff"][1, 9, 6] = np.inf
    p = copy.deepcopy(U.load_golden("cfg1_example")[2])
    p["ejection"] = U.example_bursts_params()
    p["power_laws"]["q_T"] = -0.5
    p["grid"].update(n_x=shape[0], n_y=shape[1], n_z=shape[2])
    jet = orc.OracleJet.from_fields(p, g["nd"], g["xi"], g["temp"], g["ff"], g["areas"],
                                    g["ts"], g["rr"], g["vy"])
    jet.time = 0.6 * YEAR
    fields = eng.upload_fields(g["nd"], g["xi"], g["temp"], g["ff"], g["areas"], g["ts"],
                               g["rr"] < 0, vy=g["vy"], csize_au=jet.csize, dtype=8)
    nu0 = rrls.rrl_nu_0("H", 66, 1)
    band = orc.chan_freqs(nu0, 120 * 2e5, 2e5)
    rng = np.random.default_rng(4)
    lists = {"shuffled": rng.permutation(band),
             "with far outliers": np.concatenate([band[:70], [nu0 + 5e13, nu0 + 2e13], band[70:]]),
             "descending": band[::-1].copy()}
    line = _lib.Line(**rrls.line_constants("H66a"))
    with np.errstate(all="ignore"):
        for name, rf in lists.items():
            tau = eng.rrl_scan(fields, U.bursts_from_oracle(jet), jet.time, line, rf)
            eng.synchronize()
            ref = jet.optical_depth_rrl("H66a", np.asarray(rf))
            got = tau.cpu().numpy().reshape(ref.shape)
            assert np.array_equal(np.isnan(got), np.isnan(ref)), name
            assert np.array_equal(np.isinf(got), np.isinf(ref)), name
            ok = np.isfinite(ref)
            np.testing.assert_allclose(got[ok], ref[ok], rtol=U.k3_rtol(len(rf)), err_msg=name)


def _pipeline_worker(rank, world, port, dcy, ret):
    import torch.distributed as dist
    from rajepy_amd import classes, fits, logger
    from tests.test_host_logic import example_params, pline_params
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RJP_DEVICE="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        log = logger.Log(os.path.join(dcy, "rank%d.log" % rank), verbose=False)
        pp = pline_params(dcy)
        pp["continuum"]["times"] = np.array([0., 0.5, 1., 2., 3.])
        pl = classes.Pipeline(classes.JetModel(example_params(), log=log), pp, log=log)
        pl.execute(simobserve=False, verbose=False, dryrun=False, resume=False, clobber=True)
        ok = all(r.completed for r in pl.runs) and all("flux" in r.results for r in pl.runs)
        dist.barrier()
        if rank == 0:
            jm = classes.JetModel(example_params(), log=log)
            for r in pl.runs:
                jm.time = r.year * YEAR
                for path in (r.fits_em, r.fits_tau, r.fits_flux):
                    ok = ok and os.path.exists(path)
                if r.obs_type == "continuum":
                    want = np.nansum(np.nanmean(jm.flux_ff(r.chan_freqs), axis=0))
                    ok = ok and np.isclose(r.results["flux"], want, rtol=1e-12)
                    data = fits.read(r.fits_flux)[0]
                    ok = ok and np.isclose(np.nansum(np.nanmean(data, axis=0)), want, rtol=1e-12)
            ok = ok and os.path.exists(os.path.join(dcy, "pipeline.save"))
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_pipeline_execute_on_two_ranks_sharing_the_gpu(tmp_path):
    """Pipeline.execute inside a 2-rank process group (gloo; both ranks on the one GPU of the
    box): the epochs of the run table are dealt to the ranks, each writes the products of its
    runs, results are exchanged and rank 0 writes the state files; every run's flux equals
    what a single model computes."""
    import torch.multiprocessing as mp
    dcy = str(tmp_path / "out")
    os.makedirs(dcy)
    ret = mp.Manager().dict()
    mp.spawn(_pipeline_worker, args=(2, _free_port(), dcy, ret), nprocs=2, join=True)
    assert ret[0] is True and ret[1] is True, dict(ret)


@pytest.mark.parametrize("n_ep,want_em", [(32, False), (32, True), (16, True), (41, False)])
@pytest.mark.parametrize("compact", [True, False])
@pytest.mark.parametrize("shape", [(5, 45, 38), (2, 21, 256)])
def test_lds_dma_tile_kernel_equals_the_register_path_bit_for_bit(eng, n_ep, want_em, compact,
                                                                  shape):
    """Tiles of >= 16 uniformly spaced epochs prefetch their rows by LDS-DMA when the f64
    fields are 16-byte aligned and n_z is even (ff_scan_tile_kernel); fields that start 8
    bytes off take the register-load kernel.  Same arithmetic in the same order: the maps
    must be IDENTICAL -- odd row counts (a half-used last request), a partial last workgroup,
    NaN cells, occupied y-ranges attached, both field layouts.  The first shape has 190
    sightlines (one partial workgroup, every wave straddles the red/blue plane: the
    three-operation recurrence); the second has waves that lie inside one jet (the
    two-operation recurrence with its step table in SGPRs)."""
    import torch
    from rajepy_amd import engine as E
    jet, p = _jet(shape, 20240521, 0, 0., U.example_bursts_params())
    g = U.synth_host(shape, 20240521, 0)
    rng = np.random.default_rng(5)
    g["nd"].ravel()[rng.random(g["nd"].size) < 0.2] = np.nan
    g["nd"][:, :7, :] = np.nan                 # empty leading rows: y-ranges clip them
    fields = eng.upload_fields(g["nd"], g["xi"], g["temp"], g["ff"], g["areas"], g["ts"],
                               g["rr"] < 0, csize_au=jet.csize, dtype=8)
    eng.compute_y_bounds(fields)
    if not compact:
        fields.em0 = None
    assert all(t is None or t.data_ptr() % 16 == 0
               for t in (fields.nd, fields.xi, fields.temp, fields.pf, fields.ts, fields.em0))

    def shifted(t):
        if t is None:
            return None
        buf = torch.empty(t.numel() + 3, dtype=t.dtype, device=t.device)
        k = 1 if buf.data_ptr() % 16 == 0 else 2          # start 8 bytes off a 16-byte line
        v = buf[k:k + t.numel()]
        v.copy_(t.reshape(-1))
        assert v.data_ptr() % 16 == 8
        return v

    off = E.DeviceFields(shape, 8, fields.csize_au, shifted(fields.nd), shifted(fields.xi),
                         shifted(fields.temp), shifted(fields.pf), shifted(fields.ts))
    off.em0 = shifted(fields.em0)
    off.ylo, off.yhi = fields.ylo, fields.yhi
    bursts = U.bursts_from_oracle(jet)
    ep = [float(y) * YEAR for y in np.linspace(0., 4., n_ep)]
    a1, e1, t1 = (x.clone() if x is not None else None
                  for x in eng.ff_scan(fields, bursts, ep, E.RJP_GFF_SCALAR, want_em=want_em))
    a0, e0, t0 = eng.ff_scan(off, bursts, ep, E.RJP_GFF_SCALAR, want_em=want_em)
    eng.synchronize()
    assert np.array_equal(a1.cpu().numpy(), a0.cpu().numpy())
    assert np.array_equal(t1.cpu().numpy(), t0.cpu().numpy(), equal_nan=True)
    if want_em:
        assert np.array_equal(e1.cpu().numpy(), e0.cpu().numpy())
    # and both follow the oracle
    ctau, _ = E.ff_channel_coeffs([5e9], jet.csize, p["target"]["dist"], E.RJP_GFF_SCALAR,
                                  [orc.gff(5e9, p["properties"]["T_0"])])
    jet2 = orc.OracleJet.from_fields(p, g["nd"], g["xi"], g["temp"], g["ff"], g["areas"],
                                     g["ts"], g["rr"], g["vy"])
    got = a1.cpu().numpy().reshape(n_ep, shape[0], shape[2]) * ctau[0]
    with np.errstate(all="ignore"):
        for e in (0, n_ep // 2, n_ep - 1):
            jet2.time = ep[e]
            np.testing.assert_allclose(got[e], jet2.optical_depth_ff(5e9), rtol=1e-11)


def test_field_builder_mask_on_a_grid_where_most_cells_take_the_early_out(eng):
    """The example jet on 100 x 800 x 100 cells (0.4 % filled): the builder skips the
    eight-vertex test for cells a conservative bound puts outside the jet; fill factor and
    areas must still equal the oracle builder's (the reference's vertex test) cell for cell."""
    from rajepy_amd.classes import geometry_struct
    p = copy.deepcopy(U.load_golden("cfg1_example")[2])
    p["grid"].update(n_x=100, n_y=800, n_z=100)
    jet = orc.OracleJet(p)
    geom = geometry_struct(jet.params, jet.nx, jet.ny, jet.nz)
    f = eng.build_fields(geom, 8, want_ts=False, want_vxz=False)
    eng.synchronize()
    ff = f.ff_raw.cpu().numpy().reshape(jet.nx, jet.ny, jet.nz)
    ar = f.areas_raw.cpu().numpy().reshape(ff.shape)
    ref_ff, ref_ar = jet.fill_factor, jet.areas
    assert np.array_equal(np.isnan(ff), np.isnan(ref_ff))
    assert np.array_equal(np.nan_to_num(ff), np.nan_to_num(ref_ff))
    assert np.array_equal(np.nan_to_num(ar), np.nan_to_num(ref_ar))
    filled = np.isfinite(ref_ff).mean()
    assert 0.001 < filled < 0.02, filled
