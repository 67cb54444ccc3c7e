"""Full-size GPU checks at BASELINE.json's grid sizes and channel / epoch counts (cfg2
256x1024x256 x 32 ch; cfg3 512x2048x512 x 256 H66a channels; cfg4 512x4096x512 continuum;
cfg5 512x4096x512 x 64 ch x 32 epochs), so that every kernel instantiation a bench
configuration dispatches is checked at the size it runs at.  The oracle cannot run a billion cells, so parity is checked (a) EXACTLY on a random
sample of sightlines -- sightlines are independent, so the sampled columns are regenerated on
the host from the counter hash, stacked as a (k, n_y, 1) grid and run through the oracle --
and (b) through size-independent properties: an 8-epoch fused pass equals eight single-epoch
passes bit for bit, the compact scan layout equals the wide one bit for bit, the on-device
flux-vs-time reduction equals the sum of the flux cube."""
import copy

import numpy as np
import pytest

from oracle import rt_oracle as orc
from tests import gpu_util as U

pytestmark = pytest.mark.gpu
SEED = 20240504


@pytest.fixture(scope="module")
def eng():
    from rajepy_amd.engine import RTEngine
    e = RTEngine(0)
    e.cache_moments = False        # (paths are asserted sweep by sweep; the cache: test_gpu_moments)
    yield e
    e.close()


def _sample_jet(shape, pix, temp_mode, q_T, seed=SEED):
    nx, ny, nz = shape
    cells = np.array([(x * ny + y) * nz + z for (x, z) in pix for y in range(ny)],
                     dtype=np.uint64)
    g = U.synth_host((len(pix), ny, 1), seed, temp_mode, cells=cells, nz_full=nz)
    p = copy.deepcopy(U.load_golden("cfg1_example")[2])
    p["ejection"] = U.example_bursts_params()
    p["power_laws"]["q_T"] = q_T
    p["grid"].update(n_x=len(pix), n_y=ny, n_z=1)
    return orc.OracleJet.from_fields(p, g["nd"], g["xi"], g["temp"], g["ff"], g["areas"],
                                     g["ts"], g["rr"], g["vy"])


@pytest.mark.parametrize("dtype,tol", [(8, 1e-10), (4, 1e-5)])
def test_cfg4_continuum_full_size(eng, dtype, tol):
    from rajepy_amd import engine as E
    from rajepy_amd.maths import physics as ph
    shape = (512, 4096, 512)
    nx, ny, nz = shape
    fields = eng.synth_fields(shape, SEED, 0, dtype, csize_au=0.5)
    rng = np.random.default_rng(5)
    pix = [(int(rng.integers(nx)), int(rng.integers(nz))) for _ in range(20)]
    pix += [(0, 0), (nx - 1, nz - 1), (17, nz // 2 - 1), (17, nz // 2)]   # corners, jet boundary
    jet = _sample_jet(shape, pix, 0, 0.)
    bursts = U.bursts_from_oracle(jet)
    years = [0.3, 0.9, 1.0, 1.7, 2.2, 2.9, 3.6, 4.4]
    ep = [y * orc.YEAR for y in years]
    freqs = np.array([1e9, 5e9, 2.2e10, 5e10])
    gv = [ph.gff(nu, 1e4) for nu in freqs]
    ctau, cflux = E.ff_channel_coeffs(freqs, 0.5, 120., E.RJP_GFF_SCALAR, gv)

    # (b1) one fused 8-epoch pass == eight single passes, bit for bit
    sumA8, em8, tavg = eng.ff_scan(fields, bursts, ep, E.RJP_GFF_SCALAR)
    for e in (0, 3, 7):
        s1, e1, _ = eng.ff_scan(fields, bursts, [ep[e]], E.RJP_GFF_SCALAR)
        eng.synchronize()
        assert bool((s1[0] == sumA8[e]).all()) and bool((e1[0] == em8[e]).all())

    # (b1') the compact 3-field layout == the wide 5-field layout over all 1.07e9 cells: bit
    # for bit in f64, to the extra float rounding of the product in f32
    assert fields.em0 is not None
    em0, fields.em0 = fields.em0, None
    sw, ew, tw = eng.ff_scan(fields, bursts, ep[:2], E.RJP_GFF_SCALAR)
    eng.synchronize()
    fields.em0 = em0
    if dtype == 8:
        assert bool((sw == sumA8[:2]).all()) and bool((ew == em8[:2]).all())
    else:
        assert float(((sw - sumA8[:2]).abs() / sumA8[:2]).max().item()) < 2e-7
    assert bool((tw == tavg).all())
    del sw, ew, tw

    # (b2) device flux-vs-time reduction == sum of the flux cube
    tau, flux, ftot = eng.ff_maps(sumA8[:2].contiguous(), tavg, ctau, cflux)
    eng.synchronize()
    np.testing.assert_allclose(ftot.cpu().numpy(), flux.nansum(dim=2).cpu().numpy(), rtol=1e-12)

    # (a) sampled sightlines == oracle
    idx = [x * nz + z for (x, z) in pix]
    tau_s = tau.cpu().numpy()[:, :, idx]
    flux_s = flux.cpu().numpy()[:, :, idx]
    em_s = em8.cpu().numpy()[:, idx]
    for e in range(2):
        jet.time = ep[e]
        np.testing.assert_allclose(tau_s[e], jet.optical_depth_ff(freqs)[:, :, 0], rtol=tol)
        np.testing.assert_allclose(flux_s[e], jet.flux_ff(freqs)[:, :, 0], rtol=tol)
        np.testing.assert_allclose(em_s[e], jet.emission_measure()[:, 0], rtol=tol)
    # every sightline of the dense set is optically relevant: no zeros, no NaNs
    assert bool(torch_all_finite(sumA8)) and float(sumA8.min().item()) > 0.0
    del tau, flux

    # (a') K2 exactly as the bench launches it: ONE epoch x the 256 channels of cfg4 (the
    # channel axis is cut into 4 chunks of 64 over gridDim.z): the first and the last channel
    # of the first, a middle and the last chunk against the oracle on the sampled sightlines
    # (classes.py:1395-1397, 1473-1475, 1519-1521)
    f256 = np.geomspace(1e9, 5e10, 256)
    ct, cf = E.ff_channel_coeffs(f256, 0.5, 120., E.RJP_GFF_SCALAR, [ph.gff(nu, 1e4) for nu in f256])
    tau, flux, ftot = eng.ff_maps(sumA8[1:2].contiguous(), tavg, ct, cf)
    eng.synchronize()
    chk = [0, 63, 64, 100, 191, 192, 255]
    jet.time = ep[1]
    np.testing.assert_allclose(tau[0][chk][:, idx].cpu().numpy(),
                               jet.optical_depth_ff(f256[chk])[:, :, 0], rtol=tol)
    np.testing.assert_allclose(flux[0][chk][:, idx].cpu().numpy(),
                               jet.flux_ff(f256[chk])[:, :, 0], rtol=tol)
    np.testing.assert_allclose(ftot.cpu().numpy(), flux.nansum(dim=2).cpu().numpy(), rtol=1e-12)


@pytest.mark.parametrize("temp_mode", [0, 1])
def test_cfg4_tau_layout_is_bit_identical_at_full_size(eng, temp_mode):
    """512x4096x512: the tau layout (a0, ts: 16 B/cell; + em0 with EM maps) against the compact
    (3-field) and wide (5-field) scans of the same 1.07e9 cells, bit for bit over every
    sightline -- single epoch as the bench launches it (no EM, no T_avg), single epoch with EM,
    an 8-epoch direct tile and the 32-epoch uniform tile; T_avg from rjp_tavg == the scans'.
    Both Gaunt branches (classes.py:1388-1397, 1421-1429)."""
    import torch
    from rajepy_amd import engine as E
    shape = (512, 4096, 512)
    mode = E.RJP_GFF_SCALAR if temp_mode == 0 else E.RJP_GFF_POWERLAW
    fields = eng.synth_fields(shape, SEED, temp_mode, 8, csize_au=0.5, tau_mode=mode)
    assert fields.a0 is not None and fields.scan_fields(mode, False) == 2
    eng.use_moments = False        # the epoch TILES are what is compared bit for bit here
    eng.use_chi_table = False      # ... with the Gaussians (the table path: its own test below)
    jet = _sample_jet(shape, [(0, 0)], temp_mode, 0. if temp_mode == 0 else -0.5)
    bursts = U.bursts_from_oracle(jet)
    e1 = [1.0 * orc.YEAR]
    e8 = [y * orc.YEAR for y in (0.3, 0.9, 1.0, 1.7, 2.2, 2.9, 3.6, 4.4)]
    e32 = [float(t) for t in np.linspace(0., 5., 32) * orc.YEAR]
    cases = [(e1, False), (e1, True), (e8, False), (e32, False)]
    if temp_mode == 0:
        cases.append((e32, True))

    def run():
        out = []
        for ep, em in cases:
            a, g, _ = eng.ff_scan(fields, bursts, ep, mode, want_em=em, want_tavg=False)
            out.append((a.clone(), None if g is None else g.clone()))
        return out
    got = run()
    tav = eng.tavg(fields).clone()
    a0, fields.a0 = fields.a0, None                      # compact layout
    cmp_ = run()
    t_cmp = eng.ff_scan(fields, bursts, e1, mode)[2].clone()
    em0, fields.em0 = fields.em0, None                   # wide layout
    wide = [eng.ff_scan(fields, bursts, ep, mode, want_em=em) for ep, em in cases[:2]]
    eng.synchronize()
    for (a, g), (ac, gc) in zip(got, cmp_):
        assert torch.equal(a, ac)
        assert (g is None) == (gc is None) and (g is None or torch.equal(g, gc))
    for (a, g), (aw, gw, tw) in zip(got[:2], wide):
        assert torch.equal(a, aw) and (g is None or torch.equal(g, gw))
        assert torch.equal(tav, tw)
    assert torch.equal(tav, t_cmp)
    fields.em0, fields.a0 = em0, a0
    eng.use_moments = True
    eng.use_chi_table = True
    # round 4: the shipped single-epoch scan takes the burst factor from a table in LDS and sums
    # each sightline in ONE y-range: equal to the Gaussian scan's map to rounding (its bound on
    # chi^2 is 2e-13; the other order of the sum adds ~1e-14), exactly zero where that is zero
    tab = eng.ff_scan(fields, bursts, e1, mode, want_em=False, want_tavg=False)[0]
    assert eng.last_scan_path()[0] == "table"
    eng.synchronize()
    ref = got[0][0]
    assert torch.equal(tab == 0, ref == 0)
    assert ((tab - ref).abs() / ref).max().item() < 3e-12
    # round 5: the two HEADLINE kernels against the ORACLE at the headline
    # size, in one assertion each -- ff_scan_table_kernel (a0, ts: what the timed step launches)
    # and ff_scan_table_wide_kernel (the five model fields: SURVEY 8(d)'s byte model, tau + EM +
    # T_avg in one pass) on 28 sampled sightlines incl. the corners of the map, both sides of the
    # jet plane and the first / last lanes of a workgroup (classes.py:1101-1128, 1388-1432,
    # 1471-1472); both Gaunt branches through the parametrisation
    from rajepy_amd.maths import physics as ph
    nx, ny, nz = shape
    rng = np.random.default_rng(41 + temp_mode)
    pix = [(int(rng.integers(nx)), int(rng.integers(nz))) for _ in range(20)]
    pix += [(0, 0), (nx - 1, nz - 1), (17, nz // 2 - 1), (17, nz // 2), (0, 510), (1, 0),
            (255, 511), (256, 0)]
    idx = [x * nz + z for (x, z) in pix]
    sj = _sample_jet(shape, pix, temp_mode, 0. if temp_mode == 0 else -0.5)
    sj.time = e1[0]
    nu = np.array([1e9, 5e10])
    gv = [ph.gff(f, 1e4) for f in nu] if temp_mode == 0 else None
    ctau, _ = E.ff_channel_coeffs(nu, 0.5, 120., mode, gv)
    want_tau = sj.optical_depth_ff(nu)[:, :, 0]
    np.testing.assert_allclose(ctau[:, None] * tab[0].cpu().numpy()[idx][None, :], want_tau,
                               rtol=1e-10, err_msg="ff_scan_table_kernel vs oracle")
    a0, em0, fields.a0, fields.em0 = fields.a0, fields.em0, None, None     # the five model fields
    wA, wE, wT = eng.ff_scan(fields, bursts, e1, mode, want_em=True, want_tavg=True)
    assert eng.last_scan_path()[0] == "table"
    eng.synchronize()
    fields.a0, fields.em0 = a0, em0
    np.testing.assert_allclose(ctau[:, None] * wA[0].cpu().numpy()[idx][None, :], want_tau,
                               rtol=1e-10, err_msg="ff_scan_table_wide_kernel vs oracle (tau)")
    np.testing.assert_allclose(wE[0].cpu().numpy()[idx], sj.emission_measure()[:, 0], rtol=1e-10,
                               err_msg="ff_scan_table_wide_kernel vs oracle (EM)")
    # (T_avg of the same pass: one y-range here, eight in rjp_tavg -- equal to rounding)
    assert ((wT - tav).abs() / tav).max().item() < 1e-13


def test_cfg5_epoch_sweep_by_launch_time_moments_full_size(eng):
    """BASELINE configs[4] as benchmarked since round 3: 512x4096x512, 32 uniformly spaced
    epochs, no EM maps, tau layout -> the moment path (one pass over the grid + one
    contraction).  Sampled sightlines (incl. the edges of the first / last 16-sightline groups
    and both sides of the jet plane) against the oracle at epochs 0 / 15 / 31; the whole
    [32, P] result against the 32-epoch tile of the same sweep at 5e-11; and 40 irregularly
    spaced epochs (two contraction passes) against the oracle."""
    from rajepy_amd import engine as E
    from rajepy_amd.maths import physics as ph
    shape = (512, 4096, 512)
    nx, ny, nz = shape
    fields = eng.synth_fields(shape, SEED, 0, 8, csize_au=0.5, tau_mode=E.RJP_GFF_SCALAR)
    rng = np.random.default_rng(17)
    pix = [(int(rng.integers(nx)), int(rng.integers(nz))) for _ in range(10)]
    pix += _edge_pixels(nx, nz, 16)
    jet = _sample_jet(shape, pix, 0, 0.)
    bursts = U.bursts_from_oracle(jet)
    idx = [x * nz + z for (x, z) in pix]
    ctau, _ = E.ff_channel_coeffs([5e9], 0.5, 120., E.RJP_GFF_SCALAR, [ph.gff(5e9, 1e4)])
    ep = [float(t) for t in np.linspace(0., 5., 32) * orc.YEAR]
    eng.use_moments = True
    mom = eng.ff_scan(fields, bursts, ep, E.RJP_GFF_SCALAR, want_em=False, want_tavg=False)[0].clone()
    path, err = eng.last_scan_path()
    assert path == "moments" and err <= 1e-11
    eng.use_moments = False
    til = eng.ff_scan(fields, bursts, ep, E.RJP_GFF_SCALAR, want_em=False, want_tavg=False)[0]
    eng.use_moments = True
    eng.synchronize()
    rel = ((mom - til).abs() / til).max().item()
    assert rel < 5e-11, rel
    m_s = mom.cpu().numpy()[:, idx]
    for e in (0, 15, 31):
        jet.time = ep[e]
        np.testing.assert_allclose(ctau[0] * m_s[e], jet.optical_depth_ff(5e9)[:, 0], rtol=1e-10)
    # round 4: the same sweep on the launch-time-ordered layout (register moments fused with the
    # contraction): the whole [32, P] result against the LDS moments and the tiles at 5e-11, the
    # sampled sightlines against the oracle, two sweeps of one layout bit for bit
    import torch
    lt = eng.build_lt(fields, 32)
    assert lt["rows"] * 64 < 1.3 * nx * ny * nz                   # padding: 1.22 x on this grid
    ltr = eng.ff_scan(fields, bursts, ep, E.RJP_GFF_SCALAR, want_em=False, want_tavg=False)[0].clone()
    path, err = eng.last_scan_path()
    assert path == "lt" and err <= 1e-11 and eng.last_moment_shape[0] == 32
    ltr2 = eng.ff_scan(fields, bursts, ep, E.RJP_GFF_SCALAR, want_em=False, want_tavg=False)[0]
    eng.synchronize()
    assert torch.equal(ltr.view(torch.int64), ltr2.view(torch.int64))
    assert ((ltr - til).abs() / til).max().item() < 5e-11
    assert ((ltr - mom).abs() / mom).max().item() < 5e-11
    l_s = ltr.cpu().numpy()[:, idx]
    for e in (0, 15, 31):
        jet.time = ep[e]
        np.testing.assert_allclose(ctau[0] * l_s[e], jet.optical_depth_ff(5e9)[:, 0], rtol=1e-10)
    fields.lt = None
    del mom, til, ltr, ltr2, lt
    ep2 = sorted(float(t) for t in rng.uniform(0., 5., 40) * orc.YEAR)
    mom2 = eng.ff_scan(fields, bursts, ep2, E.RJP_GFF_SCALAR, want_em=False, want_tavg=False)[0]
    assert eng.last_scan_path()[0] == "moments"
    m2 = mom2.cpu().numpy()[:, idx]
    for e in (0, 21, 39):
        jet.time = ep2[e]
        np.testing.assert_allclose(ctau[0] * m2[e], jet.optical_depth_ff(5e9)[:, 0], rtol=1e-10)


@pytest.mark.parametrize("temp_mode", [0, 1])
@pytest.mark.parametrize("shape", [(128, 96, 256), (64, 131, 512), (32, 77, 1030)])
def test_single_epoch_table_scan_vs_gaussians_and_oracle(eng, temp_mode, shape):
    """The single-epoch tau-layout scan with the burst factor from a table in LDS
    (ff_scan_tab.hip) on maps large enough to take it, one and several y-ranges, an odd number of
    rows and a map whose width is no multiple of the workgroup: against the Gaussian scan at
    3e-12 (the table's bound is 2e-13 on chi^2), sampled sightlines against the oracle at 1e-11,
    NaN launch times / weights masked alike, epochs inside, at the edge of and far outside the
    bursts' support, occupied y-ranges honoured; bursts with a dip keep the Gaussians."""
    import torch
    from rajepy_amd import engine as E
    from rajepy_amd.maths import physics as ph
    nx, ny, nz = shape
    mode = E.RJP_GFF_SCALAR if temp_mode == 0 else E.RJP_GFF_POWERLAW
    fields = eng.synth_fields(shape, SEED + 5, temp_mode, 8, csize_au=0.5, tau_mode=mode)
    # NaN sprinkles in the launch times and the weights (a device-side edit of the fields)
    g = torch.Generator(device=eng.device)
    g.manual_seed(5)
    m_ts = torch.rand(fields.ncells, device=eng.device, generator=g) < 0.02
    m_a = torch.rand(fields.ncells, device=eng.device, generator=g) < 0.02
    fields.ts[m_ts] = float("nan")
    fields.a0[m_a] = float("nan")
    fields.em0[m_a] = float("nan")           # (the oracle's cells lose their density there)
    rng = np.random.default_rng(3)
    pix = [(int(rng.integers(nx)), int(rng.integers(nz))) for _ in range(6)] + [(0, 0), (nx - 1, nz - 1)]
    jet = _sample_jet(shape, pix, temp_mode, 0. if temp_mode == 0 else -0.5, seed=SEED + 5)
    idx = [x * nz + z for (x, z) in pix]
    # the sampled oracle cells get the same NaNs
    flat = [np.ravel_multi_index((x, np.arange(ny), z), shape) for (x, z) in pix]
    mts, ma = m_ts.cpu().numpy(), m_a.cpu().numpy()
    for k, fl in enumerate(flat):
        jet._ts[k, mts[fl], 0] = np.nan
        jet._nd[k, ma[fl], 0] = np.nan
    bursts = U.bursts_from_oracle(jet)
    gv = [ph.gff(5e9, 1e4)] if temp_mode == 0 else None
    ctau, _ = E.ff_channel_coeffs([5e9], 0.5, 120., mode, gv)
    for years in (1.0, 0.0, 2.6, 40.0, -30.0):
        ep = [years * orc.YEAR]
        eng.use_chi_table = True
        tab = eng.ff_scan(fields, bursts, ep, mode, want_em=False, want_tavg=False)[0].clone()
        assert eng.last_scan_path()[0] == "table"
        eng.use_chi_table = False
        ref = eng.ff_scan(fields, bursts, ep, mode, want_em=False, want_tavg=False)[0].clone()
        assert eng.last_scan_path()[0] == "tiles"
        eng.use_chi_table = True
        eng.synchronize()
        assert torch.equal(tab == 0, ref == 0)
        ok = ref != 0
        # (the Gaussians' own 2^f polynomial is good to 1.1e-12 on chi, rjp_device.h exp2_gauss:
        # the two scans may differ by 2.2e-12 + the table's 2e-13)
        assert ((tab - ref).abs()[ok] / ref[ok]).max().item() < 3e-12
        jet.time = ep[0]
        np.testing.assert_allclose(ctau[0] * tab.cpu().numpy()[0, idx],
                                   jet.optical_depth_ff(5e9)[:, 0], rtol=1e-11)
    # occupied y-ranges: the same map
    eng.compute_y_bounds(fields)
    yb = eng.ff_scan(fields, bursts, [1.0 * orc.YEAR], mode, want_em=False, want_tavg=False)[0]
    assert eng.last_scan_path()[0] == "table"
    fields.ylo = fields.yhi = None
    full = eng.ff_scan(fields, bursts, [1.0 * orc.YEAR], mode, want_em=False, want_tavg=False)[0]
    eng.synchronize()
    assert ((yb - full).abs() / full).max().item() < 1e-13
    # a dip (negative amplitude): chi may come close to 0, the bound is on chi itself -> Gaussians
    dip = E.make_bursts([(1.0 * orc.YEAR, -0.5, 2e6)], [(1.2 * orc.YEAR, 3.0, 4e6)])
    eng.ff_scan(fields, dip, [1.0 * orc.YEAR], mode, want_em=False, want_tavg=False)
    assert eng.last_scan_path()[0] == "tiles"
    # with the emission-measure map of the epoch (a third stream, em0): table path as well
    ep = [1.3 * orc.YEAR]
    a_t, e_t, _ = eng.ff_scan(fields, bursts, ep, mode, want_em=True, want_tavg=False)
    assert eng.last_scan_path()[0] == "table"
    a_t, e_t = a_t.clone(), e_t.clone()
    eng.use_chi_table = False
    a_g, e_g, _ = eng.ff_scan(fields, bursts, ep, mode, want_em=True, want_tavg=False)
    assert eng.last_scan_path()[0] == "tiles"
    eng.use_chi_table = True
    eng.synchronize()
    for got, ref in ((a_t, a_g), (e_t, e_g)):
        assert torch.equal(got == 0, ref == 0)
        ok = ref != 0
        assert ((got - ref).abs()[ok] / ref[ok]).max().item() < 3e-12
    jet.time = ep[0]
    np.testing.assert_allclose(e_t.cpu().numpy()[0, idx], jet.emission_measure()[:, 0], rtol=1e-11)
    # T_avg asked for WITH the scan: the caller is served by the ordinary kernels
    eng.ff_scan(fields, bursts, [1.0 * orc.YEAR], mode, want_em=False, want_tavg=True)
    assert eng.last_scan_path()[0] == "tiles"
    # the five MODEL fields alone (no a0 / em0: SURVEY 8(d)'s byte model): the wide table scan
    # gives the optical-depth sums, the EM and T_avg of the epoch in one pass
    a0, em0 = fields.a0, fields.em0
    fields.a0 = fields.em0 = None
    fields.nd[m_a] = float("nan")                # (the weights' NaNs, on the wide fields)
    w_t = [t.clone() for t in eng.ff_scan(fields, bursts, ep, mode, want_em=True, want_tavg=True)]
    assert eng.last_scan_path()[0] == "table"
    eng.use_chi_table = False
    w_g = [t.clone() for t in eng.ff_scan(fields, bursts, ep, mode, want_em=True, want_tavg=True)]
    assert eng.last_scan_path()[0] == "tiles"
    eng.use_chi_table = True
    eng.synchronize()
    for got, ref in zip(w_t[:2], w_g[:2]):
        assert torch.equal(got == 0, ref == 0)
        ok = ref != 0
        assert ((got - ref).abs()[ok] / ref[ok]).max().item() < 3e-12
    assert torch.equal(torch.isnan(w_t[2]), torch.isnan(w_g[2]))
    okt = ~torch.isnan(w_g[2])
    assert ((w_t[2] - w_g[2]).abs()[okt] / w_g[2][okt]).max().item() < 1e-13
    np.testing.assert_allclose(ctau[0] * w_t[0].cpu().numpy()[0, idx],
                               jet.optical_depth_ff(5e9)[:, 0], rtol=1e-11)
    np.testing.assert_allclose(w_t[1].cpu().numpy()[0, idx], jet.emission_measure()[:, 0], rtol=1e-11)
    fields.a0, fields.em0 = a0, em0

def torch_all_finite(t):
    import torch
    return torch.isfinite(t).all().item()


def test_cfg3_rrl_full_size(eng):
    from rajepy_amd import _lib
    from rajepy_amd.maths import rrls
    shape = (512, 2048, 512)
    nx, ny, nz = shape
    fields = eng.synth_fields(shape, SEED, 1, 8, csize_au=0.5, with_vy=True)
    rng = np.random.default_rng(11)
    pix = [(int(rng.integers(nx)), int(rng.integers(nz))) for _ in range(10)] + [(3, nz // 2)]
    jet = _sample_jet(shape, pix, 1, -0.5)
    jet.time = 1.2 * orc.YEAR
    nchan = 40
    nu0 = rrls.rrl_nu_0("H", 66, 1)
    rf = orc.chan_freqs(nu0, nchan * 6e5, 6e5)            # +-12 MHz: core, wings and far field
    line = _lib.Line(**rrls.line_constants("H66a"))
    tau = eng.rrl_scan(fields, U.bursts_from_oracle(jet), jet.time, line, rf)
    eng.synchronize()
    idx = [x * nz + z for (x, z) in pix]
    got = tau.cpu().numpy()[:, idx]
    ref = jet.optical_depth_rrl("H66a", np.asarray(rf))[:, :, 0]
    np.testing.assert_allclose(got, ref, rtol=U.K3_RTOL_WAVE)
    assert bool(torch_all_finite(tau))


def _edge_pixels(nx, nz, tile):
    """Sightlines the random sample would miss by construction: the first and the last
    workgroup of the scan kernels (K1: 256 lanes x VEC sightlines = whole x-rows; K3: `tile`
    z-adjacent sightlines per workgroup), both sides of the red/blue jet plane (waves that
    straddle it select burst parameters per lane) and both sides of a tile boundary.  Every
    sightline crosses every y-split boundary (a split is a y-range of all sightlines)."""
    return [(0, 0), (0, tile - 1), (0, tile), (0, nz - 1), (nx - 1, 0), (nx - 1, nz - tile),
            (nx - 1, nz - tile - 1), (nx - 1, nz - 1), (nx // 2, nz // 2 - 1),
            (nx // 2, nz // 2), (nx - 1, nz // 2 - 1), (nx - 1, nz // 2)]


def test_cfg2_continuum_full_size(eng):
    """BASELINE configs[1]: 256x1024x256 grid, 32 continuum channels 1-50 GHz, one epoch --
    the two-sightlines-per-lane kernel with 16 y-splits of 64 rows.  Sampled sightlines
    against the oracle exactly; flux totals against the cube."""
    from rajepy_amd import engine as E
    from rajepy_amd.maths import physics as ph
    shape = (256, 1024, 256)
    nx, ny, nz = shape
    fields = eng.synth_fields(shape, SEED, 0, 8, csize_au=0.5)
    rng = np.random.default_rng(2)
    pix = [(int(rng.integers(nx)), int(rng.integers(nz))) for _ in range(16)]
    pix += _edge_pixels(nx, nz, 2)
    jet = _sample_jet(shape, pix, 0, 0.)
    jet.time = 1.0 * orc.YEAR
    bursts = U.bursts_from_oracle(jet)
    freqs = np.geomspace(1e9, 5e10, 32)
    gv = [ph.gff(nu, 1e4) for nu in freqs]
    ctau, cflux = E.ff_channel_coeffs(freqs, 0.5, 120., E.RJP_GFF_SCALAR, gv)
    sumA, em, tavg = eng.ff_scan(fields, bursts, [jet.time], E.RJP_GFF_SCALAR)
    tau, flux, ftot = eng.ff_maps(sumA, tavg, ctau, cflux)
    eng.synchronize()
    idx = [x * nz + z for (x, z) in pix]
    np.testing.assert_allclose(tau.cpu().numpy()[0][:, idx],
                               jet.optical_depth_ff(freqs)[:, :, 0], rtol=1e-10)
    np.testing.assert_allclose(flux.cpu().numpy()[0][:, idx], jet.flux_ff(freqs)[:, :, 0],
                               rtol=1e-10)
    np.testing.assert_allclose(em.cpu().numpy()[0][idx], jet.emission_measure()[:, 0],
                               rtol=1e-10)
    np.testing.assert_allclose(ftot.cpu().numpy(), flux.nansum(dim=2).cpu().numpy(), rtol=1e-12)
    assert bool(torch_all_finite(tau)) and float(tau.min().item()) > 0.0


def test_cfg5_epoch_sweep_full_size(eng):
    """BASELINE configs[4]: 512x4096x512, 64 channels x 32 uniformly spaced epochs,
    flux-vs-time output (no EM maps) -> ONE pass of the 32-epoch tile
    (ff_scan_kernel<double,1,32,..,EM=false>).  Sampled sightlines against the oracle at
    epochs 0/15/31; the whole [32, P] result against the same sweep done with EM maps, which
    takes two 16-epoch tiles anchored at other epochs (same numbers to rounding, 1e-11, not
    bit for bit: the recurrences start from different anchors); light curves against the
    sum of the flux maps."""
    from rajepy_amd import engine as E
    from rajepy_amd.maths import physics as ph
    shape = (512, 4096, 512)
    nx, ny, nz = shape
    fields = eng.synth_fields(shape, SEED, 0, 8, csize_au=0.5)
    rng = np.random.default_rng(7)
    pix = [(int(rng.integers(nx)), int(rng.integers(nz))) for _ in range(12)]
    pix += _edge_pixels(nx, nz, 1)
    jet = _sample_jet(shape, pix, 0, 0.)
    bursts = U.bursts_from_oracle(jet)
    ep = [float(t) for t in np.linspace(0., 5., 32) * orc.YEAR]
    freqs = np.geomspace(1e9, 5e10, 64)
    gv = [ph.gff(nu, 1e4) for nu in freqs]
    ctau, cflux = E.ff_channel_coeffs(freqs, 0.5, 120., E.RJP_GFF_SCALAR, gv)
    a32, none, tavg = eng.ff_scan(fields, bursts, ep, E.RJP_GFF_SCALAR, want_em=False)
    assert none is None
    _, _, ftot = eng.ff_maps(a32, tavg, ctau, cflux, want_tau=False, want_flux=False)
    eng.synchronize()
    assert tuple(ftot.shape) == (32, 64)
    idx = [x * nz + z for (x, z) in pix]
    a_s = a32.cpu().numpy()[:, idx]
    t_s = tavg.cpu().numpy()[idx]
    for e in (0, 15, 31):
        jet.time = ep[e]
        ref_tau = jet.optical_depth_ff(freqs)[:, :, 0]
        ref_flux = jet.flux_ff(freqs)[:, :, 0]
        np.testing.assert_allclose(ctau[:, None] * a_s[e][None, :], ref_tau, rtol=1e-10)
        got_flux = cflux[:, None] * (t_s[None, :] * (1. - np.exp(-ctau[:, None] * a_s[e][None, :])))
        np.testing.assert_allclose(got_flux, ref_flux, rtol=1e-10)
    # light curves == sums of the flux maps (three epochs; the cubes of all 32 would be 17 GB)
    for e in (0, 15, 31):
        _, fl, _ = eng.ff_maps(a32[e:e + 1].contiguous(), tavg, ctau, cflux, want_tau=False,
                               want_ftot=False)
        np.testing.assert_allclose(ftot[e].cpu().numpy(), fl[0].nansum(dim=1).cpu().numpy(),
                                   rtol=1e-12)
        del fl
    # the same sweep through the 16-epoch tiles (with EM maps)
    a32 = a32.clone()
    a16, em16, _ = eng.ff_scan(fields, bursts, ep, E.RJP_GFF_SCALAR, want_em=True)
    eng.synchronize()
    rel = ((a16 - a32).abs() / a32).max().item()
    assert rel < 1e-11, rel
    jet.time = ep[31]
    np.testing.assert_allclose(em16.cpu().numpy()[31][idx], jet.emission_measure()[:, 0],
                               rtol=1e-10)


def test_cfg3_rrl_256_channels_full_size(eng):
    """BASELINE configs[2] as benchmarked: 512x2048x512, 256 H66a channels of 100 kHz -> the
    256-lane kernel (rrl_scan_kernel<double,256,true>).  Sampled sightlines, including the
    first and last 8-sightline tiles, against the oracle (scipy's wofz)."""
    from rajepy_amd import _lib
    from rajepy_amd.maths import rrls
    shape = (512, 2048, 512)
    nx, ny, nz = shape
    fields = eng.synth_fields(shape, SEED, 0, 8, csize_au=0.5, with_vy=True)
    rng = np.random.default_rng(13)
    pix = [(int(rng.integers(nx)), int(rng.integers(nz))) for _ in range(6)]
    pix += _edge_pixels(nx, nz, 8)
    jet = _sample_jet(shape, pix, 0, 0.)
    jet.time = 1.0 * orc.YEAR
    nchan = 256
    nu0 = rrls.rrl_nu_0("H", 66, 1)
    rf = orc.chan_freqs(nu0, nchan * 1e5, 1e5)
    assert len(rf) == nchan
    line = _lib.Line(**rrls.line_constants("H66a"))
    tau = eng.rrl_scan(fields, U.bursts_from_oracle(jet), jet.time, line, rf)
    eng.synchronize()
    idx = [x * nz + z for (x, z) in pix]
    got = tau.cpu().numpy()[:, idx]
    ref = jet.optical_depth_rrl("H66a", np.asarray(rf))[:, :, 0]
    np.testing.assert_allclose(got, ref, rtol=U.K3_RTOL_WAVE)
    assert bool(torch_all_finite(tau))

    # the map stage of the same cube at the same size: continuum scan + K2 for the 256 channels
    # + rrl_maps, flux_rrl with and without the continuum (classes.py:1319-1343;
    # rrls.py:444-449) on the sampled sightlines, and the per-channel totals against the cube
    from rajepy_amd import engine as E
    from rajepy_amd.maths import physics as ph
    rfa = np.asarray(rf)
    gv = [ph.gff(nu, 1e4) for nu in rfa]
    ctau, cflux = E.ff_channel_coeffs(rfa, 0.5, 120., E.RJP_GFF_SCALAR, gv)
    cfl, hnu = E.rrl_channel_coeffs(rfa, 0.5, 120.)
    sumA, _, tavg = eng.ff_scan(fields, U.bursts_from_oracle(jet), [jet.time], E.RJP_GFF_SCALAR)
    tau_ff, flux_ff, _ = eng.ff_maps(sumA, tavg, ctau, cflux, want_ftot=False)
    P = nx * nz
    for contsub in (True, False):
        flux, ftot = eng.rrl_maps(tau, tau_ff.reshape(nchan, P), tavg,
                                  None if contsub else flux_ff.reshape(nchan, P), cfl, hnu)
        eng.synchronize()
        ref = jet.flux_rrl("H66a", rfa, contsub=contsub)[:, :, 0]
        np.testing.assert_allclose(flux.cpu().numpy()[:, idx], ref, rtol=U.K3_RTOL_WAVE)
        np.testing.assert_allclose(ftot.cpu().numpy(), flux.nansum(dim=1).cpu().numpy(),
                                   rtol=1e-12)
        del flux
