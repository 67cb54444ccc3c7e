#!/opt/conda/bin/python3.9
"""Generate golden input/output vectors for the line-of-sight RT path by importing the
UNMODIFIED reference from /root/reference.  Build-container only; never runs on the GPU box.

Run (from the repo root):

    PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 tests/golden/make_golden.py

Interpreter: /opt/conda/bin/python3.9 (numpy 1.26.4, scipy 1.7.1, astropy 4.3.1 -- the
reference's own pins, requirements.txt:1-9).  The system python (numpy 2.2 / scipy 1.15, no
astropy) cannot import the reference (np.NaN, np.float, interp2d are gone).

What this harness does around the reference, and nothing more:
  * a scratch dir holds a symlink `RaJePy -> /root/reference` because the package imports
    itself by that name (classes.py:28-34); nothing is copied;
  * the PyPI package `uncertainties` is not installed; maths/physics.py:11 imports
    `ufloat` from it at module level but only calls it when `errors=True`
    (physics.py:658-659), which nothing on the RT path does.  A 2-line placeholder
    module whose `ufloat` raises if ever called satisfies the import; it cannot
    change any number written here;
  * numpy 1.26 lacks `np.asscalar/np.alen` (astropy 4.3.1 wants them) and `np.float/np.str`
    (miscellaneous/functions.py:98-111): process-local aliases to the builtins;
  * params are passed as dicts (skips the validator that demands `n_0`, classes.py:157-158);
  * NO pickle shipped inside the reference is ever loaded: maths/physics.py:620 reads
    files/atomic_masses.pkl with pandas.read_pickle on every atomic_mass() call; here
    pandas.read_pickle is replaced (process-locally, before the import) by a parser of the
    TEXT table files/atomic_masses.data (the AME2003 mass table the pickle was made from)
    that returns the three columns physics.py:621-622 uses.  The twelve isotopes the path can
    ask for are checked against rajepy_amd/_constants.py:ATOMIC_MASS_MICRO_U on the way.

Outputs: tests/golden/*.npz + pipeline_cfg1.json.  Fields are stored sparsely (flat index of
jet cells + values) -- they are data (inputs / expected outputs), not reference source.
"""
import json
import os
import runpy
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _bootstrap():
    scratch = tempfile.mkdtemp(prefix="rjp_oracle_")
    os.symlink(REF, os.path.join(scratch, "RaJePy"))
    stubs = os.path.join(scratch, "stubs")
    os.makedirs(stubs)
    with open(os.path.join(stubs, "uncertainties.py"), "w") as f:
        f.write("def ufloat(*a, **k):\n"
                "    raise NotImplementedError('placeholder: uncertainties is not installed')\n")
    home = os.path.join(scratch, "home")
    os.makedirs(home)
    os.environ["HOME"] = home
    sys.dont_write_bytecode = True
    sys.path[:0] = [scratch, stubs]
    import numpy as np
    for nm, v in (("asscalar", lambda a: a.item()), ("alen", len), ("float", float),
                  ("str", str)):
        if not hasattr(np, nm):
            setattr(np, nm, v)
    import matplotlib
    matplotlib.use("Agg")
    import warnings
    warnings.filterwarnings("ignore")
    _no_pickles()
    return scratch


def _atomic_mass_table():
    """N, Z, mass[micro-u] of every nuclide in files/atomic_masses.data (AME2003 mass table,
    plain text with CR line ends; '#' marks estimated values)."""
    import re
    import pandas as pd
    raw = open(os.path.join(REF, "files", "atomic_masses.data"), "rb").read().decode("latin-1")
    rows = []
    for line in re.split(r"[\r\n]+", raw):
        m = re.match(r"^.\s*(-?\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+([A-Za-z]{1,3})\b", line)
        if not m:
            continue
        tok = line.replace("#", ".").split()
        try:
            mass = int(tok[-3]) * 1e6 + float(tok[-2])
        except ValueError:
            continue
        rows.append((int(m.group(2)), int(m.group(3)), mass))
    return pd.DataFrame(rows, columns=["N", "Z", "mass[micro-u]"])


def _no_pickles():
    import pandas as pd
    table = _atomic_mass_table()

    def read_pickle(path, *a, **k):
        if os.path.basename(str(path)) == "atomic_masses.pkl":
            return table
        raise RuntimeError("refusing to unpickle %s from the reference tree" % path)
    pd.read_pickle = read_pickle
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    try:
        import importlib.util
        spec = importlib.util.spec_from_file_location(
            "_rjp_constants", os.path.join(os.path.dirname(os.path.dirname(HERE)),
                                           "rajepy_amd", "_constants.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        for el, (z, n) in mod.NZ.items():
            got = table[(table["N"] == n) & (table["Z"] == z)]["mass[micro-u]"].values
            assert len(got) == 1 and abs(got[0] - mod.ATOMIC_MASS_MICRO_U[el]) < 1e-6, (el, got)
    finally:
        sys.path.pop(0)


SCRATCH = _bootstrap()
import numpy as np  # noqa: E402
import scipy.constants as con  # noqa: E402
import RaJePy  # noqa: E402
from RaJePy import logger  # noqa: E402
from RaJePy.maths import physics as mphys, rrls as mrrl  # noqa: E402


def example_params():
    p = runpy.run_path(os.path.join(REF, "files", "example-model-params.py"))["params"]
    p["grid"]["l_z"] = None      # BASELINE config 1 means the 50x400x50 grid
    return p


def tilted_params():
    """Small inclined/rotated jet that exercises every non-default branch: q_T != 0
    (power-law Gaunt branch), q^d_v != 0 (hyp2f1 in t_rw), asymmetric bursts, CW rotation."""
    return {
        "target": {"name": "tilted", "ra": "04:31:34.07736", "dec": "+18:08:04.9020",
                   "epoch": "J2000", "dist": 140., "v_lsr": -3.5, "M_star": 0.8,
                   "R_1": .3, "R_2": 2.0},
        "grid": {"n_x": 36, "n_y": 64, "n_z": 48, "l_z": None, "c_size": 1.0},
        "geometry": {"epsilon": 0.6, "opang": 30., "w_0": 1.5, "r_0": 2., "inc": 60.,
                     "pa": 25., "rotation": "CW"},
        "power_laws": {"q_v": -0.1, "q_T": -0.05, "q_x": -0.2, "q^d_n": -0.5,
                       "q^d_T": -0.1, "q^d_v": -0.4, "q^d_x": 0.2},
        "properties": {"v_0": 200., "x_0": 0.2, "T_0": 8e3, "mu": 1.3,
                       "mlr_bj": 1e-8, "mlr_rj": 7.5e-9},
        "ejection": {"t_0": np.array([0.3, 0.8]), "hl": np.array([0.2, 0.3]),
                     "chi": np.array([4., 6.]), "which": np.array(["RB", "B"])},
    }


def scalar_params(p):
    out = {}
    for sec, d in p.items():
        out[sec] = {}
        for k, v in d.items():
            if isinstance(v, np.ndarray):
                out[sec][k] = v.tolist()
            elif isinstance(v, (np.floating, np.integer)):
                out[sec][k] = v.item()
            else:
                out[sec][k] = v
    return out


def new_model(params, name):
    log = logger.Log(os.path.join(SCRATCH, name + ".log"), verbose=False)
    return RaJePy.JetModel(params, log=log)


def dump_fields(jm):
    """Sparse dump of every grid the RT path reads (classes.py:1116-1118, 1160-1171,
    1375, 1395-1397, 1471)."""
    ff = jm.fill_factor
    areas = jm.areas
    _ = jm.number_density          # forces _nd
    nd = jm._nd
    xi = jm.ion_fraction
    temp = jm.temperature
    _ = jm.ts                      # forces _ts (launch times, s)
    ts0 = jm._ts
    vy = jm.vel[1]
    rr = jm.rr
    jet = np.isfinite(ff)
    # everything the path multiplies by ff/areas must be NaN wherever ff is NaN, or the
    # sparse form would lose information; verify instead of assuming.
    for nm, a in (("areas", areas), ("nd", nd), ("xi", xi), ("temp", temp), ("vy", vy)):
        assert not np.isfinite(a[~jet]).any(), nm
    idx = np.flatnonzero(jet.ravel())
    take = lambda a: np.ascontiguousarray(a.ravel()[idx])
    return dict(idx=idx.astype(np.int64), ff=take(ff), areas=take(areas), nd=take(nd),
                xi=take(xi), temp=take(temp), ts0=take(ts0), vy=take(vy), rr=take(rr),
                ww=take(jm.ww), pp=take(jm.pp), rreff=take(jm.rreff),
                vx=take(jm.vel[0]), vz=take(jm.vel[2]))


def rt_products(jm, years, freqs, rrl=None, rrl_freqs=None):
    out = {}
    out["years"] = np.asarray(years, float)
    out["freqs"] = np.asarray(freqs, float)
    E, F = len(years), len(freqs)
    nx, nz = jm.nx, jm.nz
    em = np.empty((E, nx, nz))
    tau = np.empty((E, F, nx, nz))
    inten = np.empty((E, F, nx, nz))
    flux = np.empty((E, F, nx, nz))
    chi_max = np.empty(E)
    for e, yr in enumerate(years):
        jm.time = yr * con.year
        em[e] = jm.emission_measure()
        tau[e] = jm.optical_depth_ff(np.asarray(freqs, float))
        inten[e] = jm.intensity_ff(np.asarray(freqs, float))
        flux[e] = jm.flux_ff(np.asarray(freqs, float))
        chi_max[e] = np.nanmax(jm.chi_xyz)
        # scalar-call form must agree with the array-call form
        assert np.array_equal(jm.optical_depth_ff(float(freqs[0])), tau[e, 0])
    out.update(em=em, tau_ff=tau, intensity_ff=inten, flux_ff=flux, chi_max=chi_max)
    if rrl is not None:
        jm.time = years[0] * con.year
        rf = np.asarray(rrl_freqs, float)
        out["rrl_freqs"] = rf
        out["tau_rrl"] = jm.optical_depth_rrl(rrl, rf)
        out["flux_rrl_contsub"] = jm.flux_rrl(rrl, rf, contsub=True)
        out["flux_rrl_total"] = jm.flux_rrl(rrl, rf, contsub=False)
        out["intensity_rrl0"] = jm.intensity_rrl(rrl, float(rf[0]))
        el, n, dn = mrrl.rrl_parser(rrl)
        out["rrl_nu0"] = np.array(mrrl.rrl_nu_0(el, n, dn))
        out["rrl_fn1n2"] = np.array(mrrl.f_n1n2(n, dn))
        out["rrl_en"] = np.array(mrrl.energy_n(n, el))
        out["rrl_ni_per_ne"] = np.array(mrrl.ni_from_ne(1.0, el))
        out["atomic_mass"] = np.array(mphys.atomic_mass(el))
    return out


def golden_gff():
    nus = np.array([1e9, 1.5e9, 3e9, 5e9, 8.4e9, 2.2364174326e10, 4.3e10, 5e10, 1e11, 3e11])
    temps = np.array([3e3, 5e3, 8e3, 1e4, 1.5e4, 2e4])
    g = np.array([[float(mphys.gff(nu, t)) for t in temps] for nu in nus])
    np.savez_compressed(os.path.join(HERE, "gff.npz"), nus=nus, temps=temps, gff=g)
    print("gff anchors", g[3, 3], g[0, 3])


def golden_scalars():
    """Reference values of the scalar host-side helpers that feed the path."""
    from RaJePy.maths import geometry as mgeom
    rec = {}
    rec["mod_r_0"] = float(mgeom.mod_r_0(25., 7. / 9., 1.))
    rec["n_0_from_mlr"] = [float(mphys.n_0_from_mlr(1e-7, 150., 1., 1.3, qn, qv, .25, 2.5))
                           for qn, qv in ((0., 0.), (-0.5, -0.5), (-1., 0.), (-1.5, -0.5),
                                          (0.5, -1.), (-0.25, 0.75))]
    rec["rrl_nu_0"] = {l: float(mrrl.rrl_nu_0(*mrrl.rrl_parser(l)))
                       for l in ("H66a", "H58a", "He42b", "H110g", "H30d")}
    rec["doppler_shift"] = float(mphys.doppler_shift(2.2364174326e10, 12.5))
    rec["blackbody_nu"] = float(mphys.blackbody_nu(2.2e10, 9e3))
    rec["deltanu_g"] = float(mrrl.deltanu_g(2.2364e10, 1e4, "H"))
    rec["deltanu_l"] = float(mrrl.deltanu_l(1e6, 66, 1))
    xs = np.array([-12.3, -3.0, -0.4, 0., 0.7, 2.5, 6.1, 9.0, 30.0])
    rec["phi_voigt"] = [float(mrrl.phi_voigt_nu(2.2364e10, 3e5, 1.6e6, 2.2364e10 + x * 1e6))
                        for x in xs]
    rec["phi_voigt_x_mhz"] = xs.tolist()
    rec["freq_str"] = {str(f): RaJePy.miscellaneous.functions.freq_str(f)
                       for f in (5e9, 2.2364e10, 1.4e9, 3.3e8, 2.3e11)}
    # host-side JetModel behaviour without any RT: grid override by l_z, burst-less table
    p = example_params()
    p["grid"]["l_z"] = 2.
    jm = new_model(p, "lz")
    rec["lz_grid_dims"] = [jm.nx, jm.ny, jm.nz]
    p = example_params()
    p["ejection"] = {k: np.array([]) for k in ("t_0", "hl", "chi", "which")}
    rec["jetmodel_str_no_bursts"] = str(new_model(p, "nob"))
    with open(os.path.join(HERE, "scalars.json"), "w") as f:
        json.dump(rec, f, indent=1, sort_keys=True)


def golden_r86():
    """The analytic Reynolds (1986) cross-checks of maths/physics.py:93-374 (tau_r, r_tau1,
    approx_flux_expected_r86, flux_expected_r86) as sed_plot calls them
    (plotting/functions.py:1194-1227: y_max = l_z / 2 per lobe), config 1, 3 frequencies."""
    p = example_params()
    jm = new_model(p, "r86")
    freqs = [1e9, 5e9, 5e10]
    g, pl, pr = jm.params["geometry"], jm.params["power_laws"], jm.params["properties"]
    rec = {"freqs": freqs, "y_max_arcsec": 1.0,
           "params": scalar_params(jm.params),
           "ss_jml": {"R": float(jm.ss_jml("R")), "B": float(jm.ss_jml("B"))}}
    for which in ("R", "B"):
        rec["approx_" + which] = [float(mphys.approx_flux_expected_r86(jm, f, which))
                                  for f in freqs]
        rec["exact_" + which] = [float(mphys.flux_expected_r86(jm, f, which, 1.0))
                                 for f in freqs]
        rec["exact_ymin_" + which] = [float(mphys.flux_expected_r86(jm, f, which, 1.0, 0.05))
                                      for f in freqs]
    rec["approx_array_B"] = [float(v) for v in
                             mphys.approx_flux_expected_r86(jm, list(freqs), "B")]
    args = (g["r_0"], g["w_0"], pr["n_0"], pr["x_0"], pr["T_0"])
    tail = (g["inc"], g["epsilon"], pl["q_n"], pl["q_x"], pl["q_T"], g["opang"])
    rec["r_tau1_au"] = [float(mphys.r_tau1(*args, f, *tail)) for f in freqs]
    rec["r_tau1_arcsec"] = [float(mphys.r_tau1(*args, f, *tail,
                                               dist=jm.params["target"]["dist"]))
                            for f in freqs]
    rec["tau_r"] = [[float(mphys.tau_r(r, *args, f, *tail)) for r in (1., 5., 40.)]
                    for f in freqs]
    # a disc-wind density prescription (q^d_n != 0) sends both flux functions to the key
    # "mlr", which today's params files do not have (they carry mlr_bj / mlr_rj)
    try:
        mphys.flux_expected_r86(new_model(tilted_params(), "r86t"), 5e9, "B", 1.0)
        rec["tilted_raises"] = None
    except Exception as exc:
        rec["tilted_raises"] = type(exc).__name__ + ": " + str(exc)
    with open(os.path.join(HERE, "r86.json"), "w") as f:
        json.dump(rec, f, indent=1, sort_keys=True)
    print("r86: exact B", rec["exact_B"], "approx B", rec["approx_B"], rec["tilted_raises"])


def golden_model(tag, params, years, freqs, rrl, rrl_nchan, rrl_cw):
    t0 = time.time()
    jm = new_model(params, tag)
    fields = dump_fields(jm)
    el, n, dn = mrrl.rrl_parser(rrl)
    nu0 = mrrl.rrl_nu_0(el, n, dn)
    # channel grid as ContinuumRun.chan_freqs builds it (classes.py:1897-1900)
    bw = rrl_nchan * rrl_cw
    rf = nu0 - bw / 2. + rrl_cw / 2. + np.arange(rrl_nchan) * rrl_cw
    prods = rt_products(jm, years, freqs, rrl, rf)
    meta = dict(params=scalar_params(jm.params), shape=[jm.nx, jm.ny, jm.nz],
                csize=jm.csize, rrl=rrl,
                ss_jml_bj=jm._ss_jml_bj, ss_jml_rj=jm._ss_jml_rj,
                ejections=jm.ejections)
    np.savez_compressed(os.path.join(HERE, tag + ".npz"), meta=json.dumps(meta),
                        **{"f_" + k: v for k, v in fields.items()}, **prods)
    f0 = prods["flux_ff"]
    print("%s: %d jet cells, %.1fs; sum flux[e,0] = %s; tau max %.6e; EM max %.6e; "
          "tau_rrl max %.6e sum %.6e" % (
              tag, len(fields["idx"]), time.time() - t0,
              [float(np.nansum(f0[e, 0])) for e in range(len(years))],
              prods["tau_ff"][0, 0].max(), prods["em"][0].max(),
              prods["tau_rrl"].max(), prods["tau_rrl"][len(rf) // 2].sum()))
    return jm


def golden_pipeline(model_params):
    """Run the reference's Pipeline.execute RT section (classes.py:2386-2479) on config 1
    and record what a drop-in must reproduce: run order, results['flux'], FITS header cards
    and data."""
    from astropy.io import fits
    dcy = os.path.join(SCRATCH, "pline_out")
    pl_params = {
        "min_el": 20.,
        "dcys": {"model_dcy": dcy},
        "continuum": {"times": np.array([0., 1.]), "freqs": np.array([5.]) * 1e9,
                      "t_obs": np.array([1200]), "tscps": np.array([("VLA", "A")]),
                      "t_ints": np.array([5]), "bws": np.array([4e8]),
                      "chanws": np.array([2e8])},
        "rrls": {"times": np.array([0.]), "lines": np.array(["H66a"]),
                 "t_obs": np.array([1200]), "tscps": np.array([("VLA", "A")]),
                 "t_ints": np.array([60]), "bws": np.array([4e5]),
                 "chanws": np.array([1e5])},
    }
    os.makedirs(dcy)
    log = logger.Log(os.path.join(dcy, "model.log"), verbose=False)
    jm = RaJePy.JetModel(model_params, log=log)
    pl = RaJePy.Pipeline(jm, pl_params, log=log)
    pl.execute(simobserve=False, verbose=False, dryrun=False, resume=False, clobber=True)
    rec = {"runs": [], "tree": []}
    arrays = {}
    for root, _, files in os.walk(dcy):
        for fn in sorted(files):
            rec["tree"].append(os.path.relpath(os.path.join(root, fn), dcy))
    rec["tree"].sort()
    for i, run in enumerate(pl.runs):
        r = {"year": float(run.year), "obs_type": run.obs_type, "freq": float(run.freq),
             "day": int(run.day), "nchan": int(run.nchan),
             "chan_freqs": [float(_) for _ in run.chan_freqs],
             "line": getattr(run, "line", None), "completed": bool(run.completed),
             "rt_dcy": os.path.relpath(run.rt_dcy, dcy),
             "flux": np.atleast_1d(run.results["flux"]).tolist(), "fits": {}}
        for kind, path in (("em", run.fits_em), ("tau", run.fits_tau),
                           ("flux", run.fits_flux)):
            with fits.open(path) as hdul:
                hdr = hdul[0].header
                r["fits"][kind] = {"name": os.path.relpath(path, dcy),
                                   "cards": [str(c) for c in hdr.cards]}
                arrays["run%d_%s" % (i, kind)] = np.array(hdul[0].data, dtype=">f8").astype("<f8")
            with open(path, "rb") as fh:
                raw = fh.read()
            r["fits"][kind]["nbytes"] = len(raw)
            import hashlib
            r["fits"][kind]["sha256"] = hashlib.sha256(raw).hexdigest()
        rec["runs"].append(r)
    rec["jetmodel_str"] = str(jm)
    rec["pipeline_str"] = str(pl)
    with open(os.path.join(HERE, "pipeline_cfg1.json"), "w") as f:
        json.dump(rec, f, indent=1)
    np.savez_compressed(os.path.join(HERE, "pipeline_cfg1.npz"), **arrays)
    print("pipeline: runs", [(r["obs_type"], r["year"], r["flux"][:2]) for r in rec["runs"]])


def golden_as_shipped():
    """files/example-model-params.py exactly as shipped: l_z = 2 arcsec overrides the grid to
    108 x 110 x 588 (7 M cells).  Outputs only (the grids are built from the parameters), one
    epoch inside the burst history."""
    p = runpy.run_path(os.path.join(REF, "files", "example-model-params.py"))["params"]
    t0 = time.time()
    jm = new_model(p, "shipped")
    jm.time = 1.0 * con.year
    freqs = np.array([5e9, 4.3e10])
    nu0 = mrrl.rrl_nu_0("H", 66, 1)
    rf = nu0 - 4 * 5e5 / 2. + 5e5 / 2. + np.arange(4) * 5e5
    out = dict(meta=json.dumps(dict(params=scalar_params(jm.params), shape=[jm.nx, jm.ny, jm.nz])),
               year=1.0, freqs=freqs, rrl_freqs=rf,
               em=jm.emission_measure(), tau_ff=jm.optical_depth_ff(freqs),
               flux_ff=jm.flux_ff(freqs), tau_rrl=jm.optical_depth_rrl("H66a", rf),
               flux_rrl_total=jm.flux_rrl("H66a", rf, contsub=False),
               n_jet_cells=np.array(int(np.isfinite(jm.fill_factor).sum())))
    np.savez_compressed(os.path.join(HERE, "example_as_shipped.npz"), **out)
    print("as shipped: grid %s, %d jet cells, %.0f s; sum flux %s" % (
        (jm.nx, jm.ny, jm.nz), int(out["n_jet_cells"]), time.time() - t0,
        np.nansum(out["flux_ff"], axis=(1, 2))))


if __name__ == "__main__":
    if sys.argv[1:] == ["scalars"]:          # regenerate scalars.json only
        golden_scalars()
        sys.exit(0)
    if sys.argv[1:] == ["r86"]:              # the analytic Reynolds-86 cross-checks only
        golden_r86()
        sys.exit(0)
    if sys.argv[1:] == ["shipped"]:          # the 7 M-cell as-shipped example only (~4 min)
        golden_as_shipped()
        sys.exit(0)
    golden_gff()
    golden_scalars()
    golden_model("cfg1_example", example_params(), years=[0., 0.5, 1., 2., 3.],
                 freqs=[5e9, 1e9, 2.236417432622781e10, 5e10], rrl="H66a",
                 rrl_nchan=8, rrl_cw=2.5e5)
    golden_model("tilted", tilted_params(), years=[0., 0.4, 0.9],
                 freqs=[5e9, 1.5e9, 4.3e10], rrl="H58a", rrl_nchan=6, rrl_cw=4e5)
    golden_pipeline(example_params())
    golden_r86()
    golden_as_shipped()
