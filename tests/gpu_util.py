"""Shared helpers for the GPU parity tests (tests only: may import the oracle)."""
import json
import os

import numpy as np

from oracle import rt_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# K3 against scipy.special.wofz (through the oracle / the reference's golden cubes).  The
# kernels whose waves work on one cell (more than 16 channels: the 64- and 256-lane layouts)
# evaluate Re w(x + i y) to <= 1e-8 relative by design since round 3 (worst path 4.1e-9,
# tools/voigt_design.py; SURVEY.md section 7 asks <= 1e-7, the bar on the maps is 1e-5); an
# optical depth is a sum of same-signed terms, so the bound carries over to tau unamplified.
# The generic per-lane code (<= 16 channels, collapse=False) keeps 1e-11 per evaluation.
K3_RTOL_WAVE = 1e-8
K3_RTOL_LANE = 1e-9


def k3_rtol(nchan):
    return K3_RTOL_WAVE if nchan > 16 else K3_RTOL_LANE


def load_golden(tag):
    z = np.load(os.path.join(GOLDEN, tag + ".npz"))
    meta = json.loads(str(z["meta"]))
    p = meta["params"]
    for k in ("t_0", "hl", "chi", "which"):
        p["ejection"][k] = np.array(p["ejection"][k])
    return z, meta, p


def golden_dense(tag):
    """Reference grids of a golden model as dense float64 arrays + an oracle object."""
    z, meta, p = load_golden(tag)
    shape = tuple(meta["shape"])
    idx = z["f_idx"]
    d = lambda k, fill=np.nan: orc.dense_from_sparse(shape, idx, z["f_" + k], fill)
    g = dict(nd=d("nd"), xi=d("xi"), temp=d("temp"), ff=d("ff"), areas=d("areas"),
             ts=d("ts0", 0.), rr=d("rr", 1.), vy=d("vy"))
    jet = orc.OracleJet.from_fields(p, g["nd"], g["xi"], g["temp"], g["ff"], g["areas"],
                                    g["ts"], g["rr"], g["vy"])
    return z, meta, p, g, jet


def bursts_from_oracle(jet):
    """rjp_bursts parameters out of an OracleJet's burst list (classes.py:442-448)."""
    from rajepy_amd.engine import make_bursts
    out = []
    for which, ss in (("R", jet._ss_jml_rj), ("B", jet._ss_jml_bj)):
        lst = []
        for t0, peak, hl in jet.bursts[which]:
            sigma = hl * 2. / (2. * np.sqrt(2. * np.log(2.)))
            lst.append((t0, (peak - ss) / ss, sigma))
        out.append(lst)
    return make_bursts(out[0], out[1])


def synth_host(shape, seed, temp_mode=0, cell0=0, cells=None, nz_full=None):
    """Host restatement of rjp_synth_fields -- splitmix64 counter hash.
    With `cells` (flat indices into a grid whose z-extent is `nz_full`) it generates exactly
    those cells, returned in `shape`."""
    nx, ny, nz = shape
    n = nx * ny * nz
    if cells is None:
        cell = (np.arange(n, dtype=np.uint64) + np.uint64(cell0))
    else:
        cell = np.asarray(cells, dtype=np.uint64).ravel()
        assert cell.size == n
        nz = nz_full

    def u01(field):
        x = np.uint64(seed) ^ (np.uint64(field) << np.uint64(60)) ^ cell
        with np.errstate(over="ignore"):
            x = x + np.uint64(0x9E3779B97F4A7C15)
            x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            x = x ^ (x >> np.uint64(31))
        return (x >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)

    iz = (cell % np.uint64(nz)).astype(np.int64)
    red = iz < nz // 2
    nd = 10.0 ** (5.0 + 2.5 * u01(1))
    xi = 0.05 + 0.45 * u01(2)
    temp = np.full(n, 1e4) if temp_mode == 0 else 5e3 + 1.5e4 * u01(3)
    pf = np.where(u01(4) < 0.25, 0.5, 1.0)
    ts = 5.0 * u01(5) * 31536000.0
    vy = 6.2 + 60.0 * (u01(6) - 0.5)
    nx, ny, nz = shape
    r = lambda a: a.reshape(shape)
    return dict(nd=r(nd), xi=r(xi), temp=r(temp), ff=r(pf), areas=r(np.ones(n)), ts=r(ts),
                rr=r(np.where(red, -1.0, 1.0)), vy=r(vy))


def example_bursts_params():
    """The four bursts of the reference's example model (files/example-model-params.py:51-54)."""
    return {"t_0": np.array([0.5, 0.75, 1., 2.]), "hl": np.array([0.15, 0.15, 0.45, 0.5]),
            "chi": np.array([5., 5., 2.5, 10.]), "which": np.array(["R", "B", "B", "RB"])}
