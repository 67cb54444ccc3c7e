"""Jet geometry scalars and the launch-time integral (reference: maths/geometry.py).

The grid-sized geometry (vertex tests, r/w/phi, power-law fields) is built on the GPU by
`rjp_build_fields`.  The one piece kept on the host is `t_rw` for q^d_v != 0, which needs
Gauss' hypergeometric function (scipy.special.hyp2f1, as in the reference); with q^d_v == 0
the integral is closed-form and is evaluated on the device."""
import numpy as np

from .. import _constants as con


def mod_r_0(opang, epsilon, w_0):
    """'Modified' launching radius (geometry.py:12-31)."""
    return epsilon * w_0 / np.tan(np.radians(opang) / 2.)


def rho(r, r_0, mr0=None):
    """geometry.py:34-61."""
    if mr0:
        return (np.abs(r) + mr0 - r_0) / mr0
    return np.abs(r) / r_0


def w_r(r, w_0, mr0, r_0, eps):
    """Jet half-width at r (geometry.py:95-118)."""
    return w_0 * rho(r, r_0, mr0) ** eps


def r_eff(w, r_1, r_2, w_0, r, mr0, r_0, eps):
    """Disc radius a streamline at (w, r) came from (geometry.py:305-336)."""
    return r_1 + ((r_2 - r_1) * w) / w_r(r, w_0, mr0, r_0, eps)


def rotation_terms(alpha_deg, beta_deg):
    a, b = np.radians(alpha_deg), np.radians(beta_deg)
    return np.cos(a), np.sin(a), np.cos(b), np.sin(b)


def xyz_to_rwp(x, y, z, inc, pa):
    """(x, y, z) -> jet coordinates (r, w, phi) (geometry.py:181-302)."""
    ca, sa, cb, sb = rotation_terms(inc - 90., pa)
    x1, z1 = cb * x + sb * z, cb * z - sb * x
    x2, y2, r = x1, ca * y - sa * z1, sa * y + ca * z1
    with np.errstate(invalid="ignore", divide="ignore"):
        w = np.sqrt(x2 ** 2. + y2 ** 2.)
        p = np.arcsin(y2 / w)
    p = np.where(np.asarray(x2) < 0, -p + np.pi, p)
    return r, w, p


def t_rw(r, w, params):
    """Flow time [yr] from the launch radius to (r, w) (geometry.py:121-178), array form."""
    from scipy.special import hyp2f1
    g, pr, tg, pl = (params['geometry'], params['properties'], params['target'],
                     params['power_laws'])
    w_0, r_0, mr0 = g['w_0'] * con.au, g['r_0'] * con.au, g['mod_r_0'] * con.au
    v_0, eps = pr["v_0"] * 1e3, g['epsilon']
    r_1, r_2 = tg["R_1"] * con.au, tg["R_2"] * con.au
    q_v, q_vd = pl["q_v"], pl["q^d_v"]
    b = (1. - q_v + eps * q_vd) / eps

    def antiderivative(r_, w_):
        rad = r_ + mr0 - r_0
        lead = mr0 ** q_v / (v_0 * (1. - q_v + eps * q_vd)) * rad ** (1. - q_v)
        on_axis = (w_ == 0.)
        ws = np.where(on_axis, 1., w_)
        a_ = (r_1 * w_0 * rad ** eps) / (ws * mr0 ** eps * (r_2 - r_1))
        reff_term = (r_eff(w_, r_1, r_2, w_0, r_, mr0, r_0, eps) / r_1) ** -q_vd
        off = (a_ + 1.) ** q_vd * hyp2f1(q_vd, b, b + 1., (r_1 * w_0 * rad ** eps) /
                                         (ws * mr0 ** eps * (r_1 - r_2)))
        return lead * reff_term * np.where(on_axis, 1. + q_vd / (1. - q_v), off)

    w_m = np.asarray(w, dtype=np.float64) * con.au
    r_m = np.abs(np.asarray(r, dtype=np.float64)) * con.au + 0. * w_m
    return (antiderivative(r_m, w_m) - antiderivative(r_0 + 0. * w_m, w_m)) / con.year
