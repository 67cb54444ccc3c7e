"""CPU ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A literal NumPy restatement of the reference's line-of-sight radiative-transfer path
(SimonP2207/RaJePy, `classes.py:1101-1541` and the `maths/` helpers it calls), with the same
per-channel re-streaming structure, the same formulas in the same multiplication order and
the same NaN conventions.  Every function cites the reference lines it follows (paths
relative to the reference root).  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import this module; the product package `rajepy_amd`
never does (tests/test_host_logic.py::test_product_never_imports_the_oracle checks that).

Parity status: PINNED.  tests/test_oracle_golden.py checks every function here against
golden vectors produced by importing the unmodified reference in the build container
(tests/golden/make_golden.py; the reference's own tests hold no fixtures for this path,
SURVEY.md section 4).

Third-party arithmetic the reference delegates (not under the reference tree):
  * scipy.special.wofz (Faddeeva; maths/rrls.py:5,353)           -> scipy.special.wofz
  * scipy.interpolate.interp2d(kind='cubic') on scattered 5x5 points
    (maths/physics.py:677-697) = FITPACK surfit via bisplrep(kx=ky=3, s=0) + bisplev
                                                                  -> the same two calls
  * scipy.special.hyp2f1 (maths/geometry.py:168)                  -> scipy.special.hyp2f1
  * scipy.constants (CODATA-2018 at the reference's pinned scipy 1.7.1) -> literals below
"""
import os
import warnings

import numpy as np
from scipy.interpolate import bisplev, bisplrep
from scipy.special import hyp2f1, wofz

# --- scipy.constants @ scipy 1.7.1 (CODATA-2018) ------------------------------------------
AU = 149597870700.0
PARSEC = 3.085677581491367e+16
K_B = 1.380649e-23
H_PL = 6.62607015e-34
C_LIGHT = 299792458.0
YEAR = 31536000.0
M_E = 9.1093837015e-31
AMU = 1.6605390666e-27
RYDBERG = 10973731.56816
EPS0 = 8.8541878128e-12
G_NEWT = 6.6743e-11
E_CH = 1.602176634e-19
MSOL = 1.98847e30                      # _constants.py:5
C_CGS = C_LIGHT * 1e2                  # maths/rrls.py:7-11
H_CGS = H_PL * 1e7
K_CGS = K_B * 1e7

_NZ = {"H": (1, 0), "He": (2, 2), "Li": (3, 4), "Be": (4, 5), "B": (5, 6), "C": (6, 6),
       "N": (7, 7), "O": (8, 8)}
# AME2003 masses [micro-u] of those isotopes = what physics.py:620-623 reads from its table
_MASS_MICRO_U = {"H": 1007825.03207, "He": 4002603.25415, "Li": 7016004.548,
                 "Be": 9012182.201, "B": 11009305.406, "C": 12000000.0,
                 "N": 14003074.00478, "O": 15994914.61956}

_GAUNT_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir,
                           "rajepy_amd", "files", "vanHoofetal2014.data")


# ------------------------------------------------------------------------------------------
# maths/physics.py
# ------------------------------------------------------------------------------------------
def atomic_mass(atom):
    """physics.py:607-624 -- mass [kg] = table micro-u * 1e-6 * u."""
    m = _MASS_MICRO_U[atom]
    m *= 1e-6 * AMU
    return m


def z_number(atom):
    """physics.py:523-532."""
    return {"H": 1, "He": 2, "Li": 3, "Be": 4, "B": 5, "C": 6, "N": 7, "O": 8}[atom]


def rydberg_constant(atom):
    """physics.py:535-544."""
    m_atom = atomic_mass(atom)
    return RYDBERG * (m_atom / (m_atom + M_E))


def doppler_shift(nu_0, v_lsr):
    """physics.py:547-558."""
    v = v_lsr * 1000.
    return nu_0 * (1. - v / C_LIGHT)


def blackbody_nu(freq, temp):
    """physics.py:561-574 (cgs)."""
    p1 = 2. * H_PL * 1e7 * freq ** 3. / (C_LIGHT * 1e2) ** 2.
    p2 = np.exp(H_PL * 1e7 * freq / (K_B * 1e7 * temp)) - 1.
    return p1 * p2 ** -1.


def q_n(epsilon, q_v):
    """physics.py:17-36."""
    return -q_v - (2.0 * epsilon)


def q_tau(epsilon, q_x, q_n_, q_T):
    """physics.py:39-63."""
    return epsilon + 2.0 * q_x + 2.0 * q_n_ - 1.35 * q_T


def n_0_from_mlr(mlr, v_0, w_0, mu, q_nd, q_nv, R_1, R_2):
    """physics.py:474-517."""
    a = q_nd + q_nv
    if a == -1. or a == -2.:
        a *= 1. + 1e-12
    r2 = R_2 * AU
    r1 = R_1 * AU
    mlr_si = mlr * MSOL / YEAR
    constant = 2. * np.pi * (mu * atomic_mass('H')) * (v_0 * 1e3) * (w_0 * AU) ** 2.
    return mlr_si / constant / \
        ((r1 ** 2. + r2 * (r2 * (a + 1.) - r1 * (a + 2.)) * (r2 / r1) ** a) /
         ((r2 - r1) ** 2. * (a + 1.) * (a + 2.))) / 1e6


def v_rot(r, reff, rho_, epsilon, m_star):
    """physics.py:66-90 -- Keplerian speed at r_eff scaled by rho**-eps [km/s]."""
    return np.sqrt(G_NEWT * m_star * MSOL / (reff * AU)) * rho_ ** -epsilon / 1e3


_GAUNT = None


def import_vanHoof2014():
    """physics.py:626-663 (errors=False branch): 146 rows of log u x 81 columns of log g2."""
    global _GAUNT
    if _GAUNT is None:
        with open(_GAUNT_FILE, "rt") as f:
            lines = f.readlines()
        loggam2_start = float(lines[30].split('#')[0])
        logu_start = float(lines[31].split('#')[0])
        step = float(lines[32].split('#')[0])
        data = np.array([[float(_) for _ in l.split()] for l in lines[42:188]])
        n_logu, n_lg2 = data.shape
        logus = np.linspace(np.round(logu_start, decimals=1),
                            np.round(logu_start + (step * (n_logu - 1)), decimals=1), n_logu)
        lg2s = np.linspace(np.round(loggam2_start, decimals=1),
                           np.round(loggam2_start + (step * (n_lg2 - 1)), decimals=1), n_lg2)
        lg2s, logus = np.meshgrid(lg2s, logus)
        _GAUNT = (lg2s, logus, data)
    return _GAUNT


def gff(freq, temp, z=1.):
    """physics.py:666-698.  interp2d(kind='cubic') with 2-D coordinate arrays takes scipy's
    scattered-data branch = bisplrep(kx=3, ky=3, s=0) + bisplev (scipy 1.7.1
    interpolate.py, class interp2d).  Includes the row clamp against len(logus[0]) (=81,
    not 146) of physics.py:687-690."""
    Ry = M_E * E_CH ** 4. / (8 * EPS0 ** 2. * H_PL ** 2.)
    logg2 = np.log10(z ** 2. * Ry / (K_B * temp))
    logu = np.log10(H_PL * freq / (K_B * temp))
    logg2s, logus, gffs = import_vanHoof2014()
    col = int(np.argmin(np.abs(logg2s[0] - logg2)))
    row = int(np.argmin(np.abs(logus[:, 0] - logu)))
    if col < 2:
        col = 2
    elif col > len(logg2s[0]) - 3:
        col = len(logg2s[0]) - 3
    if row < 2:
        row = 2
    elif row > len(logus[0]) - 3:
        row = len(logus[0]) - 3
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tck = bisplrep(logg2s[row - 2: row + 3, col - 2: col + 3].ravel(),
                       logus[row - 2: row + 3, col - 2: col + 3].ravel(),
                       gffs[row - 2: row + 3, col - 2: col + 3].ravel(),
                       kx=3, ky=3, s=0.0)
    return float(bisplev(logg2, logu, tck))


# ------------------------------------------------------------------------------------------
# maths/rrls.py (LTE path)
# ------------------------------------------------------------------------------------------
def rrl_parser(rrl_str):
    """rrls.py:605-624."""
    dn = {'a': 1, 'b': 2, 'g': 3, 'd': 4}[rrl_str[-1].lower()]
    el, n = '', ''
    for ch in rrl_str[:-1]:
        if ch.isalpha():
            el += ch
        else:
            n += ch
    return el, int(n), dn


def rrl_nu_0(atom, n, delta_n=1):
    """rrls.py:14-29."""
    z = z_number(atom)
    r = rydberg_constant(atom)
    return r * C_LIGHT * z ** 2. * (1. / n ** 2. - 1. / (n + delta_n) ** 2.)


def energy_n(n, atom):
    """rrls.py:32-41."""
    return -2.17989724e-11 * z_number(atom) ** 2. / n ** 2.


def f_n1n2(n_1, delta_n):
    """rrls.py:44-59."""
    m = {1: 0.190775, 2: 0.026332, 3: 0.0081056, 4: 0.0034918}[delta_n]
    return n_1 * m * (1. + 1.5 * delta_n / n_1)


def ni_from_ne(n_e, atom='H'):
    """rrls.py:62-83."""
    xyz = {'H': 0.710, 'He': 0.276, 'CNO': 0.014}
    mu = (xyz['H'] / atomic_mass("H") * AMU + xyz['He'] / atomic_mass("He") * AMU +
          xyz['CNO'] / 14.24) ** -1.
    m_atom = atomic_mass(atom) / AMU
    return xyz[atom] * n_e * mu / m_atom


def deltanu_l(n_e, n, delta_n, gamma=4.5):
    """rrls.py:86-101."""
    return 8.2 * n_e * (n / 100.) ** gamma * (1. + gamma / 2. * delta_n / n)


def deltanu_g(nu_0, temp, atom):
    """rrls.py:104-118."""
    m = atomic_mass(atom)
    return np.sqrt(4. * np.log(2.) * 2. * K_B * temp / (m * C_LIGHT ** 2.)) * nu_0


def phi_voigt_nu(nu_0, fwhm_stark, fwhm_thermal):
    """rrls.py:329-359 -- returns phi_V(nu)."""
    def func(nu):
        sigma = fwhm_thermal / 2. / np.sqrt(2. * np.log(2))
        return np.real(wofz(((nu - nu_0) + 1j * fwhm_stark / 2.) /
                            sigma / np.sqrt(2.))) / sigma / np.sqrt(2. * np.pi)
    return func


def kappa_l(freq, n, oscillator_strength, line_profile_contribution, n_e, n_i, temp, z,
            energy_n1):
    """rrls.py:362-389."""
    p0 = 1.0991132675738456e-17
    p1 = n ** 2. * oscillator_strength * line_profile_contribution
    p2 = n_e * n_i / temp ** 1.5
    p3 = np.exp((z ** 2. * energy_n1) / (K_CGS * temp))
    p4 = 1. - np.exp(-H_CGS * freq / (K_CGS * temp))
    return p0 * p1 * p2 * p3 * p4


def line_intensity_lte(freq, temp, tau_c, tau_l):
    """rrls.py:428-449."""
    b_nu = blackbody_nu(freq, temp)
    i_l_cgs = b_nu * np.exp(-tau_c) * (1. - np.exp(-tau_l))
    return i_l_cgs * 1e-7 * 1e4


# ------------------------------------------------------------------------------------------
# maths/geometry.py
# ------------------------------------------------------------------------------------------
def mod_r_0(opang, epsilon, w_0):
    """geometry.py:12-31."""
    return epsilon * w_0 / np.tan(np.radians(opang) / 2.)


def rho(r, r_0, mr0=None):
    """geometry.py:34-61."""
    if mr0:
        return (np.abs(r) + mr0 - r_0) / mr0
    return np.abs(r) / r_0


def cell_value(zero_val, rho_, r_eff_, r1, q, qd):
    """geometry.py:64-92."""
    return zero_val * rho_ ** q * (r_eff_ / r1) ** qd


def w_r(r, w_0, mr0, r_0, eps):
    """geometry.py:95-118."""
    return w_0 * rho(r, r_0, mr0) ** eps


def r_eff(w, r_1, r_2, w_0, r, mr0, r_0, eps):
    """geometry.py:305-336."""
    return r_1 + ((r_2 - r_1) * w) / w_r(r, w_0, mr0, r_0, eps)


def xyz_rotate(x, y, z, alpha, beta, order='xy'):
    """geometry.py:214-266."""
    a = np.radians(alpha)
    b = np.radians(beta)
    cos_a, sin_a = np.cos(a), np.sin(a)
    cos_b, sin_b = np.cos(b), np.sin(b)

    def x_rot(x_, y_, z_):
        return x_, cos_a * y_ - sin_a * z_, sin_a * y_ + cos_a * z_

    def y_rot(x_, y_, z_):
        return cos_b * x_ + sin_b * z_, y_, cos_b * z_ - sin_b * x_

    if order.lower() == 'xy':
        return y_rot(*x_rot(x, y, z))
    elif order.lower() == 'yx':
        return x_rot(*y_rot(x, y, z))
    raise ValueError(order)


def cartesian_to_cylindrical(x, y, z):
    """geometry.py:269-302 (array branch)."""
    with np.errstate(invalid="ignore", divide="ignore"):
        rho_ = np.sqrt(x ** 2. + y ** 2.)
        phi_ = np.arcsin(y / rho_)
    phi_ = np.where(x < 0, -phi_ + np.pi, phi_)
    return rho_, phi_, z


def xyz_to_rwp(x, y, z, inc, pa):
    """geometry.py:181-211."""
    xyz = xyz_rotate(x, y, z, inc - 90., pa, order='yx')
    w, p, r = cartesian_to_cylindrical(*xyz)
    return r, w, p


def t_rw(r, w, params):
    """geometry.py:121-178 -- flow time [yr] from r_0 to (r, w); np.vectorize'd scalar
    closure in the reference, evaluated here on whole arrays with the same branches."""
    w_0 = params['geometry']['w_0'] * AU
    r_0 = params['geometry']['r_0'] * AU
    v_0 = params["properties"]["v_0"] * 1e3
    mr0 = params['geometry']['mod_r_0'] * AU
    eps = params['geometry']['epsilon']
    r_1 = params["target"]["R_1"] * AU
    r_2 = params["target"]["R_2"] * AU
    q_v = params["power_laws"]["q_v"]
    q_vd = params["power_laws"]["q^d_v"]

    def indef(r_, w_):
        r_ = np.asarray(r_, dtype=float) + 0. * w_
        const = mr0 ** q_v / (v_0 * (1. - q_v + eps * q_vd))
        rad = r_ + mr0 - r_0
        p1 = rad ** (1. - q_v)
        p2 = (r_eff(w_, r_1, r_2, w_0, r_, mr0, r_0, eps) / r_1) ** -q_vd
        zero = (w_ == 0.)
        w_safe = np.where(zero, 1., w_)
        arg = (r_1 * w_0 * rad ** eps) / (w_safe * mr0 ** eps * (r_1 - r_2))
        p3 = ((r_1 * w_0 * rad ** eps) / (w_safe * mr0 ** eps * (r_2 - r_1)) + 1.) ** q_vd
        p4 = hyp2f1(q_vd, (1. - q_v + eps * q_vd) / eps,
                    (1. - q_v + eps + eps * q_vd) / eps, arg)
        p3 = np.where(zero, 1.0, p3)
        p4 = np.where(zero, 1. + q_vd / (1. - q_v), p4)
        return const * p1 * p2 * p3 * p4

    w_m = w * AU
    return (indef(np.abs(r) * AU, w_m) - indef(r_0, w_m)) / YEAR


# ------------------------------------------------------------------------------------------
# classes.py -- JetModel fields + RT
# ------------------------------------------------------------------------------------------
class OracleJet:
    """Mirror of the parts of `JetModel` (classes.py:42-1541) the RT path touches.

    Two ways to obtain the 3-D input grids:
      * `OracleJet(params)`            -- build them from the geometry (classes.py:465-1099)
      * `OracleJet.from_fields(...)`   -- inject dense arrays (the reference exposes setters
                                          for ts / ion_fraction / temperature, 857-1000)
    """

    def __init__(self, params):
        import copy
        p = copy.deepcopy(params)
        self.params = p
        self.csize = p['grid']['c_size']
        # classes.py:169-180
        mr0 = mod_r_0(p['geometry']['opang'], p['geometry']['epsilon'], p['geometry']['w_0'])
        qn = q_n(p["geometry"]["epsilon"], p["power_laws"]["q_v"])
        qt = q_tau(p["geometry"]["epsilon"], p["power_laws"]["q_x"], qn,
                   p["power_laws"]["q_T"])
        p["geometry"]["mod_r_0"] = mr0
        p["power_laws"]["q_n"] = qn
        p["power_laws"]["q_tau"] = qt
        # classes.py:201-213 (l_z=None branch only: even cell counts)
        assert p['grid'].get('l_z') is None, "oracle restates the l_z=None branch only"
        self.nx = (p['grid']['n_x'] + 1) // 2 * 2
        self.ny = (p['grid']['n_y'] + 1) // 2 * 2
        self.nz = (p['grid']['n_z'] + 1) // 2 * 2
        # classes.py:228-242
        self._ss_jml_rb_frac = p["properties"]["mlr_rj"] / p["properties"]["mlr_bj"]
        self._ss_jml_bj = p["properties"]["mlr_bj"]
        self._ss_jml_bj *= 1.989e30 / YEAR
        self._ss_jml_rj = self._ss_jml_bj * self._ss_jml_rb_frac
        p["properties"]["n_0"] = n_0_from_mlr(
            p["properties"]["mlr_bj"], p["properties"]["v_0"], p["geometry"]["w_0"],
            p["properties"]["mu"], p["power_laws"]["q^d_n"], p["power_laws"]["q^d_v"],
            p["target"]["R_1"], p["target"]["R_2"])
        # classes.py:245-264: burst list per jet, in registration order
        self.bursts = {'R': [], 'B': []}
        for idx, t0 in enumerate(p['ejection']['t_0']):
            which = str(p['ejection']['which'][idx])
            if 'R' in which:
                self.bursts['R'].append((t0 * YEAR, self._ss_jml_rj * p['ejection']['chi'][idx],
                                         p['ejection']['hl'][idx] * YEAR))
            if 'B' in which:
                self.bursts['B'].append((t0 * YEAR, self._ss_jml_bj * p['ejection']['chi'][idx],
                                         p['ejection']['hl'][idx] * YEAR))
        self.time = 0. * YEAR
        self._ff = self._areas = self._nd = self._xi = self._temp = None
        self._ts = self._vy = self._rr = None
        self._rwp = None

    # -- injection -------------------------------------------------------------------------
    @classmethod
    def from_fields(cls, params, nd, xi, temp, ff, areas, ts0, rr, vy=None):
        self = cls(params)
        self.nx, self.ny, self.nz = nd.shape
        self._nd, self._xi, self._temp = nd, xi, temp
        self._ff, self._areas, self._ts, self._rr, self._vy = ff, areas, ts0, rr, vy
        return self

    # -- geometry (classes.py:465-569) -------------------------------------------------------
    def _grid(self):
        ix, iy, iz = np.meshgrid(np.arange(self.nx), np.arange(self.ny), np.arange(self.nz),
                                 indexing='ij')
        return (self.csize * (ix - self.nx // 2), self.csize * (iy - self.ny // 2),
                self.csize * (iz - self.nz // 2))

    @property
    def grid_rwp(self):
        if self._rwp is None:
            xx, yy, zz = self._grid()
            self._rwp = xyz_to_rwp(xx + self.csize / 2., yy + self.csize / 2.,
                                   zz + self.csize / 2., self.params["geometry"]["inc"],
                                   self.params["geometry"]["pa"])
        return self._rwp

    @property
    def rr(self):
        if self._rr is None:
            self._rr = self.grid_rwp[0]
        return self._rr

    @property
    def ww(self):
        return self.grid_rwp[1]

    @property
    def pp(self):
        return self.grid_rwp[2]

    @property
    def rreff(self):
        g, t = self.params['geometry'], self.params['target']
        return r_eff(self.ww, t["R_1"], t["R_2"], g['w_0'], np.abs(self.rr), g['mod_r_0'],
                     g['r_0'], g["epsilon"])

    @property
    def fill_factor(self):
        """classes.py:571-769."""
        if self._ff is not None:
            return self._ff
        g = self.params['geometry']
        cs = self.csize
        xx, yy, zz = self._grid()
        n_in = np.zeros(xx.shape, dtype=int)
        verts = ((0., 0., 0.), (cs, 0., 0.), (0., cs, 0.), (cs, cs, 0.),
                 (0., 0., cs), (cs, 0., cs), (0., cs, cs), (cs, cs, cs))
        for dx, dy, dz in verts:
            rv, wv = xyz_to_rwp(xx + dx, yy + dy, zz + dz, g['inc'], g['pa'])[:2]
            wrv = w_r(rv, g['w_0'], g['mod_r_0'], g['r_0'], g['epsilon'])
            n_in = np.where((wrv >= wv) & (np.abs(rv) >= g['r_0']), n_in + 1, n_in)
        ffs = np.zeros(xx.shape)
        areas = np.zeros(xx.shape)
        ffs = np.where(n_in == 8, 1.0, ffs)
        ffs = np.where((0 < n_in) & (n_in < 8), 0.5, ffs)
        areas = np.where(0 < n_in, 1.0, areas)
        self._ff = np.where(ffs > 1e-6, ffs, np.nan)
        self._areas = np.where(areas > 1e-6, areas, np.nan)
        return self._ff

    @property
    def areas(self):
        if self._areas is None:
            _ = self.fill_factor
        return self._areas

    def _r_clamped(self, r):
        r_0 = self.params['geometry']['r_0']
        return np.where((r < r_0) & ((r + self.csize / 2.) >= r_0),
                        (r_0 + r + self.csize / 2.) / 2., r)

    def _powerlaw_field(self, zero_val, q, qd, r):
        g, t = self.params['geometry'], self.params['target']
        with np.errstate(all="ignore"):
            v = cell_value(zero_val, rho(r, g['r_0'], g['mod_r_0']), self.rreff, t["R_1"],
                           q, qd)
        v = np.where(self.fill_factor > 0, v, np.nan)
        v = np.where(v == 0, np.nan, v)
        return np.nan_to_num(v, nan=np.nan, posinf=np.nan, neginf=np.nan)

    @property
    def ts0(self):
        """classes.py:847-853 -- launch-time grid `_ts` [s]."""
        if self._ts is None:
            r = self._r_clamped(np.abs(self.rr))
            with np.errstate(all="ignore"):
                self._ts = t_rw(r, self.ww, self.params) * YEAR
        return self._ts

    @property
    def nd0(self):
        """classes.py:877-897 -- steady-state density `_nd`."""
        if self._nd is None:
            pl, pr = self.params["power_laws"], self.params["properties"]
            nd = self._powerlaw_field(pr["n_0"], pl["q_n"], pl["q^d_n"],
                                      self._r_clamped(np.abs(self.rr)))
            self._nd = np.where(self.rr < 0, nd * self._ss_jml_rb_frac, nd)
        return self._nd

    @property
    def ion_fraction(self):
        """classes.py:910-936."""
        if self._xi is None:
            pl, pr = self.params["power_laws"], self.params["properties"]
            self._xi = self._powerlaw_field(pr["x_0"], pl["q_x"], pl["q^d_x"],
                                            self._r_clamped(np.abs(self.rr)))
        return self._xi

    @property
    def temperature(self):
        """classes.py:942-969, including the unit quirk: r is converted to cm before the
        comparison with r_0 [au] and before rho()."""
        if self._temp is None:
            pl, pr = self.params["power_laws"], self.params["properties"]
            r = np.abs(self.rr) * AU * 1e2
            self._temp = self._powerlaw_field(pr["T_0"], pl["q_T"], pl["q^d_T"],
                                              self._r_clamped(r))
        return self._temp

    @property
    def vel(self):
        """classes.py:1009-1095 -- (vx, vy + v_lsr, vz) [km/s]."""
        g, t = self.params['geometry'], self.params['target']
        pl, pr = self.params["power_laws"], self.params["properties"]
        vz = self._powerlaw_field(pr["v_0"], pl["q_v"], pl["q^d_v"],
                                  self._r_clamped(np.abs(self.rr))) * np.sign(self.rr)
        with np.errstate(all="ignore"):
            vr = v_rot(self.rr, self.rreff, rho(self.rr, g['r_0'], g['mod_r_0']),
                       g['epsilon'], t['M_star'])
        sgn = 1 if g["rotation"].lower() == 'ccw' else -1
        vx = -vr * np.sin(self.pp) * sgn
        vy = vr * np.cos(self.pp) * sgn
        vx = np.where(self.fill_factor > 0., vx, np.nan)
        vy = np.where(self.fill_factor > 0., vy, np.nan)
        vz = np.where(self.fill_factor > 0., vz, np.nan)
        vxs, vys, vzs = xyz_rotate(vx, vy, vz, 90. - g["inc"], -g["pa"], order='xy')
        return vxs, vys + t["v_lsr"], vzs

    @property
    def vy(self):
        if self._vy is None:
            self._vy = self.vel[1]
        return self._vy

    # -- time dependence (classes.py:399-463, 838-875) ---------------------------------------
    def _jml(self, which, t):
        ss = self._ss_jml_bj if which == 'B' else self._ss_jml_rj
        jml = ss
        for t_0, peak, hl in self.bursts[which]:
            amp = peak - ss
            sigma = hl * 2. / (2. * np.sqrt(2. * np.log(2.)))
            jml = jml + amp * np.exp(-(t - t_0) ** 2. / (2. * sigma ** 2.))
        return jml

    @property
    def ts(self):
        return self.time - self.ts0

    @property
    def chi_xyz(self):
        ts = self.ts
        return np.where(self.rr < 0, self._jml('R', ts) / self._ss_jml_rj,
                        self._jml('B', ts) / self._ss_jml_bj)

    @property
    def number_density(self):
        return self.nd0 * self.chi_xyz

    # -- RT (classes.py:1101-1541) -----------------------------------------------------------
    def emission_measure(self):
        """classes.py:1116-1120."""
        ems = (self.number_density * self.ion_fraction) ** 2. * \
              (self.csize * AU / PARSEC * (self.fill_factor / self.areas))
        return np.nansum(ems, axis=1)

    def _tau_ff_scalar(self, freq, n_es, collapse=True):
        if self.params['power_laws']['q_T'] == 0.:
            g = gff(freq, self.params['properties']['T_0'])
        else:
            g = 11.95 * self.temperature ** 0.15 * freq ** -0.1
        tff = (0.018 * self.temperature ** -1.5 * freq ** -2. * n_es ** 2. *
               (self.csize * AU * 1e2 * (self.fill_factor / self.areas)) * g)
        if collapse:
            tff = np.nansum(tff, axis=1)
        return tff

    def optical_depth_ff(self, freq, collapse=True):
        """classes.py:1375-1447."""
        n_es = self.number_density * self.ion_fraction
        if not np.isscalar(freq):
            return np.array([self._tau_ff_scalar(nu, n_es, collapse) for nu in freq])
        return self._tau_ff_scalar(freq, n_es, collapse)

    def intensity_ff(self, freq):
        """classes.py:1466-1496."""
        ts = self.temperature
        if not np.isscalar(freq):
            return np.array([self.intensity_ff(nu) for nu in freq])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            temp_b = np.nanmean(np.where(ts > 0., ts, np.nan), axis=1) * \
                (1. - np.exp(-self.optical_depth_ff(freq)))
        return 2. * freq ** 2. * K_B * temp_b / C_LIGHT ** 2.

    def solid_angle(self):
        return np.arctan((self.csize * AU) / (self.params["target"]["dist"] * PARSEC)) ** 2.

    def flux_ff(self, freq):
        """classes.py:1515-1541."""
        if not np.isscalar(freq):
            return np.array([self.flux_ff(nu) for nu in freq])
        return self.intensity_ff(freq) * np.arctan(
            (self.csize * AU) / (self.params["target"]["dist"] * PARSEC)) ** 2. / 1e-26

    def optical_depth_rrl(self, rrl, freq, lte=True, collapse=True):
        """classes.py:1159-1229."""
        element, rrl_n, rrl_dn = rrl_parser(rrl)
        rest_freq = doppler_shift(rrl_nu_0(element, rrl_n, rrl_dn), self.vy)
        n_es = self.number_density * self.ion_fraction
        fwhm_thermal = deltanu_g(rest_freq, self.temperature, element)
        fn1n2 = f_n1n2(rrl_n, rrl_dn)
        en = energy_n(rrl_n, element)
        z_atom = z_number(element)
        fwhm_stark = deltanu_l(n_es, rrl_n, rrl_dn)
        phi_v = phi_voigt_nu(rest_freq, fwhm_stark, fwhm_thermal)

        def one(f):
            with np.errstate(all="ignore"):
                kap = kappa_l(f, rrl_n, fn1n2, phi_v(f), n_es, ni_from_ne(n_es, element),
                              self.temperature, z_atom, en)
                taus = kap * (self.csize * AU * 1e2 * (self.fill_factor / self.areas))
            return np.nansum(taus, axis=1) if collapse else taus

        if not np.isscalar(freq):
            return np.array([one(f) for f in freq])
        return one(freq)

    def intensity_rrl(self, rrl, freq, lte=True):
        """classes.py:1254-1290, scalar branch (the array branch is broken in the reference,
        1266-1271, and is unreachable from Pipeline)."""
        if not lte:
            raise ValueError("Non-LTE RRL calculations not yet supported")
        assert np.isscalar(freq)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            av_temp = np.nanmean(np.where(self.temperature > 0., self.temperature, np.nan),
                                 axis=1)
        tau_rrl = self.optical_depth_rrl(rrl, freq, lte=lte, collapse=True)
        tau_ff = self.optical_depth_ff(freq, collapse=True)
        with np.errstate(all="ignore"):
            return line_intensity_lte(freq, av_temp, tau_ff, tau_rrl)

    def flux_rrl(self, rrl, freq, lte=True, contsub=True):
        """classes.py:1319-1351."""
        if not np.isscalar(freq):
            return np.array([self.flux_rrl(rrl, nu, lte, contsub) for nu in freq])
        i_rrl = self.intensity_rrl(rrl, freq, lte=lte)
        fluxes = i_rrl * np.arctan((self.csize * AU) /
                                   (self.params["target"]["dist"] * PARSEC)) ** 2. / 1e-26
        if not contsub:
            fluxes = fluxes + self.flux_ff(freq)
        return fluxes


def chan_freqs(freq, bandwidth, chanwidth):
    """classes.py:1893-1900 (ContinuumRun.nchan / chan_freqs)."""
    nchan = int(bandwidth / chanwidth)
    chan1 = freq - bandwidth / 2. + chanwidth / 2.
    return chan1 + np.arange(nchan) * chanwidth


def pipeline_flux_result(fluxes, obs_type):
    """classes.py:2461-2472."""
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if obs_type == 'continuum':
            return np.nansum(np.nanmean(fluxes, axis=0))
        return np.nansum(np.nansum(fluxes, axis=1), axis=1)


def dense_from_sparse(shape, idx, vals, fill=np.nan):
    a = np.full(int(np.prod(shape)), fill, dtype=np.float64)
    a[idx] = vals
    return a.reshape(shape)
