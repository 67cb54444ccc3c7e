"""Scalar physics helpers (reference: maths/physics.py).  Host side only -- one value per
channel or per model, never per cell."""
import functools
import os
import warnings

import numpy as np

from .. import _constants as con

_FILES = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "files")


def q_n(epsilon, q_v):
    """Density index along the jet from mass conservation (physics.py:17-36)."""
    return -q_v - 2.0 * epsilon


def q_tau(epsilon, q_x, q_n_, q_T):
    """Optical-depth index along the jet (physics.py:39-63)."""
    return epsilon + 2.0 * q_x + 2.0 * q_n_ - 1.35 * q_T


def atomic_mass(atom):
    """Mass [kg] of the isotope the reference uses for `atom` (physics.py:607-624)."""
    return con.ATOMIC_MASS_MICRO_U[atom] * (1e-6 * con.u)


def z_number(atom):
    """physics.py:523-532."""
    return con.NZ[atom][0]


def rydberg_constant(atom):
    """Finite-mass Rydberg constant [1/m] (physics.py:535-544)."""
    m = atomic_mass(atom)
    return con.Rydberg * (m / (m + con.m_e))


def doppler_shift(nu_0, v_lsr):
    """physics.py:547-558; v in km/s."""
    return nu_0 * (1. - v_lsr * 1000. / con.c)


def blackbody_nu(freq, temp):
    """Planck function in cgs (physics.py:561-574)."""
    x = con.h * 1e7 * freq / (con.k * 1e7 * temp)
    return (2. * con.h * 1e7 * freq ** 3. / (con.c * 1e2) ** 2.) / (np.exp(x) - 1.)


def _mlr_shape(q_nd, q_nv, R_1, R_2):
    a = q_nd + q_nv
    if a == -1. or a == -2.:
        a *= 1. + 1e-12                      # physics.py:442-444, 507-509
    r1, r2 = R_1 * con.au, R_2 * con.au
    return ((r1 ** 2. + r2 * (r2 * (a + 1.) - r1 * (a + 2.)) * (r2 / r1) ** a) /
            ((r2 - r1) ** 2. * (a + 1.) * (a + 2.)))


def n_0_from_mlr(mlr, v_0, w_0, mu, q_nd, q_nv, R_1, R_2):
    """Axis density at the jet base [cm^-3] for a mass-loss rate [Msol/yr]
    (physics.py:474-517)."""
    k = 2. * np.pi * (mu * atomic_mass('H')) * (v_0 * 1e3) * (w_0 * con.au) ** 2.
    return mlr * con.MSOL / con.year / k / _mlr_shape(q_nd, q_nv, R_1, R_2) / 1e6


def mlr_from_n_0(n_0, v_0, w_0, mu, q_nd, q_nv, R_1, R_2):
    """Inverse of n_0_from_mlr (physics.py:413-471)."""
    k = 2. * np.pi * (mu * atomic_mass('H')) * (n_0 * 1e6) * (v_0 * 1e3) * \
        (w_0 * con.au) ** 2.
    return k * _mlr_shape(q_nd, q_nv, R_1, R_2) / con.MSOL * con.year


@functools.lru_cache(maxsize=1)
def import_vanHoof2014():
    """van Hoof et al. (2014) thermally averaged Gaunt factors: (log gamma^2 axis,
    log u axis, table[146, 81]).  Parsed once (the reference re-reads the file per call,
    physics.py:626-663)."""
    with open(os.path.join(_FILES, "vanHoofetal2014.data"), "rt") as f:
        lines = f.readlines()
    lg2_0 = float(lines[30].split('#')[0])
    lu_0 = float(lines[31].split('#')[0])
    step = float(lines[32].split('#')[0])
    table = np.array([[float(v) for v in ln.split()] for ln in lines[42:188]])
    n_u, n_g = table.shape
    lus = np.linspace(np.round(lu_0, 1), np.round(lu_0 + step * (n_u - 1), 1), n_u)
    lg2s = np.linspace(np.round(lg2_0, 1), np.round(lg2_0 + step * (n_g - 1), 1), n_g)
    return lg2s, lus, table


def gff(freq, temp, z=1.):
    """Free-free Gaunt factor at one (frequency, temperature) (physics.py:666-698); results
    are memoised (the reference re-reads its table and refits the spline on every call)."""
    return _gff_cached(float(freq), float(temp), float(z))


@functools.lru_cache(maxsize=65536)
def _gff_cached(freq, temp, z):
    """

    The reference interpolates the 5x5 table window nearest to (log gamma^2, log u) with
    scipy's interp2d(kind='cubic') on scattered points, i.e. FITPACK surfit through
    bisplrep(kx=ky=3, s=0) + bisplev; the same two FITPACK calls are made here.  The row
    clamp deliberately reproduces physics.py:687-690 (it clamps against 81, not 146)."""
    from scipy.interpolate import bisplev, bisplrep
    ry = con.m_e * con.e ** 4. / (8 * con.epsilon_0 ** 2. * con.h ** 2.)
    lg2 = np.log10(z ** 2. * ry / (con.k * temp))
    lu = np.log10(con.h * freq / (con.k * temp))
    lg2s, lus, table = import_vanHoof2014()
    col = int(np.argmin(np.abs(lg2s - lg2)))
    row = int(np.argmin(np.abs(lus - lu)))
    col = min(max(col, 2), len(lg2s) - 3)
    row = min(max(row, 2), len(lg2s) - 3)
    gx, gy = np.meshgrid(lg2s[col - 2:col + 3], lus[row - 2:row + 3])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tck = bisplrep(gx.ravel(), gy.ravel(), table[row - 2:row + 3, col - 2:col + 3].ravel(),
                       kx=3, ky=3, s=0.0)
    return float(bisplev(lg2, lu, tck))


def v_rot(r, reff, rho, epsilon, m_star):
    """Rotation speed [km/s] (physics.py:66-90)."""
    return np.sqrt(con.G * m_star * con.MSOL / (reff * con.au)) * rho ** -epsilon / 1e3


A_K = 0.212        # Reynolds (1986) free-free absorption constant (_constants.py:13)


def tau_r(r, r_0, w_0, n_0, chi_0, T_0, freq, inc, epsilon, q_n, q_x, q_T, opang):
    """Analytic optical depth through the jet at distance r [au] along its axis, equations
    4-5 of Reynolds (1986) (physics.py:93-142).  Used as a physics cross-check of the maps
    (the reference's sed_plot does the same, plotting/functions.py:1194-1227)."""
    from . import geometry as geom
    cm = con.au * 1e2
    mr0 = geom.mod_r_0(opang, epsilon, w_0 * cm)
    q = epsilon + 2. * q_n + 2. * q_x - 1.35 * q_T
    return (2. * A_K * (w_0 * cm) * n_0 ** 2. * chi_0 ** 2. * T_0 ** -1.35 *
            geom.rho(r * cm, r_0 * cm, mr0) ** q * freq ** -2.1 / np.sin(np.radians(inc)))


A_J = 6.5e-38      # Reynolds (1986) free-free emission constant (_constants.py:14)
ARCSEC = con.pi / 648000.        # scipy.constants.arcsec [rad]


def r_tau1(r_0, w_0, n_0, chi_0, T_0, freq, inc, epsilon, q_n, q_x, q_T, opang, dist=None):
    """Distance along the jet axis of the tau = 1 surface, equation 4 of Reynolds (1986)
    (physics.py:145-236): [au], or [arcsec] when `dist` [pc] is given -- with the reference's
    own small-angle conversion r[au] / dist."""
    from . import geometry as geom
    cm = con.au * 1e2
    mr0 = geom.mod_r_0(opang, epsilon, w_0 * cm)
    q = epsilon + 2. * q_n + 2. * q_x - 1.35 * q_T
    rho = (2. * A_K * (w_0 * cm) * n_0 ** 2. * chi_0 ** 2. * T_0 ** -1.35 * freq ** -2.1 *
           np.sin(np.radians(inc)) ** -1.) ** (-1. / q)
    r = rho * mr0 + (r_0 * cm) - mr0
    if dist is None:
        return r
    return r / cm / dist


class _R86Lobe:
    """Base quantities of one lobe in cgs for the Reynolds (1986) closed forms below: jet width
    and base distances [cm], distance to the source [cm], sin(inclination), the power-law
    indices, and the base electron density n_0 x_0 [cm^-3].  The density carries the two
    adjustments the reference applies -- the red lobe's mass-loss ratio and, for a disc wind
    (q^d_n != 0), the density implied by the mass-loss rate -- in the ORDER the calling formula
    uses in the reference (the two formulas differ there, physics.py:263-273 vs 340-345; with a
    disc wind the second adjustment overrides the first or is scaled by it accordingly)."""

    ERG_TO_JY = 1e-7 * 1e2 ** 2. / 1e-26        # erg cm^-2 s^-1 Hz^-1 -> Jy

    def __init__(self, jm, which, wind_density_first):
        par = jm.params
        geo, laws, props = par['geometry'], par['power_laws'], par['properties']
        cm = con.au * 1e2
        self.width = geo['w_0'] * cm
        self.mod_r0 = geo['mod_r_0'] * cm
        self.r0 = geo['r_0'] * cm
        self.dist = par['target']['dist'] * con.parsec * 1e2
        self.sin_i = np.sin(np.radians(geo['inc']))
        self.eps, self.q_T, self.q_tau = geo['epsilon'], laws['q_T'], laws['q_tau']
        self.T0, self.x0 = props['T_0'], props['x_0']
        lobe_ratio = (jm.ss_jml('R') / jm.ss_jml('B')) if which == 'R' else 1.

        def wind_density():
            mdot = props["mlr"] * 1.989e30 / con.year
            return mdot / (np.pi * props['mu'] * atomic_mass("H") * self.width ** 2. *
                           props["v_0"] * 1e5)
        disc_wind = laws["q^d_n"] != 0.
        if wind_density_first:
            self.n0 = (wind_density() if disc_wind else props['n_0']) * lobe_ratio
        else:
            self.n0 = wind_density() if disc_wind else props['n_0'] * lobe_ratio

    def tau_base(self, freq):
        """Optical depth through the jet base (equation 4 of Reynolds 1986 at rho = 1)."""
        return (2. * A_K * self.width * (self.n0 * self.x0) ** 2. * self.T0 ** -1.35 *
                freq ** -2.1 / self.sin_i)


def approx_flux_expected_r86(jm, freq, which):
    """Approximate total flux [Jy] of one lobe, equation 16 of Reynolds (1986)
    (physics.py:239-297); `freq` scalar, list or array [Hz]; which = 'R' or 'B'."""
    freq = np.asarray(freq, dtype=float) if isinstance(freq, (list, tuple)) else freq
    lobe = _R86Lobe(jm, which, wind_density_first=True)
    s = 1. + lobe.eps + lobe.q_T
    c = s / lobe.q_tau
    alpha = 2. + 2.1 * c                                  # spectral index
    factors = (2 ** (1. - c), lobe.dist ** -2., A_J, A_K ** (-1. - c), lobe.T0 ** (1. + 1.35 * c),
               lobe.mod_r0, lobe.width ** (1. - c), (lobe.n0 * lobe.x0) ** (-2. * c),
               lobe.sin_i ** (1. + c) / (c * (s + lobe.q_tau)))
    flux = 1.
    for fac in factors:
        flux = flux * fac
    return flux * freq ** alpha * _R86Lobe.ERG_TO_JY


def flux_expected_r86(jm, freq, which, y_max, y_min=None):
    """Exact total flux [Jy] of one lobe between projected distances y_min and y_max
    [arcsec] from the jet base, equation 8 of Reynolds (1986) (physics.py:300-374).  The
    incomplete gamma function of negative order is mpmath's, as in the reference."""
    from mpmath import gammainc
    lobe = _R86Lobe(jm, which, wind_density_first=False)
    y_base = lobe.mod_r0 * lobe.sin_i                     # projected |r_0|
    shift = y_base - lobe.r0 * lobe.sin_i

    def projected(arcsec):
        return np.tan(arcsec * ARCSEC) * lobe.dist + shift
    upper = projected(y_max)
    lower = y_base if y_min is None else projected(y_min)
    tau0 = lobe.tau_base(freq)
    s = 1. + lobe.eps + lobe.q_T
    k = s / lobe.q_tau
    scale = 2. * lobe.width * lobe.dist ** -2. * A_J / A_K * lobe.T0 * freq ** 2.

    def primitive(y):
        """Antiderivative of the brightness integral at projected distance y."""
        rho = y / y_base
        tau = tau0 * rho ** lobe.q_tau
        head = y / (lobe.q_tau * s) * rho ** (s - 1.) * tau ** (-k)
        tail = lobe.q_tau * tau ** k + s * gammainc(k, tau)
        return scale * (float(head) * float(tail))
    return (primitive(upper) - primitive(lower)) * _R86Lobe.ERG_TO_JY
