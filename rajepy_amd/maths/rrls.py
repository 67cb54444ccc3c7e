"""Radio-recombination-line scalars, LTE path (reference: maths/rrls.py).  The per-cell
Voigt profile and absorption coefficient are evaluated on the GPU (csrc/rrl_scan.hip); this
module supplies the per-line constants the kernel takes (include/rjprt.h `rjp_line`)."""
import numpy as np

from .. import _constants as con
from . import physics as phys

_DN = {'a': 1, 'b': 2, 'g': 3, 'd': 4}
_M_DN = {1: 0.190775, 2: 0.026332, 3: 0.0081056, 4: 0.0034918}
_MASS_FRACTIONS = {'H': 0.710, 'He': 0.276, 'CNO': 0.014}     # Nieva & Przybilla (2012)


def rrl_parser(rrl_str):
    """'H58a' -> ('H', 58, 1) (rrls.py:605-624)."""
    element = ''.join(ch for ch in rrl_str[:-1] if ch.isalpha())
    n = int(''.join(ch for ch in rrl_str[:-1] if not ch.isalpha()))
    return element, n, _DN[rrl_str[-1].lower()]


def rrl_nu_0(atom, n, delta_n=1):
    """Rest frequency [Hz] (rrls.py:14-29)."""
    return phys.rydberg_constant(atom) * con.c * phys.z_number(atom) ** 2. * \
        (1. / n ** 2. - 1. / (n + delta_n) ** 2.)


def energy_n(n, atom):
    """Level energy [erg] (rrls.py:32-41)."""
    return -2.17989724e-11 * phys.z_number(atom) ** 2. / n ** 2.


def f_n1n2(n_1, delta_n):
    """Oscillator strength (rrls.py:44-59)."""
    return n_1 * _M_DN[delta_n] * (1. + 1.5 * delta_n / n_1)


def ni_from_ne(n_e, atom='H'):
    """Ion density from electron density (rrls.py:62-83)."""
    mu = (_MASS_FRACTIONS['H'] / phys.atomic_mass("H") * con.u +
          _MASS_FRACTIONS['He'] / phys.atomic_mass("He") * con.u +
          _MASS_FRACTIONS['CNO'] / 14.24) ** -1.
    return _MASS_FRACTIONS[atom] * n_e * mu / (phys.atomic_mass(atom) / con.u)


def deltanu_l(n_e, n, delta_n, gamma=4.5):
    """Stark (Lorentzian) FWHM [Hz] (rrls.py:86-101)."""
    return 8.2 * n_e * (n / 100.) ** gamma * (1. + gamma / 2. * delta_n / n)


def deltanu_g(nu_0, temp, atom):
    """Thermal (Gaussian) FWHM [Hz] (rrls.py:104-118)."""
    return np.sqrt(4. * np.log(2.) * 2. * con.k * temp /
                   (phys.atomic_mass(atom) * con.c ** 2.)) * nu_0


def line_constants(rrl):
    """Everything about one line that does not depend on the cell or the channel, in the
    form `rjp_line` wants:
      nu_rest, kG = deltanu_g/(nu0 sqrt T), kL = deltanu_l/n_e,
      kappa0 = 1.0991132675738456e-17 n^2 f (n_i/n_e), Z^2 E_n / k_cgs, h_cgs / k_cgs."""
    element, n, dn = rrl_parser(rrl)
    z = phys.z_number(element)
    return dict(
        nu_rest=rrl_nu_0(element, n, dn),
        kG=float(deltanu_g(1.0, 1.0, element)),
        kL=float(deltanu_l(1.0, n, dn)),
        kappa0=1.0991132675738456e-17 * n ** 2. * f_n1n2(n, dn) * ni_from_ne(1.0, element),
        en_over_k=z ** 2. * energy_n(n, element) / con.k_cgs,
        h_over_k=con.h_cgs / con.k_cgs,
    )
