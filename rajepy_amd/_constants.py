"""Physical constants used on the RT path.

The reference takes these from ``scipy.constants`` at its pinned scipy 1.7.1 (CODATA-2018;
requirements.txt:3, classes.py:20, maths/physics.py:5).  The system scipy ships CODATA-2022,
which differs at the 1e-9 level, so the 2018 values are written out here and nothing on the
path imports ``scipy.constants``.
"""
import math

pi = math.pi
au = 149597870700.0               # m
parsec = 3.085677581491367e+16    # m
k = 1.380649e-23                  # J/K
h = 6.62607015e-34                # J s
c = 299792458.0                   # m/s
year = 31536000.0                 # s (365 d; scipy.constants.year)
m_e = 9.1093837015e-31            # kg
u = 1.6605390666e-27              # kg
Rydberg = 10973731.56816          # 1/m
epsilon_0 = 8.8541878128e-12      # F/m
G = 6.6743e-11
e = 1.602176634e-19               # C

# _constants.py:3-14 of the reference
AU2CM = au * 1e2
KM2CM = 1e5
MSOL = 1.98847e30                 # kg
# (protons, neutrons) of the isotope used per element, _constants.py:7-10
NZ = {"H": (1, 0), "He": (2, 2), "Li": (3, 4), "Be": (4, 5), "B": (5, 6), "C": (6, 6),
      "N": (7, 7), "O": (8, 8), "F": (9, 10), "Ne": (10, 10), "Na": (11, 12), "Mg": (12, 12)}

# Atomic masses in micro-u (AME2003, Audi, Wapstra & Thibault 2003) of those isotopes: what
# maths/physics.py:620-623 looks up in files/atomic_masses.pkl.
ATOMIC_MASS_MICRO_U = {
    "H": 1007825.03207, "He": 4002603.25415, "Li": 7016004.548, "Be": 9012182.201,
    "B": 11009305.406, "C": 12000000.0, "N": 14003074.00478, "O": 15994914.61956,
    "F": 18998403.224, "Ne": 19992440.17542, "Na": 22989769.28087, "Mg": 23985041.699,
}

# cgs forms used by maths/rrls.py:7-11
c_cgs = c * 1e2
h_cgs = h * 1e7
k_cgs = k * 1e7
