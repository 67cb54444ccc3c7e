"""rajepy_amd -- MI355X-native line-of-sight radiative-transfer core behind RaJePy's
JetModel / Pipeline API.  All grid arithmetic runs in hand-written HIP kernels
(rajepy_amd/csrc -> librjprt.so, C-ABI in include/rjprt.h); there is no CPU fallback."""
__version_info__ = (0, 1, 0)
__version__ = '.'.join(map(str, __version_info__))

from . import _constants as cnsts  # noqa: F401,E402
