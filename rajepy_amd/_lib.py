"""ctypes binding of librjprt.so (include/rjprt.h).

There is NO CPU fallback: if the shared library is missing or a call fails this module
raises.  Build the library with ``python -c "import __graft_entry__ as g; g.build()"`` or
``rajepy_amd/csrc/build.sh`` (hipcc, --offload-arch=gfx950).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# One debug flag: only with RJP_DEBUG=1 in the environment do the experiment variables count
# (RJP_LIB = path of an experimental build of the same library, e.g. csrc/build.sh
# --debug-switches; RJP_NO_COMPACT, see engine.py).  Never a CPU path.
DEBUG = os.environ.get("RJP_DEBUG") == "1"
LIB_PATH = (DEBUG and os.environ.get("RJP_LIB")) or os.path.join(_HERE, "librjprt.so")

RJP_F32, RJP_F64 = 4, 8
RJP_GFF_SCALAR, RJP_GFF_POWERLAW = 0, 1
RJP_MAX_EPOCH_TILE = 32
RJP_RANGE_BLOCKS = 2048
RJP_VERSION = 109             # include/rjprt.h; the binding below matches exactly this ABI
RJP_OK = 0
RJP_ERR_ARG, RJP_ERR_HIP, RJP_ERR_NODEVICE, RJP_ERR_WORKSPACE, RJP_ERR_DEGENERATE = \
    -1, -2, -3, -4, -5


class RjprtError(RuntimeError):
    """A librjprt call returned a negative status (`.status`, enum rjp_status)."""

    def __init__(self, message, status=None):
        super().__init__(message)
        self.status = status


class Fields(C.Structure):
    _fields_ = [("d_nd", C.c_void_p), ("d_xi", C.c_void_p), ("d_temp", C.c_void_p),
                ("d_pf", C.c_void_p), ("d_ts", C.c_void_p), ("d_vy", C.c_void_p),
                ("nx", C.c_int32), ("ny", C.c_int32), ("nz", C.c_int32),
                ("dtype", C.c_int32), ("csize_au", C.c_double),
                ("d_ylo", C.c_void_p), ("d_yhi", C.c_void_p), ("d_em0", C.c_void_p),
                ("d_a0", C.c_void_p), ("a0_mode", C.c_int32), ("reserved_", C.c_int32),
                ("ts_lo", C.c_double), ("ts_hi", C.c_double), ("occupied_cells", C.c_int64),
                ("d_lt_cells", C.c_void_p), ("d_lt_rowoff", C.c_void_p),
                ("d_lt_aux", C.c_void_p), ("lt_K", C.c_int32), ("reserved2_", C.c_int32),
                ("d_mom_cache", C.c_void_p), ("mom_cache_K", C.c_int32),
                ("mom_cache_N", C.c_int32)]


class Bursts(C.Structure):
    """rjp_bursts: per-jet counts + host pointers to n[j] doubles each.  The arrays the
    pointers refer to are kept alive in `_keep` (engine.make_bursts)."""
    _fields_ = [("n", C.c_int32 * 2),
                ("t0", C.POINTER(C.c_double) * 2),
                ("amp_rel", C.POINTER(C.c_double) * 2),
                ("inv2s2", C.POINTER(C.c_double) * 2)]


class Line(C.Structure):
    _fields_ = [("nu_rest", C.c_double), ("kG", C.c_double), ("kL", C.c_double),
                ("kappa0", C.c_double), ("en_over_k", C.c_double), ("h_over_k", C.c_double)]


class Geometry(C.Structure):
    _fields_ = [("nx", C.c_int32), ("ny", C.c_int32), ("nz", C.c_int32),
                ("rotation_ccw", C.c_int32), ("csize", C.c_double),
                ("inc", C.c_double), ("pa", C.c_double),
                ("w_0", C.c_double), ("r_0", C.c_double), ("mod_r_0", C.c_double),
                ("epsilon", C.c_double), ("R_1", C.c_double), ("R_2", C.c_double),
                ("M_star", C.c_double), ("v_lsr", C.c_double),
                ("n_0", C.c_double), ("x_0", C.c_double), ("T_0", C.c_double),
                ("v_0", C.c_double),
                ("q_n", C.c_double), ("q_x", C.c_double), ("q_T", C.c_double),
                ("q_v", C.c_double),
                ("qd_n", C.c_double), ("qd_x", C.c_double), ("qd_T", C.c_double),
                ("qd_v", C.c_double), ("rb_frac", C.c_double),
                ("ix0", C.c_int32), ("nx_total", C.c_int32)]


_P = C.c_void_p
_DP = C.POINTER(C.c_double)

# name -> (restype, argtypes); every symbol include/rjprt.h declares
SIGNATURES = {
    "rjp_version": (C.c_int, []),
    "rjp_device_count": (C.c_int, []),
    "rjp_ctx_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "rjp_ctx_destroy": (C.c_int, [_P]),
    "rjp_last_error": (C.c_char_p, [_P]),
    "rjp_pack_field": (C.c_int, [_P, _P, _P, _P, _P, C.c_int64, C.c_int, _P]),
    "rjp_compact_fields": (C.c_int, [_P, C.POINTER(Fields), _P, _P, _P]),
    "rjp_tau_field": (C.c_int, [_P, C.POINTER(Fields), C.c_int32, _P, _P]),
    "rjp_tavg": (C.c_int, [_P, C.POINTER(Fields), _P, _P, C.c_size_t, _P]),
    "rjp_field_range": (C.c_int, [_P, _P, C.c_int64, C.c_int, _P, _P]),
    "rjp_unmask_launch_times": (C.c_int, [_P, C.POINTER(Fields), C.c_int32, _P, _P]),
    "rjp_y_bounds": (C.c_int, [_P, C.POINTER(Fields), _P, _P, _P]),
    "rjp_occupied_cells": (C.c_int, [_P, _P, _P, C.c_int64, C.POINTER(C.c_int64), _P]),
    "rjp_ff_scan_workspace": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "rjp_ff_scan": (C.c_int, [_P, C.POINTER(Fields), C.POINTER(Bursts), _DP, C.c_int32,
                              C.c_int32, _P, _P, _P, _P, C.c_size_t, _P]),
    "rjp_range_guard": (C.c_int, [_P]),
    "rjp_last_scan_path": (C.c_int, [_P, _DP, C.POINTER(C.c_int32)]),
    "rjp_last_table_build_ms": (C.c_double, [_P]),
    "rjp_moment_cache_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "rjp_lt_rowoff_entries": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "rjp_lt_count": (C.c_int, [_P, C.POINTER(Fields), C.c_int32, _P, C.POINTER(C.c_int64), _P]),
    "rjp_lt_fill": (C.c_int, [_P, C.POINTER(Fields), C.c_int32, _P, _P, _P, _P]),
    "rjp_ff_maps_workspace": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32]),
    "rjp_ff_maps": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int32, _DP, _DP, C.c_int32,
                              _P, _P, _P, _P, C.c_size_t, _P]),
    "rjp_ff_step": (C.c_int, [_P, C.POINTER(Fields), C.POINTER(Bursts), _DP, C.c_int32,
                              C.c_int32, _P, _DP, _DP, C.c_int32, _P, _P, _P, _P, _P,
                              _P, C.c_size_t, _P, C.c_size_t, _P]),
    "rjp_rrl_scan": (C.c_int, [_P, C.POINTER(Fields), C.POINTER(Bursts), C.c_double,
                               C.POINTER(Line), _DP, C.c_int32, _P, _P]),
    "rjp_ff_cells": (C.c_int, [_P, C.POINTER(Fields), C.POINTER(Bursts), C.c_double, C.c_int32,
                               _DP, C.c_int32, _P, _P]),
    "rjp_rrl_cells": (C.c_int, [_P, C.POINTER(Fields), C.POINTER(Bursts), C.c_double,
                                C.POINTER(Line), _DP, C.c_int32, _P, _P]),
    "rjp_rrl_maps": (C.c_int, [_P, _P, _P, _P, _P, C.c_int64, _DP, _DP, C.c_int32,
                               _P, _P, _P, C.c_size_t, _P]),
    "rjp_build_fields": (C.c_int, [_P, C.POINTER(Geometry), C.c_int, _P, _P, _P, _P, _P,
                                   _P, _P, _P, _P, _P, _P, _P, C.c_int32, _P]),
    "rjp_synth_fields": (C.c_int, [_P, C.c_uint64, C.c_int32, C.c_int32, C.c_int64,
                                   C.c_int64, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P,
                                   C.c_int32, _P]),
    "rjp_time_ff_scan": (C.c_int, [_P, C.POINTER(Fields), C.POINTER(Bursts), _DP,
                                   C.c_int32, C.c_int32, _P, _P, _P, _P, C.c_size_t, _P,
                                   C.c_int32, _DP]),
}

_lib = None


def load():
    """Load librjprt.so (once) and declare every prototype.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch ships its own ROCm runtime (torch/lib/libamdhip64.so); it must be the one the
    # process binds, so torch is imported BEFORE librjprt.so pulls in a HIP runtime by
    # SONAME -- two runtimes in one process do not see each other's devices or pointers.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RjprtError(
            "%s not found: the HIP library has not been built (rajepy_amd/csrc/build.sh). "
            "rajepy_amd has no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.rjp_version() != RJP_VERSION:
        raise RjprtError("librjprt.so is version %d, this binding is for %d: rebuild it "
                         "(rajepy_amd/csrc/build.sh)" % (lib.rjp_version(), RJP_VERSION))
    _lib = lib
    return lib


def check(status, ctx=None, what=""):
    if status != RJP_OK:
        msg = load().rjp_last_error(ctx)
        raise RjprtError("%s failed (status %d): %s"
                         % (what or "librjprt call", status,
                            msg.decode() if msg else "?"), status=status)


def dbl_array(values):
    """Host table of float64 as a ctypes array (kept alive by the caller).  A ctypes array of
    doubles is returned as it is: a caller that repeats a step with the same tables (a sweep
    over epochs at a fixed channel list) converts them once."""
    if isinstance(values, C.Array) and values._type_ is C.c_double:
        return values
    import numpy as np
    arr = np.ascontiguousarray(values, dtype=np.float64).ravel()
    return (C.c_double * arr.size).from_buffer_copy(arr.tobytes()) if arr.size else (C.c_double * 0)()
