"""Parameter-file schema checks, frequency strings and FITS axis ordering (reference:
miscellaneous/functions.py).  Validators RETURN the exception (callers raise it), as the
reference's do (classes.py:136-138)."""
from collections.abc import Iterable

import numpy as np

_FLOAT = (float, np.floating)
_INT = (int, np.integer)
_STR = (str, np.str_)

MODEL_SCHEMA = {
    'target': (('name', str), ('ra', str), ('dec', str), ('epoch', str), ('dist', float),
               ('v_lsr', float), ('M_star', float), ('R_1', float), ('R_2', float)),
    'grid': (('n_x', int), ('n_y', int), ('n_z', int), ('l_z', float), ('c_size', float)),
    'geometry': (('epsilon', float), ('opang', float), ('w_0', float), ('r_0', float),
                 ('inc', float), ('pa', float), ('rotation', str)),
    'power_laws': (('q_v', float), ('q_T', float), ('q_x', float), ('q^d_n', float),
                   ('q^d_T', float), ('q^d_v', float), ('q^d_x', float)),
    'properties': (('v_0', float), ('x_0', float), ('T_0', float), ('mu', float),
                   ('mlr_bj', float), ('mlr_rj', float)),
    'ejection': (('t_0', (np.ndarray, _FLOAT)), ('hl', (np.ndarray, _FLOAT)),
                 ('chi', (np.ndarray, _FLOAT)), ('which', (np.ndarray, _STR))),
}

_BAND = (('times', (np.ndarray, _FLOAT)), ('t_obs', (np.ndarray, _INT)),
         ('tscps', (np.ndarray, np.ndarray)), ('t_ints', (np.ndarray, _INT)),
         ('bws', (np.ndarray, _FLOAT)), ('chanws', (np.ndarray, _FLOAT)))
PIPELINE_SCHEMA = {
    'min_el': float,
    'dcys': (('model_dcy', str),),
    'continuum': _BAND + (('freqs', (np.ndarray, _FLOAT)),),
    'rrls': _BAND + (('lines', (np.ndarray, _STR)),),
}


def _param_key_check(params, schema):
    """Same messages and checks as miscellaneous/functions.py:46-89."""
    for section, spec in schema.items():
        if section not in params:
            return KeyError("{} keyword not found in params dict".format(section))
        if isinstance(spec, type):
            if not isinstance(params[section], spec):
                return ValueError("value of {} section of params must be of type {}, not {}"
                                  "".format(section, spec, type(params[section])))
            continue
        for key, typ in spec:
            if key not in params[section]:
                return KeyError("{} keyword not found in {} section of params dict"
                                "".format(key, section))
            val = params[section][key]
            if val is None:
                continue
            if isinstance(typ, type):
                if not isinstance(val, typ):
                    return ValueError("{} value of {} section of params must be of type {}, "
                                      "not {}".format(key, section, typ, type(val)))
            else:
                if not isinstance(val, Iterable):
                    return ValueError("{} value of {} section of params must be of type {}, "
                                      "not {}".format(key, section, typ[0], type(val)))
                if len(val) != 0 and not isinstance(val[0], typ[1]):
                    return ValueError("{} of params's section {}'s value, {}, must contain "
                                      "objects of type {}, not {}".format(typ[0], section, key,
                                                                          typ[1], type(val[0])))
    return None


def parse_sexagesimal(s):
    """'[+-]AA:BB:CC.C' -> AA + BB/60 + CC/3600 with the sign applied."""
    s = s.strip()
    sign = -1.0 if s.startswith('-') else 1.0
    parts = s.lstrip('+-').replace(' ', ':').split(':')
    if len(parts) != 3:
        raise ValueError("bad sexagesimal coordinate " + repr(s))
    a, b, c = (float(p) for p in parts)
    if not (0 <= b < 60 and 0 <= c < 60):
        raise ValueError("bad sexagesimal coordinate " + repr(s))
    return sign * (a + b / 60. + c / 3600.)


def sexagesimal_to_deg(ra, dec):
    """'HH:MM:SS.S', '+DD:MM:SS.S' -> degrees.  Raises ValueError on malformed input."""
    h, d = parse_sexagesimal(ra), parse_sexagesimal(dec)
    if not (0 <= h < 24 and -90 <= d <= 90):
        raise ValueError("coordinate out of range")
    return h * 15., d


def check_model_params(params):
    """miscellaneous/functions.py:127-190.  One deliberate leniency: `properties.n_0` is
    optional -- the reference's own example file omits it (files/example-model-params.py:
    44-50) although its validator demands it, and JetModel recomputes it anyway
    (classes.py:234-242)."""
    if not isinstance(params, dict):
        return TypeError("model params must be dict")
    e = _param_key_check(params, MODEL_SCHEMA)
    if isinstance(e, Exception):
        return e
    try:
        if params['target']['epoch'].upper() not in ('J2000', 'B1950'):
            return ValueError("Only epochs B1950 and J2000 are supported as values for epoch "
                              "within model parameters' target specifications")
        sexagesimal_to_deg(params["target"]["ra"], params["target"]["dec"])
    except ValueError:
        return ValueError("Please check validity of sexagesimal coordinates within ra/dec "
                          "fields of target section of model params, as well as a valid "
                          "value for frame")
    return None


def check_pline_params(params):
    """miscellaneous/functions.py:92-124."""
    if not isinstance(params, dict):
        return TypeError("model params must be dict")
    e = _param_key_check(params, PIPELINE_SCHEMA)
    if isinstance(e, Exception):
        return e
    for band in ('continuum', 'rrls'):
        shape = np.shape(params[band]['tscps'])
        if shape != (0,) and shape != () and shape[1] != 2:
            return ValueError("np.ndarray of params's section {}'s value, tscps, must be of "
                              "shape (n, 2)".format(band))
    return None


_UNITS = (('Hz', 1.), ('kHz', 1e3), ('MHz', 1e6), ('GHz', 1e9), ('THz', 1e12), ('PHz', 1e15))


def freq_str(freq, fmt='.0f'):
    """'5GHz'-style strings (miscellaneous/functions.py:193-233)."""
    def one(f):
        for unit, lo in _UNITS:
            if lo <= f < lo * 1e3:
                return '{:{}}{}'.format(f / lo, fmt, unit)
        raise ValueError("frequency out of range: {}".format(f))
    if isinstance(freq, Iterable):
        return [one(f) for f in freq]
    return one(freq)


def reorder_axes(data, ra_axis, dec_axis, axis3=None, axis4=None, axis3_type=None,
                 axis4_type=None):
    """Copy of `data` with axes in FITS order (..., dec, ra)
    (miscellaneous/functions.py:236-301)."""
    order = [dec_axis, ra_axis]
    if axis3 is not None:
        order.insert(0, axis3)
        if axis4 is not None:
            order.insert(0, axis4)
    return np.ascontiguousarray(np.transpose(np.asarray(data), order))


def is_iter(x):
    return isinstance(x, Iterable)
