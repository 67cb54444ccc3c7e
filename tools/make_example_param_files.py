#!/usr/bin/env python3
"""Writes model-params.py / pipeline-params.py for the example jet (config 1) into a directory:
    python tools/make_example_param_files.py <dir>"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_host_logic import example_params
out = sys.argv[1]
p = example_params()
def lit(v): return "np.array(%r)" % v.tolist() if isinstance(v, np.ndarray) else repr(v)
body = ",\n".join("  %r: {%s}" % (sec, ", ".join("%r: %s" % (k, lit(v)) for k, v in d.items())) for sec, d in p.items())
open(out + "/model-params.py", "w").write("import numpy as np\nparams = {\n" + body + "\n}\n")
open(out + "/pipeline-params.py", "w").write("""import numpy as np
params = {'min_el': 20., 'dcys': {'model_dcy': %r},
 'continuum': {'times': np.array([0., 0.5, 1., 2., 3.]), 'freqs': np.array([5e9]), 't_obs': np.array([1200]),
   'tscps': np.array([('VLA', 'A')]), 't_ints': np.array([5]), 'bws': np.array([4e8]), 'chanws': np.array([2e8])},
 'rrls': {'times': np.array([0.]), 'lines': np.array(['H66a']), 't_obs': np.array([1200]),
   'tscps': np.array([('VLA', 'A')]), 't_ints': np.array([60]), 'bws': np.array([4e5]), 'chanws': np.array([1e5])}}
""" % (out + "/out"))
