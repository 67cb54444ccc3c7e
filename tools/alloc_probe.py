import time, torch
torch.cuda.init()
torch.cuda.synchronize()
n = 512*4096*512
for k in range(2):
    torch.cuda.empty_cache()
    t0 = time.perf_counter()
    bufs = [torch.empty(n, dtype=torch.float64, device="cuda") for _ in range(10)]
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print("10 x 8.6 GB torch.empty: %.3f s" % (t1 - t0))
    del bufs
import sys
sys.path.insert(0, "/root/repo")
from rajepy_amd import classes, logger
import json, numpy as np, os, tempfile
meta = json.loads(str(np.load("/root/repo/tests/golden/cfg1_example.npz")["meta"]))
par = meta["params"]
for k in ("t_0", "hl", "chi", "which"):
    par["ejection"][k] = np.array(par["ejection"][k])
par["geometry"].pop("mod_r_0", None)
for k in ("q_n", "q_tau"):
    par["power_laws"].pop(k, None)
par["properties"].pop("n_0", None)
scale = 512.0 / par["grid"]["n_x"]
par["grid"].update(n_x=512, n_y=4096, n_z=512, c_size=par["grid"]["c_size"] / scale)
log = logger.Log(os.path.join(tempfile.mkdtemp(), "run.log"), verbose=False)
import cProfile, pstats
torch.cuda.empty_cache()
pr = cProfile.Profile()
pr.enable()
t0 = time.perf_counter()
jm = classes.JetModel(par, log=log)
_ = jm.device_fields
torch.cuda.synchronize()
print("construct %.3f s" % (time.perf_counter() - t0))
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
