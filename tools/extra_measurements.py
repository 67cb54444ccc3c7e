#!/usr/bin/env python3
"""Side measurements quoted in DESIGN.md (not the bench contract): PCIe-inclusive product
hand-over, K4 field build time, end-to-end config-1 pipeline wall time."""
import json, os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from rajepy_amd import classes, logger, engine as E
from tests.test_host_logic import example_params, pline_params

out = {}
eng = E.RTEngine(0)
# D2H of the cfg4 tau+flux cubes (2 x 256 x 512 x 512 float64 = 1.07 GB)
cube = torch.empty((2, 256, 512 * 512), dtype=torch.float64, device=eng.device)
cube.fill_(1.0)
torch.cuda.synchronize()
t0 = time.perf_counter(); host = cube.cpu(); dt = time.perf_counter() - t0
out["d2h_1.07GB_pageable_s"] = dt
pinned = torch.empty(cube.shape, dtype=cube.dtype, pin_memory=True)
torch.cuda.synchronize()
t0 = time.perf_counter(); pinned.copy_(cube, non_blocking=True); torch.cuda.synchronize()
out["d2h_1.07GB_pinned_s"] = time.perf_counter() - t0
del cube, host, pinned
# K4: build the cfg4-size example-jet fields on the device
p = example_params()
p["grid"].update(n_x=512, n_y=4096, n_z=512)
tmp = tempfile.mkdtemp()
jm = classes.JetModel(p, log=logger.Log(os.path.join(tmp, "a.log"), verbose=False), engine=eng)
t0 = time.perf_counter(); dev = jm.device_fields; eng.synchronize()
out["k4_build_512x4096x512_s"] = time.perf_counter() - t0
# sparse-model shortcut: 8-epoch scan of the example jet with / without the occupied y-ranges
ep = list(np.linspace(0., 3.5, 8) * 31536000.0)
b = jm._rjp_bursts()
for tag in ("with_ybounds", "without_ybounds"):
    if tag == "without_ybounds":
        keep = (dev.ylo, dev.yhi); dev.ylo = dev.yhi = None
    eng.ff_scan(dev, b, ep, jm.gff_mode); eng.synchronize()
    t0 = time.perf_counter(); eng.ff_scan(dev, b, ep, jm.gff_mode); eng.synchronize()
    out["jet_8epoch_scan_512x4096x512_%s_s" % tag] = time.perf_counter() - t0
dev.ylo, dev.yhi = keep
t0 = time.perf_counter(); jm.time = 0.; f = jm.flux_ff(np.geomspace(1e9, 5e10, 256))
out["jetmodel_flux_ff_256ch_512x4096x512_incl_d2h_s"] = time.perf_counter() - t0
out["jet_filled_fraction"] = float(np.isfinite(f[0]).mean())
# the same call writing its 0.54 GB FITS cube: payload laid out on the GPU vs on the host
for tag, thr in (("device", 0), ("host", 1 << 62)):
    classes.JetModel.FITS_DEVICE_MIN_BYTES = thr
    path = os.path.join(tmp, "cube_%s.fits" % tag)
    t0 = time.perf_counter(); jm.flux_ff(np.geomspace(1e9, 5e10, 256), savefits=path)
    out["jetmodel_flux_ff_256ch_savefits_%s_payload_s" % tag] = time.perf_counter() - t0
    os.remove(path)
classes.JetModel.FITS_DEVICE_MIN_BYTES = 8 << 20
del jm, dev, f
# config 1 end to end (reference: 31 s incl. plots)
dcy = os.path.join(tmp, "out"); os.makedirs(dcy)
log = logger.Log(os.path.join(dcy, "model.log"), verbose=False)
t0 = time.perf_counter()
pl = classes.Pipeline(classes.JetModel(example_params(), log=log, engine=eng), pline_params(dcy), log=log)
pl.execute(simobserve=False, verbose=False, dryrun=False, resume=False, clobber=True)
out["pipeline_cfg1_3runs_wall_s"] = time.perf_counter() - t0
print(json.dumps(out))
