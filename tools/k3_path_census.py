#!/usr/bin/env python3
"""Host census of K3's per-(cell, wave) path decision on cfg3's synthetic fields (2e5 random
cells, the four waves of a 256-channel block each): which fraction of the evaluations takes the
far-field series, the plain lattice, the lattice + pole term, the centred lattice -- for today's
kernel parameters and for alternatives (profiles/r03_k3_census.md).  Restates path_code() and
cell_line() of rajepy_amd/csrc/rrl_scan.hip; costs = VALU instructions per block from the ISA.

    python tools/k3_path_census.py
"""
import os
import sys

import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rajepy_amd.maths import rrls
import bench
lc = rrls.line_constants("H66a")
rng = np.random.default_rng(1)
n = 200000
u = rng.random((6, n))
nd = 10**(5+2.5*u[0]); xi = 0.05+0.45*u[1]; T = np.full(n, 1e4); ts = 5*u[4]*bench.YEAR
red = rng.random(n) < 0.5
vy = 6.2+60*(u[5]-0.5)
# chi at t = 1 yr with the example bursts
ej = bench.EXAMPLE_BURSTS
chi = np.ones(n)
for t0, hl, c, which in zip(ej["t_0"], ej["hl"], ej["chi"], ej["which"]):
    sig = hl*bench.YEAR*2/(2*np.sqrt(2*np.log(2)))
    g = (c-1)*np.exp(-((1.0*bench.YEAR-ts) - t0*bench.YEAR)**2/(2*sig**2))
    m = (red & ("R" in which)) | (~red & ("B" in which))
    chi += np.where(m, g, 0)
ne = nd*chi*xi
nu0 = lc["nu_rest"]*(1-vy*1000/299792458.0)
fwhm_g = lc["kG"]*np.sqrt(T)*nu0
sigma = fwhm_g/2/1.1774100225154747
is2 = 1/(sigma*np.sqrt(2))
y = 0.5*lc["kL"]*ne*is2
nchan = 256
nu = lc["nu_rest"] - nchan*1e5/2 + 1e5/2 + np.arange(nchan)*1e5
print("y percentiles", np.percentile(y, [1,5,25,50,75,95,99]))
print("x scale: is2*1e5 =", np.median(is2)*1e5, " band half-width in x:", np.median(is2)*12.8e6)

def census(h, tol, far_rules, ycen=0.03, cost=None, N=8, verbose=True):
    piH = np.pi/h
    lnq = -2*piH*y
    q = np.where(y < piH, np.exp(lnq), -1.0)
    omq = -np.expm1(lnq)
    # pole needed iff x^2 < cq:  |P| <= 6 e^{y^2-x^2} q/(1-q)^2  vs tol * y/(4*(r2max+1))
    cq = y*y + lnq + np.log(6.0) - 2*np.log(omq) - np.log(0.25*y) + np.log(1/tol) + np.log(67.0)
    tot = {}
    for w in range(4):
        fl = np.arange(64*w, 64*w+64)
        fi = np.where(fl & 1, nchan-1-(fl>>1), fl>>1)
        ev = nu[fi[::2]]; od = nu[fi[1::2]]
        xmin = np.full(n, np.inf); xmax = np.zeros(n)
        for run in (ev, od):
            lo = (run.min()-nu0)*is2; hi = (run.max()-nu0)*is2
            alo, ahi = np.abs(lo), np.abs(hi)
            xmin = np.minimum(xmin, np.where((lo<=0)&(hi>=0), 0.0, np.minimum(alo,ahi)))
            xmax = np.maximum(xmax, np.maximum(alo,ahi))
        x2min = xmin*xmin; r2min = y*y + x2min
        code = np.full(n, "", dtype=object)
        done = np.zeros(n, bool)
        for name, r2thr, x2thr in far_rules:      # ordered from cheapest (largest threshold)
            m = ~done & (r2min > r2thr) & ((x2min > x2thr) | (y > 1.0))
            code[m] = name; done |= m
        m = ~done & (y < ycen); code[m] = "cen"; done |= m
        m = ~done & (q >= 0) & (x2min < cq); code[m] = "pole"; done |= m
        code[~done] = "plain"
        for k in np.unique(code):
            tot[k] = tot.get(k, 0) + np.sum(code == k)
    tot = {k: v/(4*n) for k, v in tot.items()}
    if verbose: print("h=%.2f tol=%g" % (h, tol), {k: round(v,4) for k,v in sorted(tot.items())})
    if cost:
        avg = sum(cost[k]*v for k, v in tot.items()) + cost["tail"]
        print("   avg VALU/eval = %.1f" % avg)
    return tot



if __name__ == "__main__":
    print("round 2 kernel (h = 0.6, pole bound 1e-11, far 8 / 5 terms):")
    census(0.6, 1e-11, [("far5", 144, 64), ("far8", 64, 64)],
           cost=dict(far5=35, far8=47, plain=73, pole=153, cen=191, tail=11))
    print("round 3 kernel (h = 0.675, pole bound 3e-8, far 6 / 4 terms):")
    census(0.675, 3e-8, [("far4", 196, 64), ("far6", 64, 64)],
           cost=dict(far4=31, far6=39, plain=59, pole=131, cen=157, tail=10))
