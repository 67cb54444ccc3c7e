// K3: LTE radio-recombination-line optical-depth cube, and its map stage.
//
// The reference evaluates scipy.special.wofz over the WHOLE 3-D grid once per channel and
// recomputes every channel-independent per-cell quantity each time
// (classes.py:1159-1214; maths/rrls.py:350-354, 383-389).  Here a 256-thread workgroup owns a
// tile of z-adjacent sightlines of one x-row -- 8 sightlines for the 256-channel-lane layout,
// 16 for the 64- and 16-lane ones -- and a block of up to 256 channels:
//   phase 1  the 256 threads turn a slab of 256 cells (32 y x 8 z, or 16 y x 16 z) into
//            per-cell line constants (Doppler-shifted nu0, 1/(sigma sqrt2), Voigt y, LTE
//            prefactor, h/kT, pole-term constants) staged in LDS -- once per cell, not once
//            per channel -- and, for the layouts whose waves work on one cell at a time, into
//            one path code per (cell, wave): which of the Faddeeva paths below the whole wave
//            takes for that cell;
//   phase 2  lanes run over CHANNELS (folded about the block centre, so that a wave holds a
//            narrow |x| range): every lane reads the same cell's constants from LDS
//            (broadcast), branches on the wave's path code (a scalar), evaluates Re w(x+iy)
//            for its channel and accumulates tau in FP64; the per-(sightline, channel)
//            accumulators live in LDS, so the sightline loop is not unrolled (< 128 VGPRs,
//            4 waves per SIMD).  Cells outside the jet are skipped with a scalar branch.
// Compute-bound (vector FP64) by construction -- 102.5 VALU instructions per (cell, channel) on
// cfg3's fields (counted: profiles/r03e_cfg3_k3_sq.json), census in profiles/r03_k3_census.md; HBM traffic is 6 fields per cell, read
// once per block of 256 channels.
//
// Accuracy budget.  The wave-uniform paths (far-field series, plain lattice with or without the
// pole term, centred lattice) are designed for <= 1e-8 relative on Re w against
// scipy.special.wofz over their whole domain -- one order inside SURVEY.md section 7's 1e-7,
// three inside BASELINE.json's 1e-5 on the maps: lattice step h = 0.675 with 8 node pairs
// (3.4e-9), 6- and 4-term far-field series (4.1e-9 / 1.2e-9), pole term skipped where a
// rigorous bound puts it below 3e-8 Re w (measured <= 1.5e-9), centred lattice with 7 nodes a
// side (7e-10); tools/voigt_design.py restates each path in NumPy and prints this table.
// Rounds 1-2 held them to 1e-11 (h = 0.6, 10 pairs, 8/5 terms): 131 instead of ~102
// instructions.  The generic per-lane path (16-lane layout, collapse=False, irregular cells)
// keeps h = 0.6 / 10 pairs / 1e-11.
#include "rjp_host.h"

namespace rjp {

constexpr int kRB = 256;     // threads per workgroup

// ---- Faddeeva: Re w(x + i y), y > 0 ----------------------------------------------------
// Core: trapezoidal rule with step h on w(z) = (i/pi) Int exp(-t^2)/(z-t) dt plus the residue
// ("pole") correction for y < pi/h (Matta & Reichel 1971; Hunter & Regan 1972).
//  * y >= 0.03: plain lattice t = n h, nodes paired (+t,-t) to halve the divisions;
//  * y < 0.03, where sum and pole term of the plain lattice would cancel near a node:
//    - kernels whose waves work on one cell: lattice centred on x (voigt_centred below);
//    - otherwise: lattice shifted by h/2 whenever x is within h/4 of a node.
// Generic per-lane code (voigt_rew): h = 0.6, 10 node pairs: relative error of Re w < 1e-11
// for 1e-10 <= y <= 1e3, 0 <= x <= 1e4 (measured against scipy.special.wofz, which the reference
// calls); far field (|z|^2 > 64 and (x^2 > 64 or y > 1)): 6-level Laplace continued fraction,
// relative error < 3e-10 there.
constexpr double kH = 0.6;
constexpr int kNPair = 10;
// Wave-uniform paths (voigt_plain_wave, voigt_centred, voigt_far_series): h = 0.675, 8 node
// pairs: <= 1e-8 (tools/voigt_design.py)
constexpr double kHW = 0.675;
constexpr int kNPairW = 8;
// node tables for the two lattices (delta = 0 and delta = 1/2): tau = t^2, w = 2 exp(-tau)
// (the self-paired node t = 0 carries half weight), wt = w tau
__device__ __constant__ double c_tau0[kNPair] = {0.0, 0.36, 1.44, 3.2399999999999993, 5.76, 9.0, 12.959999999999997, 17.64, 23.04, 29.159999999999993};
__device__ __constant__ double c_tau1[kNPair] = {0.09, 0.8099999999999998, 2.25, 4.41, 7.289999999999998, 10.889999999999999, 15.209999999999999, 20.25, 26.009999999999998, 32.49};
__device__ __constant__ double c_w0[kNPair] = {1.0, 1.395352652142062, 0.47385551736424353, 0.0783277901979742, 0.006302223196888882, 0.0002468196081733591, 4.705150400019559e-06, 4.3659155902509556e-08, 1.9719011151983032e-10, 4.3351377652379543e-13};
__device__ __constant__ double c_w1[kNPair] = {1.8278623705424564, 0.8897161324458824, 0.21079844912372867, 0.02431035665982987, 0.0013646561055127555, 3.72874846630337e-05, 4.959192036090064e-07, 3.2104561103712233e-09, 1.0116505485687606e-11, 1.5516804151392108e-14};
__device__ __constant__ double c_wt0[kNPair] = {0.0, 0.5023269547711423, 0.6823519450045107, 0.25378204024143636, 0.03630080561407996, 0.002221376473560232, 6.097874918425347e-05, 7.701475101202686e-07, 4.54326016941689e-09, 1.2641261723433871e-11};
__device__ __constant__ double c_wt1[kNPair] = {0.16450761334882108, 0.7206700672811647, 0.4742965105283895, 0.10720867286984972, 0.009948343009187986, 0.000406060707980437, 7.542931086892987e-06, 6.501173623501727e-08, 2.631303076827346e-10, 5.041409668787296e-13};

// sin(2 pi u), cos(2 pi u): quarter-turn reduction, Taylor polynomials on |w| <= pi/4
// (truncation < 2e-14 / 1e-15).  |u| < 2^30.
__device__ __forceinline__ void sincos_2pi(double u, double& sn, double& cs) {
  const double k = __builtin_rint(4.0 * u);
  const double w = 6.28318530717958647692 * __builtin_fma(-0.25, k, u);
  const double w2 = w * w;
  double ps = -7.6471637318198164759e-13;                 // -1/15!
  ps = __builtin_fma(ps, w2, 1.6059043836821614599e-10);  //  1/13!
  ps = __builtin_fma(ps, w2, -2.5052108385441718775e-08); // -1/11!
  ps = __builtin_fma(ps, w2, 2.7557319223985890653e-06);  //  1/9!
  ps = __builtin_fma(ps, w2, -1.9841269841269841253e-04); // -1/7!
  ps = __builtin_fma(ps, w2, 8.3333333333333332177e-03);  //  1/5!
  ps = __builtin_fma(ps, w2, -1.6666666666666665741e-01); // -1/3!
  const double s0 = __builtin_fma(ps * w2, w, w);
  double pc = 4.7794773323873852974e-14;                  //  1/16!
  pc = __builtin_fma(pc, w2, -1.1470745597729724714e-11); // -1/14!
  pc = __builtin_fma(pc, w2, 2.0876756987868098979e-09);  //  1/12!
  pc = __builtin_fma(pc, w2, -2.7557319223985888276e-07); // -1/10!
  pc = __builtin_fma(pc, w2, 2.4801587301587301566e-05);  //  1/8!
  pc = __builtin_fma(pc, w2, -1.3888888888888889419e-03); // -1/6!
  pc = __builtin_fma(pc, w2, 4.1666666666666664354e-02);  //  1/4!
  pc = __builtin_fma(pc, w2, -0.5);
  const double c0 = __builtin_fma(pc, w2, 1.0);
  const int q = (int)k & 3;
  const double a = (q & 1) ? c0 : s0;       // q=0: s,c  q=1: c,-s  q=2: -s,-c  q=3: -c,s
  const double b = (q & 1) ? s0 : c0;
  sn = (q & 2) ? -a : a;
  cs = (q == 1 || q == 2) ? -b : b;
}

// cos(2 pi u) alone: half-turn reduction, one even polynomial on |w| <= pi/2 (truncation
// 2e-17), no quadrant selects.  |u| < 2^30.
__device__ __forceinline__ double cos_2pi(double u) {
  const double k = __builtin_rint(2.0 * u);
  const double w = 6.28318530717958647692 * __builtin_fma(-0.5, k, u);
  const double w2 = w * w;
  double p = 4.1103176233121648585e-19;                  //  1/20!
  p = __builtin_fma(p, w2, -1.5619206968586226462e-16);  // -1/18!
  p = __builtin_fma(p, w2, 4.7794773323873852974e-14);   //  1/16!
  p = __builtin_fma(p, w2, -1.1470745597729724714e-11);  // -1/14!
  p = __builtin_fma(p, w2, 2.0876756987868098979e-09);   //  1/12!
  p = __builtin_fma(p, w2, -2.7557319223985888276e-07);  // -1/10!
  p = __builtin_fma(p, w2, 2.4801587301587301566e-05);   //  1/8!
  p = __builtin_fma(p, w2, -1.3888888888888889419e-03);  // -1/6!
  p = __builtin_fma(p, w2, 4.1666666666666664354e-02);   //  1/4!
  p = __builtin_fma(p, w2, -0.5);
  p = __builtin_fma(p, w2, 1.0);
  return ((int)k & 1) ? -p : p;
}

// cos(2 pi u) of three arguments in lockstep: half-turn reduction, one even polynomial on
// |w| <= pi/2 (truncation 2e-17); the coefficients are shared and live in SGPRs (see fma_k
// below -- declared here because the pole term of the wave-uniform lattice uses it).
__device__ __forceinline__ double fma_k(double a, double b, double K);
__device__ __forceinline__ double kfma(double K, double b, double c);
__device__ __forceinline__ double kadd(double K, double b);
__device__ __forceinline__ double kmul(double K, double b);
// The top coefficients of the pole term's two polynomials, held in VGPRs for the whole kernel
// (set once through an asm statement, so the compiler can neither fold nor rematerialise
// them): with one operand in a VGPR the first Horner step is ONE FMA with the next
// coefficient as its SGPR operand, instead of a multiply and an add (a VOP3 instruction
// reads at most one SGPR pair).  Four instructions per pole-term evaluation for 4 VGPRs.
struct PoleTop { double cos_top, exp_top; };
__device__ __forceinline__ PoleTop pole_top() {
  PoleTop t;
  // (the coefficients of w^10 and r^7 of the two near-minimax fits below, in the variables the
  // polynomials now run in: half-turns d = w / pi and binary exponents f = r / ln 2)
  asm volatile("v_mov_b64 %0, %1" : "=v"(t.cos_top) : "s"(-2.46275154502513423e-02));
  asm volatile("v_mov_b64 %0, %1" : "=v"(t.exp_top) : "s"(1.33498754716926590e-05));
  return t;
}

__device__ __forceinline__ void cos_2pi_x3(double h0, double h1, double h2, double top,
                                           double& c0, double& c1, double& c2) {
  // the arguments come in HALF-turns (h = 2 u): k = rint(h), d = h - k in [-1/2, 1/2],
  // cos(2 pi u) = (-1)^k cos(pi d), and the polynomial runs in d^2 with pi^2j folded into its
  // coefficients -- no multiplication by 2 before the rounding, none by 2 pi after it
  const double h[3] = {h0, h1, h2};
  double k[3], d2[3], p[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    k[i] = __builtin_rint(h[i]);
    const double d = h[i] - k[i];
    d2[i] = d * d;
  }
  // near-minimax on |w| <= pi/2 (w = pi d) with the two leading coefficients kept at 1, -1/2:
  // degree 10, max abs error 1.1e-9 (tools/minimax_fit.py) -- the worst error of
  // the whole path stays the lattice's 3.1e-9 (tools/voigt_design.py; the pole term enters
  // Re w amplified by at most ~10 where sum and pole term cancel, and is itself <= 1e-1 of
  // it there).  Degree 12 (3.9e-12) in the first half of round 3, 14 in round 2, a degree-20
  // Taylor polynomial in round 1.  Coefficients times pi^8, pi^6, pi^4, pi^2 here.
  constexpr double cf[4] = {2.35081807469869619e-01, -1.33524270773875631e+00,
                            4.05871167107504327e+00, -4.93480220054467900e+00};
#pragma unroll
  for (int i = 0; i < 3; ++i) p[i] = fma_k(top, d2[i], cf[0]);
#pragma unroll
  for (int j = 1; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 3; ++i) p[i] = fma_k(p[i], d2[i], cf[j]);
#pragma unroll
  for (int i = 0; i < 3; ++i) p[i] = __builtin_fma(p[i], d2[i], 1.0);
  // (-1)^k: the parity of k is ADDED into the sign bit of the high word (shift-and-add is one
  // instruction; the carry out of bit 31 is dropped)
  auto flip = [](double v, double kk) __attribute__((always_inline)) {
    const uint64_t b = __builtin_bit_cast(uint64_t, v);
    const uint32_t hi = ((uint32_t)(int)kk << 31) + (uint32_t)(b >> 32);
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | (uint32_t)b);
  };
  c0 = flip(p[0], k[0]);
  c1 = flip(p[1], k[1]);
  c2 = flip(p[2], k[2]);
}

// exp(-x2) for 0 <= x2 <= 700 with the polynomial constants in SGPRs: t = -x2 log2(e) = k + f,
// |f| <= 1/2, 2^f from the degree-7 near-minimax polynomial of exp on |r| <= ln2/2 (1, 1, 1/2
// kept; relative error 3.9e-10, tools/minimax_fit.py; degree 8 / 1.6e-12 before -- the path's
// worst error is unchanged, tools/voigt_design.py) with ln2^j folded into its coefficients: the
// two-constant reduction of the natural-exponent form is one subtraction here (t carries a
// rounding error of 1e-14 at most).  Only the pole term uses this exp.
__device__ __forceinline__ double exp_neg_k(double x2, double top) {
  const double t = kmul(-1.4426950408889634074, x2);
  const double kd = __builtin_rint(t);
  const double f = t - kd;
  double p = fma_k(top, f, 1.54527372032223936e-04);
  p = fma_k(p, f, 1.33407291670661681e-03);
  p = fma_k(p, f, 9.61808661438522637e-03);
  p = fma_k(p, f, 5.55040466349984302e-02);
  p = fma_k(p, f, 2.40226506959100694e-01);
  p = fma_k(p, f, 6.93147180559945286e-01);
  p = __builtin_fma(p, f, 1.0);
  return __builtin_ldexp(p, (int)kd);
}

__device__ __forceinline__ double rcp_fast(double d) {
#if defined(RJP_RCP_F32)
  double r = (double)__builtin_amdgcn_rcpf((float)d);   // f32 seed (d within f32 range)
#else
  double r = __builtin_amdgcn_rcp(d);                   // hardware v_rcp_f64 seed
#endif
  return r * __builtin_fma(-d, r, 2.0);                 // + one Newton step
}

// Shifted-lattice evaluation, used only for y < 0.03 (see voigt_rew): nodes (n + delta) h
// with delta = 1/2 whenever x is within h/4 of a node of the plain lattice, so that the
// trapezoid sum and the pole term never cancel, however small y is.
// noinline: its six node tables would otherwise compete for scalar registers with the
// common path inside the channel loop (the compiler spilled ~160 SGPRs per iteration).
__device__ __attribute__((noinline)) double voigt_core_shifted(double ax, double y, double q,
                                                              double lnq) {
  const double r2 = __builtin_fma(ax, ax, y * y);
  const double u = ax * (1.0 / kH);
  const double fr = u - __builtin_floor(u);
  const bool half = !(fr >= 0.25 && fr < 0.75);
  const double U = r2 * r2;
  const double W = 2.0 * __builtin_fma(-ax, ax, y * y);
  double num[kNPair], den[kNPair];
#pragma unroll
  for (int n = 0; n < kNPair; ++n) {
    const double tau = half ? c_tau1[n] : c_tau0[n];
    const double c2 = half ? c_w1[n] : c_w0[n];
    const double c2t = half ? c_wt1[n] : c_wt0[n];
    den[n] = __builtin_fma(tau, W + tau, U);
    num[n] = __builtin_fma(c2, r2, c2t);
  }
  double s = 0.0;
#pragma unroll
  for (int n = 0; n + 4 <= kNPair; n += 4) {       // one reciprocal per four pairs
    const double d01 = den[n] * den[n + 1], d23 = den[n + 2] * den[n + 3];
    const double a = __builtin_fma(num[n], den[n + 1], num[n + 1] * den[n]);
    const double b = __builtin_fma(num[n + 2], den[n + 3], num[n + 3] * den[n + 2]);
    s = __builtin_fma(__builtin_fma(a, d23, b * d01), rcp_fast(d01 * d23), s);
  }
  {
    constexpr int n = kNPair - 2;
    s = __builtin_fma(__builtin_fma(num[n], den[n + 1], num[n + 1] * den[n]),
                      rcp_fast(den[n] * den[n + 1]), s);
  }
  s *= y * (kH / 3.14159265358979323846);
  const double e = y * y - ax * ax;
  if (e + lnq > (double)__logf((float)s) - 31.0) {
    // Re[ 2 exp(-z^2) q / (q - exp(-i theta)) ], theta = 2 pi (x/h - delta); the shift keeps
    // |q - e^{-i theta}| >= 1
    double st, ct, s2, c2;
    sincos_2pi(fr - (half ? 0.5 : 0.0), st, ct);
    sincos_2pi(0.31830988618379067154 * ax * y, s2, c2);
    const double den = __builtin_fma(q, q - 2.0 * ct, 1.0);
    s += 2.0 * exp_any(e) * q * (c2 * (q - ct) - s2 * st) * rcp_fast(den);
  }
  return s;
}

// Pole term of the plain lattice, P = Re[ 2 exp(-z^2) q / (q - exp(-i theta)) ] with
// theta = 2 pi x / h = 2 pi fr (mod 2 pi), written without cancellation near a node
// (1 - cos theta = 2 sin^2(theta/2), 1 - q from expm1).  Out of line: only the waves near the
// line core need it.
__device__ __attribute__((noinline)) double pole_term_plain(double ax, double y, double q) {
  // Re[e^{-i phi} conj(q - e^{-i theta})] = q cos(phi) - cos(theta - phi), phi = 2 x y;
  // |q - e^{-i theta}|^2 = 1 - 2 q cos(theta) + q^2 >= (1 - q)^2 >= 0.07 (y >= 0.03 here), so
  // three cosines do: no half-angle forms needed against cancellation
  const double e = y * y - ax * ax;
  const double u = ax * (1.0 / kH);
  const double fr = u - __builtin_floor(u);                     // theta / 2 pi
  const double ph = 0.31830988618379067154 * ax * y;            // phi / 2 pi
  const double cth = cos_2pi(fr), cph = cos_2pi(ph), cps = cos_2pi(fr - ph);
  const double den = __builtin_fma(q, q - 2.0 * cth, 1.0);
  const double num = __builtin_fma(q, cph, -cps);
  return 2.0 * exp_any(e) * q * num * rcp_fast(den);
}

// ---- small y, one cell per wave: lattice centred on x ------------------------------------
// Nodes t_k = x + (k + 1/2) h: x always sits midway between two nodes, so
//   Re w = (h y / pi) sum_k exp(-t_k^2) / ((k + 1/2)^2 h^2 + y^2)
//          + 2 exp(y^2 - x^2) cos(2 x y) / (1 + exp(2 pi y / h))
// holds for every y > 0 with no cancellation between the sum and the pole term (its
// denominator 1 - exp(-2 pi i (z - t_0)/h) is the REAL number 1 + exp(2 pi y/h)).  The
// denominators depend on the cell only (y is the same in every lane): the wave keeps
// 1/((k+1/2)^2 h^2 + y^2) in a 64-entry LDS table, one entry per lane.  The Gaussian weights
// of a lane follow a recurrence outward from the node nearest t = 0 (|t_m| <= h/2):
// E_{j+1} = E_j R_j, R_{j+1} = R_j exp(-2 h^2) -- two short polynomials instead of 21 exp.
// h = 0.675, 7 nodes on each side of the middle one, exp polynomials of degree 6 / 10: relative
// error < 1e-9 for 1e-10 <= y < 0.03, 0 <= x <= 16 (against scipy.special.wofz;
// tools/voigt_design.py -- rounds 1-2: h = 0.6, 10 nodes a side, 3e-12).
// The centred lattice is valid for every y < pi/h (the parity tests pass with any bound); it is
// USED below y = 0.03, where the plain lattice would cancel: above, the paired plain lattice is
// cheaper (cfg3: 775 ms with the bound at 0.03, 815 at 0.1, 880 at 0.3, 970 at 1.0;
// -DRJP_CEN_YMAX=... moves the bound for such A/B runs).
#ifndef RJP_CEN_YMAX
#define RJP_CEN_YMAX 0.03
#endif
constexpr double kCenYMax = RJP_CEN_YMAX;
static_assert(kCenYMax >= 0.03 && kCenYMax <= 5.0, "plain lattice needs y >= 0.03; q > 0 needs y < pi/h");
constexpr int kCenJ = 7;             // nodes on each side of the middle one
constexpr int kCenOff = 38;          // table index of k = 0; window [km-7, km+7], km >= -25
constexpr double kCenXMax = 16.0;    // beyond: continued fraction (the table ends)

__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// w = (i/sqrt(pi)) / (z - (1/2)/(z - 1/(z - (3/2)/(z - 2/(z - (5/2)/(z - 3/z))))))
__device__ __forceinline__ double voigt_far(double ax, double y) {
  double wr = ax, wi = y;
#pragma unroll
  for (int k = 6; k >= 1; --k) {
    const double s = (0.5 * k) * rcp_fast(__builtin_fma(wr, wr, wi * wi));
    wr = __builtin_fma(-s, wr, ax);
    wi = __builtin_fma(s, wi, y);
  }
  return 0.56418958354775628695 * wi * rcp_fast(__builtin_fma(wr, wr, wi * wi));
}

// `gq` = 2 q exp(y^2) / (1 + q), staged per cell (the pole term is gq exp(-x^2) cos(2 x y)).
// Polynomial and lattice constants ride in SGPRs (fma_k / kmul / kadd below), as in the other
// wave-uniform paths.
__device__ __forceinline__ double voigt_centred(double ax, double y, double ky, double gq,
                                                double cq, double* tab, double exp_top) {
  // per-cell table, written by the 64 lanes of this wave (all of them are here: y is
  // wave-uniform and so is the branch that leads here)
  {
    const int lane = threadIdx.x & (RJP_WAVE - 1);
    const double a = ((double)(lane - kCenOff) + 0.5) * kHW;
    wave_lds_fence();                       // earlier readers of the previous cell's table
    tab[lane] = rcp_fast(__builtin_fma(a, a, y * y));
    wave_lds_fence();
  }
  const double axc = fmin(ax, kCenXMax);
  const double km = __builtin_rint(kmul(-1.0 / kHW, axc) - 0.5);
  const double tm = kfma(kHW, km + 0.5, axc);                   // |tm| <= h/2
  const double w = tm * tm;                                     // <= 0.114
  double em = kadd(-1.0 / 120.0, kmul(1.0 / 720.0, w));         // exp(-w), degree 6 (5e-11)
  em = fma_k(em, w, 1.0 / 24.0);
  em = fma_k(em, w, -1.0 / 6.0);
  em = __builtin_fma(em, w, 0.5);
  em = __builtin_fma(em, w, -1.0);
  em = __builtin_fma(em, w, 1.0);
  const double v = kmul(-2.0 * kHW, tm);                        // |v| <= 0.456
  double u = kadd(2.7557319223985893e-06, kmul(2.755731922398589e-07, v));   // exp(v), degree 10 (4e-12)
  u = fma_k(u, v, 2.48015873015873e-05);
  u = fma_k(u, v, 1.984126984126984e-04);
  u = fma_k(u, v, 1.388888888888889e-03);
  u = fma_k(u, v, 8.333333333333333e-03);
  u = fma_k(u, v, 4.1666666666666664e-02);
  u = fma_k(u, v, 1.6666666666666666e-01);
  u = __builtin_fma(u, v, 0.5);
  u = __builtin_fma(u, v, 1.0);
  u = __builtin_fma(u, v, 1.0);
  constexpr double kC1 = 0.6340515618580675;                    // exp(-h^2), h = 0.675
  constexpr double kQ = 0.40202138309465485;                    // exp(-2 h^2)
  const double* t = tab + ((int)km + kCenOff);
  double s = em * t[0];
  double e = em, r = kmul(kC1, u);                              // towards +t
#pragma unroll
  for (int j = 1; j <= kCenJ; ++j) {
    e *= r;
    if (j < kCenJ) r = kmul(kQ, r);
    s = __builtin_fma(e, t[j], s);
  }
  e = em; r = kmul(kC1, rcp_fast(u));                           // towards -t
#pragma unroll
  for (int j = 1; j <= kCenJ; ++j) {
    e *= r;
    if (j < kCenJ) r = kmul(kQ, r);
    s = __builtin_fma(e, t[-j], s);
  }
  s *= ky;                                                      // y h / pi
  // pole term: below 3e-8 Re w by a rigorous bound (measured: 1e-9) once x^2 exceeds the
  // per-cell bound cq; skipped when no lane of the wave needs it
  const double x2 = ax * ax;
  if (__builtin_amdgcn_ballot_w64(x2 < cq) != 0) {
    static_assert(kCenYMax <= 0.03, "cos(2 x y) below is a short polynomial: 2 x y < 0.4 needs y < 0.03");
    // th = 2 x y < 0.4 wherever the term matters (x^2 < cq < 40, y < 0.03): degree 8, 5e-13
    const double th = 2.0 * ax * y, t2 = th * th;
    double c = kadd(-1.0 / 720.0, kmul(1.0 / 40320.0, t2));     // cos(th), degree 8
    c = fma_k(c, t2, 1.0 / 24.0);
    c = __builtin_fma(c, t2, -0.5);
    c = __builtin_fma(c, t2, 1.0);
    const double pterm = exp_neg_k(x2, exp_top) * c * gq;       // x^2 <= 256
    s += (x2 < cq) ? pterm : 0.0;
  }
  if (__builtin_amdgcn_ballot_w64(ax > kCenXMax) != 0) {
    const double vf = voigt_far(ax, y);
    s = ax > kCenXMax ? vf : s;
  }
  return s;
}

// Re w(x + i y) for one lane (x = ax >= 0 per lane, y > 0 THE SAME IN EVERY LANE: a wave
// works on one cell).  Per-cell constants: q = exp(-2 pi y / h) (or -1 when y >= pi/h: no
// pole term), cq = x^2 below which the pole term matters.
// `tab` = this wave's 64-entry LDS table when every lane of the wave works on the same cell
// (kernels with >= 64 channel lanes), else nullptr (CEN = false).
template <bool CEN>
__device__ __forceinline__ double voigt_rew(double ax, double y, double q, double cq,
                                            double* tab) {
  const double r2 = __builtin_fma(ax, ax, y * y);
  // the far-field branch is taken only when EVERY active lane qualifies: the core formula is
  // valid everywhere, so a wave that straddles the boundary runs one path, not both
  const bool far = r2 > 64.0 && (ax * ax > 64.0 || y > 1.0);
  if (__builtin_amdgcn_ballot_w64(!far) == 0) return voigt_far(ax, y);
  static_assert(!CEN, "the wave-uniform kernels call their paths directly (path codes)");
  if (y < 0.03) return voigt_core_shifted(ax, y, q, -2.0 * (3.14159265358979323846 / kH) * y);

  // Plain lattice t = n h.  Pair (+t,-t):
  //   c [1/((x-t)^2+y^2) + 1/((x+t)^2+y^2)] = 2c (A + tau) / (A^2 + tau (W + tau)),
  //   A = x^2+y^2, W = 2 (y^2 - x^2), tau = t^2 -- all node constants are immediates.
  // A lane close to a node sees the sum and the pole term cancel and the factored
  // denominator loses digits ~ x^2/(4 y^2); for y >= 0.03 the result keeps a relative error
  // < 3e-12 (measured against scipy.special.wofz), below that the shifted lattice is used.
  const double U = r2 * r2;
  const double W = 2.0 * __builtin_fma(-ax, ax, y * y);
  constexpr double tau[kNPair] = {0.0, 0.36, 1.44, 3.2399999999999993, 5.76, 9.0,
                                  12.959999999999997, 17.64, 23.04, 29.159999999999993};
  constexpr double w2[kNPair] = {1.0, 1.395352652142062, 0.47385551736424353,
                                 0.0783277901979742, 0.006302223196888882,
                                 0.0002468196081733591, 4.705150400019559e-06,
                                 4.3659155902509556e-08, 1.9719011151983032e-10,
                                 4.3351377652379543e-13};
  constexpr double w2t[kNPair] = {0.0, 0.5023269547711423, 0.6823519450045107,
                                  0.25378204024143636, 0.03630080561407996,
                                  0.002221376473560232, 6.097874918425347e-05,
                                  7.701475101202686e-07, 4.54326016941689e-09,
                                  1.2641261723433871e-11};
  // one reciprocal per FOUR pairs: n0/d0 + n1/d1 + n2/d2 + n3/d3 over the common
  // denominator (the d's are bounded, their products stay far inside the FP64 range; the
  // hardware reciprocal is the slow instruction here)
  double num[kNPair], den[kNPair];
#pragma unroll
  for (int n = 0; n < kNPair; ++n) {
    den[n] = __builtin_fma(tau[n], W + tau[n], U);
    num[n] = __builtin_fma(w2[n], r2, w2t[n]);
  }
  double s = 0.0;
#pragma unroll
  for (int n = 0; n + 4 <= kNPair; n += 4) {
    const double d01 = den[n] * den[n + 1], d23 = den[n + 2] * den[n + 3];
    const double a = __builtin_fma(num[n], den[n + 1], num[n + 1] * den[n]);
    const double b = __builtin_fma(num[n + 2], den[n + 3], num[n + 3] * den[n + 2]);
    s = __builtin_fma(__builtin_fma(a, d23, b * d01), rcp_fast(d01 * d23), s);
  }
  static_assert(kNPair % 4 == 2, "tail below handles exactly two pairs");
  {
    constexpr int n = kNPair - 2;
    s = __builtin_fma(__builtin_fma(num[n], den[n + 1], num[n + 1] * den[n]),
                      rcp_fast(den[n] * den[n + 1]), s);
  }
  s *= y * (kH / 3.14159265358979323846);
  // Pole term P = Re[ 2 exp(-z^2) q / (q - exp(-i theta)) ], theta = 2 pi x / h.
  // |P| <= 6 exp(y^2 - x^2) q / (1 - q)^2 and Re w >= y / (4 (|z|^2 + 1)) with |z|^2 < 66 in
  // this branch: P is below 1e-13 Re w, and skipped, once x^2 exceeds the per-cell bound cq.
  if (q >= 0.0 && ax * ax < cq) s += pole_term_plain(ax, y, q);
  return s;
}

// ---- wave-uniform fast paths (kernels whose waves work on one cell) ----------------------
// FP64 VOP3 instructions take no literal: a constant operand has to sit in a register.  Left
// to itself the compiler materialises every polynomial / lattice constant with two
// v_mov_b32 per use -- vector-ALU work, a fifth of the instructions of these paths -- or,
// with the loop-invariant hoisting on, keeps ~60 of them in VGPRs and spills.  The helpers
// below pin the constant to an SGPR pair instead (two s_mov_b32 on the scalar unit, which
// issues beside the vector ALU).  Never fed straight from v_rcp_f64 / a transcendental op
// (the hazard recogniser does not see through inline asm); rcp_fast() ends in ordinary ops.
__device__ __forceinline__ double fma_k(double a, double b, double K) {      // a * b + K
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(K));
  return d;
}
__device__ __forceinline__ double kfma(double K, double b, double c) {       // K * b + c
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "s"(K), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ double kadd(double K, double b) {                 // K + b
  double d;
  asm("v_add_f64 %0, %1, %2" : "=v"(d) : "s"(K), "v"(b));
  return d;
}
__device__ __forceinline__ double kmul(double K, double b) {                 // K * b
  double d;
  asm("v_mul_f64 %0, %1, %2" : "=v"(d) : "s"(K), "v"(b));
  return d;
}

// Far field, every lane of the wave: asymptotic series of w(z) in u = 1/z^2,
//   w(z) ~ (i / (sqrt(pi) z)) sum_k (2k-1)!!/2^k u^k,   Re w = (y Re S - x Im S) / (|z|^2 sqrt pi),
// ONE reciprocal per evaluation (the Laplace continued fraction above needs one per level).
// Truncation after K terms, measured against scipy.special.wofz over 1e-10 <= y <= 1e3
// (tools/voigt_design.py): |z|^2 > 64 (the Gaussian core exp(-x^2) <= 1.6e-28 is invisible
// there): K = 6 -> 4.1e-9 (K = 8 -> 8e-11: rounds 1-2); |z|^2 > 196: K = 4 -> 1.2e-9.
template <int K>
__device__ __forceinline__ double voigt_far_series(double ax, double y) {
  // (2k-1)!!/2^k, times 1/sqrt(pi): the series is linear in its coefficients, so the constant
  // factor of w(z) rides in them (one multiplication fewer per evaluation)
  constexpr double kIsp = 0.56418958354775628695;
  constexpr double c[9] = {1.0 * kIsp, 0.5 * kIsp, 0.75 * kIsp, 1.875 * kIsp, 6.5625 * kIsp,
                           29.53125 * kIsp, 162.421875 * kIsp, 1055.7421875 * kIsp,
                           7918.06640625 * kIsp};
  static_assert(K >= 2 && K <= 8, "series length");
  const double x2 = ax * ax, y2 = y * y;
  const double r2 = x2 + y2;
  const double inv = rcp_fast(r2 * r2);                 // 1 / |z^2|^2
  const double ur = (x2 - y2) * inv;                    // u = conj(z^2) / |z^2|^2
  const double nui = 2.0 * (ax * y) * inv;              // -Im u (inline asm takes no neg modifier)
  double pr = kadd(c[K - 1], kmul(c[K], ur));           // first Horner step (Im S = 0 before it)
  double pi = kmul(-c[K], nui);
#pragma unroll
  for (int k = K - 2; k >= 0; --k) {
    // S <- S u + c[k]:  Re: pr ur - pi ui + c[k],  Im: pr ui + pi ur,  ui = -nui
    const double t = __builtin_fma(pr, ur, fma_k(pi, nui, c[k]));
    pi = __builtin_fma(-pr, nui, pi * ur);
    pr = t;
  }
  // 1/|z|^2 = |z|^2 * inv
  return __builtin_fma(y, pr, -ax * pi) * (r2 * inv);
}

// Plain lattice (y >= 0.03), every lane of the wave: the eight node pairs (h = 0.675) over ONE
// common denominator -- a single reciprocal per evaluation.  With m_n = |z|^2 + t_n^2 the pair
// (+t_n, -t_n) is
//   w_n [1/((x-t_n)^2+y^2) + 1/((x+t_n)^2+y^2)] = 2 w_n m_n / d_n,   d_n = m_n^2 - 4 t_n^2 x^2,
// so numerator and denominator share m_n, and the node weights enter as RATIOS while the
// fractions are merged (one constant multiply per merge instead of one per node).  The
// d_n are >= y^4 >= 8e-7 with at most one pair near its minimum and <= ~|z|^4 each: their
// product stays inside the FP64 range for |z| < 1e9 (path_code sends waves with |x| > 1e6
// to the generic path).  Near a node d_n loses digits ~ t_n^2 / y^2, as the factored form
// of the generic path does.  Relative error <= 3.4e-9 against wofz for 0.03 <= y, x^2 <= 64,
// pole term included (worst at x = 0 just above y = pi/h, where the pole term ends;
// tools/voigt_design.py).  Rounds 1-2: h = 0.6, ten pairs, 1e-11.
// `ky` = y h / pi, staged per cell.  `ax` may carry either sign (see the channel loop).
// POLE: 0 = no pole term, 1 = the full term, 2 = its leading order in q (cells with y >= 1.3).
constexpr double kPoleLiteY = 1.3;
template <int POLE>
__device__ __forceinline__ double voigt_plain_wave(double ax, double y, double ky, double q,
                                                   double gq, const PoleTop& top) {
  constexpr double tau[kNPairW] = {0.0, 0.45562500000000006, 1.8225000000000002,
                                   4.100625000000002, 7.290000000000001, 11.390625,
                                   16.402500000000007, 22.325625000000006};
  // w2[n] = 2 exp(-tau[n]) (w2[0] = 1): ratios w2[b]/w2[a] of the pairs (0,1) (2,3) ... and
  // of the merges
  constexpr double w2[kNPairW] = {1.0, 1.268103123716135, 0.3232423849306784,
                                  0.03312464143162309, 0.0013646561055127519,
                                  2.2601872086292614e-05, 1.5049246515289555e-07,
                                  4.028415451797932e-10};
  static_assert(kNPairW == 8, "four pairs of fractions below");
  const double x2 = ax * ax;
  const double r2 = __builtin_fma(y, y, x2);
  const double X4 = -4.0 * x2;
  double N[4], D[4];
#pragma unroll
  for (int a = 0; a < kNPairW; a += 2) {
    const double ma = a == 0 ? r2 : kadd(tau[a], r2);
    const double mb = kadd(tau[a + 1], r2);
    const double da = a == 0 ? ma * ma : kfma(tau[a], X4, ma * ma);
    const double db = kfma(tau[a + 1], X4, mb * mb);
    // true numerator = w2[a] * (ma db + rho mb da)
    N[a / 2] = __builtin_fma(ma, db, kmul(w2[a + 1] / w2[a], mb * da));
    D[a / 2] = da * db;
  }
  // merges: N01 = N0 D1 + sigma N1 D0 with sigma the ratio of the fractions' scales
  const double N01 = __builtin_fma(N[0], D[1], kmul(w2[2] / w2[0], N[1] * D[0])), D01 = D[0] * D[1];
  const double N23 = __builtin_fma(N[2], D[3], kmul(w2[6] / w2[4], N[3] * D[2])), D23 = D[2] * D[3];
  const double Nall = __builtin_fma(N01, D23, kmul(w2[4] / w2[0], N23 * D01)), Dall = D01 * D23;
  // sum = w2[0] * Nall / Dall (w2[0] = 1); Re w = (h y / pi) * sum = ky * sum
  static_assert(w2[0] == 1.0, "ky carries no node weight");
  if (POLE == 0) return Nall * rcp_fast(Dall) * ky;
  // P = Re[ 2 exp(-z^2) q / (q - exp(-i theta)) ], theta = 2 pi x / h, for every lane (it
  // is negligible where x^2 exceeds the per-cell bound cq, and harmless there):
  // Re[e^{-i phi} conj(q - e^{-i theta})] = q cos(phi) - cos(theta - phi), phi = 2 x y;
  // |q - e^{-i theta}|^2 = 1 - 2 q cos(theta) + q^2 >= (1 - q)^2 >= 0.06 for y >= 0.03
  // `gq` = 2 q exp(y^2) is staged per cell: exp(y^2 - x^2) costs the lane exp(-x^2) only
  const double u = kmul(2.0 / kHW, ax);                         // theta / pi (half-turns)
  const double ph = kmul(0.63661977236758134308, ax * y);       // phi / pi
  if constexpr (POLE == 2) {
    // y >= 1.3: q = exp(-2 pi y / h) <= 5.6e-6, and P = -2 E q cos(theta - phi) (1 + O(q)) with
    // |P| <= 1.3e-4 Re w there: the terms of order q^2 E stay below 5e-10 Re w (the path's worst
    // error remains the lattice's 3.1e-9, tools/voigt_design.py) -- ONE cosine, no denominator
    double c0, c1, cps;
    cos_2pi_x3(u - ph, 0.0, 0.0, top.cos_top, cps, c0, c1);     // (the idle slots fold away)
    const double pq = exp_neg_k(x2, top.exp_top) * gq * cps;
    return __builtin_fma(Nall, ky, -pq * Dall) * rcp_fast(Dall);
  }
  double cth, cph, cps;
  cos_2pi_x3(u, ph, u - ph, top.cos_top, cth, cph, cps);
  const double den = __builtin_fma(q, q - 2.0 * cth, 1.0);
  const double num = __builtin_fma(q, cph, -cps);
  // one reciprocal for both fractions: (Nall ky den + 2 E q num Dall) / (Dall den), with
  // Dall <= (|z|^2 + 23)^16 < 1e32 for the |x| <= 8 a wave of this path can hold
  const double pq = exp_neg_k(x2, top.exp_top) * gq * num;
  return __builtin_fma(Nall * ky, den, pq * Dall) * rcp_fast(Dall * den);
}

// Per-(cell, wave) path codes, decided once per cell in phase 1 from the |x| range of the
// wave's channels -- the channel loop then branches on a scalar instead of testing regimes
// per lane.  One byte per wave of the channel block.
enum : int {
  kPathSkip = 0,      // C == 0: the cell contributes nothing (outside the jet, NaN, ...)
  kPathFarA = 1,      // every lane far field, |z|^2 > 64: 6-term series
  kPathFarB = 2,      // every lane |z|^2 > 196: 4-term series
  kPathPlain = 3,     // plain lattice, pole term negligible in every lane
  kPathPlainPole = 4, // plain lattice + pole term
  kPathCentred = 5,   // y < 0.03: centred lattice
  kPathGeneric = 6,   // irregular constants (inf ...) or |x| > 1e6 beside core lanes:
                      // per-lane generic code with NumPy's NaN filter
  kPathPlainPoleLite = 7, // plain lattice + the pole term to leading order in q (y >= 1.3)
  kPathExpFlag = 8    // bit 3: h nu / kT is not small over the band -> exp() per lane
};

template <typename T>
struct RrlFields {
  const T* nd;
  const T* xi;
  const T* temp;
  const T* pf;
  const T* ts;
  const T* vy;
  const int32_t* ylo;      // optional occupied y-range per sightline
  const int32_t* yhi;
};

struct LineDev {
  double nu_rest, kG, kL, kappa0, en_over_k, h_over_k;
  double path0;        // csize * au * 100 [cm]
  double nu_ref;       // reference frequency of the channel block expansion
  double dnu_max;      // max |nu_f - nu_ref| over all channels
};

// Per-cell line constants: everything of kappa_L * path that does not depend on the channel.
struct CellLine {
  double C = 0.0;      // LTE prefactor * path / (sigma sqrt(2 pi)); 0 = cell contributes nothing
  double nu0 = 0.0;    // Doppler-shifted rest frequency [Hz]
  double is2 = 0.0;    // 1 / (sigma sqrt 2) [1/Hz]
  double y = 1.0;      // Voigt y = (fwhm_L / 2) / (sigma sqrt 2)
  double a = 0.0;      // h / (k T) [1/Hz]
  double E0 = 0.0;     // exp(-a nu_ref)
  double q = -1.0, cq = 0.0;   // pole-term constants (see voigt_rew)
  // the forms the wave-uniform channel loop reads (one fma each instead of sub + mul and of
  // the five-instruction stimulated-emission tail):
  double c1 = 0.0;     // -nu0 * is2:  x = nu * is2 + c1
  double A = 0.0;      // C (1 - E0):  C (1 - E0 (1 - a dnu)) = A + B dnu
  double B = 0.0;      // C E0 a
};

template <typename T, bool BURSTS, bool CEN>
__device__ __forceinline__ CellLine cell_line(const RrlFields<T>& f, int64_t o,
                                              const BurstsDev& b, double time_s,
                                              const LineDev& ln) {
  CellLine c;
  // (the layouts whose waves work on one cell take the wave-uniform paths: their lattice step)
  const double kPiOverH = 3.14159265358979323846 / (CEN ? kHW : kH);
  const double nd = (double)f.nd[o], xi = (double)f.xi[o], Tk = (double)f.temp[o],
               pf = (double)f.pf[o], vy = (double)f.vy[o];
  double chi = 1.0;
  if (BURSTS) chi = chi_cell(b, signbit_d(nd), time_s - (double)f.ts[o]);
  const double ne = fabs(nd) * chi * xi;
  c.nu0 = ln.nu_rest * (1.0 - vy * 1000.0 / 299792458.0);          // physics.py:557-558
  const double fwhm_g = ln.kG * sqrt(Tk) * c.nu0;                  // rrls.py:116-118
  const double sigma = fwhm_g / 2.0 / 1.1774100225154747;          // / sqrt(2 ln 2)
  c.is2 = 1.0 / (sigma * 1.4142135623730951);
  const double fwhm_l = ln.kL * ne;                                // rrls.py:101
  c.y = 0.5 * fwhm_l * c.is2;
  c.a = ln.h_over_k / Tk;
  // kappa_L * path without the profile and the stimulated-emission factor
  c.C = ln.kappa0 * (ne * ne / (Tk * sqrt(Tk))) * exp(ln.en_over_k / Tk) *
        (ln.path0 * pf) / (sigma * 2.5066282746310002);
  c.E0 = exp(-c.a * ln.nu_ref);
  c.c1 = -c.nu0 * c.is2;
  c.A = -c.C * expm1(-c.a * ln.nu_ref);
  c.B = c.C * c.E0 * c.a;
  const double lnq = -2.0 * kPiOverH * c.y;
  c.q = (c.y < kPiOverH) ? exp(lnq) : -1.0;
  const double omq = -expm1(lnq);                                  // 1 - q
  // pole term needed iff y^2 - x^2 + ln(6 q / (1-q)^2) > ln(tol y / (4 * 67)), i.e. iff
  // x^2 < cq; tol = 3e-8 for the wave-uniform kernels (a crude bound: the term left out is
  // <= 1.5e-9 Re w when measured, tools/voigt_design.py), 1e-13 for the generic per-lane path
  c.cq = c.y * c.y + lnq + 1.7917594692280550 - 2.0 * log(omq) - log(0.25 * c.y) +
         (CEN ? 17.3221740089 : 29.9336062089226) + 4.2046926193909657;
  // centred lattice (y < 0.03): |P| <= exp(y^2 - x^2) and Re w >= y / (4 (|z|^2 + 1)) with
  // |z|^2 <= 16^2 + 1: negligible iff x^2 > y^2 - ln y + ln(1 / 3e-8) + ln(4 * 258)
  if (CEN && c.y < kCenYMax) c.cq = c.y * c.y - log(c.y) + 17.3221740089 + 6.9392539460415;
  if (!(c.C == c.C) || c.C == 0.0 || !(c.y > 0.0)) c.C = 0.0;     // nansum drops NaN terms
  return c;
}

// Path code of one cell for a wave whose live channels are nu in [lo_e, hi_e] (even lanes)
// and [lo_o, hi_o] (odd lanes; the folded channel order gives every wave two runs).  The end
// points are lane values, and x is evaluated exactly as the lanes do, so the wave-level
// decision agrees with what each lane would decide.
__device__ __forceinline__ int path_code(const CellLine& c, const double (&rg)[4],
                                         double dnu_max) {
  if (c.C == 0.0) return kPathSkip;
  const bool regular = (c.C - c.C == 0.0) && (c.nu0 - c.nu0 == 0.0) && (c.is2 - c.is2 == 0.0) &&
                       (c.y - c.y == 0.0) && c.y > 0.0 && (c.a - c.a == 0.0) &&
                       (c.E0 - c.E0 == 0.0);
  int code;
  double xmin = __builtin_inf(), xmax = 0.0;
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const double lo = __builtin_fma(rg[2 * g], c.is2, c.c1),
                 hi = __builtin_fma(rg[2 * g + 1], c.is2, c.c1);
    if (!(rg[2 * g] <= rg[2 * g + 1])) continue;                  // no live lane in this run
    const double alo = fabs(lo), ahi = fabs(hi);
    xmin = fmin(xmin, (lo <= 0.0 && hi >= 0.0) ? 0.0 : fmin(alo, ahi));
    xmax = fmax(xmax, fmax(alo, ahi));
  }
  const double x2min = xmin * xmin;
  const double r2min = __builtin_fma(c.y, c.y, x2min);
  if (!regular || !(xmax - xmax == 0.0)) code = kPathGeneric;
  else if (r2min > 64.0 && (x2min > 64.0 || c.y > 1.0)) code = r2min > 196.0 ? kPathFarB : kPathFarA;
  else if (xmax > 1e6) code = kPathGeneric;
  else if (c.y < kCenYMax) code = kPathCentred;
  else code = (c.q >= 0.0 && x2min < c.cq)
                  ? (c.y >= kPoleLiteY ? kPathPlainPoleLite : kPathPlainPole) : kPathPlain;
  // 1 - E0 exp(-a dnu) to first order in a dnu: the dropped term (a dnu)^2 / 2 stays below
  // 2e-9 of the factor itself (which is ~ a nu_ref for h nu << k T)
  if (!(0.5 * (c.a * dnu_max) * (c.a * dnu_max) < 2e-9 * (1.0 - c.E0))) code |= kPathExpFlag;
  return code;
}

// kappa_L * path of one (cell, channel): C * Re w * (1 - exp(-h nu / kT))   (rrls.py:383-389)
template <bool CEN>
__device__ __forceinline__ double line_term(const CellLine& c, double nu_f, double dnu,
                                            double dnu_max, double* tab) {
  const double xv = (nu_f - c.nu0) * c.is2;
  const double V = voigt_rew<CEN>(fabs(xv), c.y, c.q, c.cq, tab);
  // 1 - exp(-h nu / kT) = 1 - E0 * exp(-a (nu - nu_ref))
  const double eps = c.a * dnu;
  double ex;
  if (c.a * dnu_max < 1e-3)
    ex = __builtin_fma(eps, __builtin_fma(eps, __builtin_fma(eps, -1.0 / 6.0, 0.5), -1.0), 1.0);
  else
    ex = exp(-eps);
  return c.C * V * (1.0 - c.E0 * ex);
}

// Out of line: the cold generic path of the wave-uniform kernels must not cost their channel
// loop registers.  It takes nothing from the staged constants (they are stored in the forms
// the fast paths read, and q / cq belong to the h = 0.675 lattice): the cell is evaluated
// again from the fields with the generic code's own constants (h = 0.6, pole term kept down
// to 1e-13 Re w), as cell_line<.., CEN = false>.
template <typename T, bool BURSTS>
__device__ __attribute__((noinline)) double line_term_generic(const RrlFields<T>& f, int64_t o,
                                                              const BurstsDev& b, double time_s,
                                                              const LineDev& ln, double nu_f,
                                                              double dnu) {
  const CellLine c = cell_line<T, BURSTS, false>(f, o, b, time_s, ln);
  if (c.C == 0.0) return 0.0;
  return line_term<false>(c, nu_f, dnu, ln.dnu_max, nullptr);
}

// collapse=False: the 3-D per-cell optical depths (classes.py:1176-1177, 1382-1383).  One
// thread per cell, serial over channels; out[f * ncell + cell].
template <typename T, bool BURSTS>
__global__ __launch_bounds__(kRB) void rrl_cells_kernel(RrlFields<T> f, int64_t ncell,
                                                        BurstsDev b, double time_s, LineDev ln,
                                                        const double* __restrict__ nu, int nchan,
                                                        double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kRB + threadIdx.x;
  if (i >= ncell) return;
  const CellLine c = cell_line<T, BURSTS, false>(f, i, b, time_s, ln);
  const double nan = __builtin_nan("");
  // a cell outside the jet is NaN in the reference's 3-D output (NaN fields propagate)
  const bool dead = !((double)f.nd[i] == (double)f.nd[i]) || !((double)f.xi[i] == (double)f.xi[i]) ||
                    !((double)f.temp[i] == (double)f.temp[i]) || !((double)f.pf[i] == (double)f.pf[i]) ||
                    !((double)f.vy[i] == (double)f.vy[i]);
  for (int k = 0; k < nchan; ++k) {
    double v = nan;
    if (!dead) v = c.C == 0.0 ? 0.0 : line_term<false>(c, nu[k], nu[k] - ln.nu_ref, ln.dnu_max, nullptr);
    out[(int64_t)k * ncell + i] = v;
  }
}

// LF = lanes along the channel axis (16, 64 or 256); G = kRB / LF sightline groups.
// Tile = ZT z-adjacent sightlines (8 for LF = 256, else 16), slab = 256 / ZT y-rows.
// The per-(sightline, channel) accumulators live in LDS, one slot per thread and sightline,
// so the sightline loop is NOT unrolled: one inlined copy of the Voigt code, < 128 VGPRs.
#ifndef RJP_K3_ZT256
#define RJP_K3_ZT256 8      /* sightlines per workgroup of the 256-channel-lane kernel */
#endif
template <int LF> struct RrlTile {
  static constexpr int ZT = LF == 256 ? RJP_K3_ZT256 : 16;
  static constexpr int YC = kRB / ZT;
  static constexpr int G = kRB / LF;
  static constexpr int NZP = ZT / G;        // sightlines per thread
};

#ifndef RJP_K3_WAVES
#define RJP_K3_WAVES 4      /* 128-VGPR budget: 4 waves per SIMD hide the LDS/constant waits (+8 %) */
#endif
template <typename T, int LF, bool BURSTS>
__global__ __launch_bounds__(kRB, RJP_K3_WAVES) void rrl_scan_kernel(
    RrlFields<T> f, int nx, int ny, int nz, BurstsDev b, double time_s, LineDev ln,
    const double* __restrict__ nu, int nchan, double* __restrict__ tau) {
  using TL = RrlTile<LF>;
  constexpr int ZT = TL::ZT, YC = TL::YC, NZP = TL::NZP;
  static_assert(ZT % TL::G == 0, "tile/group mismatch");

  __shared__ double s_nu0[kRB], s_is2[kRB], s_y[kRB], s_C[kRB], s_a[kRB], s_E0[kRB],
      s_q[kRB], s_cq[kRB];
  __shared__ double s_acc[NZP * kRB];
  // per-wave table of the centred Voigt lattice (kernels whose waves work on one cell)
  constexpr bool CEN = LF >= RJP_WAVE;
  __shared__ double s_tab[CEN ? kRB / RJP_WAVE : 1][RJP_WAVE];
  // path codes (one byte per wave of the channel block) and the waves' channel ranges
  constexpr int NWC = CEN ? LF / RJP_WAVE : 1;
  // one byte per (wave of the channel block, sightline of the tile, y-row of the slab), rows
  // adjacent: a wave fetches the codes of eight rows with ONE 8-byte read and two
  // readfirstlane, then shifts them out of an SGPR pair (it used to read, add an address and
  // readfirstlane per evaluation)
  static_assert(YC % 8 == 0, "eight codes per read");
  __shared__ __attribute__((aligned(8))) uint8_t s_cb[CEN ? NWC * ZT * YC : 8];
  __shared__ double s_ky[CEN ? kRB : 1];     // y h / pi of the wave-uniform lattice
  __shared__ double s_rng[NWC][4];

  const int ntz = (nz + ZT - 1) / ZT;
  // XCD-aware tile map (round 5): a tile's rows are 64-byte runs (8 sightlines x 8 B), half of a
  // 128-byte line; workgroups are dealt round-robin to the 8 XCDs, so with the identity map the
  // z-neighbour that needs the other half ran on ANOTHER XCD (its own L2) and every line came
  // from HBM twice -- FETCH_SIZE 71.7 GB raw for 26.3 GB algorithmic, profiles/r04_cfg3_f64_pmc.json.
  // Every XCD now takes a contiguous range of tiles in dispatch order: neighbours share an L2.
  // (K3 is FP64-vector-bound: this is about wasted traffic, not time.)
#ifndef RJP_K3_XCD
#define RJP_K3_XCD 1
#endif
  unsigned bx = blockIdx.x;
  if (RJP_K3_XCD) {
    const unsigned per = gridDim.x / 8;                    // (the tail past 8 * per: identity)
    if (bx < 8 * per) bx = (bx % 8) * per + bx / 8;
  }
  const int x = (int)bx / ntz;
  const int z0 = ((int)bx - x * ntz) * ZT;
  const int tid = threadIdx.x;
  // 256 channel lanes: the four waves of a workgroup hold four BANDS of |x| (the folded channel
  // order below), i.e. paths of very different cost (line core: lattice + pole term, 113
  // instructions; outermost band: 37), and they meet at a barrier per slab.  Which wave takes
  // which band ROTATES WITH THE SIGHTLINE inside a slab (band = (wave + j) mod 4, j = the
  // sightline of the tile): every wave gets every band twice per slab, so the four waves reach
  // the barrier together instead of three of them waiting for the one that holds the line core.
  // A (sightline, band) pair still belongs to exactly one wave per slab: no atomics, the same
  // summation order.  The accumulators are LDS slots per CHANNEL already; the channel
  // frequencies go to LDS too.
#ifndef RJP_K3_ROT
#define RJP_K3_ROT 1
#endif
  constexpr bool ROT = LF == 256 && RJP_K3_ROT != 0;
  __shared__ double s_nu[ROT ? kRB : 1];
  const int fl = tid % LF;
  const int g = tid / LF;
  // Lanes take the channels of this block folded about the block centre: lane 0 -> first,
  // lane 1 -> last, lane 2 -> second, ...  A band centred on the line then gives each wave
  // a narrow range of |x|: the outermost wave is entirely far-field (asymptotic series) and
  // only the innermost needs the pole term, instead of every wave straddling both regimes.
  const int fbase = blockIdx.y * LF;
  const int nblk = min(LF, nchan - fbase);
  const int fi = fbase + ((fl & 1) ? nblk - 1 - (fl >> 1) : (fl >> 1));
  const bool chan_live = fl < nblk;
  const double nu_f0 = chan_live ? nu[fi] : ln.nu_ref;
  if constexpr (ROT) s_nu[tid] = nu_f0;          // (visible after the barrier of the range block)

#pragma unroll
  for (int j = 0; j < NZP; ++j) s_acc[j * kRB + tid] = 0.0;

  if constexpr (CEN) {
    // frequency range of this wave's even and odd lanes (its two runs of channels)
    const double inf = __builtin_inf();
    double r0 = (chan_live && !(fl & 1)) ? nu_f0 : inf, r1 = (chan_live && !(fl & 1)) ? nu_f0 : -inf;
    double r2 = (chan_live && (fl & 1)) ? nu_f0 : inf, r3 = (chan_live && (fl & 1)) ? nu_f0 : -inf;
#pragma unroll
    for (int d = RJP_WAVE / 2; d > 0; d >>= 1) {
      r0 = fmin(r0, __shfl_xor(r0, d, RJP_WAVE));
      r1 = fmax(r1, __shfl_xor(r1, d, RJP_WAVE));
      r2 = fmin(r2, __shfl_xor(r2, d, RJP_WAVE));
      r3 = fmax(r3, __shfl_xor(r3, d, RJP_WAVE));
    }
    if ((tid & (RJP_WAVE - 1)) == 0) {
      const int w = fl / RJP_WAVE;               // LF = 64: every wave holds the same channels
      s_rng[w][0] = r0; s_rng[w][1] = r1; s_rng[w][2] = r2; s_rng[w][3] = r3;
    }
    __syncthreads();
  }

  const PoleTop ptop = pole_top();
  const int cy = tid / ZT, cz = tid % ZT;       // this thread's cell in the slab (phase 1)

  int ya = 0, ye = ny;
  if (f.ylo) {
    // sparse models: only the slabs that intersect the tile's occupied y-range
    __shared__ int s_lo, s_hi;
    if (tid == 0) { s_lo = ny; s_hi = 0; }
    __syncthreads();
    if (tid < ZT && z0 + tid < nz) {
      const int64_t p = (int64_t)x * nz + z0 + tid;
      const int lo = f.ylo[p], hi = f.yhi[p];
      if (lo < hi) { atomicMin(&s_lo, lo); atomicMax(&s_hi, hi); }
    }
    __syncthreads();
    ya = (s_lo / YC) * YC;
    ye = s_hi;
  }

  // (a scalar: the path code must reach the branches below as a wave-uniform value)
  const int wave = (CEN && LF > RJP_WAVE) ? __builtin_amdgcn_readfirstlane(fl / RJP_WAVE) : 0;

  for (int yb = ya; yb < ye; yb += YC) {
    // ---- phase 1: per-cell line constants --------------------------------------------
    {
      const int yy = yb + cy, zz = z0 + cz;
      CellLine cl;
      if (yy < ny && zz < nz)
        cl = cell_line<T, BURSTS, (LF >= RJP_WAVE)>(f, ((int64_t)x * ny + yy) * nz + zz, b,
                                                    time_s, ln);
      const double C = cl.C, nu0 = cl.nu0, is2 = cl.is2, yv = cl.y, a = cl.a, E0 = cl.E0,
                   q = cl.q, cq = cl.cq;
      // wave-uniform kernels stage the derived forms in the same slots: s_C <- A, s_nu0 <- c1,
      // s_E0 <- B (the per-lane layouts keep the plain constants)
      s_C[tid] = CEN ? cl.A : C; s_nu0[tid] = CEN ? cl.c1 : nu0; s_is2[tid] = is2; s_y[tid] = yv;
      // (wave-uniform kernels: the bound cq is used up by path_code below for cells of the
      // plain lattice, y >= 0.03 -- their slot carries 2 q exp(y^2) for the pole term instead;
      // q < 0 marks y >= pi/h, where no pole term exists)
      // cells of the centred lattice, y < 0.03: q is needed as 2 q exp(y^2) / (1 + q) only
      const bool cen_cell = CEN && yv < kCenYMax;
      const double gq = q >= 0.0 ? 2.0 * q * exp(yv * yv) : 0.0;
      s_a[tid] = a; s_E0[tid] = CEN ? cl.B : E0; s_q[tid] = cen_cell ? gq / (1.0 + q) : q;
      s_cq[tid] = (CEN && !cen_cell) ? gq : cq;
      if constexpr (CEN) {
        s_ky[tid] = yv * (kHW / 3.14159265358979323846);
#pragma unroll
        for (int w = 0; w < NWC; ++w) {
          const double rg[4] = {s_rng[w][0], s_rng[w][1], s_rng[w][2], s_rng[w][3]};
          s_cb[(w * ZT + cz) * YC + cy] = (uint8_t)path_code(cl, rg, ln.dnu_max);
        }
      }
    }
    __syncthreads();

    // ---- phase 2: lanes over channels ------------------------------------------------
#pragma unroll 1
    for (int j = 0; j < NZP; ++j) {
      const int wq = ROT ? (wave + j) & 3 : wave;                          // this wave's band
      const int slot = ROT ? (wq << 6) | (tid & (RJP_WAVE - 1)) : tid;     // channel slot of this lane
      const double nu_f = ROT ? s_nu[slot] : nu_f0;
      const double dnu = nu_f - ln.nu_ref;
      double acc = s_acc[j * kRB + slot];
      if constexpr (CEN) {
        // the wave works on ONE cell per trip: its path was decided in phase 1
        const uint8_t* cb = s_cb + (wq * ZT + g * NZP + j) * YC;
        uint32_t q_lo = 0, q_hi = 0;            // the codes of eight rows, in an SGPR pair
#pragma unroll 1
        for (int r = 0; r < YC; ++r) {
          const int ci = r * ZT + g * NZP + j;
          if ((r & 7) == 0) {
            const uint2 v = *reinterpret_cast<const uint2*>(cb + r);
            q_lo = __builtin_amdgcn_readfirstlane(v.x);
            q_hi = __builtin_amdgcn_readfirstlane(v.y);
          }
          const int pc = (int)(q_lo & 0xffu);
          q_lo = (q_lo >> 8) | (q_hi << 24);
          q_hi >>= 8;
          if (pc == kPathSkip) continue;
          const int path = pc & 7;
          if (path == kPathGeneric) {
            // irregular constants or absurd |x| beside core lanes: per-lane generic code,
            // NaN terms dropped as numpy.nansum does
            static_assert(kCenYMax == 0.03, "generic path assumes the centred bound ends at 0.03");
            const int64_t o = ((int64_t)x * ny + (yb + r)) * nz + (z0 + g * NZP + j);
            const double term = line_term_generic<T, BURSTS>(f, o, b, time_s, ln, nu_f, dnu);
            if (term == term) acc += term;
            continue;
          }
          const double yv = s_y[ci];
          // Re w is even in x, and the lattice and its pole term are written in x^2 and in
          // cosines of angles odd in x: they take the SIGNED x (no |x| to materialise for the
          // inline-asm constant multiplies, which carry no source modifiers)
          const double xs = __builtin_fma(nu_f, s_is2[ci], s_nu0[ci]);            // s_nu0 holds c1
          const double ax = fabs(xs);
          double V;
          if (path == kPathFarA) V = voigt_far_series<6>(ax, yv);
          else if (path == kPathFarB) V = voigt_far_series<4>(ax, yv);
          else if (path == kPathPlain) V = voigt_plain_wave<0>(xs, yv, s_ky[ci], 0.0, 0.0, ptop);
          else if (path == kPathPlainPole)
            V = voigt_plain_wave<1>(xs, yv, s_ky[ci], s_q[ci], s_cq[ci], ptop);     // s_cq: 2 q e^(y^2)
          else if (path == kPathPlainPoleLite)
            V = voigt_plain_wave<2>(xs, yv, s_ky[ci], 0.0, s_cq[ci], ptop);
          else V = voigt_centred(ax, yv, s_ky[ci], s_q[ci], s_cq[ci], s_tab[tid / RJP_WAVE],
                                 ptop.exp_top);                               // s_q: 2 q e^(y^2) / (1 + q)
          // C V (1 - exp(-h nu / kT)) with 1 - exp(...) = 1 - E0 exp(-a (nu - nu_ref)); to first
          // order in a dnu over the band (unless the path code says otherwise) that is
          // V (A + B dnu), A = C (1 - E0), B = C E0 a staged per cell: two fmas
          if (pc & kPathExpFlag) {
            const double a = s_a[ci];
            const double ce0 = s_E0[ci] / a;                  // C E0
            acc = __builtin_fma(V, (s_C[ci] + ce0) - ce0 * exp(-a * dnu), acc);
          } else {
            acc = __builtin_fma(V, __builtin_fma(s_E0[ci], dnu, s_C[ci]), acc);
          }
        }
      } else {
#pragma unroll 1
        for (int r = 0; r < YC; ++r) {
          const int ci = r * ZT + g * NZP + j;
          const double C = s_C[ci];
          if (C != 0.0) {
            CellLine cl;
            cl.C = C; cl.nu0 = s_nu0[ci]; cl.is2 = s_is2[ci]; cl.y = s_y[ci]; cl.a = s_a[ci];
            cl.E0 = s_E0[ci]; cl.q = s_q[ci]; cl.cq = s_cq[ci];
            const double term = line_term<false>(cl, nu_f, dnu, ln.dnu_max, nullptr);
            if (term == term) acc += term;
          }
        }
      }
      s_acc[j * kRB + slot] = acc;
    }
    __syncthreads();
  }

  if (chan_live) {
    const int64_t base = (int64_t)fi * nx * nz + (int64_t)x * nz + z0 + g * NZP;
#pragma unroll
    for (int j = 0; j < NZP; ++j)
      if (z0 + g * NZP + j < nz) tau[base + j] = s_acc[j * kRB + tid];
  }
}

// ---- map stage (intensity_rrl / flux_rrl at map level) ---------------------------------
__global__ __launch_bounds__(kRB) void rrl_maps_kernel(
    const double* __restrict__ tau_rrl, const double* __restrict__ tau_ff,
    const double* __restrict__ tavg, const double* __restrict__ flux_ff, int64_t npix,
    const double* __restrict__ cflux, const double* __restrict__ hnu_k, int nchan,
    double* __restrict__ flux, double* __restrict__ part) {
  const int64_t p = (int64_t)blockIdx.x * kRB + threadIdx.x;
  const int fch = blockIdx.y;
  const bool live = p < npix;
  const int64_t o = (int64_t)fch * npix + p;
  double s = 0.0;
  if (live) {
    // B_nu(T) ~ 1/(exp(h nu/kT) - 1)   (physics.py:571-574)
    const double bnu = 1.0 / (exp(hnu_k[fch] / tavg[p]) - 1.0);
    // rrls.py:445-447
    s = cflux[fch] * bnu * exp(-tau_ff[o]) * one_minus_exp_neg(tau_rrl[o]);
    if (flux_ff) s += flux_ff[o];
    if (flux) flux[o] = s;
  }
  if (part) {
    __shared__ double red[kRB / RJP_WAVE];
    double v = (live && s == s) ? s : 0.0;
#pragma unroll
    for (int d = RJP_WAVE / 2; d > 0; d >>= 1) v += __shfl_down(v, d, RJP_WAVE);
    if ((threadIdx.x & (RJP_WAVE - 1)) == 0) red[threadIdx.x / RJP_WAVE] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
      double tot = 0.0;
      for (int w = 0; w < kRB / RJP_WAVE; ++w) tot += red[w];
      part[(int64_t)fch * gridDim.x + blockIdx.x] = tot;
    }
  }
}

// ---- launch helpers ---------------------------------------------------------------------
template <typename T, int LF>
static hipError_t rrl_launch_t(const rjp_fields* fl, const BurstsDev& b, bool bursts,
                               double time_s, const LineDev& ln, const double* d_nu,
                               int nchan, double* tau, hipStream_t st) {
  RrlFields<T> f{(const T*)fl->d_nd, (const T*)fl->d_xi, (const T*)fl->d_temp,
                 (const T*)fl->d_pf, (const T*)fl->d_ts, (const T*)fl->d_vy, fl->d_ylo, fl->d_yhi};
  const int ntz = (fl->nz + RrlTile<LF>::ZT - 1) / RrlTile<LF>::ZT;
  dim3 grid((unsigned)(fl->nx * ntz), (unsigned)((nchan + LF - 1) / LF));
  if (bursts)
    hipLaunchKernelGGL((rrl_scan_kernel<T, LF, true>), grid, dim3(kRB), 0, st, f, fl->nx,
                       fl->ny, fl->nz, b, time_s, ln, d_nu, nchan, tau);
  else
    hipLaunchKernelGGL((rrl_scan_kernel<T, LF, false>), grid, dim3(kRB), 0, st, f, fl->nx,
                       fl->ny, fl->nz, b, time_s, ln, d_nu, nchan, tau);
  return hipGetLastError();
}

template <typename T>
static hipError_t rrl_launch_lf(const rjp_fields* fl, const BurstsDev& b, bool bursts,
                                double time_s, const LineDev& ln, const double* d_nu,
                                int nchan, double* tau, hipStream_t st) {
  // 65-128 channels: two blocks of the 64-lane layout (phase 1 runs twice: ~9 % more work) beat
  // one 256-lane block with half its lanes idle (a channel shard of a 256-channel cube on 2 ranks:
  // 338 -> see profiles/r05_cfg4_f64_tau2_bench.json rank_share.cfg3.channels)
  if (nchan > 128) return rrl_launch_t<T, 256>(fl, b, bursts, time_s, ln, d_nu, nchan, tau, st);
  if (nchan > 16) return rrl_launch_t<T, 64>(fl, b, bursts, time_s, ln, d_nu, nchan, tau, st);
  return rrl_launch_t<T, 16>(fl, b, bursts, time_s, ln, d_nu, nchan, tau, st);
}

static void fill_line(const rjp_fields* fl, const rjp_line* line, const double* h_nu, int nchan,
                      LineDev& ln) {
  ln.nu_rest = line->nu_rest; ln.kG = line->kG; ln.kL = line->kL; ln.kappa0 = line->kappa0;
  ln.en_over_k = line->en_over_k; ln.h_over_k = line->h_over_k;
  ln.path0 = fl->csize_au * 149597870700.0 * 1e2;
  double lo = h_nu[0], hi = h_nu[0];
  for (int i = 1; i < nchan; ++i) { lo = h_nu[i] < lo ? h_nu[i] : lo; hi = h_nu[i] > hi ? h_nu[i] : hi; }
  ln.nu_ref = 0.5 * (lo + hi);
  ln.dnu_max = 0.5 * (hi - lo);
}

hipError_t rrl_cells_launch(const rjp_fields* fl, const rjp_bursts* hb, const double* d_ext,
                            double time_s, const rjp_line* line, const double* h_nu,
                            const double* d_nu, int nchan, double* out, hipStream_t st) {
  BurstsDev b;
  const bool bursts = bursts_to_dev(hb, b, d_ext);
  if (bursts && !fl->d_ts) return hipErrorInvalidValue;
  LineDev ln;
  fill_line(fl, line, h_nu, nchan, ln);
  const int64_t n = (int64_t)fl->nx * fl->ny * fl->nz;
  const unsigned blocks = (unsigned)((n + kRB - 1) / kRB);
  auto go = [&](auto tag) {
    using T = decltype(tag);
    RrlFields<T> f{(const T*)fl->d_nd, (const T*)fl->d_xi, (const T*)fl->d_temp,
                   (const T*)fl->d_pf, (const T*)fl->d_ts, (const T*)fl->d_vy, nullptr, nullptr};
    if (bursts)
      hipLaunchKernelGGL((rrl_cells_kernel<T, true>), dim3(blocks), dim3(kRB), 0, st, f, n, b,
                         time_s, ln, d_nu, nchan, out);
    else
      hipLaunchKernelGGL((rrl_cells_kernel<T, false>), dim3(blocks), dim3(kRB), 0, st, f, n, b,
                         time_s, ln, d_nu, nchan, out);
  };
  if (fl->dtype == RJP_F64) go(double{}); else go(float{});
  return hipGetLastError();
}

hipError_t rrl_scan_launch(const rjp_fields* fl, const rjp_bursts* hb, const double* d_ext,
                           double time_s, const rjp_line* line, const double* h_nu,
                           const double* d_nu, int nchan, double* tau, hipStream_t st) {
  BurstsDev b;
  const bool bursts = bursts_to_dev(hb, b, d_ext);
  if (bursts && !fl->d_ts) return hipErrorInvalidValue;
  LineDev ln;
  fill_line(fl, line, h_nu, nchan, ln);
  if (fl->dtype == RJP_F64)
    return rrl_launch_lf<double>(fl, b, bursts, time_s, ln, d_nu, nchan, tau, st);
  return rrl_launch_lf<float>(fl, b, bursts, time_s, ln, d_nu, nchan, tau, st);
}

hipError_t rrl_maps_launch(const double* tau_rrl, const double* tau_ff, const double* tavg,
                           const double* flux_ff, int64_t npix, const double* d_cflux,
                           const double* d_hnu_k, int nchan, double* flux, double* ftot,
                           double* part, hipStream_t st) {
  const unsigned nblk = (unsigned)((npix + kRB - 1) / kRB);
  hipLaunchKernelGGL(rrl_maps_kernel, dim3(nblk, (unsigned)nchan), dim3(kRB), 0, st, tau_rrl,
                     tau_ff, tavg, flux_ff, npix, d_cflux, d_hnu_k, nchan, flux,
                     ftot ? part : nullptr);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return err;
  if (ftot) err = sum_partials_launch(part, nchan, (int)nblk, ftot, st);
  return err;
}

}  // namespace rjp
