// Device-side helpers shared by the gfx950 kernels of librjprt.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rjprt.h"

#define RJP_WAVE 64

// ---- vector loads: 16 B per lane -------------------------------------------------------
#ifndef RJP_NT_LOADS
#define RJP_NT_LOADS 1   /* the fields are streamed exactly once: non-temporal loads, +5 % on cfg4 (6.1 -> 6.4 TB/s) */
#endif
typedef double rjp_d2 __attribute__((ext_vector_type(2)));
typedef float rjp_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void load_vec(const double* __restrict__ p, double (&out)[2]) {
#if RJP_NT_LOADS
  rjp_d2 t = __builtin_nontemporal_load(reinterpret_cast<const rjp_d2*>(p));
#else
  rjp_d2 t = *reinterpret_cast<const rjp_d2*>(p);
#endif
  out[0] = t.x; out[1] = t.y;
}
__device__ __forceinline__ void load_vec(const float* __restrict__ p, double (&out)[4]) {
#if RJP_NT_LOADS
  rjp_f4 t = __builtin_nontemporal_load(reinterpret_cast<const rjp_f4*>(p));
#else
  rjp_f4 t = *reinterpret_cast<const rjp_f4*>(p);
#endif
  out[0] = (double)t.x; out[1] = (double)t.y; out[2] = (double)t.z; out[3] = (double)t.w;
}
__device__ __forceinline__ void load_vec(const double* __restrict__ p, double (&out)[1]) {
  out[0] = *p;
}
__device__ __forceinline__ void load_vec(const float* __restrict__ p, double (&out)[1]) {
  out[0] = (double)*p;
}

// cached (not non-temporal) vector loads / stores of small maps
__device__ __forceinline__ void load_plain(const double* __restrict__ p, double (&out)[2]) {
  rjp_d2 t = *reinterpret_cast<const rjp_d2*>(p);
  out[0] = t.x; out[1] = t.y;
}
__device__ __forceinline__ void load_plain(const double* __restrict__ p, double (&out)[1]) {
  out[0] = *p;
}
__device__ __forceinline__ void store_plain(double* __restrict__ p, const double (&v)[2]) {
  rjp_d2 t; t.x = v[0]; t.y = v[1];
  *reinterpret_cast<rjp_d2*>(p) = t;
}
__device__ __forceinline__ void store_plain(double* __restrict__ p, const double (&v)[1]) {
  p[0] = v[0];
}
// stores of the product cubes (written once, never read back by a kernel of the step)
#ifndef RJP_K2_NT
#define RJP_K2_NT 0
#endif
__device__ __forceinline__ void store_cube(double* __restrict__ p, const double (&v)[2]) {
  rjp_d2 t; t.x = v[0]; t.y = v[1];
#if RJP_K2_NT
  __builtin_nontemporal_store(t, reinterpret_cast<rjp_d2*>(p));
#else
  *reinterpret_cast<rjp_d2*>(p) = t;
#endif
}
__device__ __forceinline__ void store_cube(double* __restrict__ p, const double (&v)[1]) {
#if RJP_K2_NT
  __builtin_nontemporal_store(v[0], p);
#else
  p[0] = v[0];
#endif
}

// ---- burst factor chi(t) (classes.py:442-448, 866-868) ---------------------------------
// Kernel-argument copy of rjp_bursts.  The first RJP_SGPR_BURSTS bursts of each jet travel by
// value (they live in SGPRs / the scalar cache: the fast path of every shipped example); a jet
// with more keeps the rest in a small device table `ext`, read with wave-uniform scalar
// loads: ext[(jet * 3 + k) * next + (i - RJP_SGPR_BURSTS)], k = 0 t0, 1 amp_rel, 2 k2.
// The reference registers any number of bursts (classes.py:245-264, 399-463).
#define RJP_SGPR_BURSTS 8
#define RJP_LOG2E 1.4426950408889634074
struct BurstsDev {
  int n[2];
  double t0[2][RJP_SGPR_BURSTS];
  double amp_rel[2][RJP_SGPR_BURSTS];
  double k2[2][RJP_SGPR_BURSTS];          // -inv2s2 * log2(e): the Gaussians are evaluated in base 2
  const double* ext;      // overflow bursts (nullptr when every jet has <= RJP_SGPR_BURSTS)
  int next;               // overflow capacity per jet = max(n) - RJP_SGPR_BURSTS, or 0
};

static inline int bursts_overflow(const rjp_bursts* hb) {
  if (!hb) return 0;
  const int m = hb->n[0] > hb->n[1] ? hb->n[0] : hb->n[1];
  return m > RJP_SGPR_BURSTS ? m - RJP_SGPR_BURSTS : 0;
}

// doubles of the device table the overflow bursts need (parameters only)
static inline size_t bursts_ext_doubles(const rjp_bursts* hb) {
  return (size_t)bursts_overflow(hb) * 6;
}

// host: fill the parameter part of the overflow table (unused slots contribute 0)
static inline void bursts_fill_ext(const rjp_bursts* hb, double* tab) {
  const int next = bursts_overflow(hb);
  for (int j = 0; j < 2; ++j)
    for (int i = 0; i < next; ++i) {
      const int src = RJP_SGPR_BURSTS + i;
      const bool live = src < hb->n[j];
      tab[(j * 3 + 0) * next + i] = live ? hb->t0[j][src] : 0.0;
      tab[(j * 3 + 1) * next + i] = live ? hb->amp_rel[j][src] : 0.0;
      tab[(j * 3 + 2) * next + i] = live ? -hb->inv2s2[j][src] * RJP_LOG2E : 0.0;
    }
}

// host: rjp_bursts -> kernel-argument copy; returns true when any burst is present.
// `d_ext` = device copy of the table bursts_fill_ext() wrote (nullptr without overflow).
static inline bool bursts_to_dev(const rjp_bursts* hb, BurstsDev& b,
                                 const double* d_ext = nullptr) {
  bool any = false;
  for (int j = 0; j < 2; ++j) {
    b.n[j] = hb ? hb->n[j] : 0;
    if (b.n[j] > 0) any = true;
    for (int i = 0; i < RJP_SGPR_BURSTS; ++i) {
      const bool live = hb && i < hb->n[j];
      b.t0[j][i] = live ? hb->t0[j][i] : 0.0;
      b.amp_rel[j][i] = live ? hb->amp_rel[j][i] : 0.0;   // unused slots contribute 0
      b.k2[j][i] = live ? -hb->inv2s2[j][i] * RJP_LOG2E : 0.0;
    }
  }
  b.next = bursts_overflow(hb);
  b.ext = b.next > 0 ? d_ext : nullptr;
  return any;
}

// Polynomial of exp(r) on |r| <= ln2/2: degree 10, near-minimax with the leading 1, 1, 1/2
// kept (they are inline constants), relative error 7e-16 before rounding -- one FMA shorter
// and closer than the degree-11 Taylor polynomial it replaced (tools/minimax_fit.py).
#define RJP_EXP_C10 2.77658272960759129e-07
#define RJP_EXP_C9 2.76399272005369552e-06
#define RJP_EXP_C8 2.48011840775311536e-05
#define RJP_EXP_C7 1.98411738635095533e-04
#define RJP_EXP_C6 1.38888891559099128e-03
#define RJP_EXP_C5 8.33333337887889360e-03
#define RJP_EXP_C4 4.16666666661282964e-02
#define RJP_EXP_C3 1.66666666665949731e-01

// 2^f on |f| <= 1/2: degree 10, near-minimax with the constant term kept at 1 (relative error
// 4e-16 before rounding, tools/minimax_fit.py).  The burst Gaussians exp(-(t - t0)^2 / 2 s^2)
// are evaluated as 2^(k2 (t - t0)^2) with k2 = -log2(e) / 2 s^2 prepared on the host: no
// multiplication by log2(e), no Cody-Waite pair -- the fraction f = t - rint(t) is exact.
#define RJP_EXP2_C10 7.11175884550745750e-09
#define RJP_EXP2_C9 1.02105064723726014e-07
#define RJP_EXP2_C8 1.32152356340024657e-06
#define RJP_EXP2_C7 1.52526477476260911e-05
#define RJP_EXP2_C6 1.54035308008194563e-04
#define RJP_EXP2_C5 1.33335582464808402e-03
#define RJP_EXP2_C4 9.61812910738071326e-03
#define RJP_EXP2_C3 5.55041086643465811e-02
#define RJP_EXP2_C2 2.40226506959104719e-01
#define RJP_EXP2_C1 6.93147180559951614e-01

__device__ __forceinline__ double exp2_poly(double f) {
  double p = RJP_EXP2_C10;
  p = __builtin_fma(p, f, RJP_EXP2_C9);
  p = __builtin_fma(p, f, RJP_EXP2_C8);
  p = __builtin_fma(p, f, RJP_EXP2_C7);
  p = __builtin_fma(p, f, RJP_EXP2_C6);
  p = __builtin_fma(p, f, RJP_EXP2_C5);
  p = __builtin_fma(p, f, RJP_EXP2_C4);
  p = __builtin_fma(p, f, RJP_EXP2_C3);
  p = __builtin_fma(p, f, RJP_EXP2_C2);
  p = __builtin_fma(p, f, RJP_EXP2_C1);
  return __builtin_fma(p, f, 1.0);
}

// 2^t for t <= 0 (NaN counts as -inf), clamped below at 2^-1021 ~ 4e-308: 15 DP instructions,
// no denormal/overflow paths.  The relative error is that of t itself (|t| * 1.1e-16 * ln 2)
// plus 1e-15: below 6e-15 wherever the Gaussian is above 1e-16.
__device__ __forceinline__ double exp2_nonpos(double t) {
  t = fmax(t, -1021.0);
  const double kd = __builtin_rint(t);
  return __builtin_ldexp(exp2_poly(t - kd), (int)kd);
}

// 2^t for |t| <= 1000 (no clamp)
__device__ __forceinline__ double exp2_any(double t) {
  const double kd = __builtin_rint(t);
  return __builtin_ldexp(exp2_poly(t - kd), (int)kd);
}

// 2^t for t <= 0 to float accuracy (rel. err ~1e-7): 2^k by ldexp, 2^f by the hardware
// v_exp_f32.  Used only with f32 field storage, whose inputs carry a 6e-8 rounding already.
__device__ __forceinline__ double exp2_nonpos_f32acc(double t) {
  t = fmax(t, -1021.0);
  const double kd = __builtin_rint(t);
  const float e = __builtin_amdgcn_exp2f((float)(t - kd));
  return __builtin_ldexp((double)e, (int)kd);
}

template <bool F32ACC>
__device__ __forceinline__ double exp2_burst(double t) {
  return F32ACC ? exp2_nonpos_f32acc(t) : exp2_nonpos(t);
}

// The burst Gaussian where it is evaluated once per (cell, epoch, burst) -- the direct scans,
// K3's per-cell burst factor: 2^(k2 (tl - t0)^2).  The single-epoch scan is bound by HBM AND
// by vector-ALU issue at a power-limited clock (61 instructions per cell, 87 % issue at
// 1.65 GHz: profiles/r03b_cfg4_k1_tau_sq.json), so the instruction count is bandwidth: 2^f
// by a degree-8 near-minimax polynomial (relative error 1.1e-12 before rounding,
// tools/minimax_fit.py; degree 10 / 4e-16 in rounds 1-2): 18 instructions per Gaussian
// instead of 20.  chi carries at most that error, tau = sum a0 chi^2 at most 2.2e-12 -- seven
// orders inside BASELINE.json's 1e-5, and inside the 1e-11 the parity tests hold against the
// reference's maps.  The anchors of the uniform-epoch recurrences keep degree 10 (two
// exponentials per cell and burst serve a whole tile there, and the step ratio's error is
// raised to the power of the step; with a second constant set in the tile kernels they ran
// 1-2 % slower).  A NaN argument counts as -inf (fmax), as before.
// (Forming u = tl sk + c0 with one fma and 2^(-u^2) was tried: both constants are SGPR pairs
// and a VOP3 instruction reads one, so the fma costs two moves -- no gain.)
#define RJP_EXP2_D8 1.33441841430774186e-06
#define RJP_EXP2_D7 1.53142092581519469e-05
#define RJP_EXP2_D6 1.54030982836854641e-04
#define RJP_EXP2_D5 1.33334341574215041e-03
#define RJP_EXP2_D4 9.61812955884247880e-03
#define RJP_EXP2_D3 5.55041095922895744e-02
#define RJP_EXP2_D2 2.40226506947425783e-01
#define RJP_EXP2_D1 6.93147180541232588e-01
// 2^t, t <= 0 or NaN (the argument of a burst Gaussian)
template <bool F32ACC>
__device__ __forceinline__ double exp2_gauss(double targ) {
  const double t = __builtin_fmax(targ, -1021.0);
  const double kd = __builtin_rint(t);
  const double f = t - kd;
  if (F32ACC) return __builtin_ldexp((double)__builtin_amdgcn_exp2f((float)f), (int)kd);
  double p = RJP_EXP2_D8;
  p = __builtin_fma(p, f, RJP_EXP2_D7);
  p = __builtin_fma(p, f, RJP_EXP2_D6);
  p = __builtin_fma(p, f, RJP_EXP2_D5);
  p = __builtin_fma(p, f, RJP_EXP2_D4);
  p = __builtin_fma(p, f, RJP_EXP2_D3);
  p = __builtin_fma(p, f, RJP_EXP2_D2);
  p = __builtin_fma(p, f, RJP_EXP2_D1);
  return __builtin_ldexp(__builtin_fma(p, f, 1.0), (int)kd);
}
template <bool F32ACC>
__device__ __forceinline__ double gauss2(double tl, double t0, double k2) {
  const double d = tl - t0;
  return exp2_gauss<F32ACC>((d * d) * k2);
}

// exp(x) for |x| <= 700, relative error < 1e-14: Cody-Waite reduction + the degree-10
// polynomial of exp(r) above (K3's pole term and the power-law Gaunt factor)
__device__ __forceinline__ double exp_any(double x) {
  const double L2E = 1.4426950408889634074;
  const double LN2_HI = 6.93147180369123816490e-01;
  const double LN2_LO = 1.90821492927058770002e-10;
  double kd = __builtin_rint(x * L2E);
  double r = __builtin_fma(-kd, LN2_HI, x);
  r = __builtin_fma(-kd, LN2_LO, r);
  double p = RJP_EXP_C10;
  p = __builtin_fma(p, r, RJP_EXP_C9);
  p = __builtin_fma(p, r, RJP_EXP_C8);
  p = __builtin_fma(p, r, RJP_EXP_C7);
  p = __builtin_fma(p, r, RJP_EXP_C6);
  p = __builtin_fma(p, r, RJP_EXP_C5);
  p = __builtin_fma(p, r, RJP_EXP_C4);
  p = __builtin_fma(p, r, RJP_EXP_C3);
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  return __builtin_ldexp(p, (int)kd);
}

// 1 - exp(-tau) for tau >= 0 (the map stages: T_b = T_avg (1 - e^-tau), classes.py:1473-1475;
// rrls.py:445-447), relative error < 4e-15 for EVERY tau -- also the thin columns where
// 1 - exp() through libm's exp cancels (tau ~ 1e-6 keeps 10 digits that way): with
// -tau = k ln2 + r,  1 - e^-tau = (1 - 2^k) - 2^k expm1(r), and 1 - 2^k is exact.  ~21
// instructions against ~45 for libm's exp and the subtraction; tau = inf gives 1, NaN gives 1
// (a tau map holds no NaN: its sums are nansums).
__device__ __forceinline__ double one_minus_exp_neg(double tau) {
  const double L2E = 1.4426950408889634074;
  const double LN2_HI = 6.93147180369123816490e-01;
  const double LN2_LO = 1.90821492927058770002e-10;
  const double x = fmax(-tau, -800.0);
  const double kd = __builtin_rint(x * L2E);
  double r = __builtin_fma(-kd, LN2_HI, x);
  r = __builtin_fma(-kd, LN2_LO, r);
  double g = RJP_EXP_C10;                       // exp(r) = 1 + r + r^2/2 + r^3 g(r)
  g = __builtin_fma(g, r, RJP_EXP_C9);
  g = __builtin_fma(g, r, RJP_EXP_C8);
  g = __builtin_fma(g, r, RJP_EXP_C7);
  g = __builtin_fma(g, r, RJP_EXP_C6);
  g = __builtin_fma(g, r, RJP_EXP_C5);
  g = __builtin_fma(g, r, RJP_EXP_C4);
  g = __builtin_fma(g, r, RJP_EXP_C3);
  const double em1 = __builtin_fma(r * r, __builtin_fma(r, g, 0.5), r);     // expm1(r)
  const double s = __builtin_ldexp(1.0, (int)kd);                           // 2^k, k <= 0
  return __builtin_fma(-s, em1, 1.0 - s);
}

// chi for one cell of one jet (wave-uniform loop count; parameters come from SGPRs, those of
// bursts beyond RJP_SGPR_BURSTS from the overflow table)
// A NaN launch time gives NaN (the reference's Gaussians propagate it, classes.py:442-448) --
// unless the jet has no burst at all: its mass-loss rate is then the steady-state constant
// whatever the launch time (classes.py:232-233), chi = 1.
__device__ __forceinline__ double chi_jet(const BurstsDev& b, int jet, double tl) {
  double chi = 1.0;
  const int nb = b.n[jet];
  if (nb > 0 && !(tl == tl)) return __builtin_nan("");
  const int n0 = nb < RJP_SGPR_BURSTS ? nb : RJP_SGPR_BURSTS;
  for (int i = 0; i < n0; ++i)
    chi = __builtin_fma(b.amp_rel[jet][i], gauss2<false>(tl, b.t0[jet][i], b.k2[jet][i]), chi);
  for (int i = RJP_SGPR_BURSTS; i < nb; ++i) {
    const double* e = b.ext + (size_t)(jet * 3) * b.next + (i - RJP_SGPR_BURSTS);
    chi = __builtin_fma(e[b.next], gauss2<false>(tl, e[0], e[2 * b.next]), chi);
  }
  return chi;
}

__device__ __forceinline__ double chi_cell(const BurstsDev& b, bool red, double tl) {
  return red ? chi_jet(b, 0, tl) : chi_jet(b, 1, tl);
}

// chi for a batch of NB independent (cell, epoch) pairs.  The burst loop is outermost and
// the NB exp() evaluations inside it are independent, so their dependent FMA chains overlap.
// A wave whose lanes all sit in one jet reads that jet's parameters from SGPRs; a wave that
// straddles the red/blue plane selects them per lane (unused slots have amp_rel = 0).
// `sg` = the high dwords of the NS = NB / ET signed fields the cells' jet flags sit in (bit 31
// = red jet; pair k belongs to cell k % NS): "any red / any blue lane in the wave" comes from one
// OR and one AND chain over the raw words (three-input ops) instead of a flag per cell.
template <int NB, int NS, bool F32ACC>
__device__ __forceinline__ void chi_batch(const BurstsDev& b, const uint32_t (&sg)[NS],
                                          const double (&tl)[NB], double (&chi)[NB]) {
  static_assert(NB % NS == 0, "pairs per cell");
  uint32_t wor = 0u, wand = 0xffffffffu;
#pragma unroll
  for (int k = 0; k < NS; ++k) { wor |= sg[k]; wand &= sg[k]; }
  const bool wave_red = __builtin_amdgcn_ballot_w64((int)wor < 0) != 0;
  const bool wave_blue = __builtin_amdgcn_ballot_w64((int)wand >= 0) != 0;
  if (!(wave_red && wave_blue)) {
    const int jet = wave_red ? 0 : 1;
    const int nb = b.n[jet];
    const int n0 = nb < RJP_SGPR_BURSTS ? nb : RJP_SGPR_BURSTS;
    // the first burst starts the sums from the constant 1 (no register initialisation)
    if (n0 > 0) {
      const double t0 = b.t0[jet][0], k2 = b.k2[jet][0], amp = b.amp_rel[jet][0];
#pragma unroll
      for (int k = 0; k < NB; ++k) chi[k] = __builtin_fma(amp, gauss2<F32ACC>(tl[k], t0, k2), 1.0);
    } else {
#pragma unroll
      for (int k = 0; k < NB; ++k) chi[k] = 1.0;
    }
    auto one = [&](double t0, double k2, double amp) __attribute__((always_inline)) {
#pragma unroll
      for (int k = 0; k < NB; ++k)
        chi[k] = __builtin_fma(amp, gauss2<F32ACC>(tl[k], t0, k2), chi[k]);
    };
    for (int i = 1; i < n0; ++i) one(b.t0[jet][i], b.k2[jet][i], b.amp_rel[jet][i]);
    for (int i = RJP_SGPR_BURSTS; i < nb; ++i) {
      const double* e = b.ext + (size_t)(jet * 3) * b.next + (i - RJP_SGPR_BURSTS);
      one(e[0], e[2 * b.next], e[b.next]);
    }
  } else {
#pragma unroll
    for (int k = 0; k < NB; ++k) chi[k] = 1.0;
    const int nb = b.n[0] > b.n[1] ? b.n[0] : b.n[1];
    const int n0 = nb < RJP_SGPR_BURSTS ? nb : RJP_SGPR_BURSTS;
    auto one = [&](double t0r, double k2r, double ampr, double t0b, double k2b,
                   double ampb) __attribute__((always_inline)) {
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const bool red = (int)sg[k % NS] < 0;
        const double t0 = red ? t0r : t0b;
        const double k2 = red ? k2r : k2b;
        const double amp = red ? ampr : ampb;
        chi[k] = __builtin_fma(amp, gauss2<F32ACC>(tl[k], t0, k2), chi[k]);
      }
    };
    for (int i = 0; i < n0; ++i)
      one(b.t0[0][i], b.k2[0][i], b.amp_rel[0][i], b.t0[1][i], b.k2[1][i],
          b.amp_rel[1][i]);
    for (int i = RJP_SGPR_BURSTS; i < nb; ++i) {
      const double* r = b.ext + (i - RJP_SGPR_BURSTS);
      const double* u = r + (size_t)3 * b.next;
      one(r[0], r[2 * b.next], r[b.next], u[0], u[2 * b.next], u[b.next]);
    }
  }
}

// Uniformly spaced epochs t_e = t_m + (e - m) dt: for one burst the Gaussian exponents of a
// cell form a quadratic in e, so E_e = exp(arg_e) obeys E_{e+1} = E_e R_e, R_{e+1} = R_e Q
// with Q = exp(-2 inv dt^2) the same for every cell.  Two exponentials per (cell, burst) serve
// the whole tile instead of one per epoch.  The anchor is the middle epoch m; when its Gaussian
// underflows (arg_m < -700) every epoch of the tile is negligible -- the launcher guarantees
// that by using this path only while the tile's half-span is below 28 sigma of the
// narrowest burst.  Base 2 throughout: log2 E_m = k2 vm^2, log2(E_{m+1} / E_m) =
// 2 k2 dt (vm + dt / 2) = a1 (vm + hdt).
#ifndef RJP_TWO_OP
#define RJP_TWO_OP 1           /* 0: A/B build with the three-operation recurrence everywhere */
#endif
#define RJP_STEP_TAB 16                      /* doubles per (jet, burst) of UnifDev::atab */
struct UnifDev {
  int on;                                   // 0 = evaluate every epoch directly
  int nbt;                                  // bursts per jet in `atab` (= max(n[0], n[1]))
  double dt;                                // epoch spacing [s]
  double hdt;                               // dt / 2
  double q[2][RJP_SGPR_BURSTS];             // exp(-2 inv2s2 dt^2)
  double a1[2][RJP_SGPR_BURSTS];            // 2 k2 dt
  const double* qext;                       // q of the overflow bursts: qext[jet * next + i]
  // step table of the two-operation recurrence (waves inside one jet):
  // atab[(jet * nbt + i) * RJP_STEP_TAB + k - 1] = q^(k (k + 1) / 2), k = 1 .. 16
  const double* atab;
};

template <int ET, int UV>
__device__ __forceinline__ void chi_batch_uniform(const BurstsDev& b, const UnifDev& un,
                                                  const bool (&red)[UV],
                                                  const double (&tlm)[UV],   // anchor epoch
                                                  double (&chi)[ET * UV]) {
  constexpr int M = ET / 2;
  constexpr double kDead = -1009.0;         // log2 of the anchor Gaussian below which it is dropped
  bool any_red = false, any_blue = false;
#pragma unroll
  for (int c = 0; c < UV; ++c) { any_red |= red[c]; any_blue |= !red[c]; }
  const bool wave_red = __builtin_amdgcn_ballot_w64(any_red) != 0;
  const bool wave_blue = __builtin_amdgcn_ballot_w64(any_blue) != 0;
  const bool mixed = wave_red && wave_blue;
  const int jet = wave_red ? 0 : 1;
  const int nb = mixed ? (b.n[0] > b.n[1] ? b.n[0] : b.n[1]) : b.n[jet];
  // (a wave inside one jet starts from chi = 1 inside its first burst: the 1 is the addend of
  // that burst's FMAs instead of ET moves per cell)
  if (!RJP_TWO_OP || mixed || nb == 0) {
#pragma unroll
    for (int k = 0; k < ET * UV; ++k) chi[k] = 1.0;
  }
  const int n0 = nb < RJP_SGPR_BURSTS ? nb : RJP_SGPR_BURSTS;
  int i = 0;
#if RJP_TWO_OP
  if (!mixed) {
    // Wave inside one jet (the usual case): E_{m+j} = E_m rup^j q^{j(j-1)/2} and
    // E_{m-j} = E_m rup^{-j} q^{j(j+1)/2}.  The powers of q are the same for every cell of the
    // wave: they come from the step table by SCALAR loads (SGPR operands of the FMAs), so a
    // step costs one multiply + one FMA per direction instead of two multiplies + one FMA.
    // Magnitudes: the launcher keeps the half-span below 28 sigma, so rup^{+-j} E_m stays
    // below exp(420) and the table entries above exp(-420).
    typedef double tab8 __attribute__((ext_vector_type(8)));
    auto apply2 = [&](int c, double t0, double k2, double amp, double a1,
                      const double* __restrict__ tab, bool first) __attribute__((always_inline)) {
      double vm = tlm[c] - t0;
      // The step table is requested HERE, one SMEM round trip ahead of its first use: issued
      // by hand because the scheduler otherwise sinks the scalar loads to the recurrence (to
      // save SGPRs during the exponentials) and every trip then waits for them.  `vm` passes
      // through the statement so that the exponentials stay behind the request.
      tab8 ta, tb;
      if (M > 8)
        asm volatile("s_load_dwordx16 %0, %3, 0x0\n\ts_load_dwordx16 %1, %3, 0x40"
                     : "=&s"(ta), "=&s"(tb), "+v"(vm) : "s"(tab));
      else
        asm volatile("s_load_dwordx16 %0, %2, 0x0" : "=&s"(ta), "+v"(vm) : "s"(tab));
      const double tm = (vm * vm) * k2;
      const bool dead = tm < kDead;
      const double em = exp2_nonpos(tm);
      const double rup = exp2_any((vm + un.hdt) * a1);               // E_{m+1} / E_m
      double ir = __builtin_amdgcn_rcp(rup);
      ir = ir * __builtin_fma(-rup, ir, 2.0);
      const double ae = dead ? 0.0 : amp * em;    // a dead cell adds exactly nothing
      chi[M * UV + c] = (first ? 1.0 : chi[M * UV + c]) + ae;
      double eu = ae, ed = ae;
      const double ru = dead ? 0.0 : rup, rd = dead ? 0.0 : ir;
      if (M > 8) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(ta), "+s"(tb));
      else asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(ta));
      auto T = [&](int k) __attribute__((always_inline)) { return k < 8 ? ta[k] : tb[k - 8]; };
#pragma unroll
      for (int j = 1; j <= M; ++j) {
        if (M + j < ET) {
          eu *= ru;
          const double cu = first ? 1.0 : chi[(M + j) * UV + c];
          chi[(M + j) * UV + c] = j == 1 ? cu + eu : __builtin_fma(eu, T(j - 2), cu);
        }
        ed *= rd;
        chi[(M - j) * UV + c] = __builtin_fma(ed, T(j - 1), first ? 1.0 : chi[(M - j) * UV + c]);
      }
    };
    // the parameters of the NEXT burst are requested before the current one is worked on (a
    // trip is ~100 FP64 instructions and would otherwise start by waiting for the SMEM round
    // trip); the step table of the current burst is requested at the top of its trip and
    // first needed after the two exponentials
    struct BP { double t0, k2, amp, a1; };
    auto ld = [&](int k) __attribute__((always_inline)) {
      const int kk = k < RJP_SGPR_BURSTS - 1 ? k : RJP_SGPR_BURSTS - 1;   // stay inside the table
      return BP{b.t0[jet][kk], b.k2[jet][kk], b.amp_rel[jet][kk], un.a1[jet][kk]};
    };
    if (nb == 0) return;
    BP cur = ld(0);
    {
      const BP nxt = ld(1);
      const double* tab = un.atab + (size_t)(jet * un.nbt) * RJP_STEP_TAB;
#pragma unroll
      for (int c = 0; c < UV; ++c) apply2(c, cur.t0, cur.k2, cur.amp, cur.a1, tab, true);
      cur = nxt;
    }
    for (i = 1; i < n0; ++i) {
      const BP nxt = ld(i + 1);
      const double* tab = un.atab + (size_t)(jet * un.nbt + i) * RJP_STEP_TAB;
#pragma unroll
      for (int c = 0; c < UV; ++c) apply2(c, cur.t0, cur.k2, cur.amp, cur.a1, tab, false);
      cur = nxt;
    }
    for (i = RJP_SGPR_BURSTS; i < nb; ++i) {
      const double* tab = un.atab + (size_t)(jet * un.nbt + i) * RJP_STEP_TAB;
      const double* e = b.ext + (size_t)(jet * 3) * b.next + (i - RJP_SGPR_BURSTS);
      const double k2 = e[2 * b.next];
#pragma unroll
      for (int c = 0; c < UV; ++c) apply2(c, e[0], k2, e[b.next], 2.0 * k2 * un.dt, tab, false);
    }
    return;
  }
#endif
  // Waves that straddle the red/blue plane (and the RJP_TWO_OP = 0 build): one burst, its
  // parameters for the red and the blue jet selected per lane, three operations per step
  auto apply = [&](int c, double t0r, double k2r, double ampr, double qr, double a1r, double t0b,
                   double k2b, double ampb, double qb, double a1b) __attribute__((always_inline)) {
    const bool r = mixed ? red[c] : (jet == 0);
    const double t0 = r ? t0r : t0b;
    const double k2 = r ? k2r : k2b;
    const double amp = r ? ampr : ampb;
    const double q = r ? qr : qb;
    const double a1 = r ? a1r : a1b;
    const double vm = tlm[c] - t0;
    const double tm = (vm * vm) * k2;
    const bool dead = tm < kDead;
    const double em = exp2_nonpos(tm);
    const double rup = exp2_any((vm + un.hdt) * a1);                 // E_{m+1} / E_m
    // E_{m-1} / E_m = q / rup: hardware reciprocal + one Newton step (rup is a finite
    // normal number whenever the cell is not `dead`)
    double ir = __builtin_amdgcn_rcp(rup);
    ir = ir * __builtin_fma(-rup, ir, 2.0);
    const double rdn = q * ir;
    const double a = dead ? 0.0 : amp;          // a dead cell adds exactly nothing ...
    const double em0 = dead ? 0.0 : em;         // ... and must not turn 0 * inf into NaN
    chi[M * UV + c] = __builtin_fma(a, em0, chi[M * UV + c]);
    // the chains towards later and earlier epochs advance in the same trip: two independent
    // dependency chains per burst in program order
    double eu = em0, ru = dead ? 0.0 : rup, ed = em0, rd = dead ? 0.0 : rdn;
#pragma unroll
    for (int j = 1; j <= M; ++j) {
      if (M + j < ET) {
        eu *= ru; ru *= q;
        chi[(M + j) * UV + c] = __builtin_fma(a, eu, chi[(M + j) * UV + c]);
      }
      ed *= rd; rd *= q;
      chi[(M - j) * UV + c] = __builtin_fma(a, ed, chi[(M - j) * UV + c]);
    }
  };
  auto one = [&](int k, int c) __attribute__((always_inline)) {
    apply(c, b.t0[0][k], b.k2[0][k], b.amp_rel[0][k], un.q[0][k], un.a1[0][k], b.t0[1][k],
          b.k2[1][k], b.amp_rel[1][k], un.q[1][k], un.a1[1][k]);
  };
  // two bursts per trip: their exponential chains are independent and interleave
  for (; i + 1 < n0; i += 2) {
#pragma unroll
    for (int c = 0; c < UV; ++c) { one(i, c); one(i + 1, c); }
  }
  for (; i < n0; ++i) {
#pragma unroll
    for (int c = 0; c < UV; ++c) one(i, c);
  }
  for (i = RJP_SGPR_BURSTS; i < nb; ++i) {
    const int k = i - RJP_SGPR_BURSTS;
    const double* r = b.ext + k;
    const double* u = r + (size_t)3 * b.next;
    const double qr = un.qext[k], qb = un.qext[b.next + k];
    const double k2r = r[2 * b.next], k2b = u[2 * b.next];
#pragma unroll
    for (int c = 0; c < UV; ++c)
      apply(c, r[0], k2r, r[b.next], qr, 2.0 * k2r * un.dt, u[0], k2b, u[b.next], qb,
            2.0 * k2b * un.dt);
  }
}

// 1/d: hardware reciprocal + one Newton step (d a finite normal number)
__device__ __forceinline__ double rcp_newton(double d) {
  const double r = __builtin_amdgcn_rcp(d);
  return r * __builtin_fma(-d, r, 2.0);
}

// T^-1.5 with an f32 rsqrt seed + one fp64 Newton step (rel. err < 1e-13); exact-ish slow
// path outside the f32 exponent range and for T == 0 / inf / negative.
__device__ __attribute__((noinline)) double pow_m1p5_slow(double T) {
  return 1.0 / (T * __builtin_sqrt(T));   // NaN for T<0 or NaN, inf for 0, 0 for inf
}
__device__ __forceinline__ double pow_m1p5(double T) {
  if (T > 1e-30 && T < 1e30) {
    double y = (double)__builtin_amdgcn_rsqf((float)T);
    double h = 0.5 * T;
    y = y * __builtin_fma(-h * y, y, 1.5);
    return y * y * y;
  }
  return pow_m1p5_slow(T);
}

// The same for N independent values without per-value branches: the Newton chains interleave
// and the (rare) out-of-range fix-up costs one wave-uniform test.  NaN and negative T come
// out of the fast path as NaN, which is what T^-1.5 is for them.
template <int N>
__device__ __forceinline__ void pow_m1p5_batch(const double (&T)[N], double (&out)[N]) {
  bool odd = false;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    double y = (double)__builtin_amdgcn_rsqf((float)T[k]);
    const double h = 0.5 * T[k];
    y = y * __builtin_fma(-h * y, y, 1.5);
    out[k] = y * y * y;
    odd |= (T[k] >= 0.0 && T[k] <= 1e-30) || T[k] >= 1e30;
  }
  if (__builtin_amdgcn_ballot_w64(odd) != 0) {
#pragma unroll
    for (int k = 0; k < N; ++k)
      if ((T[k] >= 0.0 && T[k] <= 1e-30) || T[k] >= 1e30) out[k] = pow_m1p5_slow(T[k]);
  }
}

// T^-1.35 = T^-1.5 * T^0.15 (power-law Gaunt factor, classes.py:1393, 1426) for N values, as
// w^27 with w = T^(-1/20): seed from the hardware f32 log2 / exp2 (relative error ~2e-7, up to
// 4e-7 at the ends of the f32 range), ONE third-order (Halley) step of the division-free
// inverse-root iteration  w <- w (1 + d/20 + 21 d^2/800),  d = 1 - T w^20  (error after it
// ~0.02 d^3 < 1e-16), then six multiplications.  Relative error of the result < 1e-14
// against a 40-digit pow (27 roundings), ~26 instructions against ~48 for the log/exp chain
// this replaced and ~130 for libm's pow.  T <= 1e-30 (incl. 0, negative) and T >= 1e30 take
// the exact slow path after one wave-uniform test; NaN propagates through the fast path.
template <int N>
__device__ __forceinline__ void pow_m1p35_batch(const double (&T)[N], double (&out)[N]) {
  bool odd = false;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const float lf = __builtin_amdgcn_logf((float)T[k]);                    // log2
    double w = (double)__builtin_amdgcn_exp2f(-0.05f * lf);
    const double w2 = w * w, w4 = w2 * w2, w5 = w4 * w, w10 = w5 * w5;
    const double d = __builtin_fma(-T[k], w10 * w10, 1.0);
    w *= __builtin_fma(d, __builtin_fma(d, 21.0 / 800.0, 0.05), 1.0);
    const double w3 = w * w * w, w9 = w3 * w3 * w3;
    out[k] = w9 * w9 * w9;
    odd |= T[k] <= 1e-30 || T[k] >= 1e30;
  }
  if (__builtin_amdgcn_ballot_w64(odd) != 0) {
#pragma unroll
    for (int k = 0; k < N; ++k)
      if (T[k] <= 1e-30 || T[k] >= 1e30) out[k] = pow_m1p5_slow(T[k]) * pow(T[k], 0.15);
  }
}

__device__ __forceinline__ bool signbit_d(double v) {
  return (__double_as_longlong(v) < 0);
}
__device__ __forceinline__ uint32_t hi_dword(double v) {
  return (uint32_t)((unsigned long long)__double_as_longlong(v) >> 32);
}

// The temperature factor of a cell's free-free optical depth exactly as K1 evaluates it on the
// wide and compact layouts (element-wise identical to the batched forms): T^-1.5 for the scalar
// Gaunt factor, T^-1.35 for the power-law one.  Producers of the tau field rjp_fields.d_a0 use
// it so that a0 == g0 * tpow bit for bit.
__device__ __forceinline__ double tau_weight(double T, int gff_mode) {
  const double in[1] = {T};
  double out[1];
  if (gff_mode == RJP_GFF_POWERLAW) pow_m1p35_batch<1>(in, out);
  else pow_m1p5_batch<1>(in, out);
  return out[0];
}

// ---- splitmix64 counter hash for the synthetic generator -------------------------------
__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__host__ __device__ __forceinline__ double hash_u01(uint64_t seed, uint64_t field,
                                                    uint64_t cell) {
  uint64_t h = splitmix64(seed ^ (field << 60) ^ cell);
  return (double)(h >> 11) * (1.0 / 9007199254740992.0);
}
