// Epoch sweeps by launch-time moments (round 3).
//
// For a fixed sightline the free-free sum of the reference at epoch t_e is
//     sumA_e = sum_y a0_y * chi_jet(y)(t_e - ts_y)^2        (classes.py:861-875, 1395-1432)
// -- a convolution of the sightline's distribution of a0 over LAUNCH TIME with the fixed
// function F_jet = chi_jet^2.  The epoch enters through F only.  So instead of evaluating
// chi for every (cell, epoch) pair (the tiles of ff_scan_kernels.h: 11-13 instructions per
// pair, ALU-bound from 8 epochs on), the launch-time axis [ts_lo, ts_hi] is cut into K bins and
// the per-sightline CHEBYSHEV MOMENTS of a0 are accumulated in ONE pass over the grid,
//     M[p][jet][k][n] = sum_{y in bin k, jet} a0_y T_n(xi_y),    xi = position inside the bin,
// after which ANY number of epochs, uniformly spaced or not, is the small contraction
//     sumA_e[p] = sum_{jet,k,n} M[p][jet][k][n] * W[jet][k][n][e],
// W = Chebyshev coefficients of s -> F_jet(t_e - s) on bin k, computed for the call's bursts and
// epochs (since round 4 on the device: mom_tables_kernel below).  F is entire (a polynomial in
// Gaussians): the expansion converges faster than geometrically, and it is CHECKED -- the
// interpolant is compared with F at 2N+1 points of every (jet, bin, epoch) and the path is used
// only when the worst relative error stays below 1e-11 (the agreement the recurrence tiles have
// with the direct ones); a launch-time range too wide for the narrowest burst simply keeps the
// tiles.  The moment maps themselves depend on neither epochs nor burst parameters: a caller may
// keep them (rjp_fields.d_mom_cache) and later sweeps of the model are contractions only.
//
// Moment pass: launch times are uncorrelated along y in general (and in the synthetic set), so
// a cell's (jet, bin) is random and the accumulators cannot live in registers: a workgroup owns
// 16 z-adjacent sightlines over all y and keeps their 16 x 2 x K x N moments in LDS (up to
// 159 KB), updated with f64 LDS atomics -- N per cell, ~8 LDS cycles per wave-instruction
// (profiles/r03b_cfg5_moments_sq.json: the LDS is busy 72 % of the pass, HBM is not the
// bound) -- then writes them transposed, M_T[idx][p].  The (K, N) shapes trade bins for order
// inside the LDS budget 2 K N <= 1280: the host takes the CHEAPEST shape (fewest atomics per
// cell) whose expansion passes the accuracy check -- (80, 8), (53, 12), (39, 16); at
// 512x4096x512: 3.9 / 4.1 / 5.0 ms (tools/native/moments_probe.hip).  The next rows of a
// thread are fetched (unconditionally, row index clamped) while the atomics of the current
// ones drain: two register sets in ping-pong, counted vmcnt waits.
// Sums of one sightline come from 64 threads in atomic order: results are reproducible to
// rounding, not bit for bit (the tiles are).  Contraction: one lane per sightline, four moment
// rows in flight, the W row of each coefficient through scalar loads, 32 epochs per pass over
// M_T.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "rjp_host.h"

namespace rjp {

#ifndef RJP_MOM_XCD
#define RJP_MOM_XCD 1
#endif
constexpr int kMomSL = 16;                 // sightlines per workgroup
constexpr int kMomBS = 1024;               // threads per workgroup: 16 sightlines x 64 y-rows (16
                                           // waves per CU keep more atomics in flight: 5.2 ms
                                           // against 6.8 with 256 threads, moments_probe.hip)
constexpr int kMomU = 4;                   // rows per register set (two sets in ping-pong)
// (K bins, N moments) shapes, cheapest first; 2 K N <= RJP_MOM_MAX_IDX, a multiple of 4
struct MomShape { int K, N; };
constexpr MomShape kMomShapes[] = {{80, 8}, {53, 12}, {39, 16}};
constexpr int kMomNShapes = sizeof(kMomShapes) / sizeof(kMomShapes[0]);

struct MomDev {
  double s0, inv_h;
  int has_bursts[2];
  int* guard;      // range guard: raised by a weighted cell whose launch time is outside the bins
};

template <int K, int N>
__global__ __launch_bounds__(kMomBS) void moments_kernel(const double* __restrict__ a0,
                                                      const double* __restrict__ ts,
                                                      const int32_t* __restrict__ ylo,
                                                      const int32_t* __restrict__ yhi, int ny,
                                                      int nz, int64_t npix, int64_t npixp,
                                                      MomDev md, double* __restrict__ MT) {
  constexpr int SL = kMomSL, U = kMomU;
  static_assert(2 * K * N <= RJP_MOM_MAX_IDX && (2 * K * N) % 4 == 0, "shape outside the budget");
  extern __shared__ double s_mom[];        // [2][K][N][SL]
  constexpr int TOT = 2 * K * N * SL;
  for (int i = threadIdx.x; i < TOT; i += kMomBS) s_mom[i] = 0.0;
  __syncthreads();
  const int sl = threadIdx.x % SL, yr = threadIdx.x / SL;
  constexpr int YR = kMomBS / SL;
  // tile of 16 sightlines this workgroup owns.  RJP_MOM_XCD (A/B switch, profiles/r05_mom_xcd_ab.log):
  // workgroups are dealt round-robin to the 8 XCDs, so with the identity map z-adjacent tiles
  // (the 128-byte neighbours of a row) run on different XCDs at different times; the XCD-aware
  // map gives every XCD a contiguous range of tiles in dispatch order -- the 32 CUs of an XCD
  // then read 4 KiB-contiguous rows at about the same time
  unsigned tile = blockIdx.x;
#if RJP_MOM_XCD
  {
    const unsigned per = gridDim.x / 8;                    // (the tail past 8 * per: identity)
    if (blockIdx.x < 8 * per) tile = (blockIdx.x % 8) * per + blockIdx.x / 8;
  }
#endif
  const int64_t p = (int64_t)tile * SL + sl;
  const bool live = p < npix;
  const int64_t x = live ? p / nz : 0;
  const int z = live ? (int)(p - x * nz) : 0;
  const int64_t col = x * (int64_t)ny * nz + z;
  const int ya = !live ? 0 : ylo ? ylo[p] : 0;
  const int yb = !live ? 0 : ylo ? yhi[p] : ny;

  // unconditional loads (row index clamped into the grid, the weight is zeroed at use time):
  // no branch around a load, so the waits are counted and the next rows stay in flight
  auto fetch = [&](double (&aa)[U], double (&tt)[U], int ybase) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int y = ybase + u * YR;
      const int64_t o = col + (int64_t)(y < ny ? y : ny - 1) * nz;
      aa[u] = __builtin_nontemporal_load(a0 + o);
      tt[u] = __builtin_nontemporal_load(ts + o);
    }
  };
  auto cell = [&](double av, double tv) __attribute__((always_inline)) {
    const bool red = signbit_d(av);
    double am = __builtin_fmax(__builtin_fabs(av), 0.0);              // nansum: NaN -> 0
    if (!(tv == tv)) {
      // a NaN launch time drops the cell -- unless its jet has no burst: F == 1 there,
      // whatever the bin (classes.py:232-233, 442-448)
      if (md.has_bursts[red ? 0 : 1]) am = 0.0;
      tv = md.s0;
    }
    if (am != 0.0) {
      const double w = (tv - md.s0) * md.inv_h;
      const double kw = __builtin_floor(w);
      const double kf = __builtin_fmin(__builtin_fmax(kw, 0.0), (double)(K - 1));
      // range guard (include/rjprt.h): bins cover [ts_lo, ts_hi] only (kw == K: ts == ts_hi); a
      // finite launch time outside poisons the sightline with NaN and raises the context's flag
      if ((kw < 0.0 || kw > (double)K) && __builtin_fabs(kw) < __builtin_inf()) {
        *md.guard = 1;
        am = __builtin_nan("");
      }
      const double xi = __builtin_fma(2.0, w - kf, -1.0);
      double* base = s_mom + (((red ? 0 : K) + (int)kf) * N) * SL + sl;
      // the recurrence runs on the WEIGHTED polynomials t_n = am T_n(xi) (it is linear): no
      // multiplication per moment (round 4: 63 -> 52 vector instructions per cell)
      double tm = am, tc = am * xi;
      atomicAdd(base, am);
      // (an infinite term -- T = 0 makes T^-1.5 infinite, and the reference's sum with it --
      // goes into the zeroth moment only: its coefficient is the bin average of chi^2 > 0,
      // so the sightline comes out +inf as in the tiles, not inf * T_n(xi) = NaN)
      if (am <= 1.7976931348623157e308) {           // (false for the guard's NaN as well)
        atomicAdd(base + SL, tc);
        const double x2 = 2.0 * xi;
#pragma unroll
        for (int n = 2; n < N; ++n) {
          const double tn = __builtin_fma(x2, tc, -tm);
          tm = tc;
          tc = tn;
          atomicAdd(base + n * SL, tn);
        }
      }
    }
  };

  double a[U], t[U], an[U], tn[U];
  fetch(a, t, ya + yr);
  for (int y0 = ya + yr; y0 < yb; y0 += 2 * YR * U) {        // ping-pong: no register copies
    fetch(an, tn, y0 + YR * U);
#pragma unroll
    for (int u = 0; u < U; ++u) cell(y0 + u * YR < yb ? a[u] : 0.0, t[u]);
    fetch(a, t, y0 + 2 * YR * U);
#pragma unroll
    for (int u = 0; u < U; ++u) cell(y0 + (U + u) * YR < yb ? an[u] : 0.0, tn[u]);
  }
  __syncthreads();
  // transposed flush: M_T[idx][p] (a full 128-byte segment per 16 lanes)
  for (int i = threadIdx.x; i < TOT; i += kMomBS) {
    const int idx = i / SL, s = i % SL;
    MT[(int64_t)idx * npixp + (int64_t)tile * SL + s] = s_mom[i];
  }
}

// sumA[e][p] = sum_idx M_T[idx][p] * W[idx][e], e < ne <= 32; four moment rows in flight per
// lane (one at a time left the loop latency-bound: 0.66 -> 0.47 ms at 512x512 sightlines x 1024;
// round 4, same-buffer A/B on cfg5's 1272 rows: 2 / 4 / 8 / 12 rows 0.80 / 0.58 / 0.64 / 0.74 ms, two
// register sets in ping-pong 0.89-1.07 ms: four plain rows stay).
// Round 5: a lane walks ALL rows of its sightline, so a small map -- an x-slab of a sharded grid:
// 64 x 512 sightlines = 512 waves -- left most of the chip idle and the contraction took as long
// as on the whole map (0.48 ms of a 1.04 ms step, profiles/r05_share_cfg5_8_kernel_stats.csv).
// gridDim.y now cuts the row range into `nsp` chunks (multiples of UI rows); with nsp > 1 the
// chunk sums go to `part[(c * ET + e) * npix + p]` and moments_eval_sum_kernel adds them in a
// fixed order (bit-reproducible for a given nsp).
#ifndef RJP_EVAL_UI
#define RJP_EVAL_UI 4
#endif
#ifndef RJP_EVAL_WAVES
#define RJP_EVAL_WAVES 4096      /* row chunks are added until about this many waves exist */
#endif
constexpr int kEvalMaxSplit = 16;

template <bool SPLIT>
__global__ __launch_bounds__(256) void moments_eval_kernel(const double* __restrict__ MT,
                                                           int64_t npix, int64_t npixp, int nidx,
                                                           int rows_per_chunk,
                                                           const double* __restrict__ W, int ne,
                                                           double scale,
                                                           double* __restrict__ sumA,
                                                           double* __restrict__ part) {
  constexpr int ET = RJP_MOM_TILE, UI = RJP_EVAL_UI;
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= npix) return;
  // (the unsplit instance is round 4's kernel to the instruction: one range, no partial sums)
  const int i_lo = SPLIT ? blockIdx.y * rows_per_chunk : 0;
  const int i_hi = SPLIT ? min(nidx, i_lo + rows_per_chunk) : nidx;
  double acc[ET];
#pragma unroll
  for (int e = 0; e < ET; ++e) acc[e] = 0.0;
  for (int i0 = i_lo; i0 < i_hi; i0 += UI) {
    double m[UI];
#pragma unroll
    for (int j = 0; j < UI; ++j) m[j] = MT[(int64_t)(i0 + j) * npixp + p];
#pragma unroll
    for (int j = 0; j < UI; ++j) {
      const double* w = W + (size_t)(i0 + j) * ET;          // wave-uniform: scalar loads
#pragma unroll
      for (int e = 0; e < ET; ++e) acc[e] = __builtin_fma(m[j], w[e], acc[e]);
    }
  }
  if (SPLIT) {
#pragma unroll
    for (int e = 0; e < ET; ++e)
      if (e < ne) part[((int64_t)blockIdx.y * ET + e) * npix + p] = acc[e];
    return;
  }
#pragma unroll
  for (int e = 0; e < ET; ++e)
    if (e < ne) sumA[(int64_t)e * npix + p] = acc[e] * scale;      // (scale == 1: exact)
}

__global__ __launch_bounds__(256) void moments_eval_sum_kernel(const double* __restrict__ part,
                                                               int nsp, int64_t npix, double scale,
                                                               double* __restrict__ sumA) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int e = blockIdx.y;
  if (p >= npix) return;
  double s = 0.0;
  for (int c = 0; c < nsp; ++c) s += part[((int64_t)c * RJP_MOM_TILE + e) * npix + p];
  sumA[(int64_t)e * npix + p] = s * scale;
}

// row chunks of the contraction on a map of `npix` sightlines (1 on maps that fill the chip)
static int mom_eval_split(int64_t npix) {
  const int64_t waves = (npix + RJP_WAVE - 1) / RJP_WAVE;
  int nsp = 1;
  while (nsp < kEvalMaxSplit && waves * nsp < RJP_EVAL_WAVES) nsp *= 2;
  return nsp;
}

// ---- per-block min / max of a field, NaN ignored (rjp_field_range) -----------------------------
template <typename T>
__global__ __launch_bounds__(256) void field_range_kernel(const T* __restrict__ f, int64_t n,
                                                          double* __restrict__ part) {
  double lo = __builtin_inf(), hi = -__builtin_inf();
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const double v = (double)f[i];
    if (v == v) { lo = __builtin_fmin(lo, v); hi = __builtin_fmax(hi, v); }
  }
#pragma unroll
  for (int d = RJP_WAVE / 2; d > 0; d >>= 1) {
    lo = __builtin_fmin(lo, __shfl_xor(lo, d, RJP_WAVE));
    hi = __builtin_fmax(hi, __shfl_xor(hi, d, RJP_WAVE));
  }
  __shared__ double s_lo[256 / RJP_WAVE], s_hi[256 / RJP_WAVE];
  if ((threadIdx.x & (RJP_WAVE - 1)) == 0) { s_lo[threadIdx.x / RJP_WAVE] = lo; s_hi[threadIdx.x / RJP_WAVE] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 256 / RJP_WAVE; ++w) { lo = __builtin_fmin(lo, s_lo[w]); hi = __builtin_fmax(hi, s_hi[w]); }
    part[2 * blockIdx.x] = lo;
    part[2 * blockIdx.x + 1] = hi;
  }
}

hipError_t field_range_launch(const void* d_field, int64_t n, int dtype, double* d_part,
                              hipStream_t st) {
  if (dtype == RJP_F64)
    hipLaunchKernelGGL(field_range_kernel<double>, dim3(RJP_RANGE_BLOCKS), dim3(256), 0, st,
                       (const double*)d_field, n, d_part);
  else
    hipLaunchKernelGGL(field_range_kernel<float>, dim3(RJP_RANGE_BLOCKS), dim3(256), 0, st,
                       (const float*)d_field, n, d_part);
  return hipGetLastError();
}

// any finite entry outside [lo, hi]?  -> *flag = 1 (the range guard's check of a new range)
template <typename T>
__global__ __launch_bounds__(256) void range_check_kernel(const T* __restrict__ f, int64_t n,
                                                          double lo, double hi,
                                                          int* __restrict__ flag) {
  bool bad = false;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const double v = (double)__builtin_nontemporal_load(f + i);
    bad |= (v < lo || v > hi) && __builtin_fabs(v) < __builtin_inf();
  }
  if (bad) *flag = 1;
}

hipError_t range_check_launch(const void* d_field, int64_t n, int dtype, double lo, double hi,
                              int* d_flag, hipStream_t st) {
  if (dtype == RJP_F64)
    hipLaunchKernelGGL(range_check_kernel<double>, dim3(RJP_RANGE_BLOCKS * 4), dim3(256), 0, st,
                       (const double*)d_field, n, lo, hi, d_flag);
  else
    hipLaunchKernelGGL(range_check_kernel<float>, dim3(RJP_RANGE_BLOCKS * 4), dim3(256), 0, st,
                       (const float*)d_field, n, lo, hi, d_flag);
  return hipGetLastError();
}

// ---- coefficient tables, built and checked on the device ---------------------------------------
// One thread per (shape, jet, bin, epoch): the Chebyshev coefficients of s -> F_jet(t_e - s) on the
// bin from N node values (DCT with the staged matrix), then the interpolant against F at 2N+1
// equispaced points (Clenshaw) -- the worst relative error of a shape goes to err[shape] through
// an integer atomicMax on the bits of the (non-negative) double; NaN compares above everything
// and rejects the shape.  (Rounds 1-3 built these tables on the host: ~3e5 exp per request, more
// than the sweep they served.)
struct MomTabArgs {
  int ncand, E, nb[2];
  int K[RJP_MOM_MAX_CAND], N[RJP_MOM_MAX_CAND];
  long long w_off[RJP_MOM_MAX_CAND], tab_off[RJP_MOM_MAX_CAND];
  long long off_b[2], off_e;
  double s0, span;
};

__device__ __forceinline__ double burst_F_dev(const double* __restrict__ b, int n, double tl) {
  double chi = 1.0;
  for (int i = 0; i < n; ++i) {
    const double d = tl - b[i];
    chi += b[n + i] * exp(-d * d * b[2 * n + i]);
  }
  return chi * chi;
}

__global__ __launch_bounds__(256) void mom_tables_kernel(const double* __restrict__ tab,
                                                         MomTabArgs a, double* __restrict__ W,
                                                         unsigned long long* __restrict__ err) {
  constexpr int ET = RJP_MOM_TILE, NMAX = RJP_MOM_NMAX;
  const int c = blockIdx.y;
  const int K = a.K[c], N = a.N[c];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 2 * K * a.E) return;
  const int e = i % a.E, jk = i / a.E, j = jk / K, k = jk - j * K;
  const int nidx = 2 * K * N;
  double* col = W + a.w_off[c] + (size_t)(e / ET) * nidx * ET + (size_t)((j * K + k) * N) * ET + e % ET;
  if (a.nb[j] <= 0) { col[0] = 1.0; return; }                       // F == 1 (the rest stays 0)
  const double* xn = tab + a.tab_off[c];
  const double* cs = xn + N;
  const double* bj = tab + a.off_b[j];
  const double te = tab[a.off_e + e];
  const double h = a.span / K, ck = a.s0 + (k + 0.5) * h;
  double f[NMAX], cf[NMAX];
  for (int m = 0; m < N; ++m) f[m] = burst_F_dev(bj, a.nb[j], te - (ck + 0.5 * h * xn[m]));
  for (int n = 0; n < N; ++n) {
    double s = 0.0;
    for (int m = 0; m < N; ++m) s += f[m] * cs[n * N + m];
    cf[n] = s * (n == 0 ? 1.0 : 2.0) / N;
    col[(size_t)n * ET] = cf[n];
  }
  const int NT = 2 * N + 1;
  double worst = 0.0;
  for (int m = 0; m < NT; ++m) {
    const double xv = -1.0 + 2.0 * m / (NT - 1);
    double b1 = 0.0, b2 = 0.0;
    for (int n = N - 1; n >= 1; --n) { const double b0 = 2.0 * xv * b1 - b2 + cf[n]; b2 = b1; b1 = b0; }
    const double val = xv * b1 - b2 + cf[0];
    const double ref = burst_F_dev(bj, a.nb[j], te - (ck + 0.5 * h * xv));
    const double er = fabs(val - ref) / ref;                         // F = chi^2 > 0; 0/0 -> NaN
    worst = (er <= worst) ? worst : er;
  }
  atomicMax(err + c, (unsigned long long)__double_as_longlong(worst));
}

// ---- host side --------------------------------------------------------------------------------
size_t moments_workspace_bytes(int64_t npix) {
  const int64_t npixp = (npix + kMomSL - 1) / kMomSL * kMomSL;
  return (size_t)RJP_MOM_MAX_IDX * (size_t)npixp * sizeof(double) + 256;
}

// ... plus, on small maps, the chunk sums of the split contraction behind the moment maps
size_t moments_scan_workspace_bytes(int64_t npix) {
  const int nsp = mom_eval_split(npix);
  return moments_workspace_bytes(npix) +
         (nsp > 1 ? (size_t)nsp * RJP_MOM_TILE * (size_t)npix * sizeof(double) : 0);
}

void moments_release(MomPlan& mp) {
  if (mp.d_W) (void)hipFree(mp.d_W);
  if (mp.d_err) (void)hipFree(mp.d_err);
  if (mp.h_err) (void)hipHostFree(mp.h_err);
  mp.d_W = nullptr; mp.capW = 0; mp.d_err = nullptr; mp.h_err = nullptr; mp.d_Wsel = nullptr;
  mp.key_E = -1; mp.key_ok = false; mp.ok = false;
}

// Can this scan take a moment path?  Decides between the launch-time-ordered layout (when the
// caller attached one: any order up to 32 over its bins) and the LDS moments (three shapes inside
// the LDS budget), applies the cost model, and either recognises the previous request (1) or
// prepares the staged table of a new one (2).
int moments_plan(const rjp_fields* fl, const rjp_bursts* hb, const double* epochs, int n_epochs,
                 int mode, bool want_em, size_t work_bytes, MomPlan& mp) {
  mp.ok = false;
  if (!hb || (hb->n[0] <= 0 && hb->n[1] <= 0)) return 0;
  if (n_epochs < RJP_MOM_MIN_EPOCHS) return 0;
  // (with EM maps: the tau layout with its em0 field attached -- a second pass weighs by em0)
  if (scan_layout(fl, mode, want_em) != LAY_TAU || !fl->d_ts) return 0;
  if (!(fl->ts_hi >= fl->ts_lo) || !std::isfinite(fl->ts_lo) || !std::isfinite(fl->ts_hi) ||
      (fl->ts_lo == 0.0 && fl->ts_hi == 0.0))
    return 0;                                                   // range not provided
  if (work_bytes < moments_scan_workspace_bytes((int64_t)fl->nx * fl->nz)) return 0;
  for (int e = 0; e < n_epochs; ++e)
    if (!std::isfinite(epochs[e])) return 0;
  const bool lt = fl->d_lt_cells && fl->d_lt_rowoff && fl->d_lt_aux && fl->lt_K >= 1 &&
                  fl->lt_K <= RJP_LT_MAX_K && !want_em && n_epochs <= RJP_LT_MAX_EPOCHS;
  bool lds = true;                       // may the LDS moment pass run (cost model)?
  if (fl->occupied_cells >= 0) {
    // Cost model (seconds on one MI355X, from the cfg5-size measurements of round 3): the tiles
    // pay per (cell, epoch) pair -- 0.40 ps in the uniform-epoch recurrence, 0.93 ps when every
    // epoch is evaluated directly -- on the cells inside the occupied y-ranges; the moment path
    // pays 4.7 ps per such cell once (its most expensive shape), plus per SIGHTLINE 10 KiB of
    // moments written and read back (3.6 ns) and 1.1 ns per contraction pass of 32 epochs.
    // Short or sparsely filled sightlines keep the tiles.  (A caller that attached the
    // launch-time-ordered layout has paid for it: that path is taken whenever it is accurate.)
    const double npix = (double)fl->nx * fl->nz;
    const double cells = fl->occupied_cells > 0 ? (double)fl->occupied_cells : npix * fl->ny;
    bool uniform = n_epochs >= 4;
    const double dt = n_epochs > 1 ? (epochs[n_epochs - 1] - epochs[0]) / (n_epochs - 1) : 0.0;
    for (int e = 0; e < n_epochs && uniform; ++e)
      uniform = std::fabs(epochs[e] - (epochs[0] + e * dt)) <= 1e-9 * std::fabs(dt);
    // (EM maps: a second moment pass + contraction; the tiles carry a third field and a
    // second set of sums: 0.51 ps per pair in the 32-epoch recurrence on cfg5's grid)
    const double t_tiles = cells * n_epochs * (uniform ? 0.40e-12 : 0.93e-12) * (want_em ? 1.25 : 1.0);
    const double t_mom = (cells * 4.7e-12 +
                          npix * (3.6e-9 + 1.1e-9 * ((n_epochs + RJP_MOM_TILE - 1) / RJP_MOM_TILE))) *
                         (want_em ? 2.0 : 1.0);
    lds = t_mom < 0.8 * t_tiles;
  }
  // a caller-kept cache of the moment maps of one of the LDS shapes (rjp_fields.d_mom_cache):
  // that shape is tried first -- its sweep is a contraction only, cheaper than every other path
  int cache_shape = -1;
  if (fl->d_mom_cache && !want_em)
    for (int sh = 0; sh < kMomNShapes; ++sh)
      if (kMomShapes[sh].K == fl->mom_cache_K && kMomShapes[sh].N == fl->mom_cache_N) cache_shape = sh;
  if (cache_shape >= 0) lds = true;
  if (!lt && !lds) return 0;
  const int ltK = lt ? fl->lt_K : 0;
  // same request as last time?  (the tables are still on the device)
  if (mp.key_E == n_epochs && mp.key_lo == fl->ts_lo && mp.key_hi == fl->ts_hi &&
      mp.key_ltK == (lds ? ltK : -2 - ltK) + 1000 * (cache_shape + 1) && mp.key_epochs.size() == (size_t)n_epochs &&
      std::memcmp(mp.key_epochs.data(), epochs, sizeof(double) * n_epochs) == 0 &&
      mp.key_n[0] == hb->n[0] && mp.key_n[1] == hb->n[1]) {
    bool same = true;
    size_t o = 0;
    for (int j = 0; j < 2 && same; ++j)
      for (int i = 0; i < hb->n[j] && same; ++i, o += 3)
        same = mp.key_bursts[o] == hb->t0[j][i] && mp.key_bursts[o + 1] == hb->amp_rel[j][i] &&
               mp.key_bursts[o + 2] == hb->inv2s2[j][i];
    if (same) { mp.ok = mp.key_ok; return mp.ok ? 1 : 0; }
  }
  mp.key_E = n_epochs; mp.key_lo = fl->ts_lo; mp.key_hi = fl->ts_hi;
  mp.key_ltK = (lds ? ltK : -2 - ltK) + 1000 * (cache_shape + 1);   // (the candidate set depends on all three)
  mp.key_epochs.assign(epochs, epochs + n_epochs);
  mp.key_n[0] = hb->n[0]; mp.key_n[1] = hb->n[1];
  mp.key_bursts.clear();
  double inv_max = 0.0;                                         // the narrowest burst
  for (int j = 0; j < 2; ++j)
    for (int i = 0; i < hb->n[j]; ++i) {
      mp.key_bursts.push_back(hb->t0[j][i]);
      mp.key_bursts.push_back(hb->amp_rel[j][i]);
      mp.key_bursts.push_back(hb->inv2s2[j][i]);
      if (!(hb->inv2s2[j][i] <= inv_max)) inv_max = hb->inv2s2[j][i];     // (NaN propagates)
    }
  mp.key_ok = false;
  mp.has_bursts[0] = hb->n[0] > 0;
  mp.has_bursts[1] = hb->n[1] > 0;
  mp.nchunk = (n_epochs + RJP_MOM_TILE - 1) / RJP_MOM_TILE;
  const double span = fl->ts_hi - fl->ts_lo;
  const double sigma_min = std::sqrt(0.5 / inv_max);            // NaN / 0 for degenerate widths
  // candidate shapes, cheapest first.  RJP_MOM_SHAPE=<i> pins one (debug builds' A/B switch).
  // Deterministic guard beside the sampled accuracy check: a shape whose node spacing h / N
  // exceeds the narrowest burst's sigma could let that burst fall between all nodes AND all
  // test points (every table would then pass with F ~ 1): such shapes are not even tried.
  int pin = -1;
#ifdef RJP_DEBUG_SWITCHES
  if (const char* e = getenv("RJP_MOM_SHAPE")) pin = atoi(e);
#endif
  mp.cands.clear();
  auto consider = [&](int K, int N, int path) {
    const double h = span > 0.0 ? span / K : 0.0;
    if (!(sigma_min >= h / N)) return;                          // also rejects NaN / inf widths
    if ((int)mp.cands.size() < RJP_MOM_MAX_CAND) mp.cands.push_back(MomCand{K, N, path, 0, 0});
  };
  // the cached shape first, then the layout (any order up to 32 over its bins), then the three
  // LDS shapes
  if (cache_shape >= 0) consider(kMomShapes[cache_shape].K, kMomShapes[cache_shape].N, 1);
  if (lt)
    for (int N = 8; N <= RJP_MOM_NMAX; N += 4) consider(ltK, N, 2);
  if (lds)
    for (int sh = 0; sh < kMomNShapes; ++sh)
      if (pin < 0 || sh == pin) consider(kMomShapes[sh].K, kMomShapes[sh].N, 1);
  if (mp.cands.empty()) return 0;
  // the staged table: [jet 0: t0.., amp.., inv2s2..][jet 1: ...][epochs][per shape: x[N], cs[N][N]]
  mp.stage.clear();
  for (int j = 0; j < 2; ++j) {
    mp.off_bursts[j] = mp.stage.size();
    for (int k = 0; k < 3; ++k)
      for (int i = 0; i < hb->n[j]; ++i)
        mp.stage.push_back(k == 0 ? hb->t0[j][i] : k == 1 ? hb->amp_rel[j][i] : hb->inv2s2[j][i]);
  }
  mp.off_epochs = mp.stage.size();
  mp.stage.insert(mp.stage.end(), epochs, epochs + n_epochs);
  const double pi = 3.14159265358979323846;
  mp.w_total = 0;
  for (MomCand& c : mp.cands) {
    c.tab_off = mp.stage.size();
    for (int i = 0; i < c.N; ++i) mp.stage.push_back(std::cos(pi * (i + 0.5) / c.N));
    for (int n = 0; n < c.N; ++n)
      for (int i = 0; i < c.N; ++i) mp.stage.push_back(std::cos(pi * n * (i + 0.5) / c.N));
    c.w_off = mp.w_total;
    mp.w_total += (size_t)mp.nchunk * 2 * c.K * c.N * RJP_MOM_TILE;
  }
  mp.s0 = fl->ts_lo;
  return 2;
}

hipError_t moments_build(MomPlan& mp, const double* d_stage, hipStream_t st) {
  hipError_t e;
  if (!mp.d_err) {
    e = hipMalloc((void**)&mp.d_err, RJP_MOM_MAX_CAND * sizeof(unsigned long long));
    if (e != hipSuccess) return e;
    e = hipHostMalloc((void**)&mp.h_err, RJP_MOM_MAX_CAND * sizeof(unsigned long long), hipHostMallocDefault);
    if (e != hipSuccess) return e;
  }
  if (mp.w_total > mp.capW) {
    // (tables of an earlier request may still be read by kernels in flight on this stream)
    e = hipStreamSynchronize(st);
    if (e != hipSuccess) return e;
    if (mp.d_W) (void)hipFree(mp.d_W);
    mp.d_W = nullptr; mp.capW = 0;
    e = hipMalloc((void**)&mp.d_W, mp.w_total * sizeof(double));
    if (e != hipSuccess) return e;
    mp.capW = mp.w_total;
  }
  e = hipMemsetAsync(mp.d_W, 0, mp.w_total * sizeof(double), st);
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(mp.d_err, 0, RJP_MOM_MAX_CAND * sizeof(unsigned long long), st);
  if (e != hipSuccess) return e;
  MomTabArgs a;
  a.ncand = (int)mp.cands.size();
  a.E = mp.key_E;
  a.nb[0] = mp.key_n[0]; a.nb[1] = mp.key_n[1];
  int kmax = 1;
  for (int c = 0; c < a.ncand; ++c) {
    a.K[c] = mp.cands[c].K; a.N[c] = mp.cands[c].N;
    a.w_off[c] = (long long)mp.cands[c].w_off; a.tab_off[c] = (long long)mp.cands[c].tab_off;
    kmax = std::max(kmax, a.K[c]);
  }
  a.off_b[0] = (long long)mp.off_bursts[0]; a.off_b[1] = (long long)mp.off_bursts[1];
  a.off_e = (long long)mp.off_epochs;
  a.s0 = mp.key_lo; a.span = mp.key_hi - mp.key_lo;
  const unsigned nbx = (unsigned)((2 * kmax * a.E + 255) / 256);
  hipLaunchKernelGGL(mom_tables_kernel, dim3(nbx, (unsigned)a.ncand), dim3(256), 0, st, d_stage, a,
                     mp.d_W, mp.d_err);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  e = hipMemcpyAsync(mp.h_err, mp.d_err, RJP_MOM_MAX_CAND * sizeof(unsigned long long),
                     hipMemcpyDeviceToHost, st);
  if (e != hipSuccess) return e;
  e = hipStreamSynchronize(st);                                  // the one sync of a new request
  if (e != hipSuccess) return e;
  mp.key_ok = false;
  for (int c = 0; c < a.ncand && !mp.key_ok; ++c) {
    double w;
    std::memcpy(&w, &mp.h_err[c], sizeof(double));
    mp.worst = w;
    mp.K = mp.cands[c].K; mp.N = mp.cands[c].N; mp.path = mp.cands[c].path;
    mp.d_Wsel = mp.d_W + mp.cands[c].w_off;
    mp.key_ok = w <= RJP_MOM_TOL;                                // (NaN fails)
  }
  const double span = mp.key_hi - mp.key_lo;
  mp.inv_h = span > 0.0 ? mp.K / span : 1.0;
  mp.s0 = mp.key_lo;
  mp.ok = mp.key_ok;
  return hipSuccess;
}

template <int K, int N>
static hipError_t moments_pass(const rjp_fields* fl, const double* weights, const MomDev& md,
                               int64_t npix, int64_t npixp, double* ws, bool& attr_set,
                               hipStream_t st) {
  const size_t shm = (size_t)2 * K * N * kMomSL * sizeof(double);
  // (more than 64 KB of dynamic LDS must be allowed explicitly: once per context, i.e. per device)
  if (!attr_set) {
    const hipError_t e = hipFuncSetAttribute((const void*)moments_kernel<K, N>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL((moments_kernel<K, N>), dim3((unsigned)(npixp / kMomSL)), dim3(kMomBS), shm,
                     st, weights, (const double*)fl->d_ts, fl->d_ylo, fl->d_yhi,
                     fl->ny, fl->nz, npix, npixp, md, ws);
  return hipGetLastError();
}

// `weights` = the field whose launch-time moments are taken (a0 for the optical-depth sums,
// em0 for the emission measure: both carry the jet flag in their sign bit), `scale` = the
// constant factor of the result (1 for the sums of a0).
// `ws`: the moment maps (the caller's workspace, or its moment cache); `part`: room for the chunk
// sums of the split contraction (moments_scan_workspace_bytes() - moments_workspace_bytes() bytes;
// inside the caller's workspace, never inside a moment cache).
hipError_t moments_run(const rjp_fields* fl, const MomPlan& mp, int n_epochs,
                       double* sumA, double* ws, double* part, hipStream_t st,
                       const double* weights, double scale, int* d_guard, bool skip_pass) {
  const double* d_W = mp.d_Wsel;
  const int64_t npix = (int64_t)fl->nx * fl->nz;
  const int64_t npixp = (npix + kMomSL - 1) / kMomSL * kMomSL;
  MomDev md;
  md.s0 = mp.s0; md.inv_h = mp.inv_h;
  md.has_bursts[0] = mp.has_bursts[0]; md.has_bursts[1] = mp.has_bursts[1];
  md.guard = d_guard;
  // `skip_pass`: `ws` already holds this model's moment maps of this shape (a caller-kept cache)
  hipError_t err = skip_pass ? hipSuccess : hipErrorInvalidValue;
  if (skip_pass) {}
  else if (mp.K == 80 && mp.N == 8) err = moments_pass<80, 8>(fl, weights, md, npix, npixp, ws, mp.attr_set[0], st);
  else if (mp.K == 53 && mp.N == 12) err = moments_pass<53, 12>(fl, weights, md, npix, npixp, ws, mp.attr_set[1], st);
  else if (mp.K == 39 && mp.N == 16) err = moments_pass<39, 16>(fl, weights, md, npix, npixp, ws, mp.attr_set[2], st);
  if (err != hipSuccess) return err;
  const int nidx = 2 * mp.K * mp.N;
  const int nsp = part ? mom_eval_split(npix) : 1;
  // (chunks of whole UI-row groups; nidx is a multiple of 4)
  const int rows = ((nidx + nsp - 1) / nsp + RJP_EVAL_UI - 1) / RJP_EVAL_UI * RJP_EVAL_UI;
  const int nsp_used = (nidx + rows - 1) / rows;
  for (int c = 0; c < mp.nchunk; ++c) {
    const int ne = std::min(RJP_MOM_TILE, n_epochs - c * RJP_MOM_TILE);
    double* out = sumA + (int64_t)c * RJP_MOM_TILE * npix;
    if (nsp_used > 1)
      hipLaunchKernelGGL(moments_eval_kernel<true>, dim3((unsigned)((npix + 255) / 256), (unsigned)nsp_used),
                         dim3(256), 0, st, ws, npix, npixp, nidx, rows,
                         d_W + (size_t)c * nidx * RJP_MOM_TILE, ne, scale, out, part);
    else
      hipLaunchKernelGGL(moments_eval_kernel<false>, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0,
                         st, ws, npix, npixp, nidx, rows, d_W + (size_t)c * nidx * RJP_MOM_TILE, ne,
                         scale, out, (double*)nullptr);
    err = hipGetLastError();
    if (err != hipSuccess) return err;
    if (nsp_used > 1) {
      hipLaunchKernelGGL(moments_eval_sum_kernel, dim3((unsigned)((npix + 255) / 256), (unsigned)ne),
                         dim3(256), 0, st, part, nsp_used, npix, scale, out);
      err = hipGetLastError();
      if (err != hipSuccess) return err;
    }
  }
  return hipSuccess;
}

}  // namespace rjp
