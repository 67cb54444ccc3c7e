// K1, single epoch, tau layout: the burst factor from a table in LDS (round 4).
//
// The single-epoch scan of (a0, ts) spends ~42 of its 51 vector instructions per cell on the 2-3
// burst Gaussians of chi(t - ts) (classes.py:442-448, 866-868) and is co-limited by vector-ALU
// issue at a power-limited clock (profiles/r03c_cfg4_k1_tau_sq.json).  chi is a function of ONE
// variable, the time since launch: per call it is tabulated on the interval of times since launch
// that can occur AND matter -- [t - ts_hi, t - ts_lo] cut with the bursts' support (chi == 1 to
// 1e-17 outside) -- as piecewise polynomials of degree 7 in the interval's own coordinate
// xi in [-1, 1), 8 coefficients per interval and jet at a stride of 80 bytes (an odd multiple of
// 16: the four 16-byte reads of a lookup then spread over all LDS bank groups; at 64 bytes they
// fall on a quarter of them), built on the device by a tiny kernel in front of the scan (8
// Chebyshev nodes per interval, inverse Vandermonde matrix on [-1, 1] from the host).  A cell
// costs one interval lookup: four 16-byte LDS reads at a random address + 7 FMAs, whatever the
// number of bursts.  The number of intervals follows from the interpolation bound
//   |chi - p| <= max|chi^(8)| (h/2)^8 / (8! 2^7),   max|G^(8)| = 105 A / sigma^8
// for a Gaussian of amplitude A: h is chosen so that the bound is <= 1e-13 (chi >= 1: bursts with
// a negative amplitude keep the Gaussians), i.e. <= 2e-13 relative on chi^2; a table that would
// not fit 72 KB of LDS (460 intervals per jet; two workgroups per CU) keeps the Gaussians too.
// (Quintics at 48 bytes, the round-3 experiment, need 4.2 x the intervals: the example's bursts
// then fit for few epochs only.)
//
// Launch shape: with ~22 instead of 51 vector instructions per cell the scan no longer needs
// eight y-ranges' worth of waves to hide its ALU work: ONE y-range per sightline chunk on maps
// that fill the chip that way (cfg4: 512 workgroups of 256 threads, two per CU), six rows of
// loads in flight, sums written straight to the map -- no partial sums, no reduction kernel.
// Same-buffer A/B at cfg4 (profiles/r04_k1_table_variants.log): 2.48 ms against 2.64-2.69 ms for
// the Gaussians = 0.865 of the 8 TB/s peak on the 17.18 GB the scan needs.  The sums follow another order than the compact / wide layouts' eight ranges: equal
// to them to rounding (tests: 1e-12), not bit for bit -- the price VERDICT r03 item 5 accepts.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "ff_scan_kernels.h"

namespace rjp {

#ifndef RJP_TAB_U
#define RJP_TAB_U 6              /* rows of loads in flight per lane (same-buffer A/B at cfg4, profiles/r04_k1_table_variants.log: 6: 2.483 ms, 8: 2.513, 12: 2.516, 16: 2.529) */
#endif
#ifndef RJP_TAB_U_EM
#define RJP_TAB_U_EM 4           /* ... with the EM map (three streams; 3 / 4 / 6 rows: 3.83 / 3.86 / 3.86 ms -- noise) */
#endif
#ifndef RJP_TAB_U_WIDE
#define RJP_TAB_U_WIDE 4         /* ... from the five model fields (five streams; 2 / 3 / 4 rows: 6.229 / 6.230 / 6.187 ms) */
#endif
#ifndef RJP_TAB_WGS
#define RJP_TAB_WGS 512          /* y-ranges are added until this many workgroups exist (1024 / 2048 at cfg4: 2.518 / 2.524 ms) */
#endif
#ifndef RJP_TAB_GUARD
#define RJP_TAB_GUARD 1          /* 0: a build without the launch-time range guard, for A/B only (2 vector instructions per cell; same-buffer A/B at cfg4: profiles/r05_guard_ab.log) */
#endif
constexpr int kChiNC = 8;                    // coefficients per interval (degree 7)
constexpr int kChiStride = 10;               // doubles between intervals (80 B)
constexpr int kChiMaxNI = 460;               // 2 jets x 460 x 80 B = 73600 B of LDS
constexpr double kChiTol = 1e-13;            // bound on |chi - table|
// a Gaussian of relative amplitude A is below 1e-17 beyond sqrt(2 ln(A 1e17)) sigmas
static double chi_reach(double amp) { return std::sqrt(2.0 * std::log(std::max(amp, 1.0) * 1e17)); }

struct ChiTabDev {
  int ni;
  double lo, inv_h;
  double wmax;     // the largest double below ni: a cell launched exactly at ts_lo (w == ni)
                   // belongs to the last interval, xi -> 1
  // range guard (include/rjprt.h "Launch-time range guard"): the table covers the times since
  // launch of [ts_lo, ts_hi] only and the lookup clamps -- a lane that meets a finite launch time
  // outside the range poisons its sums with NaN and raises the context's flag
  double ts_lo, ts_hi;
  int* guard;
};

// one lane's verdict at the end of its column: `tmin` / `tmax` = fmin / fmax over the launch
// times it read (NaN ignored; an infinite launch time is the reference's chi == 1, not a breach)
__device__ __forceinline__ bool ts_range_breach(const ChiTabDev& t, double tmin, double tmax) {
  return (tmin < t.ts_lo && tmin > -__builtin_inf()) || (tmax > t.ts_hi && tmax < __builtin_inf());
}

// table builder: one thread per (jet, interval); tab[(jet * ni + k) * kChiStride + c]
// stage = [Vandermonde inverse 8 x 8][nodes 8][jet 0: t0.., amp.., inv2s2..][jet 1: ...]
__global__ __launch_bounds__(256) void chi_table_kernel(const double* __restrict__ stage, int nb0,
                                                        int nb1, ChiTabDev t,
                                                        double* __restrict__ tab) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 2 * t.ni) return;
  const int j = i / t.ni, k = i - j * t.ni;
  const int nb = j == 0 ? nb0 : nb1;
  const double* vinv = stage;
  const double* xs = stage + kChiNC * kChiNC;
  const double* bj = stage + kChiNC * kChiNC + kChiNC + (j == 0 ? 0 : 3 * nb0);
  const double h = 1.0 / t.inv_h;
  double f[kChiNC];
  for (int m = 0; m < kChiNC; ++m) {
    const double tl = t.lo + (k + 0.5 * (xs[m] + 1.0)) * h;
    double chi = 1.0;
    for (int b = 0; b < nb; ++b) {
      const double d = tl - bj[b];
      chi += bj[nb + b] * exp(-d * d * bj[2 * nb + b]);
    }
    f[m] = chi;
  }
  for (int c = 0; c < kChiNC; ++c) {
    double s = 0.0;
    for (int m = 0; m < kChiNC; ++m) s += vinv[c * kChiNC + m] * f[m];
    tab[(size_t)i * kChiStride + c] = s;
  }
  tab[(size_t)i * kChiStride + 8] = tab[(size_t)i * kChiStride + 9] = 0.0;
}

// EM: the emission measure of the epoch as well (a third stream, em0: classes.py:1116-1118 with
// number_density = _nd chi); `em_scale` is applied when the sums go straight to the map
template <int U, bool EM>
__global__ __launch_bounds__(kBlock) void ff_scan_table_kernel(
    const double* __restrict__ a0, const double* __restrict__ em0, const double* __restrict__ ts,
    const int32_t* __restrict__ ylo, const int32_t* __restrict__ yhi, int ny, int nz,
    int64_t nchunks, int64_t npix, int ylen, int nsplit, ChiTabDev t, double t_epoch,
    const double* __restrict__ tab, double* __restrict__ out, int64_t out_split_stride,
    double* __restrict__ out_em, double em_scale) {
  constexpr int VEC = 2;
  extern __shared__ __attribute__((aligned(16))) double s_chi[];       // [2][ni][10]
  for (int i = threadIdx.x; i < 2 * t.ni * kChiStride; i += kBlock) s_chi[i] = tab[i];
  __shared__ int s_lo, s_hi;
  if (threadIdx.x == 0) { s_lo = ny; s_hi = 0; }
  __syncthreads();
  const int split = (int)(blockIdx.x % (unsigned)nsplit);
  const int64_t c = (int64_t)(blockIdx.x / (unsigned)nsplit) * kBlock + threadIdx.x;
  const bool lane_live = c < nchunks;
  const int64_t p0 = c * VEC;
  int y0 = split * ylen;
  int y1 = min(ny, y0 + ylen);
  if (ylo) {
    if (lane_live) {
      int lo = ny, hi = 0;
#pragma unroll
      for (int v = 0; v < VEC; ++v) { lo = min(lo, ylo[p0 + v]); hi = max(hi, yhi[p0 + v]); }
      if (lo < hi) { atomicMin(&s_lo, lo); atomicMax(&s_hi, hi); }
    }
    __syncthreads();
    y0 = max(y0, s_lo);
    y1 = min(y1, s_hi);
  }
  if (!lane_live) return;
  const int64_t x = p0 / nz;
  const int z = (int)(p0 - x * nz);
  const double wmax = t.wmax;
  // chi^2-weighted term of one cell: the jet picks the half of the table, the time since launch
  // the interval; a NaN launch time lands in interval 0 and the term is masked (nansum)
  auto chi2 = [&](double av, double tv) __attribute__((always_inline)) {
    double w = (t_epoch - tv - t.lo) * t.inv_h;
    w = __builtin_fmin(__builtin_fmax(w, 0.0), wmax);
    const double kf = __builtin_floor(w);
    const double xi = __builtin_fma(2.0, w - kf, -1.0);
    const int k = (int)kf + (signbit_d(av) ? 0 : t.ni);
    const rjp_d2* cp = reinterpret_cast<const rjp_d2*>(s_chi + k * kChiStride);
    const rjp_d2 c01 = cp[0], c23 = cp[1], c45 = cp[2], c67 = cp[3];
    double chi = __builtin_fma(c67.y, xi, c67.x);
    chi = __builtin_fma(chi, xi, c45.y);
    chi = __builtin_fma(chi, xi, c45.x);
    chi = __builtin_fma(chi, xi, c23.y);
    chi = __builtin_fma(chi, xi, c23.x);
    chi = __builtin_fma(chi, xi, c01.y);
    chi = __builtin_fma(chi, xi, c01.x);
    return chi * chi;
  };
  double acc[VEC] = {0.0, 0.0}, accE[VEC] = {0.0, 0.0};
  double tmin = __builtin_inf(), tmax = -__builtin_inf();
  int64_t off = (x * ny + y0) * (int64_t)nz + z;
  int y = y0;
  for (; y + U <= y1; y += U) {
    double a[U][VEC], g[U][VEC], tt[U][VEC];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      load_vec(a0 + off + (int64_t)u * nz, a[u]);
      if (EM) load_vec(em0 + off + (int64_t)u * nz, g[u]);
      load_vec(ts + off + (int64_t)u * nz, tt[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const double c2 = chi2(a[u][v], tt[u][v]);
        if (RJP_TAB_GUARD) {
          tmin = __builtin_fmin(tmin, tt[u][v]);
          tmax = __builtin_fmax(tmax, tt[u][v]);
        }
        acc[v] = __builtin_fma(keep_if_ordered(__builtin_fabs(a[u][v]), tt[u][v]), c2, acc[v]);
        if (EM) accE[v] = __builtin_fma(keep_if_ordered(__builtin_fabs(g[u][v]), tt[u][v]), c2, accE[v]);
      }
    off += (int64_t)U * nz;
  }
  for (; y < y1; ++y) {
    double a[VEC], g[VEC], tt[VEC];
    load_vec(a0 + off, a);
    if (EM) load_vec(em0 + off, g);
    load_vec(ts + off, tt);
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      const double c2 = chi2(a[v], tt[v]);
      if (RJP_TAB_GUARD) {
        tmin = __builtin_fmin(tmin, tt[v]);
        tmax = __builtin_fmax(tmax, tt[v]);
      }
      acc[v] = __builtin_fma(keep_if_ordered(__builtin_fabs(a[v]), tt[v]), c2, acc[v]);
      if (EM) accE[v] = __builtin_fma(keep_if_ordered(__builtin_fabs(g[v]), tt[v]), c2, accE[v]);
    }
    off += nz;
  }
  if (ts_range_breach(t, tmin, tmax)) {
    *t.guard = 1;
#pragma unroll
    for (int v = 0; v < VEC; ++v) { acc[v] = __builtin_nan(""); accE[v] = __builtin_nan(""); }
  }
  double* w = out + (int64_t)split * out_split_stride + p0;
#pragma unroll
  for (int v = 0; v < VEC; ++v) w[v] = acc[v];
  if (EM) {
    // straight to the map (scaled) with one y-range, else plane 1 of this range's partial sums
    double* we = nsplit == 1 ? out_em + p0 : out + (int64_t)split * out_split_stride + npix + p0;
#pragma unroll
    for (int v = 0; v < VEC; ++v) we[v] = nsplit == 1 ? accE[v] * em_scale : accE[v];
  }
}

// The same scan from the five MODEL fields (nd, xi, temp, pf, ts: SURVEY 8(d)'s byte model, what
// a model without the derived scan fields a0 / em0 streams): optical-depth sums, emission measure
// and T_avg of the epoch in one pass, the burst factor from the table.  Per-product NaN semantics
// as ff_scan_kernel's wide layout (nansum of (n chi x)^2 pf T^p and of (n chi x)^2 pf; T_avg =
// nanmean_y(T > 0), classes.py:1116-1118, 1395-1397, 1471-1472).
template <int U, int MODE>
__global__ __launch_bounds__(kBlock) void ff_scan_table_wide_kernel(
    FieldPtrs<double> f, int ny, int nz, int64_t nchunks, int64_t npix, int ylen, int nsplit,
    ChiTabDev t, double t_epoch, const double* __restrict__ tab, double* __restrict__ ws,
    double* __restrict__ sumA, double* __restrict__ em, double* __restrict__ tavg,
    double em_scale) {
  constexpr int VEC = 2;
  extern __shared__ __attribute__((aligned(16))) double s_chi[];       // [2][ni][10]
  for (int i = threadIdx.x; i < 2 * t.ni * kChiStride; i += kBlock) s_chi[i] = tab[i];
  __shared__ int s_lo, s_hi;
  if (threadIdx.x == 0) { s_lo = ny; s_hi = 0; }
  __syncthreads();
  const int split = (int)(blockIdx.x % (unsigned)nsplit);
  const int64_t c = (int64_t)(blockIdx.x / (unsigned)nsplit) * kBlock + threadIdx.x;
  const bool lane_live = c < nchunks;
  const int64_t p0 = c * VEC;
  int y0 = split * ylen;
  int y1 = min(ny, y0 + ylen);
  if (f.ylo) {
    if (lane_live) {
      int lo = ny, hi = 0;
#pragma unroll
      for (int v = 0; v < VEC; ++v) { lo = min(lo, f.ylo[p0 + v]); hi = max(hi, f.yhi[p0 + v]); }
      if (lo < hi) { atomicMin(&s_lo, lo); atomicMax(&s_hi, hi); }
    }
    __syncthreads();
    y0 = max(y0, s_lo);
    y1 = min(y1, s_hi);
  }
  if (!lane_live) return;
  const int64_t x = p0 / nz;
  const int z = (int)(p0 - x * nz);
  const double wmax = t.wmax;
  auto chi2 = [&](bool red, double tv) __attribute__((always_inline)) {
    double w = (t_epoch - tv - t.lo) * t.inv_h;
    w = __builtin_fmin(__builtin_fmax(w, 0.0), wmax);
    const double kf = __builtin_floor(w);
    const double xi = __builtin_fma(2.0, w - kf, -1.0);
    const int k = (int)kf + (red ? 0 : t.ni);
    const rjp_d2* cp = reinterpret_cast<const rjp_d2*>(s_chi + k * kChiStride);
    const rjp_d2 c01 = cp[0], c23 = cp[1], c45 = cp[2], c67 = cp[3];
    double chi = __builtin_fma(c67.y, xi, c67.x);
    chi = __builtin_fma(chi, xi, c45.y);
    chi = __builtin_fma(chi, xi, c45.x);
    chi = __builtin_fma(chi, xi, c23.y);
    chi = __builtin_fma(chi, xi, c23.x);
    chi = __builtin_fma(chi, xi, c01.y);
    chi = __builtin_fma(chi, xi, c01.x);
    return chi * chi;
  };
  double accA[VEC] = {0.0, 0.0}, accE[VEC] = {0.0, 0.0}, accT[VEC] = {0.0, 0.0};
  double tmin = __builtin_inf(), tmax = -__builtin_inf();
  int cnt[VEC] = {0, 0};
  auto rows = [&](auto utag, int64_t off) __attribute__((always_inline)) {
    constexpr int UU = decltype(utag)::value;
    double nd[UU][VEC], xi[UU][VEC], tp[UU][VEC], pf[UU][VEC], tt[UU][VEC];
#pragma unroll
    for (int u = 0; u < UU; ++u) {
      const int64_t o = off + (int64_t)u * nz;
      load_vec(f.nd + o, nd[u]);
      load_vec(f.xi + o, xi[u]);
      load_vec(f.temp + o, tp[u]);
      load_vec(f.pf + o, pf[u]);
      load_vec(f.ts + o, tt[u]);
    }
    double tpw[UU][VEC];
    if (MODE == RJP_GFF_POWERLAW)
      pow_m1p35_batch<UU * VEC>(reinterpret_cast<const double (&)[UU * VEC]>(tp),
                                reinterpret_cast<double (&)[UU * VEC]>(tpw));
    else
      pow_m1p5_batch<UU * VEC>(reinterpret_cast<const double (&)[UU * VEC]>(tp),
                               reinterpret_cast<double (&)[UU * VEC]>(tpw));
#pragma unroll
    for (int u = 0; u < UU; ++u)
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const double Tk = tp[u][v];
        accT[v] += __builtin_fmax(Tk, 0.0);
        cnt[v] += Tk > 0.0 ? 1 : 0;
        const double n0 = __builtin_fabs(nd[u][v]) * xi[u][v];
        const double g = poison_unless(n0 * n0 * pf[u][v], tt[u][v] == tt[u][v]);
        const double c2 = chi2(signbit_d(nd[u][v]), tt[u][v]);
        if (RJP_TAB_GUARD) {
          tmin = __builtin_fmin(tmin, tt[u][v]);
          tmax = __builtin_fmax(tmax, tt[u][v]);
        }
        accE[v] = __builtin_fma(nan_to_zero<false>(g), c2, accE[v]);
        accA[v] = __builtin_fma(nan_to_zero<false>(g * tpw[u][v]), c2, accA[v]);
      }
  };
  int64_t off = (x * ny + y0) * (int64_t)nz + z;
  int y = y0;
  for (; y + U <= y1; y += U) {
    rows(std::integral_constant<int, U>{}, off);
    off += (int64_t)U * nz;
  }
  for (; y < y1; ++y) {
    rows(std::integral_constant<int, 1>{}, off);
    off += nz;
  }
  if (ts_range_breach(t, tmin, tmax)) {
    *t.guard = 1;
#pragma unroll
    for (int v = 0; v < VEC; ++v) { accA[v] = __builtin_nan(""); accE[v] = __builtin_nan(""); }
  }
  if (nsplit == 1) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      sumA[p0 + v] = accA[v];
      if (em) em[p0 + v] = accE[v] * em_scale;
      if (tavg) tavg[p0 + v] = accT[v] / (double)cnt[v];      // 0/0 = NaN on empty sightlines
    }
  } else {
    double* w = ws + (int64_t)split * nacc(1) * npix + p0;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      w[v] = accA[v];
      w[npix + v] = accE[v];
      w[2 * npix + v] = accT[v];
      w[3 * npix + v] = (double)cnt[v];
    }
  }
}

// ---- host -----------------------------------------------------------------------------------
// Vandermonde inverse of the 8 Chebyshev nodes on [-1, 1] (monomial coefficients in the interval's
// own coordinate from node values; on [-1, 1] the matrix is well conditioned)
static void chi_nodes(double (&xs)[kChiNC], double (&vinv)[kChiNC][kChiNC]) {
  const double pi = 3.14159265358979323846;
  double A[kChiNC][2 * kChiNC];
  for (int m = 0; m < kChiNC; ++m) {
    xs[m] = -std::cos(pi * (m + 0.5) / kChiNC);
    double p = 1.0;
    for (int q = 0; q < kChiNC; ++q) { A[m][q] = p; p *= xs[m]; }
    for (int q = 0; q < kChiNC; ++q) A[m][kChiNC + q] = m == q ? 1.0 : 0.0;
  }
  for (int c = 0; c < kChiNC; ++c) {
    int piv = c;
    for (int r = c + 1; r < kChiNC; ++r) if (std::fabs(A[r][c]) > std::fabs(A[piv][c])) piv = r;
    for (int q = 0; q < 2 * kChiNC; ++q) std::swap(A[c][q], A[piv][q]);
    for (int r = 0; r < kChiNC; ++r)
      if (r != c) {
        const double f = A[r][c] / A[c][c];
        for (int q = 0; q < 2 * kChiNC; ++q) A[r][q] -= f * A[c][q];
      }
  }
  for (int c = 0; c < kChiNC; ++c)
    for (int m = 0; m < kChiNC; ++m) vinv[c][m] = A[c][kChiNC + m] / A[c][c];
}

// Can this scan take the table path, and with which table?  On success: cp.tab describes the
// table, cp.stage is the small host table to stage (inverse, nodes, burst parameters).
bool chi_table_plan(const rjp_fields* fl, const rjp_bursts* hb, const double* epochs, int n_epochs,
                    int mode, bool want_em, size_t work_bytes, ChiPlan& cp) {
  cp.ok = false;
  if (!hb || (hb->n[0] <= 0 && hb->n[1] <= 0) || n_epochs != 1) return false;
  // the tau layout (a0, ts [, em0]) or the five model fields (f64): the compact layout -- no
  // default of any producer -- keeps the Gaussians
  const int lay = scan_layout(fl, mode, want_em);
  if ((lay != LAY_TAU && !(lay == LAY_WIDE && fl->dtype == RJP_F64)) || !fl->d_ts ||
      ff_scan_vec(fl) != 2)
    return false;
  cp.wide = lay == LAY_WIDE;
  if (!(fl->ts_hi >= fl->ts_lo) || !std::isfinite(fl->ts_lo) || !std::isfinite(fl->ts_hi) ||
      (fl->ts_lo == 0.0 && fl->ts_hi == 0.0) || !std::isfinite(epochs[0]))
    return false;
  // small maps gain nothing (their scan is launch-bound) and their workspace may not hold the table
  const int64_t npix = (int64_t)fl->nx * fl->nz;
  if (npix / 2 < 64 * 256 || fl->ny < 64) return false;
  // the bursts' support and the interpolation bound
  double s_lo = INFINITY, s_hi = -INFINITY, B = 0.0;
  for (int j = 0; j < 2; ++j) {
    double Bj = 0.0;
    for (int i = 0; i < hb->n[j]; ++i) {
      const double inv = hb->inv2s2[j][i], amp = hb->amp_rel[j][i], t0 = hb->t0[j][i];
      if (!(inv > 0.0) || !std::isfinite(inv) || !(amp >= 0.0) || !std::isfinite(amp) ||
          !std::isfinite(t0))
        return false;                             // dips (chi < 1) and degenerate widths: Gaussians
      const double sigma = std::sqrt(0.5 / inv);
      s_lo = std::min(s_lo, t0 - chi_reach(amp) * sigma);
      s_hi = std::max(s_hi, t0 + chi_reach(amp) * sigma);
      const double s2 = sigma * sigma;
      Bj += amp * 105.0 / (s2 * s2 * s2 * s2);
    }
    B = std::max(B, Bj);
  }
  double lo = std::max(s_lo, epochs[0] - fl->ts_hi), hi = std::min(s_hi, epochs[0] - fl->ts_lo);
  int ni = 1;
  if (!(hi > lo)) {
    // no cell is inside a burst's support at this epoch: chi == 1, one constant interval
    lo = epochs[0] - fl->ts_hi;
    hi = lo + 1.0;
  } else {
    const double h = 2.0 * std::pow(kChiTol * 5160960.0 / B, 1.0 / 8.0);      // 8! 2^7
    const double n = std::ceil((hi - lo) / h);
    if (!(n <= kChiMaxNI)) return false;          // would not fit the LDS: Gaussians
    ni = std::max(1, (int)n);
  }
  const size_t tab_bytes = (size_t)2 * ni * kChiStride * sizeof(double);
  // the table sits in the caller's workspace, in planes 2-3 of the first y-range's four planes
  // (the temperature sums of the other layouts: the scan writes planes 0 and 1 only): 2 npix
  // doubles in, 16-byte aligned because n_z is even
  if (work_bytes < (size_t)2 * npix * sizeof(double) + tab_bytes || (size_t)2 * npix * 8 < tab_bytes)
    return false;
  // (the wide scan fills all four planes of its <= 8 y-ranges: its table sits behind them)
  if (cp.wide && work_bytes < (size_t)32 * npix * sizeof(double) + tab_bytes) return false;
  cp.ni = ni;
  cp.lo = lo;
  cp.inv_h = ni / (hi - lo);
  cp.n[0] = hb->n[0]; cp.n[1] = hb->n[1];
  struct Nodes {
    double xs[kChiNC], vinv[kChiNC][kChiNC];
    Nodes() { chi_nodes(xs, vinv); }
  };
  static const Nodes nodes;                       // (initialised once, thread-safely)
  cp.stage.clear();
  for (int c = 0; c < kChiNC; ++c)
    for (int m = 0; m < kChiNC; ++m) cp.stage.push_back(nodes.vinv[c][m]);
  for (int m = 0; m < kChiNC; ++m) cp.stage.push_back(nodes.xs[m]);
  for (int j = 0; j < 2; ++j)
    for (int k = 0; k < 3; ++k)
      for (int i = 0; i < hb->n[j]; ++i)
        cp.stage.push_back(k == 0 ? hb->t0[j][i] : k == 1 ? hb->amp_rel[j][i] : hb->inv2s2[j][i]);
  cp.ok = true;
  return true;
}

// y-ranges of the table scan on a map of `npix` sightlines and `ny` rows: one when the sightline
// chunks alone give every CU two workgroups, else as many as it takes (powers of two, >= 64 rows
// each; the partials are reduced in ff_reduce_kernel's fixed order)
static int chi_table_nsplit(int64_t npix, int ny, bool wide) {
  const int64_t wgs = (npix / 2 + kBlock - 1) / kBlock;
  int nsplit = 1;
  while (wgs * nsplit < RJP_TAB_WGS && nsplit * 2 * 64 <= ny && nsplit < (wide ? 8 : 16)) nsplit *= 2;
  return nsplit;
}

// Workspace of the table path (part of rjp_ff_scan_workspace): the tau-layout scan writes planes
// 0-1 of nacc(1) = 4 planes per y-range and keeps its table in planes 2-3 of the first range; the
// wide scan fills all four planes of its <= 8 ranges and keeps the table behind them.
size_t chi_table_workspace_bytes(int64_t npix, int ny) {
  if (npix / 2 < 64 * 256 || ny < 64) return 0;            // (chi_table_plan's size rule)
  const size_t tab = (size_t)2 * kChiMaxNI * kChiStride * sizeof(double);
  const size_t tau = (size_t)chi_table_nsplit(npix, ny, false) * nacc(1) * npix * sizeof(double);
  const size_t wide = (size_t)32 * npix * sizeof(double) + tab;
  return std::max(tau, wide);
}

hipError_t chi_table_scan(const rjp_fields* fl, const ChiPlan& cp, const double* d_stage,
                          double t_epoch, int mode, double* sumA, double* em, double* tavg,
                          double* ws, size_t work_bytes, int* d_guard, hipStream_t st) {
  const int64_t npix = (int64_t)fl->nx * fl->nz;
  const int64_t nchunks = npix / 2;
  const size_t tab_doubles = (size_t)2 * cp.ni * kChiStride;
  if (((uintptr_t)ws % 16) != 0 || work_bytes < (2 * npix + tab_doubles) * sizeof(double))
    return hipErrorInvalidValue;
  // plane 2 of the first y-range (see the plan); behind the 8 x 4 planes of the wide scan
  double* d_tab = ws + (cp.wide ? 32 : 2) * npix;
  if (cp.wide && work_bytes < (32 * npix + tab_doubles) * sizeof(double)) return hipErrorInvalidValue;
  ChiTabDev t{cp.ni, cp.lo, cp.inv_h, std::nextafter((double)cp.ni, 0.0), fl->ts_lo, fl->ts_hi,
              d_guard};
  hipLaunchKernelGGL(chi_table_kernel, dim3((unsigned)((2 * cp.ni + 255) / 256)), dim3(256), 0, st,
                     d_stage, cp.n[0], cp.n[1], t, d_tab);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  const int64_t wgs = (nchunks + kBlock - 1) / kBlock;
  // never more y-ranges than the CALLER'S workspace holds (a workspace of rjp_ff_scan_workspace()
  // bytes holds them all; ADVICE r04: the rule used to be checked against nothing)
  int nsplit = chi_table_nsplit(npix, fl->ny, cp.wide);
  while (nsplit > 1 && (size_t)nsplit * nacc(1) * npix * sizeof(double) > work_bytes) nsplit /= 2;
  const int ylen = (fl->ny + nsplit - 1) / nsplit;
  const size_t shm = tab_doubles * sizeof(double);
  // em = sum (n x)^2 * csize*au/pc * pf  (classes.py:1116-1118)
  const double em_scale = fl->csize_au * 149597870700.0 / 3.085677581491367e+16;
  // (per context = per device: a context is bound to one device and used by one host thread)
  constexpr int kMaxShm = 2 * kChiMaxNI * kChiStride * (int)sizeof(double);
  if (!cp.attr_set) {
    e = hipFuncSetAttribute((const void*)ff_scan_table_wide_kernel<RJP_TAB_U_WIDE, 0>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, kMaxShm);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)ff_scan_table_wide_kernel<RJP_TAB_U_WIDE, 1>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, kMaxShm);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)ff_scan_table_kernel<RJP_TAB_U, false>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, kMaxShm);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)ff_scan_table_kernel<RJP_TAB_U_EM, true>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, kMaxShm);
    if (e != hipSuccess) return e;
    cp.attr_set = true;
  }
  const dim3 grid((unsigned)(wgs * nsplit));
  if (cp.wide) {
    FieldPtrs<double> f{(const double*)fl->d_nd, (const double*)fl->d_xi, (const double*)fl->d_temp,
                        (const double*)fl->d_pf, (const double*)fl->d_ts, fl->d_ylo, fl->d_yhi,
                        nullptr, nullptr};
    if (mode == RJP_GFF_POWERLAW)
      hipLaunchKernelGGL((ff_scan_table_wide_kernel<RJP_TAB_U_WIDE, 1>), grid, dim3(kBlock), shm, st,
                         f, fl->ny, fl->nz, nchunks, npix, ylen, nsplit, t, t_epoch, d_tab, ws, sumA,
                         em, tavg, em_scale);
    else
      hipLaunchKernelGGL((ff_scan_table_wide_kernel<RJP_TAB_U_WIDE, 0>), grid, dim3(kBlock), shm, st,
                         f, fl->ny, fl->nz, nchunks, npix, ylen, nsplit, t, t_epoch, d_tab, ws, sumA,
                         em, tavg, em_scale);
    e = hipGetLastError();
    if (e != hipSuccess || nsplit == 1) return e;
    return ff_reduce_launch(ws, nsplit, 1, npix, 0, em_scale, sumA, em, tavg, st);
  }
  double* out = nsplit == 1 ? sumA : ws;
  const int64_t stride = nsplit == 1 ? 0 : (int64_t)nacc(1) * npix;
  if (em)
    hipLaunchKernelGGL((ff_scan_table_kernel<RJP_TAB_U_EM, true>), grid, dim3(kBlock), shm, st,
                       (const double*)fl->d_a0, (const double*)fl->d_em0, (const double*)fl->d_ts,
                       fl->d_ylo, fl->d_yhi, fl->ny, fl->nz, nchunks, npix, ylen, nsplit, t,
                       t_epoch, d_tab, out, stride, em, em_scale);
  else
    hipLaunchKernelGGL((ff_scan_table_kernel<RJP_TAB_U, false>), grid, dim3(kBlock), shm, st,
                       (const double*)fl->d_a0, (const double*)nullptr, (const double*)fl->d_ts,
                       fl->d_ylo, fl->d_yhi, fl->ny, fl->nz, nchunks, npix, ylen, nsplit, t,
                       t_epoch, d_tab, out, stride, (double*)nullptr, em_scale);
  e = hipGetLastError();
  if (e != hipSuccess || nsplit == 1) return e;
  return ff_reduce_launch(ws, nsplit, 1, npix, 0, em_scale, sumA, em, nullptr, st);
}

}  // namespace rjp
