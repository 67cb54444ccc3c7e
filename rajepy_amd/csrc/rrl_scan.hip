// K3: LTE radio-recombination-line optical-depth cube, and its map stage.
//
// The reference evaluates scipy.special.wofz over the WHOLE 3-D grid once per channel and
// recomputes every channel-independent per-cell quantity each time
// (classes.py:1159-1214; maths/rrls.py:350-354, 383-389).  Here a 256-thread workgroup owns a
// tile of 16 z-adjacent sightlines of one x-row and up to 256 channels:
//   phase 1  the 256 threads turn a slab of 16 y x 16 z cells into per-cell line constants
//            (Doppler-shifted nu0, 1/(sigma sqrt2), Voigt y, LTE prefactor, h/kT ...) staged
//            in LDS -- each cell's constants are computed once, not once per channel;
//   phase 2  lanes run over CHANNELS: every lane reads the same cell's constants from LDS
//            (broadcast), evaluates Re w(x+iy) for its channel and accumulates tau in FP64
//            registers, one accumulator per sightline of the tile.  Cells outside the jet
//            (NaN prefactor) are skipped with a wave-uniform branch.
// Compute-bound (vector FP64): ~200 FP64 instructions per (cell, channel); HBM traffic is
// 6 fields per cell, read once per block of 256 channels.
#include "rjp_device.h"

namespace rjp {

constexpr int kRB = 256;     // threads per workgroup
constexpr int kZT = 16;      // sightlines (z) per tile
constexpr int kYC = 16;      // y-rows per LDS slab

// ---- Faddeeva: Re w(x + i y), y > 0 ----------------------------------------------------
// Core: trapezoidal rule with step h on w(z) = (i/pi) Int exp(-t^2)/(z-t) dt plus the residue
// ("pole") correction for y < pi/h (Matta & Reichel 1971; Hunter & Regan 1972).  The node
// lattice is shifted by h/2 whenever x is within h/4 of a node, so the pole term and the sum
// never cancel.  Nodes are paired (+t,-t) to halve the divisions.  With h = 0.6 and 10 pairs
// the relative error of Re w is < 1e-11 for 1e-10 <= y <= 1e3, 0 <= x <= 1e4 (measured
// against scipy.special.wofz, which the reference calls).
// Far field (|z|^2 > 64 and (x^2 > 64 or y > 1)): 6-term Laplace continued fraction,
// relative error < 3e-10 there.
constexpr double kH = 0.6;
constexpr int kNPair = 10;
__device__ __constant__ double c_node0[kNPair] = {          // exp(-(n h)^2), n = 0 halved
    0.5, 0.697676326071031, 0.23692775868212176, 0.0391638950989871,
    0.003151111598444441, 0.00012340980408667956, 2.3525752000097794e-06,
    2.1829577951254778e-08, 9.859505575991516e-11, 2.1675688826189771e-13};
__device__ __constant__ double c_node1[kNPair] = {          // exp(-((n + 1/2) h)^2)
    0.9139311852712282, 0.4448580662229412, 0.10539922456186433, 0.012155178329914935,
    0.0006823280527563778, 1.864374233151685e-05, 2.479596018045032e-07,
    1.6052280551856116e-09, 5.058252742843803e-12, 7.758402075696054e-15};

__device__ __forceinline__ double rcp_fast(double d) {
  // f32 seed + one fp64 Newton step: rel. err < 4e-14 for d within the f32 exponent range
  double r = (double)__builtin_amdgcn_rcpf((float)d);
  return r * __builtin_fma(-d, r, 2.0);
}

__device__ double voigt_rew(double ax, double y, double q) {
  const double r2 = __builtin_fma(ax, ax, y * y);
  if (r2 > 64.0 && (ax * ax > 64.0 || y > 1.0)) {
    // w = (i/sqrt(pi)) / (z - (1/2)/(z - 1/(z - (3/2)/(z - 2/(z - (5/2)/(z - 3/z))))))
    double wr = ax, wi = y;
#pragma unroll
    for (int k = 6; k >= 1; --k) {
      const double s = (0.5 * k) * rcp_fast(__builtin_fma(wr, wr, wi * wi));
      wr = __builtin_fma(-s, wr, ax);
      wi = __builtin_fma(s, wi, y);
    }
    return 0.56418958354775628695 * wi * rcp_fast(__builtin_fma(wr, wr, wi * wi));
  }
  const double u = ax * (1.0 / kH);
  const double fr = u - __builtin_floor(u);
  const bool half = !(fr >= 0.25 && fr < 0.75);
  const double dh = half ? 0.5 * kH : 0.0;
  double s = 0.0;
#pragma unroll
  for (int n = 0; n < kNPair; ++n) {
    const double t = n * kH + dh;
    const double c = half ? c_node1[n] : c_node0[n];
    const double S = __builtin_fma(t, t, r2);
    const double D = 2.0 * ax * t;
    // c * [1/((x-t)^2+y^2) + 1/((x+t)^2+y^2)] = c * 2S / (S^2 - D^2)
    s = __builtin_fma(c * (S + S), rcp_fast((S - D) * (S + D)), s);
  }
  s *= y * (kH / 3.14159265358979323846);
  if (q >= 0.0) {
    const double e = y * y - ax * ax;
    if (e > -80.0) {
      // Re[ 2 exp(-z^2) q / (q - exp(-i theta)) ], theta = 2 pi (x/h - delta)
      double st, ct, s2, c2;
      sincos(6.28318530717958647692 * (fr - (half ? 0.5 : 0.0)), &st, &ct);
      sincos(2.0 * ax * y, &s2, &c2);
      const double den = __builtin_fma(q, q - 2.0 * ct, 1.0);
      s += 2.0 * exp(e) * q * (c2 * (q - ct) - s2 * st) / den;
    }
  }
  return s;
}

template <typename T>
struct RrlFields {
  const T* nd;
  const T* xi;
  const T* temp;
  const T* pf;
  const T* ts;
  const T* vy;
};

struct LineDev {
  double nu_rest, kG, kL, kappa0, en_over_k, h_over_k;
  double path0;        // csize * au * 100 [cm]
  double nu_ref;       // reference frequency of the channel block expansion
  double dnu_max;      // max |nu_f - nu_ref| over all channels
};

// LF = lanes along the channel axis (16, 64 or 256); G = kRB / LF sightline groups.
template <typename T, int LF, bool BURSTS>
__global__ __launch_bounds__(kRB) void rrl_scan_kernel(
    RrlFields<T> f, int nx, int ny, int nz, BurstsDev b, double time_s, LineDev ln,
    const double* __restrict__ nu, int nchan, double* __restrict__ tau) {
  constexpr int G = kRB / LF;
  constexpr int NZP = kZT / G;       // sightlines per thread
  static_assert(kZT % G == 0, "tile/group mismatch");

  __shared__ double s_nu0[kRB], s_is2[kRB], s_y[kRB], s_C[kRB], s_a[kRB], s_E0[kRB],
      s_q[kRB];

  const int ntz = (nz + kZT - 1) / kZT;
  const int x = blockIdx.x / ntz;
  const int z0 = (blockIdx.x - x * ntz) * kZT;
  const int tid = threadIdx.x;
  const int fl = tid % LF;
  const int g = tid / LF;
  const int fi = blockIdx.y * LF + fl;
  const bool chan_live = fi < nchan;
  const double nu_f = chan_live ? nu[fi] : ln.nu_ref;
  const double dnu = nu_f - ln.nu_ref;

  double acc[NZP];
#pragma unroll
  for (int j = 0; j < NZP; ++j) acc[j] = 0.0;

  const int cy = tid / kZT, cz = tid % kZT;     // this thread's cell in the slab (phase 1)
  const double kPiOverH = 3.14159265358979323846 / kH;

  for (int yb = 0; yb < ny; yb += kYC) {
    // ---- phase 1: per-cell line constants --------------------------------------------
    {
      const int yy = yb + cy, zz = z0 + cz;
      double C = 0.0, nu0 = 0.0, is2 = 0.0, yv = 1.0, a = 0.0, E0 = 0.0, q = -1.0;
      if (yy < ny && zz < nz) {
        const int64_t o = ((int64_t)x * ny + yy) * nz + zz;
        const double nd = (double)f.nd[o], xi = (double)f.xi[o], Tk = (double)f.temp[o],
                     pf = (double)f.pf[o], vy = (double)f.vy[o];
        double chi = 1.0;
        if (BURSTS) chi = chi_cell(b, signbit_d(nd), time_s - (double)f.ts[o]);
        const double ne = fabs(nd) * chi * xi;
        nu0 = ln.nu_rest * (1.0 - vy * 1000.0 / 299792458.0);        // physics.py:557-558
        const double fwhm_g = ln.kG * sqrt(Tk) * nu0;                // rrls.py:116-118
        const double sigma = fwhm_g / 2.0 / 1.1774100225154747;      // / sqrt(2 ln 2)
        is2 = 1.0 / (sigma * 1.4142135623730951);
        const double fwhm_l = ln.kL * ne;                            // rrls.py:101
        yv = 0.5 * fwhm_l * is2;
        a = ln.h_over_k / Tk;
        // kappa_L * path without the profile and the stimulated-emission factor
        C = ln.kappa0 * (ne * ne / (Tk * sqrt(Tk))) * exp(ln.en_over_k / Tk) *
            (ln.path0 * pf) / (sigma * 2.5066282746310002);
        E0 = exp(-a * ln.nu_ref);
        q = (yv < kPiOverH) ? exp(-2.0 * kPiOverH * yv) : -1.0;
        if (!(C == C) || C == 0.0 || !(yv > 0.0)) C = 0.0;           // nansum drops NaN terms
      }
      s_C[tid] = C; s_nu0[tid] = nu0; s_is2[tid] = is2; s_y[tid] = yv; s_a[tid] = a;
      s_E0[tid] = E0; s_q[tid] = q;
    }
    __syncthreads();

    // ---- phase 2: lanes over channels ------------------------------------------------
    for (int r = 0; r < kYC; ++r) {
#pragma unroll
      for (int j = 0; j < NZP; ++j) {
        const int ci = r * kZT + g * NZP + j;
        double C = s_C[ci];
        bool live = C != 0.0;
        if (LF >= RJP_WAVE) live = __builtin_amdgcn_readfirstlane((int)live) != 0;
        if (live) {
          const double xv = (nu_f - s_nu0[ci]) * s_is2[ci];
          const double V = voigt_rew(fabs(xv), s_y[ci], s_q[ci]);
          // 1 - exp(-h nu / kT) = 1 - E0 * exp(-a (nu - nu_ref))          (rrls.py:387)
          const double a = s_a[ci];
          const double eps = a * dnu;
          double ex;
          if (a * ln.dnu_max < 1e-3)
            ex = __builtin_fma(eps, __builtin_fma(eps, __builtin_fma(eps, -1.0 / 6.0, 0.5), -1.0), 1.0);
          else
            ex = exp(-eps);
          const double term = C * V * (1.0 - s_E0[ci] * ex);
          if (term == term) acc[j] += term;
        }
      }
    }
    __syncthreads();
  }

  if (chan_live) {
    const int64_t base = (int64_t)fi * nx * nz + (int64_t)x * nz + z0 + g * NZP;
#pragma unroll
    for (int j = 0; j < NZP; ++j)
      if (z0 + g * NZP + j < nz) tau[base + j] = acc[j];
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
    s = cflux[fch] * bnu * exp(-tau_ff[o]) * (1.0 - exp(-tau_rrl[o]));
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
                 (const T*)fl->d_pf, (const T*)fl->d_ts, (const T*)fl->d_vy};
  const int ntz = (fl->nz + kZT - 1) / kZT;
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
  if (nchan > 64) return rrl_launch_t<T, 256>(fl, b, bursts, time_s, ln, d_nu, nchan, tau, st);
  if (nchan > 16) return rrl_launch_t<T, 64>(fl, b, bursts, time_s, ln, d_nu, nchan, tau, st);
  return rrl_launch_t<T, 16>(fl, b, bursts, time_s, ln, d_nu, nchan, tau, st);
}

hipError_t rrl_scan_launch(const rjp_fields* fl, const rjp_bursts* hb, double time_s,
                           const rjp_line* line, const double* h_nu, const double* d_nu,
                           int nchan, double* tau, hipStream_t st) {
  BurstsDev b;
  bool bursts = false;
  for (int j = 0; j < 2; ++j) {
    b.n[j] = hb ? hb->n[j] : 0;
    if (b.n[j] > 0) bursts = true;
    for (int i = 0; i < RJP_MAX_BURSTS; ++i) {
      b.t0[j][i] = hb ? hb->t0[j][i] : 0.0;
      b.amp_rel[j][i] = hb ? hb->amp_rel[j][i] : 0.0;
      b.inv2s2[j][i] = hb ? hb->inv2s2[j][i] : 0.0;
    }
  }
  if (bursts && !fl->d_ts) return hipErrorInvalidValue;
  LineDev ln;
  ln.nu_rest = line->nu_rest; ln.kG = line->kG; ln.kL = line->kL; ln.kappa0 = line->kappa0;
  ln.en_over_k = line->en_over_k; ln.h_over_k = line->h_over_k;
  ln.path0 = fl->csize_au * 149597870700.0 * 1e2;
  double lo = h_nu[0], hi = h_nu[0];
  for (int i = 1; i < nchan; ++i) { lo = h_nu[i] < lo ? h_nu[i] : lo; hi = h_nu[i] > hi ? h_nu[i] : hi; }
  ln.nu_ref = 0.5 * (lo + hi);
  ln.dnu_max = 0.5 * (hi - lo);
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
  if (ftot) {
    hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)nchan), dim3(kRB), 0, st, part,
                       (int)nblk, ftot);
    err = hipGetLastError();
  }
  return err;
}

}  // namespace rjp
