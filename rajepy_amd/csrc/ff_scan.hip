// K1 (free-free / emission-measure line-of-sight scan) and K2 (per-channel map stage).
//
// K1 replaces the three y-reductions the reference performs per channel with ~15 full-grid
// NumPy temporaries each (classes.py:1116-1120, 1375-1432, 1471-1472) by ONE streaming pass
// over the device-resident fields (three in the compact layout, five in the wide one): one
// lane owns VEC adjacent (x,z) sightlines (16 B of every field per load, 1 KiB per
// wave-instruction, lanes adjacent along the contiguous z-axis), walks y serially with UNROLL
// rows of loads in flight, evaluates the burst factor chi(t) for a tile of up to 32 epochs in
// registers and keeps FP64 accumulators.  The y-range is split over workgroups so small maps
// still fill 256 CUs; partial sums go to a workspace and a tiny second kernel reduces them in
// a fixed order (bitwise reproducible, no atomics).
// HBM-bound: algorithmic bytes per cell and epoch tile = 5 fields * sizeof(T) in the wide layout,
// 3 fields in the compact layout (rjp_fields.d_em0 = (n x)^2 pf with the jet flag in its sign
// bit, temp, ts) and 2 fields in the tau layout (rjp_fields.d_a0 = em0 T^-1.5|-1.35 -- every
// factor of a cell's optical depth that depends on neither frequency nor epoch -- and ts; a
// third, em0, only when emission-measure maps are asked for as well).  The tau layout never
// touches the temperature: T_avg = nanmean_y(T > 0) does not depend on the epoch either
// (classes.py:1471-1472) and has its own one-off pass, tavg_kernel.
#include <algorithm>
#include <cmath>
#include <type_traits>
#include <vector>


#include "ff_scan_kernels.h"

namespace rjp {
// ---- T_avg = nanmean_y(T where T > 0) (classes.py:1471-1472, 1484-1485) ----------------------
// Depends on neither frequency nor epoch: one pass over the temperature field per MODEL (the
// tau layout's scans never read T).  Same lane / y-range structure and the same summation
// order as the single-epoch ff_scan_kernel, so the map is bit-identical to the one that kernel
// derives on the wide and compact layouts.  Partials: ws[split][0|1][pixel].
template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void tavg_kernel(FieldPtrs<T> f, int ny, int nz,
                                                      int64_t nchunks, int64_t npix, int ylen,
                                                      int nsplit, double* __restrict__ ws) {
  constexpr int U = 8;
  const int split = (int)(blockIdx.x % (unsigned)nsplit);
  const int64_t c = (int64_t)(blockIdx.x / (unsigned)nsplit) * kBlock + threadIdx.x;
  const bool lane_live = c < nchunks;
  const int64_t p0 = c * VEC;
  int y0 = split * ylen;
  int y1 = min(ny, y0 + ylen);
  if (f.ylo) {
    __shared__ int s_lo, s_hi;
    if (threadIdx.x == 0) { s_lo = ny; s_hi = 0; }
    __syncthreads();
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
  double accT[VEC];
  int cnt[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) { accT[v] = 0.0; cnt[v] = 0; }
  int64_t off = (x * ny + y0) * (int64_t)nz + z;
  int y = y0;
  for (; y + U <= y1; y += U) {
    double tp[U][VEC];
#pragma unroll
    for (int u = 0; u < U; ++u) load_vec(f.temp + off + (int64_t)u * nz, tp[u]);
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        accT[v] += __builtin_fmax(tp[u][v], 0.0);
        cnt[v] += tp[u][v] > 0.0 ? 1 : 0;
      }
    off += (int64_t)U * nz;
  }
  for (; y < y1; ++y) {
    double tp[VEC];
    load_vec(f.temp + off, tp);
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      accT[v] += __builtin_fmax(tp[v], 0.0);
      cnt[v] += tp[v] > 0.0 ? 1 : 0;
    }
    off += nz;
  }
  double* w = ws + (int64_t)split * 2 * npix + p0;
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    w[v] = accT[v];
    w[npix + v] = (double)cnt[v];
  }
}

__global__ __launch_bounds__(kBlock) void tavg_reduce_kernel(const double* __restrict__ ws,
                                                             int nsplit, int64_t npix,
                                                             double* __restrict__ tavg) {
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= npix) return;
  double t = 0.0, n = 0.0;
  for (int s = 0; s < nsplit; ++s) {
    t += ws[((int64_t)s * 2) * npix + p];
    n += ws[((int64_t)s * 2 + 1) * npix + p];
  }
  tavg[p] = t / n;              // 0/0 = NaN on empty sightlines = nanmean of all-NaN
}

// Fixed-order reduction over the y-splits; writes the base maps of epochs [e0, e0+et).
__global__ __launch_bounds__(kBlock) void ff_reduce_kernel(
    const double* __restrict__ ws, int nsplit, int et, int64_t npix, int e0, double em_scale,
    double* __restrict__ sumA, double* __restrict__ em, double* __restrict__ tavg) {
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= npix) return;
  const int na = nacc(et);
  for (int e = 0; e < et; ++e) {
    double a = 0.0, g = 0.0;
    for (int s = 0; s < nsplit; ++s) {
      a += ws[((int64_t)s * na + e) * npix + p];
      if (em) g += ws[((int64_t)s * na + et + e) * npix + p];     // not written without EM
    }
    sumA[(int64_t)(e0 + e) * npix + p] = a;
    if (em) em[(int64_t)(e0 + e) * npix + p] = g * em_scale;
  }
  if (tavg && e0 == 0) {
    double t = 0.0, n = 0.0;
    for (int s = 0; s < nsplit; ++s) {
      t += ws[((int64_t)s * na + 2 * et) * npix + p];
      n += ws[((int64_t)s * na + 2 * et + 1) * npix + p];
    }
    tavg[p] = t / n;            // 0/0 = NaN on empty sightlines = nanmean of all-NaN
  }
}

hipError_t ff_reduce_launch(const double* ws, int nsplit, int et, int64_t npix, int e0,
                            double em_scale, double* sumA, double* em, double* tavg,
                            hipStream_t st) {
  hipLaunchKernelGGL(ff_reduce_kernel, dim3((unsigned)((npix + kBlock - 1) / kBlock)), dim3(kBlock),
                     0, st, ws, nsplit, et, npix, e0, em_scale, sumA, em, tavg);
  return hipGetLastError();
}

// ---- K2 ------------------------------------------------------------------------------
// One thread per VEC adjacent pixels, serial over the channels of its slice: every store
// instruction is a coalesced row segment of one (epoch, channel) map, 16 B per lane when the
// map has an even number of pixels.  Write-bound: 16 B per voxel-channel.  The per-channel
// totals are reduced per wave (shuffles, no LDS, no barrier) into `part`, one slot per wave.
template <int VEC>
__global__ __launch_bounds__(kBlock) void ff_maps_kernel(
    const double* __restrict__ sumA, const double* __restrict__ tavg, int64_t npix,
    const double* __restrict__ ctau, const double* __restrict__ cflux, int nchan, int fchunk,
    double* __restrict__ tau, double* __restrict__ flux, double* __restrict__ part,
    int nparts) {
  const int64_t p = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * VEC;
  const int e = blockIdx.y;
  const int f0 = blockIdx.z * fchunk;
  const int f1 = min(nchan, f0 + fchunk);
  const bool live = p < npix;                    // npix % VEC == 0: a lane is live or not at all
  double A[VEC], ta[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) { A[v] = 0.0; ta[v] = 0.0; }
  if (live) {
    load_plain(sumA + (int64_t)e * npix + p, A);
    load_plain(tavg + p, ta);
  }
  const int slot = blockIdx.x * (kBlock / RJP_WAVE) + threadIdx.x / RJP_WAVE;
  for (int f = f0; f < f1; ++f) {
    double t[VEC], s[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      t[v] = ctau[f] * A[v];
      s[v] = cflux[f] * (ta[v] * one_minus_exp_neg(t[v]));
    }
    const int64_t o = ((int64_t)e * nchan + f) * npix + p;
    if (live) {
      if (tau) store_cube(tau + o, t);
      if (flux) store_cube(flux + o, s);
    }
    if (part) {
      double acc = 0.0;
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc += (live && s[v] == s[v]) ? s[v] : 0.0;      // nansum
#pragma unroll
      for (int d = RJP_WAVE / 2; d > 0; d >>= 1) acc += __shfl_down(acc, d, RJP_WAVE);
      if ((threadIdx.x & (RJP_WAVE - 1)) == 0)
        part[((int64_t)e * nchan + f) * nparts + slot] = acc;
    }
  }
}

// The cube kernel for maps large enough to fill the chip with FEWER, fatter threads (round 4): a
// lane walks `kp` pixel groups for kMapsFC channels, stores as above, and keeps one flux
// accumulator per channel in registers -- the six-step wave reduction (a third of the kernel's
// issue slots when it came once per channel and pixel pair: 0.11 of cfg4's 0.24 ms) now comes
// once per channel and LANE, as in ff_ftot_kernel below.  Same arithmetic per (pixel, channel);
// the totals are summed in another (fixed) order.
constexpr int kMapsFC = 16;      // channels per workgroup (gridDim.z slices the channel axis)
template <int VEC, bool FT, int KP>
__global__ __launch_bounds__(kBlock) void ff_maps_acc_kernel(
    const double* __restrict__ sumA, const double* __restrict__ tavg, int64_t npix,
    const double* __restrict__ ctau, const double* __restrict__ cflux, int nchan,
    double* __restrict__ tau, double* __restrict__ flux, double* __restrict__ part, int nparts) {
  const int e = blockIdx.y;
  const int f0 = blockIdx.z * kMapsFC;
  const int nf = min(nchan - f0, kMapsFC);
  double acc[kMapsFC];
#pragma unroll
  for (int j = 0; j < kMapsFC; ++j) acc[j] = 0.0;
  // the base maps of all KP pixel groups first (their loads are in flight together), then the
  // stores of group after group
  double A[KP][VEC], ta[KP][VEC];
  int64_t pk[KP];
#pragma unroll
  for (int k = 0; k < KP; ++k) {
    pk[k] = (((int64_t)blockIdx.x * KP + k) * kBlock + threadIdx.x) * VEC;
    const int64_t pc = pk[k] < npix ? pk[k] : 0;               // npix % VEC == 0
    load_plain(sumA + (int64_t)e * npix + pc, A[k]);
    load_plain(tavg + pc, ta[k]);
  }
#pragma unroll
  for (int k = 0; k < KP; ++k) {
    if (pk[k] >= npix) continue;
#pragma unroll
    for (int j = 0; j < kMapsFC; ++j) {
      if (j < nf) {
        const double ct = ctau[f0 + j], cf = cflux[f0 + j];
        double t[VEC], s[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          t[v] = ct * A[k][v];
          s[v] = cf * (ta[k][v] * one_minus_exp_neg(t[v]));
          if (FT) acc[j] += s[v] == s[v] ? s[v] : 0.0;                 // nansum
        }
        const int64_t o = ((int64_t)e * nchan + f0 + j) * npix + pk[k];
        if (tau) store_cube(tau + o, t);
        if (flux) store_cube(flux + o, s);
      }
    }
  }
  if (FT) {
    const int slot = blockIdx.x * (kBlock / RJP_WAVE) + threadIdx.x / RJP_WAVE;
#pragma unroll
    for (int j = 0; j < kMapsFC; ++j) {
      if (j < nf) {
        double a = acc[j];
#pragma unroll
        for (int d = RJP_WAVE / 2; d > 0; d >>= 1) a += __shfl_down(a, d, RJP_WAVE);
        if ((threadIdx.x & (RJP_WAVE - 1)) == 0)
          part[((int64_t)e * nchan + f0 + j) * nparts + slot] = a;
      }
    }
  }
}

// Light curves (no cube is written): the per-channel totals alone.  The map kernel above pays a
// six-step wave reduction per channel and pixel pair -- a quarter of its instructions once
// there is nothing to store; here a lane walks kFtotKP pixel groups for up to kFtotFC channels
// with one accumulator per channel in registers, and the wave reduction comes once per channel
// and lane.  Same arithmetic per (pixel, channel); sums in a fixed order.
constexpr int kFtotKP = 4;       // pixel groups per lane
constexpr int kFtotFC = 16;      // channels per workgroup (gridDim.z slices the channel axis)
template <int VEC>
__global__ __launch_bounds__(kBlock) void ff_ftot_kernel(
    const double* __restrict__ sumA, const double* __restrict__ tavg, int64_t npix,
    const double* __restrict__ ctau, const double* __restrict__ cflux, int nchan,
    double* __restrict__ part, int nparts) {
  const int e = blockIdx.y;
  const int f0 = blockIdx.z * kFtotFC;
  const int nf = min(nchan - f0, kFtotFC);
  double acc[kFtotFC];
#pragma unroll
  for (int j = 0; j < kFtotFC; ++j) acc[j] = 0.0;
#pragma unroll 1
  for (int k = 0; k < kFtotKP; ++k) {
    const int64_t p = (((int64_t)blockIdx.x * kFtotKP + k) * kBlock + threadIdx.x) * VEC;
    if (p >= npix) continue;                       // npix % VEC == 0
    double A[VEC], ta[VEC];
    load_plain(sumA + (int64_t)e * npix + p, A);
    load_plain(tavg + p, ta);
    // nansum, hoisted: a flux is NaN exactly where T_avg is (an empty sightline; the sums A and
    // 1 - e^-tau are never NaN) -- such a pixel adds zero to every channel, so its T_avg is
    // zeroed ONCE and the channel loop is one multiply and one FMA around 1 - e^-tau
#pragma unroll
    for (int v = 0; v < VEC; ++v) ta[v] = ta[v] == ta[v] ? ta[v] : 0.0;
#pragma unroll
    for (int j = 0; j < kFtotFC; ++j) {
      if (j < nf) {
        const double ct = ctau[f0 + j], cf = cflux[f0 + j];
#pragma unroll
        for (int v = 0; v < VEC; ++v)
          acc[j] = __builtin_fma(cf * ta[v], one_minus_exp_neg(ct * A[v]), acc[j]);
      }
    }
  }
  const int slot = blockIdx.x * (kBlock / RJP_WAVE) + threadIdx.x / RJP_WAVE;
#pragma unroll
  for (int j = 0; j < kFtotFC; ++j) {
    if (j < nf) {
      double a = acc[j];
#pragma unroll
      for (int d = RJP_WAVE / 2; d > 0; d >>= 1) a += __shfl_down(a, d, RJP_WAVE);
      if ((threadIdx.x & (RJP_WAVE - 1)) == 0)
        part[((int64_t)e * nchan + f0 + j) * nparts + slot] = a;
    }
  }
}

// sums the per-block partials of one (epoch, channel) in a fixed order
__global__ __launch_bounds__(kBlock) void sum_partials_kernel(const double* __restrict__ part,
                                                              int nblk,
                                                              double* __restrict__ out) {
  const int64_t row = blockIdx.x;
  double v = 0.0;
  for (int i = threadIdx.x; i < nblk; i += kBlock) v += part[row * nblk + i];
  __shared__ double red[kBlock / RJP_WAVE];
#pragma unroll
  for (int d = RJP_WAVE / 2; d > 0; d >>= 1) v += __shfl_down(v, d, RJP_WAVE);
  if ((threadIdx.x & (RJP_WAVE - 1)) == 0) red[threadIdx.x / RJP_WAVE] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0.0;
    for (int w = 0; w < kBlock / RJP_WAVE; ++w) tot += red[w];
    out[row] = tot;
  }
}

// Occupied y-range per sightline: one lane per sightline, lanes adjacent along z.
template <typename T, bool CMP>
__global__ __launch_bounds__(kBlock) void y_bounds_kernel(FieldPtrs<T> f, int ny, int nz,
                                                          int64_t npix, int32_t* __restrict__ ylo,
                                                          int32_t* __restrict__ yhi) {
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= npix) return;
  const int64_t x = p / nz;
  const int z = (int)(p - x * nz);
  int64_t off = x * ny * (int64_t)nz + z;
  int lo = ny, hi = 0;
  for (int y = 0; y < ny; ++y, off += nz) {
    const double tk = (double)f.temp[off];
    bool dense;                                   // n, x and ff/areas all non-NaN
    if constexpr (CMP) {
      const double g = (double)f.em0[off];
      dense = g == g;
    } else {
      const double nd = (double)f.nd[off], xi = (double)f.xi[off], pf = (double)f.pf[off];
      dense = nd == nd && xi == xi && pf == pf;
    }
    const bool matters = (tk > 0.0) || dense;
    if (matters) { lo = min(lo, y); hi = y + 1; }
  }
  ylo[p] = lo;
  yhi[p] = hi;
}

// sum_p max(0, yhi[p] - ylo[p]) -> *count (zeroed by the caller): rjp_fields.occupied_cells
__global__ __launch_bounds__(kBlock) void occupied_kernel(const int32_t* __restrict__ ylo,
                                                          const int32_t* __restrict__ yhi,
                                                          int64_t npix,
                                                          unsigned long long* __restrict__ count) {
  long long v = 0;
  for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < npix;
       p += (int64_t)gridDim.x * kBlock)
    v += max(0, yhi[p] - ylo[p]);
#pragma unroll
  for (int d = RJP_WAVE / 2; d > 0; d >>= 1) v += __shfl_down(v, d, RJP_WAVE);
  if ((threadIdx.x & (RJP_WAVE - 1)) == 0 && v != 0) atomicAdd(count, (unsigned long long)v);
}

hipError_t occupied_launch(const int32_t* ylo, const int32_t* yhi, int64_t npix,
                           unsigned long long* d_count, hipStream_t st) {
  const unsigned blocks = (unsigned)std::min<int64_t>((npix + kBlock - 1) / kBlock, 1024);
  hipLaunchKernelGGL(occupied_kernel, dim3(blocks), dim3(kBlock), 0, st, ylo, yhi, npix, d_count);
  return hipGetLastError();
}

hipError_t y_bounds_launch(const rjp_fields* fl, int32_t* ylo, int32_t* yhi, hipStream_t st) {
  const int64_t npix = (int64_t)fl->nx * fl->nz;
  const dim3 grid((unsigned)((npix + kBlock - 1) / kBlock)), blk(kBlock);
  if (fl->dtype == RJP_F64) {
    FieldPtrs<double> f{(const double*)fl->d_nd, (const double*)fl->d_xi,
                        (const double*)fl->d_temp, (const double*)fl->d_pf, nullptr, nullptr,
                        nullptr, (const double*)fl->d_em0, nullptr};
    if (fl->d_em0)
      hipLaunchKernelGGL((y_bounds_kernel<double, true>), grid, blk, 0, st, f, fl->ny, fl->nz, npix, ylo, yhi);
    else
      hipLaunchKernelGGL((y_bounds_kernel<double, false>), grid, blk, 0, st, f, fl->ny, fl->nz, npix, ylo, yhi);
  } else {
    FieldPtrs<float> f{(const float*)fl->d_nd, (const float*)fl->d_xi, (const float*)fl->d_temp,
                       (const float*)fl->d_pf, nullptr, nullptr, nullptr, (const float*)fl->d_em0,
                       nullptr};
    if (fl->d_em0)
      hipLaunchKernelGGL((y_bounds_kernel<float, true>), grid, blk, 0, st, f, fl->ny, fl->nz, npix, ylo, yhi);
    else
      hipLaunchKernelGGL((y_bounds_kernel<float, false>), grid, blk, 0, st, f, fl->ny, fl->nz, npix, ylo, yhi);
  }
  return hipGetLastError();
}

// collapse=False: the 3-D per-cell free-free optical depths (classes.py:1382-1383, 1395-1397).
// out[f * ncell + cell] = ctau[f] * T^-1.5|-1.35 * (n chi x)^2 * pf, NaN outside the jet.
template <typename T, int MODE, bool BURSTS>
__global__ __launch_bounds__(kBlock) void ff_cells_kernel(FieldPtrs<T> f, int64_t ncell,
                                                          BurstsDev b, double time_s,
                                                          const double* __restrict__ ctau,
                                                          int nchan, double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= ncell) return;
  const double nd = (double)f.nd[i], Tk = (double)f.temp[i];
  double tpow = pow_m1p5(Tk);
  if (MODE == RJP_GFF_POWERLAW) tpow *= pow(Tk, 0.15);
  double chi = 1.0;
  if (BURSTS) chi = chi_cell(b, signbit_d(nd), time_s - (double)f.ts[i]);
  const double ne = fabs(nd) * chi * (double)f.xi[i];
  const double a = ne * ne * (double)f.pf[i] * tpow;
  for (int k = 0; k < nchan; ++k) out[(int64_t)k * ncell + i] = ctau[k] * a;
}

hipError_t ff_cells_launch(const rjp_fields* fl, const rjp_bursts* hb, const double* d_ext,
                           double time_s, int mode, const double* d_ctau, int nchan,
                           double* out, hipStream_t st) {
  BurstsDev b;
  const bool bursts = bursts_to_dev(hb, b, d_ext);
  if (bursts && !fl->d_ts) return hipErrorInvalidValue;
  const int64_t n = (int64_t)fl->nx * fl->ny * fl->nz;
  const dim3 grid((unsigned)((n + kBlock - 1) / kBlock)), blk(kBlock);
  auto go = [&](auto tag) {
    using T = decltype(tag);
    FieldPtrs<T> f{(const T*)fl->d_nd, (const T*)fl->d_xi, (const T*)fl->d_temp,
                   (const T*)fl->d_pf, (const T*)fl->d_ts, nullptr, nullptr, nullptr, nullptr};
    if (mode == RJP_GFF_SCALAR) {
      if (bursts) hipLaunchKernelGGL((ff_cells_kernel<T, 0, true>), grid, blk, 0, st, f, n, b, time_s, d_ctau, nchan, out);
      else hipLaunchKernelGGL((ff_cells_kernel<T, 0, false>), grid, blk, 0, st, f, n, b, time_s, d_ctau, nchan, out);
    } else {
      if (bursts) hipLaunchKernelGGL((ff_cells_kernel<T, 1, true>), grid, blk, 0, st, f, n, b, time_s, d_ctau, nchan, out);
      else hipLaunchKernelGGL((ff_cells_kernel<T, 1, false>), grid, blk, 0, st, f, n, b, time_s, d_ctau, nchan, out);
    }
  };
  if (fl->dtype == RJP_F64) go(double{}); else go(float{});
  return hipGetLastError();
}

// ---- launch helpers ---------------------------------------------------------------------
// y-splits for a map of `nchunks` lanes: enough workgroups for ~16 waves per SIMD chip-wide
// (measured optimum on cfg4: 8 splits, 6.2 TB/s vs 5.8 at 3 and 5.9 at 32), at least 16 rows
// per split, and at least 128 rows once every CU already has a wave without splitting (cfg2,
// 256x1024x256: 8 splits of 128 rows 0.29 ms, 16 of 64 rows 0.31-0.32 ms,
// profiles/r02_k1_tuning_ab.md).
// `et` = epochs of the tile: the uniform-epoch tiles of >= 16 epochs are ALU-bound and run
// for ~0.5 ms per epoch and block row range, 3-4 workgroups per CU at a time; with 8 y-ranges
// (32 workgroups per CU) the last wave of workgroups leaves the chip 1/4 empty for a tenth of
// the pass.  They get 4x the y-ranges (cfg5's 32-epoch tile 16.6 -> 16.0 ms including the
// longer reduction of the partial sums; the HBM-bound single-epoch scan is fastest at 8),
// fewer on maps so large that the partial sums would pass 6 GiB.
static int ysplit_rule(int64_t nchunks, int ny, int et = 1, int64_t npix = 0) {
  const int64_t waves = (nchunks + RJP_WAVE - 1) / RJP_WAVE;
  const int64_t smax = std::max(1, waves >= 256 ? ny / 128 : ny / 16);
  auto rule = [&](int64_t target) {
    int64_t s = (target + waves - 1) / waves;
    if (s > smax) s = smax;
    return s < 1 ? (int64_t)1 : s;
  };
  int64_t s = rule(256 * 64);
  if (et >= 16) {
    // finer y-ranges while their partial sums stay below 6 GiB (4.4 GiB at cfg5's size)
    int64_t fine = rule(256 * 512);
    const int64_t per_range = (int64_t)nacc(et) * std::max<int64_t>(npix, nchunks) * 8;
    while (fine > s && fine * per_range > ((int64_t)6 << 30)) fine = (fine + 1) / 2;
    s = std::max(s, fine);
  }
  return (int)s;
}

// Experiment switches (A/B runs of launch parameters) exist only in builds made with
// -DRJP_DEBUG_SWITCHES (build.sh --debug-switches); the shipped library reads no environment.
static const char* debug_env(const char* name) {
#ifdef RJP_DEBUG_SWITCHES
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

static int forced_ysplit() {
  // (read on every call: a probe may change it between scans of one process, so that the
  // same allocation of the fields is timed under every setting)
  const char* s = debug_env("RJP_YSPLIT");
  return s ? atoi(s) : 0;
}

static int choose_ysplit(int64_t nchunks, int ny, int et = 1, int64_t npix = 0) {
  if (forced_ysplit() > 0) return std::min(forced_ysplit(), ny);
  return ysplit_rule(nchunks, ny, et, npix);
}

int ff_scan_vec(const rjp_fields* fl) {
  const int full = fl->dtype == RJP_F64 ? 2 : 4;
  const size_t esz = (size_t)fl->dtype;
  bool ok = (fl->nz % full) == 0;
  const void* ptrs[7] = {fl->d_nd, fl->d_xi, fl->d_temp, fl->d_pf, fl->d_ts, fl->d_em0, fl->d_a0};
  for (const void* p : ptrs)
    if (p && ((uintptr_t)p % 16) != 0) ok = false;
  (void)esz;
  static int force1 = -1;
  if (force1 < 0) force1 = debug_env("RJP_FORCE_VEC1") ? 1 : 0;
  return (ok && !force1) ? full : 1;
}

size_t ff_scan_workspace_bytes(int nx, int ny, int nz, int n_epochs) {
  const int64_t npix = (int64_t)nx * nz;
  const int etmax = n_epochs < kMaxTile ? (n_epochs < 1 ? 1 : n_epochs) : kMaxTile;
  // worst case over the tile sizes and the lane widths the launcher may pick (1, 2 or 4
  // sightlines per lane; one for tiles of >= 16 epochs)
  int64_t need = 1;
  for (int et : {1, 2, 4, 8, 16, 32}) {
    if (et > etmax && et != 1) continue;
    int64_t s = 1;
    for (int vec : {1, 2, 4})
      s = std::max<int64_t>(s, ysplit_rule(std::max<int64_t>(1, npix / vec), ny, et, npix));
    if (forced_ysplit() > 0) s = std::max<int64_t>(s, std::min(forced_ysplit(), ny));
    need = std::max<int64_t>(need, s * nacc(et));
  }
  // (an n_epochs that is no tile size itself is cut into tiles no larger than it)
  size_t bytes = (size_t)need * npix * sizeof(double) + 256;
  // the single-epoch table scan has its own y-range rule and keeps its table here (ff_scan_tab.hip)
  bytes = std::max(bytes, chi_table_workspace_bytes(npix, ny));
  // sweeps long enough for the moment path (ff_moments.hip) keep its transposed moments here
  // (10 KiB per sightline; maps so large that this passes 6 GiB stay on the epoch tiles)
  if (n_epochs >= RJP_MOM_MIN_EPOCHS && moments_workspace_bytes(npix) <= ((size_t)6 << 30))
    bytes = std::max(bytes, moments_scan_workspace_bytes(npix));
  return bytes;
}

// Decide whether a tile of epochs may use the uniform-spacing recurrence and fill its
// per-burst constants: un.q for the bursts that travel by value, `qext` (2 * next doubles,
// may be nullptr when the model has no overflow bursts) for the rest.
static void uniform_tile(const double* t, int et, const rjp_bursts* hb, UnifDev& un,
                         double* qext, std::vector<double>* atab = nullptr) {
  un.on = 0;
  un.nbt = 0;
  un.dt = 0.0;
  un.hdt = 0.0;
  un.qext = nullptr;
  un.atab = nullptr;
  if (atab) atab->clear();
  const int next = bursts_overflow(hb);
  for (int j = 0; j < 2; ++j) {
    for (int i = 0; i < RJP_SGPR_BURSTS; ++i) { un.q[j][i] = 1.0; un.a1[j][i] = 0.0; }
    for (int i = 0; qext && i < next; ++i) qext[j * next + i] = 1.0;
  }
  static int disabled = -1;
  if (disabled < 0) disabled = debug_env("RJP_NO_UNIFORM") ? 1 : 0;
  if (et < 4 || disabled || !hb) return;
  if (next > 0 && !qext) return;
  const double dt = (t[et - 1] - t[0]) / (et - 1);
  double tmax = 0.0, dev = 0.0;
  for (int e = 0; e < et; ++e) {
    tmax = std::max(tmax, std::fabs(t[e]));
    dev = std::max(dev, std::fabs(t[e] - (t[0] + e * dt)));
  }
  if (!(dev <= 8.0 * 2.220446049250313e-16 * tmax)) return;     // not (numerically) uniform
  const int m = et / 2;
  const double half_span = std::max(m, et - 1 - m) * std::fabs(dt);
  for (int j = 0; j < 2; ++j)
    for (int i = 0; i < hb->n[j]; ++i) {
      const double inv = hb->inv2s2[j][i];
      if (!(inv > 0.0)) return;
      const double sigma = std::sqrt(0.5 / inv);
      if (!(half_span <= 28.0 * sigma)) return;   // anchor underflow would hide live epochs
      const double q = std::exp(-2.0 * inv * dt * dt);
      if (i < RJP_SGPR_BURSTS) {
        un.q[j][i] = q;
        un.a1[j][i] = 2.0 * (-inv * RJP_LOG2E) * dt;
      } else qext[j * next + (i - RJP_SGPR_BURSTS)] = q;
    }
  un.on = 1;
  un.dt = dt;
  un.hdt = 0.5 * dt;
  un.nbt = std::max(hb->n[0], hb->n[1]);
  if (atab) {
    // step table of the two-operation recurrence: q^(k (k + 1) / 2) = exp(-inv dt^2 k (k + 1));
    // unused slots (the shorter jet) hold 1, their amplitude is 0
    atab->assign((size_t)2 * un.nbt * RJP_STEP_TAB, 1.0);
    for (int j = 0; j < 2; ++j)
      for (int i = 0; i < hb->n[j]; ++i)
        for (int k = 1; k <= RJP_STEP_TAB; ++k)
          (*atab)[((size_t)j * un.nbt + i) * RJP_STEP_TAB + k - 1] =
              std::exp(-hb->inv2s2[j][i] * dt * dt * (double)(k * (k + 1)));
  }
}

// LDS-DMA row prefetch of the long tiles: 16-byte pairs of f64 cells
bool tile_dma_ok(const rjp_fields* fl) {
  static int off = -1;
  if (off < 0) off = debug_env("RJP_NO_TILE_DMA") ? 1 : 0;
  if (off || fl->dtype != RJP_F64 || (fl->nz % 2) != 0) return false;
  const void* ptrs[7] = {fl->d_nd, fl->d_xi, fl->d_temp, fl->d_pf, fl->d_ts, fl->d_em0, fl->d_a0};
  for (const void* q : ptrs)
    if (q && ((uintptr_t)q % 16) != 0) return false;
  return (int64_t)fl->nx * fl->nz >= 2;
}

// The layout a scan of `fl` in Gaunt mode `mode` uses: the tau layout when its field was built
// for that mode (f64 storage; with EM maps it also needs em0), else compact, else wide.
int scan_layout(const rjp_fields* fl, int mode, bool want_em) {
  if (fl->d_a0 && fl->a0_mode == mode && fl->dtype == RJP_F64 && (!want_em || fl->d_em0))
    return LAY_TAU;
  return fl->d_em0 ? LAY_CMP : LAY_WIDE;
}

// One tile of a scan: the kernels live in the ff_scan_inst_*.hip translation units, one per
// (storage, layout[, Gaunt mode]) slice.
static hipError_t dispatch_tile(int vec, const rjp_fields* fl, const BurstsDev& b, bool bursts,
                                int mode, const double* t, const UnifDev& un, int et, int nsplit,
                                int ylen, double* ws, bool want_em, hipStream_t st) {
  const int lay = scan_layout(fl, mode, want_em);
  if (fl->dtype == RJP_F32) return scan_f32(vec, lay, fl, b, bursts, mode, t, un, et, nsplit, ylen, ws, want_em, st);
  // tau layout: the Gaunt mode is baked into the field, one kernel serves both
  if (lay == LAY_TAU) return scan_f64_tau(vec, fl, b, bursts, t, un, et, nsplit, ylen, ws, want_em, st);
  if (lay == LAY_CMP)
    return mode == RJP_GFF_SCALAR
               ? scan_f64_cmp_scalar(vec, fl, b, bursts, t, un, et, nsplit, ylen, ws, want_em, st)
               : scan_f64_cmp_plaw(vec, fl, b, bursts, t, un, et, nsplit, ylen, ws, want_em, st);
  return scan_f64_wide(vec, fl, b, bursts, mode, t, un, et, nsplit, ylen, ws, want_em, st);
}

static bool tile32_em() {
  static int v = -1;
  if (v < 0) v = debug_env("RJP_NO_TILE32_EM") ? 0 : 1;
  return v != 0;
}

static bool use_tile32() {
  static int v = -1;
  if (v < 0) v = debug_env("RJP_NO_TILE32") ? 0 : 1;
  return v != 0;
}

void ff_scan_plan(const rjp_fields* fl, const rjp_bursts* hb, const double* epochs,
                  int n_epochs, int mode, bool want_em, ScanPlan& pl) {
  // tiles of >= 16 epochs (the uniform-epoch recurrences) exist for f64 storage on the tau and
  // compact layouts -- what every BASELINE configuration runs on; f32 storage and the wide
  // layout take tiles of <= 8 epochs (their long-tile kernels were a third of the library's
  // build time and size for no configuration that uses them)
  const bool long_tiles = fl->dtype == RJP_F64 && scan_layout(fl, mode, want_em) != LAY_WIDE;
  BurstsDev probe_b;
  pl.bursts = bursts_to_dev(hb, probe_b);
  pl.next = bursts_overflow(hb);
  const int64_t npix = (int64_t)fl->nx * fl->nz;
  pl.vec = ff_scan_vec(fl);
  pl.tiles.clear();
  pl.ext.assign(bursts_ext_doubles(hb), 0.0);
  if (pl.next > 0) bursts_fill_ext(hb, pl.ext.data());
  std::vector<double> q((size_t)2 * pl.next + 1), atab;
  int e0 = 0;
  while (e0 < n_epochs) {
    ScanTile tl;
    tl.e0 = e0;
    tl.et = 1;
    uniform_tile(epochs + e0, 1, hb, tl.un, q.data());       // un.on = 0, q = 1
    if (pl.bursts) {
      const int left = n_epochs - e0;
      tl.et = (left >= 8 && pl.vec != 4) ? 8 : left >= 4 ? 4 : left >= 2 ? 2 : 1;
      UnifDev probe;
      if (left >= 16 && long_tiles) {
        uniform_tile(epochs + e0, 16, hb, probe, q.data());
        if (probe.on) tl.et = 16;
      }
      if (left >= 32 && long_tiles && (!want_em || tile32_em()) && use_tile32()) {
        uniform_tile(epochs + e0, 32, hb, probe, q.data());
        if (probe.on) tl.et = 32;
      }
      // the tile's own constants (short tiles of f32 storage keep their float-accuracy exp:
      // launch_tile ignores `un` there)
      uniform_tile(epochs + e0, tl.et, hb, tl.un, q.data(), &atab);
    }
    // tiles of >= 16 epochs run one sightline per lane
    tl.nsplit = choose_ysplit(tl.et >= 16 ? npix : npix / pl.vec, fl->ny, tl.et, npix);
    tl.ylen = (fl->ny + tl.nsplit - 1) / tl.nsplit;
    tl.q_off = pl.ext.size();
    if (pl.next > 0) pl.ext.insert(pl.ext.end(), q.begin(), q.begin() + 2 * pl.next);
    tl.a_off = pl.ext.size();
    pl.ext.insert(pl.ext.end(), atab.begin(), atab.end());
    pl.tiles.push_back(tl);
    if (!pl.bursts) break;
    e0 += tl.et;
  }
}

// T_avg map of the model (one pass over the temperature field).  `ws`: 2 * nsplit * npix doubles
// (never more than the scan's own workspace: the same y-ranges, 2 planes instead of 4).
hipError_t tavg_launch(const rjp_fields* fl, double* tavg, double* ws, hipStream_t st) {
  const int64_t npix = (int64_t)fl->nx * fl->nz;
  const int vec = ff_scan_vec(fl);
  const int64_t nchunks = npix / vec;
  const int nsplit = choose_ysplit(nchunks, fl->ny, 1, npix);
  const int ylen = (fl->ny + nsplit - 1) / nsplit;
  dim3 grid((unsigned)(((nchunks + kBlock - 1) / kBlock) * nsplit), 1u);
  auto go = [&](auto tag, auto vtag) {
    using T = decltype(tag);
    constexpr int VEC = decltype(vtag)::value;
    FieldPtrs<T> f{nullptr, nullptr, (const T*)fl->d_temp, nullptr, nullptr, fl->d_ylo,
                   fl->d_yhi, nullptr, nullptr};
    hipLaunchKernelGGL((tavg_kernel<T, VEC>), grid, dim3(kBlock), 0, st, f, fl->ny, fl->nz,
                       nchunks, npix, ylen, nsplit, ws);
  };
  if (fl->dtype == RJP_F64) {
    if (vec == 2) go(double{}, std::integral_constant<int, 2>{});
    else go(double{}, std::integral_constant<int, 1>{});
  } else {
    if (vec == 4) go(float{}, std::integral_constant<int, 4>{});
    else go(float{}, std::integral_constant<int, 1>{});
  }
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return err;
  hipLaunchKernelGGL(tavg_reduce_kernel, dim3((unsigned)((npix + kBlock - 1) / kBlock)),
                     dim3(kBlock), 0, st, ws, nsplit, npix, tavg);
  return hipGetLastError();
}

// Enqueue the whole scan for n_epochs epochs.  `d_ext` = device copy of pl.ext (nullptr when
// it is empty).  Returns hipSuccess or the first error.
hipError_t ff_scan_run(const rjp_fields* fl, const rjp_bursts* hb, const ScanPlan& pl,
                       const double* d_ext, const double* epochs, int n_epochs, int mode,
                       double* sumA, double* em, double* tavg, double* ws, hipStream_t st) {
  BurstsDev b;
  const bool bursts = bursts_to_dev(hb, b, d_ext);
  if (bursts && !fl->d_ts) return hipErrorInvalidValue;
  if (!pl.ext.empty() && !d_ext) return hipErrorInvalidValue;
  const int64_t npix = (int64_t)fl->nx * fl->nz;
  const int vec = pl.vec;
  // em = sum (n x)^2 * csize*au/pc * pf  (classes.py:1116-1118)
  const double em_scale = fl->csize_au * 149597870700.0 / 3.085677581491367e+16;
  const unsigned rblocks = (unsigned)((npix + kBlock - 1) / kBlock);
  if (scan_layout(fl, mode, em != nullptr) == LAY_TAU) {
    // the tau layout's scans keep no temperature sums: a caller that asks for T_avg with the
    // scan (instead of once per model through rjp_tavg) gets the separate pass here
    if (tavg) {
      if (!fl->d_temp) return hipErrorInvalidValue;
      hipError_t err = tavg_launch(fl, tavg, ws, st);
      if (err != hipSuccess) return err;
      tavg = nullptr;
    }
  }

  for (size_t k = 0; k < pl.tiles.size(); ++k) {
    const int e0 = pl.tiles[k].e0, et = pl.tiles[k].et;
    const int nsplit = pl.tiles[k].nsplit, ylen = pl.tiles[k].ylen;
    UnifDev un = pl.tiles[k].un;
    un.qext = pl.next > 0 ? d_ext + pl.tiles[k].q_off : nullptr;
    un.atab = un.on ? d_ext + pl.tiles[k].a_off : nullptr;
    hipError_t err;
    const double* t = epochs + e0;
    // 16-epoch tiles are ALU-bound and register-hungry: one sightline per lane (160 VGPRs,
    // 3 waves/SIMD) beats two (256 VGPRs, 1 wave/SIMD) by 15 %
    err = dispatch_tile(et < 16 ? vec : 1, fl, b, bursts, mode, t, un, et, nsplit, ylen, ws,
                        em != nullptr, st);
    if (err != hipSuccess) return err;
    hipLaunchKernelGGL(ff_reduce_kernel, dim3(rblocks), dim3(kBlock), 0, st, ws, nsplit, et,
                       npix, e0, em_scale, sumA, em, tavg);
    err = hipGetLastError();
    if (err != hipSuccess) return err;
    if (!bursts) {
      // chi == 1 at every epoch: replicate epoch 0
      for (int e = 1; e < n_epochs; ++e) {
        err = hipMemcpyAsync(sumA + (int64_t)e * npix, sumA, npix * sizeof(double),
                             hipMemcpyDeviceToDevice, st);
        if (err != hipSuccess) return err;
        if (em) {
          err = hipMemcpyAsync(em + (int64_t)e * npix, em, npix * sizeof(double),
                               hipMemcpyDeviceToDevice, st);
          if (err != hipSuccess) return err;
        }
      }
      break;
    }
  }
  return hipSuccess;
}

size_t ff_maps_workspace_bytes(int64_t npix, int n_epochs, int n_chan) {
  // one partial per wave of the map kernels (1 or 2 pixels per lane): at most npix/64 + 4
  const int64_t nparts = ((npix + kBlock - 1) / kBlock) * (kBlock / RJP_WAVE);
  return (size_t)nparts * n_epochs * n_chan * sizeof(double) + 256;
}

hipError_t ff_maps_launch(const double* sumA, const double* tavg, int64_t npix, int n_epochs,
                          const double* d_ctau, const double* d_cflux, int n_chan, double* tau,
                          double* flux, double* ftot, double* part, hipStream_t st) {
  // two pixels per lane (16-B loads and stores) when every map row pair is 16-B aligned
  bool vec2 = (npix % 2) == 0;
  for (const void* q : {(const void*)sumA, (const void*)tavg, (const void*)tau, (const void*)flux})
    if (q && ((uintptr_t)q % 16) != 0) vec2 = false;
  const int vec = vec2 ? 2 : 1;
  const int64_t lanes = npix / vec;
  if (!tau && !flux && ftot) {
    // totals only (light curves): the register-accumulator kernel
    const unsigned nb = (unsigned)((lanes + (int64_t)kBlock * kFtotKP - 1) / ((int64_t)kBlock * kFtotKP));
    const int np = (int)nb * (kBlock / RJP_WAVE);
    dim3 g(nb, (unsigned)n_epochs, (unsigned)((n_chan + kFtotFC - 1) / kFtotFC));
    if (vec2)
      hipLaunchKernelGGL(ff_ftot_kernel<2>, g, dim3(kBlock), 0, st, sumA, tavg, npix, d_ctau,
                         d_cflux, n_chan, part, np);
    else
      hipLaunchKernelGGL(ff_ftot_kernel<1>, g, dim3(kBlock), 0, st, sumA, tavg, npix, d_ctau,
                         d_cflux, n_chan, part, np);
    hipError_t e0 = hipGetLastError();
    if (e0 != hipSuccess) return e0;
    return sum_partials_launch(part, n_epochs * n_chan, np, ftot, st);
  }
  const unsigned nblk = (unsigned)((lanes + kBlock - 1) / kBlock);
  {
    // maps that fill the chip with 16 channels (x 4 pixel groups) per lane: the kernel with
    // per-lane flux accumulators
    const unsigned nzc = (unsigned)((n_chan + kMapsFC - 1) / kMapsFC);
    const unsigned nb4 = (unsigned)((lanes + (int64_t)kBlock * 4 - 1) / ((int64_t)kBlock * 4));
    int kp = 0;
    if ((int64_t)nb4 * n_epochs * nzc >= 1024) kp = 4;
    else if ((int64_t)nblk * n_epochs * nzc >= 1024) kp = 1;
    if (kp) {
      const unsigned nbx = kp == 4 ? nb4 : nblk;
      const int np = (int)nbx * (kBlock / RJP_WAVE);
      dim3 g(nbx, (unsigned)n_epochs, nzc);
      auto go = [&](auto vtag, auto ftag) {
        constexpr int V = decltype(vtag)::value;
        constexpr bool FT = decltype(ftag)::value;
        if (kp == 4)
          hipLaunchKernelGGL((ff_maps_acc_kernel<V, FT, 4>), g, dim3(kBlock), 0, st, sumA, tavg, npix,
                             d_ctau, d_cflux, n_chan, tau, flux, ftot ? part : nullptr, np);
        else
          hipLaunchKernelGGL((ff_maps_acc_kernel<V, FT, 1>), g, dim3(kBlock), 0, st, sumA, tavg, npix,
                             d_ctau, d_cflux, n_chan, tau, flux, ftot ? part : nullptr, np);
      };
      using I1 = std::integral_constant<int, 1>;
      using I2 = std::integral_constant<int, 2>;
      if (vec2) { if (ftot) go(I2{}, std::true_type{}); else go(I2{}, std::false_type{}); }
      else { if (ftot) go(I1{}, std::true_type{}); else go(I1{}, std::false_type{}); }
      hipError_t e1 = hipGetLastError();
      if (e1 != hipSuccess) return e1;
      return ftot ? sum_partials_launch(part, n_epochs * n_chan, np, ftot, st) : hipSuccess;
    }
  }
  const int nparts = (int)nblk * (kBlock / RJP_WAVE);
  // split channels over gridDim.z so that small maps still expose >= ~2048 blocks
  int fsplit = 1;
  while ((int64_t)nblk * n_epochs * fsplit < 2048 && fsplit < n_chan) fsplit *= 2;
  const int fchunk = (n_chan + fsplit - 1) / fsplit;
  fsplit = (n_chan + fchunk - 1) / fchunk;
  dim3 grid(nblk, (unsigned)n_epochs, (unsigned)fsplit);
  if (vec2)
    hipLaunchKernelGGL(ff_maps_kernel<2>, grid, dim3(kBlock), 0, st, sumA, tavg, npix, d_ctau,
                       d_cflux, n_chan, fchunk, tau, flux, ftot ? part : nullptr, nparts);
  else
    hipLaunchKernelGGL(ff_maps_kernel<1>, grid, dim3(kBlock), 0, st, sumA, tavg, npix, d_ctau,
                       d_cflux, n_chan, fchunk, tau, flux, ftot ? part : nullptr, nparts);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return err;
  if (ftot) err = sum_partials_launch(part, n_epochs * n_chan, nparts, ftot, st);
  return err;
}

hipError_t sum_partials_launch(const double* part, int rows, int nblk, double* out,
                               hipStream_t st) {
  hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)rows), dim3(kBlock), 0, st, part, nblk,
                     out);
  return hipGetLastError();
}

}  // namespace rjp
