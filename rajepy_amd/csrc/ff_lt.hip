// Epoch sweeps on the launch-time-ordered layout (round 4).
//
// The LDS moment pass of ff_moments.hip is bound by its f64 LDS atomics (N read-modify-writes of
// 8 bytes per cell: the LDS moves 16 N bytes per cell, HBM 16), because a cell's (jet,
// launch-time bin) is random along y and the accumulators of a sightline cannot live in
// registers.  For a model that is swept many times the cells are therefore bucketed ONCE
// (rjp_lt_count + rjp_lt_fill, per model like em0 / a0): every group of 64 consecutive sightlines
// gets its cells sorted by key q = jet * K + bin into
//     cells[(rowoff[g * 2K + q] + r) * 64 + lane] = (|a0|, ts)      16 bytes, r < rows(g, q)
// with rows(g, q) = the largest count among the group's 64 sightlines, rounded up to chunks of 4
// rows; lanes with fewer cells hold (0, bin centre).  The sweep is then ONE kernel:
//   * a wave owns a group and a contiguous range of keys; every lane is in the SAME bin at the
//     same time, so the bin's N Chebyshev moments sit in registers (2N + 1 FP64 instructions per
//     cell, no LDS) and the bin's coefficient rows W[q][n][0..32) are wave-uniform: scalar loads,
//     SGPR operands;
//   * at the end of a bin the moments are contracted into 32 epoch sums per lane (N x 32 FMAs per
//     ~100 rows) -- the moment maps never exist in HBM;
//   * rows are streamed as 16-byte loads (1 KiB per wave instruction) through three rotating
//     register buffers of 4 rows: 8-12 rows in flight while the wave computes;
//   * key ranges are split over waves so that >> 256 x 12 waves exist; their partial sums are
//     reduced in a fixed order (bit-reproducible for one layout).
// HBM-bound on the PADDED bytes: Poisson noise of the per-bin counts across the 64 sightlines of
// a group costs 15-30 % (cfg5's grid, K = 32: 1.22 x) -- the price of wave-uniform bins.
// NaN semantics as in the tiles: cells with a NaN or zero a0 are dropped (nansum); a NaN launch
// time drops the cell when its jet has bursts and counts with chi = 1 when it has none (aux sums,
// classes.py:232-233); an infinite a0 (T = 0) makes the sightline +inf.
#include <algorithm>
#include <cmath>

#include "rjp_host.h"

namespace rjp {

constexpr int kLtLanes = 64;
constexpr int kLtChunk = 4;                 // rows per (group, bin) come in chunks of this

struct LtBins { double s0, inv_h; int K; };

__device__ __forceinline__ int lt_key(double av, double tv, const LtBins& b) {
  const double w = (tv - b.s0) * b.inv_h;
  const double kf = __builtin_fmin(__builtin_fmax(__builtin_floor(w), 0.0), (double)(b.K - 1));
  return (signbit_d(av) ? 0 : b.K) + (int)kf;
}
// does the cell enter the bucketed layout?  (finite non-zero weight, finite launch time)
__device__ __forceinline__ bool lt_keeps(double av, double tv) {
  const double am = __builtin_fabs(av);
  return am > 0.0 && am <= 1.7976931348623157e308 && tv == tv;
}

size_t lt_rowoff_entries(int nx, int nz, int K) {
  const int64_t npix = (int64_t)nx * nz;
  return (size_t)((npix + kLtLanes - 1) / kLtLanes) * 2 * K + 1;
}

// rows[g][q] = largest count among the group's lanes, in chunks of kLtChunk.  One workgroup of
// four waves per group (each wave every fourth row); the counters are LDS integers.
__global__ __launch_bounds__(256) void lt_count_kernel(const double* __restrict__ a0,
                                                       const double* __restrict__ ts, int ny,
                                                       int nz, int64_t npix, LtBins b,
                                                       int32_t* __restrict__ rows,
                                                       int* __restrict__ guard) {
  extern __shared__ unsigned lt_cnt[];        // [Q][64]
  const int Q = 2 * b.K;
  for (int i = threadIdx.x; i < Q * kLtLanes; i += 256) lt_cnt[i] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t p = (int64_t)blockIdx.x * kLtLanes + lane;
  if (p < npix) {
    const int64_t x = p / nz;
    const int z = (int)(p - x * nz);
    const int64_t col = x * (int64_t)ny * nz + z;
    for (int y0 = wv; y0 < ny; y0 += 4 * 8) {
      double a[8], t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int y = y0 + 4 * u;
        const int64_t o = col + (int64_t)(y < ny ? y : ny - 1) * nz;
        a[u] = __builtin_nontemporal_load(a0 + o);
        t[u] = __builtin_nontemporal_load(ts + o);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (y0 + 4 * u < ny && lt_keeps(a[u], t[u])) {
          // range guard (include/rjprt.h): the bins cover [ts_lo, ts_hi] only; rjp_lt_count
          // reports a kept cell outside them instead of building a layout with clamped bins
          const double kw = __builtin_floor((t[u] - b.s0) * b.inv_h);
          if ((kw < 0.0 || kw > (double)b.K) && __builtin_fabs(kw) < __builtin_inf()) *guard = 1;
          atomicAdd(&lt_cnt[lt_key(a[u], t[u], b) * kLtLanes + lane], 1u);
        }
    }
  }
  __syncthreads();
  for (int q = wv; q < Q; q += 4) {
    unsigned m = lt_cnt[q * kLtLanes + lane];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, d, RJP_WAVE));
    if (lane == 0)
      rows[(int64_t)blockIdx.x * Q + q] = (int32_t)((m + (kLtChunk - 1)) & ~(unsigned)(kLtChunk - 1));
  }
}

// in-place exclusive prefix of n counts (one workgroup), the total goes to off[n]
__global__ __launch_bounds__(1024) void lt_scan_kernel(int32_t* __restrict__ off, int64_t n) {
  __shared__ long long part[1024];
  const int64_t per = (n + 1023) / 1024;
  const int64_t i0 = min(n, (int64_t)threadIdx.x * per), i1 = min(n, i0 + per);
  long long s = 0;
  for (int64_t i = i0; i < i1; ++i) s += off[i];
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    long long r = 0;
    for (int i = 0; i < 1024; ++i) { const long long v = part[i]; part[i] = r; r += v; }
    off[n] = r > 2147483647ll ? -1 : (int32_t)r;                  // the total (or "too many")
  }
  __syncthreads();
  long long r = part[threadIdx.x];
  for (int64_t i = i0; i < i1; ++i) { const int32_t c = off[i]; off[i] = (int32_t)r; r += c; }
}

// One wave per group walks y once: the r-th kept cell of (lane, key q) goes to row rowoff[q] + r;
// the rest of the key's rows is padding.  aux: [0] / [1] sums of |a0| over red / blue cells with
// a NaN launch time, [2] 1 when a kept-out cell had an infinite weight and a finite launch time.
__global__ __launch_bounds__(64) void lt_fill_kernel(const double* __restrict__ a0,
                                                     const double* __restrict__ ts, int ny, int nz,
                                                     int64_t npix, LtBins b,
                                                     const int32_t* __restrict__ off,
                                                     rjp_d2* __restrict__ cells,
                                                     double* __restrict__ aux) {
  extern __shared__ unsigned short lt_pos[];  // [Q][64]
  const int Q = 2 * b.K;
  const int lane = threadIdx.x;
  for (int q = 0; q < Q; ++q) lt_pos[q * kLtLanes + lane] = 0;
  const int32_t* go = off + (int64_t)blockIdx.x * Q;
  const int64_t p = (int64_t)blockIdx.x * kLtLanes + lane;
  const bool live = p < npix;
  double nan_r = 0.0, nan_b = 0.0, has_inf = 0.0;
  if (live) {
    const int64_t x = p / nz;
    const int z = (int)(p - x * nz);
    const int64_t col = x * (int64_t)ny * nz + z;
    for (int y0 = 0; y0 < ny; y0 += 8) {
      double a[8], t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int y = y0 + u;
        const int64_t o = col + (int64_t)(y < ny ? y : ny - 1) * nz;
        a[u] = __builtin_nontemporal_load(a0 + o);
        t[u] = __builtin_nontemporal_load(ts + o);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (y0 + u >= ny) continue;
        const double am = __builtin_fabs(a[u]);
        if (lt_keeps(a[u], t[u])) {
          const int q = lt_key(a[u], t[u], b);
          const unsigned r = lt_pos[q * kLtLanes + lane];
          lt_pos[q * kLtLanes + lane] = (unsigned short)(r + 1);
          rjp_d2 c; c.x = am; c.y = t[u];
          cells[((int64_t)go[q] + r) * kLtLanes + lane] = c;
        } else if (am > 0.0) {                     // (NaN and zero weights: nansum drops them)
          if (!(t[u] == t[u])) { if (signbit_d(a[u])) nan_r += am; else nan_b += am; }
          else has_inf = 1.0;                      // finite launch time, infinite weight
        }
      }
    }
    aux[p] = nan_r;
    aux[npix + p] = nan_b;
    aux[2 * npix + p] = has_inf;
  }
  for (int q = 0; q < Q; ++q) {
    const int nrow = go[q + 1] - go[q];
    rjp_d2 c; c.x = 0.0; c.y = b.s0 + ((q >= b.K ? q - b.K : q) + 0.5) / b.inv_h;
    for (int r = lt_pos[q * kLtLanes + lane]; r < nrow; ++r)
      cells[((int64_t)go[q] + r) * kLtLanes + lane] = c;
  }
}

hipError_t lt_count_launch(const rjp_fields* fl, int K, int32_t* d_rowoff, int* d_guard,
                           hipStream_t st) {
  const int64_t npix = (int64_t)fl->nx * fl->nz;
  const int64_t G = (npix + kLtLanes - 1) / kLtLanes;
  const double span = fl->ts_hi - fl->ts_lo;
  const LtBins b{fl->ts_lo, span > 0.0 ? K / span : 1.0, K};
  hipLaunchKernelGGL(lt_count_kernel, dim3((unsigned)G), dim3(256), (size_t)2 * K * kLtLanes * 4, st,
                     (const double*)fl->d_a0, (const double*)fl->d_ts, fl->ny, fl->nz, npix, b,
                     d_rowoff, d_guard);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(lt_scan_kernel, dim3(1), dim3(1024), 0, st, d_rowoff, G * 2 * K);
  return hipGetLastError();
}

hipError_t lt_fill_launch(const rjp_fields* fl, int K, const int32_t* d_rowoff, void* d_cells,
                          double* d_aux, hipStream_t st) {
  const int64_t npix = (int64_t)fl->nx * fl->nz;
  const int64_t G = (npix + kLtLanes - 1) / kLtLanes;
  const double span = fl->ts_hi - fl->ts_lo;
  const LtBins b{fl->ts_lo, span > 0.0 ? K / span : 1.0, K};
  hipLaunchKernelGGL(lt_fill_kernel, dim3((unsigned)G), dim3(64), (size_t)2 * K * kLtLanes * 2, st,
                     (const double*)fl->d_a0, (const double*)fl->d_ts, fl->ny, fl->nz, npix, b,
                     d_rowoff, (rjp_d2*)d_cells, d_aux);
  return hipGetLastError();
}

// ---- the sweep ----------------------------------------------------------------------------------
__device__ __forceinline__ rjp_d2 lt_load(const rjp_d2* p) { return __builtin_nontemporal_load(p); }

// final write of the sweep by the pass itself, when every group is one wave's work
struct LtDirect {
  double* sumA;             // null: write partial planes, lt_reduce_kernel finishes
  const double* aux;
  int64_t npix;
  int ne, hb_red, hb_blue;
};

#ifndef RJP_LT_ONE_WAVE_GROUPS
#define RJP_LT_ONE_WAVE_GROUPS 1024   /* maps of at least this many groups: one wave per group */
#endif
#ifndef RJP_LT_WAVES
#define RJP_LT_WAVES 4096             /* smaller maps: row shares until about this many waves exist */
#endif
#ifndef RJP_LT_OCC
#define RJP_LT_OCC 1            /* minimum waves per SIMD asked of the register allocator */
#endif
template <int N>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(RJP_LT_OCC)))
void lt_moments_kernel(const rjp_d2* __restrict__ cells,
                                                        const int32_t* __restrict__ off, LtBins b,
                                                        int nsplit, const double* __restrict__ W,
                                                        int64_t npixp, double* __restrict__ part,
                                                        LtDirect dir) {
  constexpr int ET = RJP_MOM_TILE, C = kLtChunk;
  const int Q = 2 * b.K;
  const int g = blockIdx.x / nsplit, sp = blockIdx.x % nsplit;
  const int lane = threadIdx.x;
  const int32_t* go = off + (int64_t)g * Q;
  double acc[ET], M[N];
#pragma unroll
  for (int e = 0; e < ET; ++e) acc[e] = 0.0;
#pragma unroll
  for (int n = 0; n < N; ++n) M[n] = 0.0;
  const double c1 = 2.0 * b.inv_h;                     // xi = ts * c1 + c0(bin)
  const double cb = -(2.0 * b.s0 * b.inv_h + 1.0);
  // This wave's share of the group: an equal share of its ROWS (in whole chunks of C -- every
  // bin starts at a multiple of C), whatever bins they fall in.  (Round 4 cut the KEY range
  // evenly instead: a sightline group usually belongs to ONE jet, so half the key range was
  // empty -- with two ranges one wave did all the work, profiles/r05_slab_ab.log.)  A bin cut in
  // two is no problem: each wave contracts its part of the bin's moments with the same rows of W.
  const int Rg0 = go[0];
  const int nch = (go[Q] - Rg0) / C;
  const int R0 = Rg0 + (int)((long long)nch * sp / nsplit) * C;
  const int R1 = Rg0 + (int)((long long)nch * (sp + 1) / nsplit) * C;
  int q = 0;
  while (q + 1 < Q && go[q + 1] <= R0) ++q;            // the bin R0 lies in (empty ones skipped)
  int rnext = go[q + 1];
  double c0 = cb - 2.0 * (q >= b.K ? q - b.K : q);
  const rjp_d2* base = cells + lane;
  // the moments of the bin that just ended -> 32 epoch sums (coefficient rows: scalar loads)
  auto flush = [&]() __attribute__((always_inline)) {
    const double* w = W + (size_t)q * N * ET;
#pragma unroll
    for (int n = 0; n < N; ++n) {
#pragma unroll
      for (int e = 0; e < ET; ++e) acc[e] = __builtin_fma(M[n], w[n * ET + e], acc[e]);
      M[n] = 0.0;
    }
  };
  auto issue = [&](rjp_d2 (&buf)[C], int r) __attribute__((always_inline)) {
    const int rc = r < R1 ? r : R1 - C;                // clamped: never past the range
#pragma unroll
    for (int u = 0; u < C; ++u) buf[u] = lt_load(base + (int64_t)(rc + u) * kLtLanes);
  };
  auto step = [&](const rjp_d2 (&buf)[C], int r) __attribute__((always_inline)) {
    if (r >= R1) return;
    while (r == rnext) {                               // bins that ended here (empty ones too)
      flush();
      ++q;                                             // (r < R1 <= go[Q]: q stays below Q)
      rnext = go[q + 1];
      c0 = cb - 2.0 * (q >= b.K ? q - b.K : q);
    }
#pragma unroll
    for (int u = 0; u < C; ++u) {
      const double xi = __builtin_fma(buf[u].y, c1, c0);
      double tm = buf[u].x, tc = buf[u].x * xi;
      const double x2 = xi + xi;
      M[0] += tm;
      M[1] += tc;
#pragma unroll
      for (int n = 2; n < N; ++n) {
        const double tn = __builtin_fma(x2, tc, -tm);
        tm = tc; tc = tn;
        M[n] += tn;
      }
    }
  };
  if (R0 < R1) {
    rjp_d2 A[C], B[C], D[C];
    issue(A, R0); issue(B, R0 + C);
    for (int r = R0; r < R1; r += 3 * C) {
      issue(D, r + 2 * C); step(A, r);
      issue(A, r + 3 * C); step(B, r + C);
      issue(B, r + 4 * C); step(D, r + 2 * C);
    }
  }
  if (R0 < R1) flush();                                // the last (part of a) bin
  const int64_t p = (int64_t)g * kLtLanes + lane;
  if (dir.sumA) {
    // one wave per group (nsplit == 1): the epoch sums are final -- add the cells that never
    // entered the layout and write the maps, no partial planes, no reduction kernel
    if (p < dir.npix) {
      double extra = 0.0;
      if (!dir.hb_red) extra += dir.aux[p];
      if (!dir.hb_blue) extra += dir.aux[dir.npix + p];
      const bool inf = dir.aux[2 * dir.npix + p] != 0.0;
#pragma unroll
      for (int e = 0; e < ET; ++e)
        if (e < dir.ne) dir.sumA[(int64_t)e * dir.npix + p] = inf ? __builtin_inf() : acc[e] + extra;
    }
    return;
  }
#pragma unroll
  for (int e = 0; e < ET; ++e) part[((int64_t)sp * ET + e) * npixp + p] = acc[e];
}

// fixed-order sum of the key-range partials + the cells that never entered the layout
__global__ __launch_bounds__(256) void lt_reduce_kernel(const double* __restrict__ part, int nsplit,
                                                        int64_t npix, int64_t npixp, int ne,
                                                        const double* __restrict__ aux, int hb_red,
                                                        int hb_blue, double* __restrict__ sumA) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int e = blockIdx.y;
  if (p >= npix || e >= ne) return;
  double s = 0.0;
  for (int k = 0; k < nsplit; ++k) s += part[((int64_t)k * RJP_MOM_TILE + e) * npixp + p];
  // a NaN launch time drops the cell -- unless its jet has no burst: chi == 1 there
  if (!hb_red) s += aux[p];
  if (!hb_blue) s += aux[npix + p];
  if (aux[2 * npix + p] != 0.0) s = __builtin_inf();
  sumA[(int64_t)e * npix + p] = s;
}

template <int N>
static hipError_t lt_pass(const rjp_fields* fl, const MomPlan& mp, const LtBins& b, int nsplit,
                          int64_t G, int64_t npixp, const double* W, double* ws,
                          const LtDirect& dir, hipStream_t st) {
  hipLaunchKernelGGL((lt_moments_kernel<N>), dim3((unsigned)(G * nsplit)), dim3(64), 0, st,
                     (const rjp_d2*)fl->d_lt_cells, fl->d_lt_rowoff, b, nsplit, W, npixp, ws, dir);
  return hipGetLastError();
}

hipError_t lt_run(const rjp_fields* fl, const MomPlan& mp, int n_epochs, double* sumA, double* ws,
                  size_t work_bytes, hipStream_t st) {
  if (n_epochs > RJP_LT_MAX_EPOCHS || mp.K != fl->lt_K) return hipErrorInvalidValue;
  const int64_t npix = (int64_t)fl->nx * fl->nz;
  const int64_t G = (npix + kLtLanes - 1) / kLtLanes, npixp = G * kLtLanes;
  const LtBins b{mp.s0, mp.inv_h, mp.K};
  // Waves: ONE per group from RJP_LT_ONE_WAVE_GROUPS groups on (each streams its group's
  // contiguous rows end to end and writes the final sums itself: no partial planes, no reduction);
  // smaller maps -- the x-slabs of a sharded grid -- cut every group's ROWS into equal shares
  // until about RJP_LT_WAVES waves exist.  Same-buffer A/B on 1 / 2 / 4 / 8-way slabs of cfg5's
  // grid (4096 / 2048 / 1024 / 512 groups, 20 bins; profiles/r05_slab_ab2.log): 1024 groups --
  // 1 / 2 / 4 / 8 shares 0.79 / 0.90 / 0.85 / 0.92 ms; 512 groups -- 2 / 4 / 8 / 16 shares 0.50 /
  // 0.46 / 0.44 / 0.49 ms; the whole map 3.11 ms with 1, 3.28 with 2.  The partial sums are
  // nsplit x 32 planes of npixp = 64 G doubles: never more than the CALLER'S workspace holds (on
  // tiny maps 64 G exceeds the 16-sightline padding the workspace is sized with).
  int nsplit = 1;
  if (G < RJP_LT_ONE_WAVE_GROUPS)
    while (G * nsplit < RJP_LT_WAVES && 2 * nsplit <= RJP_MOM_MAX_IDX / RJP_MOM_TILE &&
           (size_t)2 * nsplit * RJP_MOM_TILE * npixp * sizeof(double) <= work_bytes)
      nsplit *= 2;
  if (nsplit > 1 && !ws) return hipErrorInvalidValue;
  const LtDirect dir{nsplit == 1 ? sumA : nullptr, fl->d_lt_aux, npix, n_epochs,
                     mp.has_bursts[0], mp.has_bursts[1]};
  hipError_t err = hipErrorInvalidValue;
  switch (mp.N) {
    case 8: err = lt_pass<8>(fl, mp, b, nsplit, G, npixp, mp.d_Wsel, ws, dir, st); break;
    case 12: err = lt_pass<12>(fl, mp, b, nsplit, G, npixp, mp.d_Wsel, ws, dir, st); break;
    case 16: err = lt_pass<16>(fl, mp, b, nsplit, G, npixp, mp.d_Wsel, ws, dir, st); break;
    case 20: err = lt_pass<20>(fl, mp, b, nsplit, G, npixp, mp.d_Wsel, ws, dir, st); break;
    case 24: err = lt_pass<24>(fl, mp, b, nsplit, G, npixp, mp.d_Wsel, ws, dir, st); break;
    case 28: err = lt_pass<28>(fl, mp, b, nsplit, G, npixp, mp.d_Wsel, ws, dir, st); break;
    case 32: err = lt_pass<32>(fl, mp, b, nsplit, G, npixp, mp.d_Wsel, ws, dir, st); break;
  }
  if (err != hipSuccess || nsplit == 1) return err;
  hipLaunchKernelGGL(lt_reduce_kernel, dim3((unsigned)((npix + 255) / 256), (unsigned)n_epochs),
                     dim3(256), 0, st, ws, nsplit, npix, npixp, n_epochs, fl->d_lt_aux,
                     mp.has_bursts[0], mp.has_bursts[1], sumA);
  return hipGetLastError();
}

}  // namespace rjp
