// K1 device code and its templated launchers (included by ff_scan.hip for the shared types and by
// the ff_scan_inst_*.hip translation units, each of which instantiates one slice of the
// {storage, layout, Gaunt mode} x {lane width, epoch tile, bursts, EM} kernel family).
#pragma once
#include <algorithm>
#include <cmath>
#include <type_traits>
#include <vector>
#include "rjp_device.h"
#include "rjp_host.h"

namespace rjp {

template <typename T>
struct FieldPtrs {
  const T* nd;
  const T* xi;
  const T* temp;
  const T* pf;
  const T* ts;
  const int32_t* ylo;      // optional occupied y-range per sightline (nullptr = all rows)
  const int32_t* yhi;
  const T* em0;            // compact layout: (|nd| xi)^2 pf, sign bit = red jet
  const T* a0;             // tau layout: em0 * T^-1.5 (or T^-1.35), sign bit = red jet
};

// A NaN launch time never reaches the jet: its cell is given chi = 1 here (a launch at
// -1e300 s is > 700 sigma from every burst) and masked when it is accumulated.
__device__ __forceinline__ double launch_or_never(double ts) { return __builtin_fmax(ts, -1e300); }

// x, or NaN when !keep: only the high dword is touched
__device__ __forceinline__ double poison_unless(double x, bool keep) {
  const long long b = __double_as_longlong(x);
  const unsigned hi = keep ? (unsigned)((unsigned long long)b >> 32) : 0x7FF80000u;
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) |
                                          ((unsigned long long)b & 0xFFFFFFFFull)));
}

// x if neither x nor t is NaN, else +0 (x >= 0 or NaN: a magnitude as load_rows leaves it)
__device__ __forceinline__ double keep_if_ordered(double x, double t) {
  return __builtin_isunordered(x, t) ? 0.0 : x;
}

// nansum: NaN -> 0.  A compact-layout term is never negative ((n x)^2 pf with pf >= 0 times a
// temperature power), so max(x, 0) does it in one instruction; the wide layout admits any
// sign of pf and selects.
template <bool NONNEG>
__device__ __forceinline__ double nan_to_zero(double x) {
  if (NONNEG) return __builtin_fmax(x, 0.0);
  return x == x ? x : 0.0;
}

template <int ET>
struct EpochTile {
  double t[ET];
  UnifDev un;        // uniform-spacing recurrence for the burst factor (ET >= 4 only)
};

constexpr int kBlock = 256;
constexpr int kMaxTile = 32;     // largest epoch tile (uniformly spaced epochs, no EM maps)
// y-rows of loads kept in flight per lane; fewer when many accumulators are live so the
// kernel stays inside the 256-VGPR budget without scratch
#ifndef RJP_UNROLL_BASE
#define RJP_UNROLL_BASE 4
#endif
// 4-wide (f32) lanes on the wide layout: 2 rows (8 cells per batch, 166 VGPRs; 4 rows need
// 256 + AGPR spills); on the compact layout 4 rows still fit 3 waves/SIMD and are 3 % faster
// ... and 2 rows in the power-law Gaunt mode, whose T^-1.35 chains are batched by eight cells
// single-epoch scan on the tau layout (a0, ts): 6 rows = 12 loads of 16 B in flight per lane,
// 126 VGPRs, still 4 waves/SIMD -- 1.2-1.8 % faster than 4 rows in an A/B on the same
// buffers (8 rows: 158 VGPRs, 3 waves, no better; profiles/r03_k1_tau_tuning_ab.log); with
// the EM accumulators (a third field) 6 rows would cost a wave: 4
#ifndef RJP_UNROLL_TAU
#define RJP_UNROLL_TAU 6
#endif
__host__ __device__ constexpr int unroll_for(int vec, int et, int lay, int mode, bool em = true) {
  return vec * et >= 16 ? 1
         : vec * et >= 8 ? 2
         : lay == LAY_TAU ? (vec * et >= 4 ? 2 : em ? 4 : RJP_UNROLL_TAU)
         : vec == 4 && (lay == LAY_WIDE || mode == RJP_GFF_POWERLAW) ? 2 : RJP_UNROLL_BASE;
}
// The fast T^-1.35 (power-law Gaunt mode) is a property of the KERNEL, not of a row batch:
// the unrolled body and the row tail must evaluate a cell identically, or a scan would depend
// on where its y-range starts.  Every tile uses it (the Halley form needs ~8 live registers,
// so the 16- and 32-epoch tiles afford it too; they used to call libm's pow out of line).
__host__ __device__ constexpr bool fast_power_law(int vec, int et, int lay, int mode) {
  (void)vec; (void)et; (void)lay;
  return mode == RJP_GFF_POWERLAW;
}

// number of accumulator planes a tile of ET epochs writes per y-split
__host__ __device__ constexpr int nacc(int et) { return 2 * et + 2; }

// U rows x VEC sightlines of one lane: loads first (U*5 independent 16-B loads in flight),
// then the burst factors of all U*VEC*ET (cell, epoch) pairs as ONE batch so their exp()
// polynomial chains interleave (FP64 FMA latency is what limits a single chain), then the
// accumulation.
// U rows x VEC sightlines of one lane as they come out of memory
template <int VEC, int U>
struct RowBatch {
  double g0[U][VEC];     // (n x)^2 * ff/areas at chi = 1
  double a[U][VEC];      // tau layout: g0 * T^-1.5|-1.35 as stored
  double tp[U][VEC];     // temperature (wide and compact layouts)
  double ts[U][VEC];     // launch time
  bool rj[U][VEC];       // red-jet flag
  uint32_t sg[U][VEC];   // high dword of the field that carries it in its sign bit
};

template <typename T, int VEC, bool BURSTS, int LAY, bool EM, int U>
__device__ __forceinline__ void load_rows(const FieldPtrs<T>& f, int64_t off, int64_t stride,
                                          RowBatch<VEC, U>& rb) {
  auto& g0 = rb.g0; auto& tp = rb.tp; auto& ts = rb.ts; auto& rj = rb.rj;
  // g and the jet flag: from three wide fields or from the one compact field
  if constexpr (LAY == LAY_TAU) {
    auto& a = rb.a;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t o = off + u * stride;
      load_vec(f.a0 + o, a[u]);
      if (EM) load_vec(f.em0 + o, g0[u]);
      if (BURSTS) load_vec(f.ts + o, ts[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        rj[u][v] = signbit_d(a[u][v]);
        rb.sg[u][v] = hi_dword(a[u][v]);
        a[u][v] = fabs(a[u][v]);
        if (EM) g0[u][v] = fabs(g0[u][v]);
      }
  } else if constexpr (LAY == LAY_CMP) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t o = off + u * stride;
      load_vec(f.em0 + o, g0[u]);
      load_vec(f.temp + o, tp[u]);
      if (BURSTS) load_vec(f.ts + o, ts[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        rj[u][v] = signbit_d(g0[u][v]);
        rb.sg[u][v] = hi_dword(g0[u][v]);
        g0[u][v] = fabs(g0[u][v]);
      }
  } else {
    double nd[U][VEC], xi[U][VEC], pf[U][VEC];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t o = off + u * stride;
      load_vec(f.nd + o, nd[u]);
      load_vec(f.xi + o, xi[u]);
      load_vec(f.temp + o, tp[u]);
      load_vec(f.pf + o, pf[u]);
      if (BURSTS) load_vec(f.ts + o, ts[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const double n0 = fabs(nd[u][v]) * xi[u][v];   // steady-state electron density
        g0[u][v] = n0 * n0 * pf[u][v];
        rj[u][v] = signbit_d(nd[u][v]);
        rb.sg[u][v] = hi_dword(nd[u][v]);
      }
  }
}

template <typename T, int VEC, int ET, int MODE, bool BURSTS, bool UNIF, int LAY, bool EM, int U>
__device__ __forceinline__ void compute_rows(const RowBatch<VEC, U>& rb, const BurstsDev& b,
                                             const EpochTile<ET>& ep, double (&accA)[ET][VEC],
                                             double (&accE)[EM ? ET : 1][VEC],
                                             double (&accT)[VEC], int (&cnt)[VEC]) {
  const auto& g0 = rb.g0; const auto& tp = rb.tp; const auto& ts = rb.ts; const auto& rj = rb.rj;
  constexpr int NB = ET * U * VEC;
  double chi[NB];
  if (BURSTS && UNIF) {
    // uniformly spaced epochs: two exp() per (cell, burst) for the whole tile
    double tlm[U * VEC];
    bool red[U * VEC];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        tlm[u * VEC + v] = ep.t[ET / 2] - launch_or_never(ts[u][v]);
        red[u * VEC + v] = rj[u][v];
      }
    chi_batch_uniform<ET, U * VEC>(b, ep.un, red, tlm, chi);
  } else if (BURSTS) {
    double tl[NB];
    uint32_t sg[U * VEC];
#pragma unroll
    for (int e = 0; e < ET; ++e)
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          const int k = (e * U + u) * VEC + v;
          // a NaN launch time needs no special care on this path: the Gaussian argument is
          // NaN, exp's clamp max(arg, -1021) returns the number, the burst adds ~1e-308 and
          // chi stays 1; the cell is masked when it is accumulated
          tl[k] = ep.t[e] - ts[u][v];
          sg[u * VEC + v] = rb.sg[u][v];
        }
    chi_batch<NB, U * VEC, sizeof(T) == 4>(b, sg, tl, chi);
  }

  if constexpr (LAY == LAY_TAU) {
    // the temperature power is part of the stored field (exactly the product the other
    // layouts form below: g0 * tpow), T_avg has its own pass: what is left per cell is the
    // mask and one FMA per epoch and sum
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        if (BURSTS) {
          // nansum: a term is dropped when the field or the launch time is NaN -- ONE unordered
          // compare of the two and a select of the magnitude (3 instructions; poisoning the
          // high dword and two fmax took 5)
          const double am = keep_if_ordered(rb.a[u][v], ts[u][v]);
          const double gm = EM ? keep_if_ordered(g0[u][v], ts[u][v]) : 0.0;
#pragma unroll
          for (int e = 0; e < ET; ++e) {
            const double c = chi[(e * U + u) * VEC + v];
            const double c2 = c * c;
            if (EM) accE[e][v] = __builtin_fma(gm, c2, accE[e][v]);
            accA[e][v] = __builtin_fma(am, c2, accA[e][v]);
          }
        } else {
          if (EM) accE[0][v] += nan_to_zero<true>(g0[u][v]);
          accA[0][v] += nan_to_zero<true>(rb.a[u][v]);
        }
      }
    }
    return;
  }

  // temperature powers of the whole batch: T^-1.5 (scalar Gaunt) or T^-1.35 = T^-1.5 * T^0.15
  // (power law)
  constexpr bool kFastPowerLaw = fast_power_law(VEC, ET, LAY, MODE);
  constexpr bool CMP = LAY != LAY_WIDE;
  double tpw[U][VEC];
  if constexpr (kFastPowerLaw && (U * VEC) % 4 == 0 && U * VEC > 4) {
    // groups of four: the log/exp chains are long, interleaving all eight costs a wave of
    // occupancy
#pragma unroll
    for (int g4 = 0; g4 < U * VEC; g4 += 4) {
      pow_m1p35_batch<4>(*reinterpret_cast<const double (*)[4]>(&tp[0][0] + g4),
                         *reinterpret_cast<double (*)[4]>(&tpw[0][0] + g4));
      __builtin_amdgcn_sched_barrier(0);
    }
  } else if constexpr (kFastPowerLaw)
    pow_m1p35_batch<U * VEC>(reinterpret_cast<const double (&)[U * VEC]>(tp),
                             reinterpret_cast<double (&)[U * VEC]>(tpw));
  else
    pow_m1p5_batch<U * VEC>(reinterpret_cast<const double (&)[U * VEC]>(tp),
                            reinterpret_cast<double (&)[U * VEC]>(tpw));

#pragma unroll
  for (int u = 0; u < U; ++u) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      const double Tk = tp[u][v];
      double tpow = tpw[u][v];
      if (MODE == RJP_GFF_POWERLAW && !kFastPowerLaw) tpow *= pow(Tk, 0.15);
      // nanmean over T > 0 (classes.py:1471): max(T, 0) adds T, or an exact zero for
      // T <= 0 and NaN
      accT[v] += __builtin_fmax(Tk, 0.0);
      cnt[v] += Tk > 0.0 ? 1 : 0;
      if (BURSTS) {
        // nansum semantics hoisted out of the epoch loop: g chi^2 is NaN iff g is NaN or
        // chi is (chi is NaN iff the launch time is -- such cells were given chi = 1 above);
        // a masked cell contributes an exact zero at every epoch.  (The one exception of the
        // reference -- a jet WITHOUT bursts has a constant mass-loss rate, classes.py:232-233,
        // so a NaN launch time does not reach its cells -- is served by the caller: it scans
        // launch times patched by rjp_unmask_launch_times; a per-cell test here cost the
        // single-epoch scan 2.7 %.)
        const double g = poison_unless(g0[u][v], ts[u][v] == ts[u][v]);
        const double gm = nan_to_zero<CMP>(g);
        const double am = nan_to_zero<CMP>(g * tpow);
#pragma unroll
        for (int e = 0; e < ET; ++e) {
          const double c = chi[(e * U + u) * VEC + v];
          const double c2 = c * c;
          if (EM) accE[e][v] = __builtin_fma(gm, c2, accE[e][v]);
          accA[e][v] = __builtin_fma(am, c2, accA[e][v]);
        }
      } else {
        if (EM) accE[0][v] += nan_to_zero<CMP>(g0[u][v]);
        accA[0][v] += nan_to_zero<CMP>(g0[u][v] * tpow);
      }
    }
  }
}

template <typename T, int VEC, int ET, int MODE, bool BURSTS, bool UNIF, int LAY, bool EM, int U>
__device__ __forceinline__ void scan_rows(const FieldPtrs<T>& f, int64_t off, int64_t stride,
                                          const BurstsDev& b, const EpochTile<ET>& ep,
                                          double (&accA)[ET][VEC],
                                          double (&accE)[EM ? ET : 1][VEC],
                                          double (&accT)[VEC], int (&cnt)[VEC]) {
  RowBatch<VEC, U> rb;
  load_rows<T, VEC, BURSTS, LAY, EM, U>(f, off, stride, rb);
  compute_rows<T, VEC, ET, MODE, BURSTS, UNIF, LAY, EM, U>(rb, b, ep, accA, accE, accT, cnt);
}

// EM = false (flux-vs-time sweeps: no emission-measure maps wanted) drops the second
// accumulator set: fewer registers, one more wave per SIMD on the 16-epoch tiles.
template <typename T, int VEC, int ET, int MODE, bool BURSTS, bool UNIF, int LAY, bool EM>
__global__ __launch_bounds__(kBlock) void ff_scan_kernel(
    FieldPtrs<T> f, int ny, int nz, int64_t nchunks, int64_t npix, int ylen, int nsplit,
    BurstsDev b, EpochTile<ET> ep, double* __restrict__ ws) {
  constexpr int kUnroll = unroll_for(VEC, ET, LAY, MODE, EM);
  // 1-D grid with the y-split index fastest: workgroups that run together stream consecutive
  // y-ranges of the same sightlines, i.e. neighbouring memory, instead of ranges 16 MiB apart
  // (n_y n_z elements) -- +5 % on cfg4 (6.0 -> 6.3 TB/s)
  const int split = (int)(blockIdx.x % (unsigned)nsplit);
  const int64_t c = (int64_t)(blockIdx.x / (unsigned)nsplit) * kBlock + threadIdx.x;
  const bool lane_live = c < nchunks;
  const int64_t p0 = c * VEC;              // first sightline (pixel) of this lane
  int y0 = split * ylen;
  int y1 = min(ny, y0 + ylen);
  if (f.ylo) {
    // sparse models: clip this workgroup's rows to the occupied range of its sightlines
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

  double accA[ET][VEC], accE[EM ? ET : 1][VEC], accT[VEC];
  int cnt[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    accT[v] = 0.0; cnt[v] = 0;
#pragma unroll
    for (int e = 0; e < ET; ++e) { accA[e][v] = 0.0; if (EM) accE[e][v] = 0.0; }
  }

  int64_t off = (x * ny + y0) * (int64_t)nz + z;
  const int64_t stride = nz;

  int y = y0;
  // (issuing the next half-batch's loads before computing the current one was tried: 165
  // VGPRs, 3 waves/SIMD, 7 % slower)
  for (; y + kUnroll <= y1; y += kUnroll) {
    scan_rows<T, VEC, ET, MODE, BURSTS, UNIF, LAY, EM, kUnroll>(f, off, stride, b, ep, accA, accE, accT, cnt);
    off += kUnroll * stride;
  }
  for (; y < y1; ++y) {
    scan_rows<T, VEC, ET, MODE, BURSTS, UNIF, LAY, EM, 1>(f, off, stride, b, ep, accA, accE, accT, cnt);
    off += stride;
  }

  // partial sums: ws[split][plane][pixel]
  double* w = ws + (int64_t)split * nacc(ET) * npix + p0;
#pragma unroll
  for (int e = 0; e < ET; ++e) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      w[(int64_t)e * npix + v] = accA[e][v];
      if (EM) w[(int64_t)(ET + e) * npix + v] = accE[e][v];
    }
  }
  if constexpr (LAY != LAY_TAU) {     // (the tau layout keeps no temperature sums)
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      w[(int64_t)(2 * ET) * npix + v] = accT[v];
      w[(int64_t)(2 * ET + 1) * npix + v] = (double)cnt[v];
    }
  }
}

// ---- epoch tiles of >= 16 epochs with the rows prefetched by LDS-DMA ------------------------
// These tiles are ALU-bound with ~1700 cycles of FP64 work per row and wave and 2-3 waves per
// SIMD; with ordinary loads a wave asks for a row and waits for it (all its registers hold
// accumulators and burst factors, none is free to hold a row in flight), so a SIMD idles
// whenever its waves wait together (VALU issue 75 %).  Here each wave keeps the NEXT two rows
// of its 64 sightlines in flight as `global_load_lds_dwordx4` requests -- no register
// destination -- into its own double-buffered slice of LDS, and reads the current rows back
// with ds_read_b64 (conflict-free, consecutive lanes).  One request moves 1 KiB = two rows of
// one field: lanes 0-31 fetch the 64 sightlines of row y as 16-byte pairs, lanes 32-63 those
// of row y + 1; the LDS image is wave-base + 16 * lane.  No barrier anywhere: a wave reads
// only what it requested itself, behind its own counted s_waitcnt vmcnt.
// Needs f64 fields, an even n_z and 16-byte aligned field pointers (what 2-wide lanes need).
template <int LAY, bool EM> struct TileDma {
  static constexpr int NF = LAY == LAY_TAU ? (EM ? 3 : 2) : LAY == LAY_CMP ? 3 : 5;
};

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// (the 32-epoch tiles with EM maps hold 96 accumulators and burst factors per lane: the register
// allocator is told to stay inside 2 waves per SIMD -- left to itself it takes 256 VGPRs plus
// AGPR copies for the tau layout and runs one wave per SIMD, 24 ms instead of 19)
template <int ET, int MODE, int LAY, bool EM>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(ET >= 32 && EM ? 2 : 1)))
void ff_scan_tile_kernel(
    FieldPtrs<double> f, int ny, int nz, int64_t npix, int ylen, int nsplit, BurstsDev b,
    EpochTile<ET> ep, double* __restrict__ ws) {
  constexpr int NF = TileDma<LAY, EM>::NF;
  constexpr int kWaves = kBlock / RJP_WAVE;
  // [wave][buffer][field][row 0: 64 sightlines | row 1: 64 sightlines]
  __shared__ double s_rows[kWaves][2][NF][2 * RJP_WAVE];
  __shared__ int s_lo, s_hi;
  const int split = (int)(blockIdx.x % (unsigned)nsplit);
  const int64_t p0 = (int64_t)(blockIdx.x / (unsigned)nsplit) * kBlock + threadIdx.x;
  const bool lane_live = p0 < npix;
  int y0 = split * ylen;
  int y1 = min(ny, y0 + ylen);
  if (f.ylo) {
    if (threadIdx.x == 0) { s_lo = ny; s_hi = 0; }
    __syncthreads();
    if (lane_live) {
      const int lo = f.ylo[p0], hi = f.yhi[p0];
      if (lo < hi) { atomicMin(&s_lo, lo); atomicMax(&s_hi, hi); }
    }
    __syncthreads();
    y0 = max(y0, s_lo);
    y1 = min(y1, s_hi);
  }

  double accA[ET][1], accE[EM ? ET : 1][1], accT[1];
  int cnt[1];
  accT[0] = 0.0; cnt[0] = 0;
#pragma unroll
  for (int e = 0; e < ET; ++e) { accA[e][0] = 0.0; if (EM) accE[e][0] = 0.0; }

  const int lane = threadIdx.x & (RJP_WAVE - 1);
  const int wave = threadIdx.x / RJP_WAVE;
  // the pair of sightlines this lane fetches (clamped into the map: the last workgroup)
  const int64_t pw = p0 - lane;                                   // the wave's first sightline
  const int64_t pd = min(pw + 2 * (lane & 31), npix - 2);
  const int64_t xd = pd / nz;
  const int zd = (int)(pd - xd * nz);
  const int half = lane >> 5;                                     // which of the two rows
  const int64_t col = xd * ny * (int64_t)nz + zd;                 // + row * nz
  const double* src[NF];
  if constexpr (LAY == LAY_TAU) { src[0] = f.a0; src[1] = f.ts; if constexpr (EM) src[2] = f.em0; }
  else if constexpr (LAY == LAY_CMP) { src[0] = f.em0; src[1] = f.temp; src[2] = f.ts; }
  else { src[0] = f.nd; src[1] = f.xi; src[2] = f.temp; src[3] = f.pf; src[4] = f.ts; }
  typedef __attribute__((address_space(3))) double lds_double;
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_double*)&s_rows[wave][0][0][0];
  const unsigned lds0u = __builtin_amdgcn_readfirstlane(lds0);
  auto request = [&](int y, int buf) __attribute__((always_inline)) {
    const int row = min(y + half, y1 - 1);                         // an odd tail re-reads a row
    const int64_t o = col + (int64_t)row * nz;
#pragma unroll
    for (int k = 0; k < NF; ++k)
      glds16(src[k] + o, lds0u + (unsigned)((buf * NF + k) * 2 * RJP_WAVE * sizeof(double)));
  };
  auto one_row = [&](int buf, int r) __attribute__((always_inline)) {
    RowBatch<1, 1> rb;
    const double* q = &s_rows[wave][buf][0][r * RJP_WAVE + lane];
    if constexpr (LAY == LAY_TAU) {
      const double a = q[0];
      rb.ts[0][0] = q[2 * RJP_WAVE];
      rb.rj[0][0] = signbit_d(a);
      rb.sg[0][0] = hi_dword(a);
      rb.a[0][0] = fabs(a);
      if constexpr (EM) rb.g0[0][0] = fabs(q[4 * RJP_WAVE]);
    } else if constexpr (LAY == LAY_CMP) {
      const double g = q[0];
      rb.tp[0][0] = q[2 * RJP_WAVE];
      rb.ts[0][0] = q[4 * RJP_WAVE];
      rb.rj[0][0] = signbit_d(g);
      rb.sg[0][0] = hi_dword(g);
      rb.g0[0][0] = fabs(g);
    } else {
      const double nd = q[0], xi = q[2 * RJP_WAVE], pf = q[6 * RJP_WAVE];
      rb.tp[0][0] = q[4 * RJP_WAVE];
      rb.ts[0][0] = q[8 * RJP_WAVE];
      const double n0 = fabs(nd) * xi;
      rb.g0[0][0] = n0 * n0 * pf;
      rb.rj[0][0] = signbit_d(nd);
      rb.sg[0][0] = hi_dword(nd);
    }
    compute_rows<double, 1, ET, MODE, true, true, LAY, EM, 1>(rb, b, ep, accA, accE, accT, cnt);
  };

  if (y0 < y1) {
    request(y0, 0);
    int buf = 0;
    for (int y = y0; y < y1; y += 2) {
      if (y + 2 < y1) {
        request(y + 2, buf ^ 1);
        // all but the NF requests just issued have landed
        if (NF == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if (NF == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      // (one copy of the row's code: the loop over the two rows is not unrolled)
#pragma unroll 1
      for (int r = 0; r < 2; ++r)
        if (y + r < y1) one_row(buf, r);
      buf ^= 1;
    }
  }

  if (!lane_live) return;
  double* w = ws + (int64_t)split * nacc(ET) * npix + p0;
#pragma unroll
  for (int e = 0; e < ET; ++e) {
    w[(int64_t)e * npix] = accA[e][0];
    if (EM) w[(int64_t)(ET + e) * npix] = accE[e][0];
  }
  if constexpr (LAY != LAY_TAU) {
    w[(int64_t)(2 * ET) * npix] = accT[0];
    w[(int64_t)(2 * ET + 1) * npix] = (double)cnt[0];
  }
}

// ---- templated launchers of one (storage, layout, Gaunt mode) slice --------------------------
template <typename T, int VEC, int ET, int MODE, bool BURSTS, int LAY>
inline hipError_t launch_tile(const rjp_fields* fl, const BurstsDev& b, const double* t,
                              const UnifDev& un, int nsplit, int ylen, double* ws, bool want_em,
                              hipStream_t st) {
  // (long tiles: f64 storage on the tau / compact layouts only -- ff_scan_plan never asks for
  // another, and nothing is instantiated for them)
  if constexpr (ET >= 16 && (sizeof(T) == 4 || LAY == LAY_WIDE)) {
    return hipErrorInvalidValue;
  } else {
  FieldPtrs<T> f{(const T*)fl->d_nd, (const T*)fl->d_xi, (const T*)fl->d_temp,
                 (const T*)fl->d_pf, (const T*)fl->d_ts, fl->d_ylo, fl->d_yhi,
                 (const T*)fl->d_em0, (const T*)fl->d_a0};
  EpochTile<ET> ep;
  for (int e = 0; e < ET; ++e) ep.t[e] = t[e];
  ep.un = un;
  if (ET < 4) ep.un.on = 0;
  const int64_t npix = (int64_t)fl->nx * fl->nz;
  const int64_t nchunks = npix / VEC;
  dim3 grid((unsigned)(((nchunks + kBlock - 1) / kBlock) * nsplit), 1u);
  // the recurrence pays with at least 4 epochs per tile; short tiles of f32 storage keep
  // their 9-instruction float-accuracy exp instead
  if constexpr (sizeof(T) == 8 && VEC == 1 && BURSTS && ET >= 16) {
    if (ep.un.on && tile_dma_ok(fl)) {
      FieldPtrs<double> fd{(const double*)fl->d_nd, (const double*)fl->d_xi,
                           (const double*)fl->d_temp, (const double*)fl->d_pf,
                           (const double*)fl->d_ts, fl->d_ylo, fl->d_yhi,
                           (const double*)fl->d_em0, (const double*)fl->d_a0};
      if (want_em)
        hipLaunchKernelGGL((ff_scan_tile_kernel<ET, MODE, LAY, true>), grid, dim3(kBlock), 0, st,
                           fd, fl->ny, fl->nz, npix, ylen, nsplit, b, ep, ws);
      else
        hipLaunchKernelGGL((ff_scan_tile_kernel<ET, MODE, LAY, false>), grid, dim3(kBlock), 0, st,
                           fd, fl->ny, fl->nz, npix, ylen, nsplit, b, ep, ws);
      return hipGetLastError();
    }
  }
  if constexpr (ET == 32) {
    // 32 epochs per pass: recurrence only
    if (ep.un.on) {
      if (want_em)
        hipLaunchKernelGGL((ff_scan_kernel<T, VEC, ET, MODE, true, true, LAY, true>), grid,
                           dim3(kBlock), 0, st, f, fl->ny, fl->nz, nchunks, npix, ylen, nsplit, b, ep, ws);
      else
        hipLaunchKernelGGL((ff_scan_kernel<T, VEC, ET, MODE, true, true, LAY, false>), grid,
                           dim3(kBlock), 0, st, f, fl->ny, fl->nz, nchunks, npix, ylen, nsplit, b, ep, ws);
      return hipGetLastError();
    }
  } else if constexpr (BURSTS && ET >= 4 && (sizeof(T) == 8 || ET == 16)) {
    if (ep.un.on) {
      if (want_em)
        hipLaunchKernelGGL((ff_scan_kernel<T, VEC, ET, MODE, true, true, LAY, true>), grid,
                           dim3(kBlock), 0, st, f, fl->ny, fl->nz, nchunks, npix, ylen, nsplit, b, ep, ws);
      else
        hipLaunchKernelGGL((ff_scan_kernel<T, VEC, ET, MODE, true, true, LAY, false>), grid,
                           dim3(kBlock), 0, st, f, fl->ny, fl->nz, nchunks, npix, ylen, nsplit, b, ep, ws);
      return hipGetLastError();
    }
  }
  if constexpr (ET <= 8) {
    // single-epoch and generic tiles always carry the emission measure (cheap there)
    // (on the tau layout the EM accumulators cost a third field: never carried unasked)
    if (want_em || (ET < 4 && LAY != LAY_TAU))
      hipLaunchKernelGGL((ff_scan_kernel<T, VEC, ET, MODE, BURSTS, false, LAY, true>), grid,
                         dim3(kBlock), 0, st, f, fl->ny, fl->nz, nchunks, npix, ylen, nsplit, b, ep, ws);
    else
      hipLaunchKernelGGL((ff_scan_kernel<T, VEC, ET, MODE, BURSTS, false, LAY, false>), grid,
                         dim3(kBlock), 0, st, f, fl->ny, fl->nz, nchunks, npix, ylen, nsplit, b, ep, ws);
    return hipGetLastError();
  }
  return hipErrorInvalidValue;       // a 16-epoch tile that is not uniform: launcher bug
  }
}

template <typename T, int VEC, int MODE, int LAY>
inline hipError_t dispatch_et(const rjp_fields* fl, const BurstsDev& b, bool bursts,
                              const double* t, const UnifDev& un, int et, int nsplit, int ylen,
                              double* ws, bool want_em, hipStream_t st) {
  if (!bursts) return launch_tile<T, VEC, 1, MODE, false, LAY>(fl, b, t, un, nsplit, ylen, ws, want_em, st);
  switch (et) {
    case 1: return launch_tile<T, VEC, 1, MODE, true, LAY>(fl, b, t, un, nsplit, ylen, ws, want_em, st);
    case 2: return launch_tile<T, VEC, 2, MODE, true, LAY>(fl, b, t, un, nsplit, ylen, ws, want_em, st);
    case 4: return launch_tile<T, VEC, 4, MODE, true, LAY>(fl, b, t, un, nsplit, ylen, ws, want_em, st);
    case 8:
      // 4 sightlines x 8 epochs x 2 sums does not fit 256 VGPRs: the launcher caps the
      // epoch tile at 4 for 4-wide (f32) lanes
      if constexpr (VEC == 4) return hipErrorInvalidValue;
      else return launch_tile<T, VEC, 8, MODE, true, LAY>(fl, b, t, un, nsplit, ylen, ws, want_em, st);
    case 16:
      // only the uniform-epoch recurrence keeps 16 epochs of state in registers
      if constexpr (VEC == 1)
        return launch_tile<T, VEC, 16, MODE, true, LAY>(fl, b, t, un, nsplit, ylen, ws, want_em, st);
      else return hipErrorInvalidValue;
    case 32:
      if constexpr (VEC == 1)
        return launch_tile<T, VEC, 32, MODE, true, LAY>(fl, b, t, un, nsplit, ylen, ws, want_em, st);
      else return hipErrorInvalidValue;
  }
  return hipErrorInvalidValue;
}

}  // namespace rjp
