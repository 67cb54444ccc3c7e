// Feasibility probe (round 3): how fast can per-sightline Chebyshev moments of
// a0 over launch-time bins be accumulated in ONE pass over the grid when launch times are
// uncorrelated along y (the synthetic set)?  tau_e(p) = sum_y a0_y F(t_e - ts_y) is a convolution
// of the sightline's launch-time distribution with F = chi^2: with K bins x N moments per jet the
// whole epoch sweep becomes this pass + a small contraction.  Accumulators: LDS, f64 atomics.
//   hipcc --offload-arch=gfx950 -O3 -o moments_probe moments_probe.hip ; ./moments_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void fill(double* a0, double* ts, size_t n, int nz) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t step = (size_t)gridDim.x * 256;
  for (; i < n; i += step) {
    unsigned long long x = i * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull;
    unsigned long long y = (i + 0x1234567) * 0xD1B54A32D192ED03ull; y ^= y >> 31; y *= 0x94D049BB133111EBull;
    const double u = (double)(x >> 11) * (1.0 / 9007199254740992.0);
    const double v = (double)(y >> 11) * (1.0 / 9007199254740992.0);
    const bool red = (int)(i % nz) < nz / 2;
    a0[i] = (red ? -1.0 : 1.0) * (1.0 + 100.0 * u);
    ts[i] = 5.0 * v;
  }
}

// SL sightlines per workgroup (z-adjacent), 256 threads: thread = (sightline, y-row offset)
template <int K, int N, int SL, int U, int BS, int JETS = 2, bool PF = false, bool PAIR = false>
__global__ __launch_bounds__(BS) void moments(const double* __restrict__ a0, const double* __restrict__ ts,
                                               int ny, int nz, double s0, double inv_h,
                                               double* __restrict__ MT, size_t npix) {
  extern __shared__ double lds[];          // [JETS][K][N][SL]
  constexpr int TOT = JETS * K * N * SL;
  for (int i = threadIdx.x; i < TOT; i += BS) lds[i] = 0.0;
  __syncthreads();
  const int sl = threadIdx.x % SL, yr = threadIdx.x / SL;
  constexpr int YR = BS / SL;
  // PAIR: tiles 2k and 2k+1 (the two 64-byte halves of a row's 128-byte lines when SL = 8) go to
  // workgroups b and b + 8 -- the same XCD, launched together: the second half hits in L2
  const unsigned bid = PAIR ? ((blockIdx.x % 8) * 2 + (blockIdx.x / 8) % 2 + 16 * (blockIdx.x / 16)) : blockIdx.x;
  const size_t p = (size_t)bid * SL + sl;
  const size_t x = p / nz;
  const int z = (int)(p - x * nz);
  const size_t col = x * (size_t)ny * nz + z;
  auto cell = [&](double av, double tv) __attribute__((always_inline)) {
      const bool red = av < 0.0;
      const double am = fmax(fabs(av), 0.0);
      const double w = (tv - s0) * inv_h;
      double kf = floor(w);
      kf = fmin(fmax(kf, 0.0), (double)(K - 1));
      const double xi = 2.0 * (w - kf) - 1.0;
      double* base = lds + ((((red || JETS == 1) ? 0 : K) + (int)kf) * N) * SL + sl;
      double tm = 1.0, tc = xi;
      atomicAdd(base, am);
      atomicAdd(base + SL, am * tc);
      const double x2 = 2.0 * xi;
#pragma unroll
      for (int n = 2; n < N; ++n) {
        const double tn = __builtin_fma(x2, tc, -tm);
        tm = tc; tc = tn;
        atomicAdd(base + n * SL, am * tn);
      }
  };
  if (PF) {
    // unconditional loads (row index clamped, weight zeroed instead): no branches around the
    // loads, so the waits can be counted and the next rows stay in flight behind the atomics
    double a[U], t[U], an[U], tn_[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int y = yr + u * YR;
      const int yc = y < ny ? y : ny - 1;
      a[u] = __builtin_nontemporal_load(a0 + col + (size_t)yc * nz);
      t[u] = __builtin_nontemporal_load(ts + col + (size_t)yc * nz);
    }
    auto fetch = [&](double (&aa)[U], double (&tt)[U], int yb) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int y = yb + u * YR;
        const int yc = y < ny ? y : ny - 1;
        aa[u] = __builtin_nontemporal_load(a0 + col + (size_t)yc * nz);
        tt[u] = __builtin_nontemporal_load(ts + col + (size_t)yc * nz);
      }
    };
    for (int y0 = yr; y0 < ny; y0 += 2 * YR * U) {       // ping-pong: no register copies
      fetch(an, tn_, y0 + YR * U);
#pragma unroll
      for (int u = 0; u < U; ++u) cell((y0 + u * YR < ny) ? a[u] : 0.0, t[u]);
      fetch(a, t, y0 + 2 * YR * U);
#pragma unroll
      for (int u = 0; u < U; ++u) cell((y0 + YR * U + u * YR < ny) ? an[u] : 0.0, tn_[u]);
    }
  } else {
  for (int y0 = yr; y0 < ny; y0 += YR * U) {
    double a[U], t[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int y = y0 + u * YR;
      const bool in = y < ny;
      a[u] = in ? __builtin_nontemporal_load(a0 + col + (size_t)y * nz) : 0.0;
      t[u] = in ? __builtin_nontemporal_load(ts + col + (size_t)y * nz) : 0.0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) cell(a[u], t[u]);
  }
  }
  __syncthreads();
  // flush, transposed: MT[idx][p]
  for (int i = threadIdx.x; i < TOT; i += BS) {
    const int idx = i / SL, s = i % SL;
    MT[(size_t)idx * npix + (size_t)(PAIR ? ((blockIdx.x % 8) * 2 + (blockIdx.x / 8) % 2 + 16 * (blockIdx.x / 16)) : blockIdx.x) * SL + s] = lds[i];
  }
}

template <int ET, int UI, int PP>
__global__ __launch_bounds__(256) void eval(const double* __restrict__ MT, size_t npix, int kn2,
                                            const double* __restrict__ W, double* __restrict__ out) {
  // PP pixels per thread (p, p + npix/PP ...), UI moment rows in flight per pixel
  const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t half = npix / PP;
  if (p >= half) return;
  double acc[PP][ET];
#pragma unroll
  for (int q = 0; q < PP; ++q)
#pragma unroll
    for (int e = 0; e < ET; ++e) acc[q][e] = 0.0;
  for (int i0 = 0; i0 < kn2; i0 += UI) {
    double m[PP][UI];
#pragma unroll
    for (int j = 0; j < UI; ++j)
#pragma unroll
      for (int q = 0; q < PP; ++q) m[q][j] = MT[(size_t)(i0 + j) * npix + p + q * half];
#pragma unroll
    for (int j = 0; j < UI; ++j) {
      const double* w = W + (size_t)(i0 + j) * ET;
#pragma unroll
      for (int e = 0; e < ET; ++e)
#pragma unroll
        for (int q = 0; q < PP; ++q) acc[q][e] = __builtin_fma(m[q][j], w[e], acc[q][e]);
    }
  }
#pragma unroll
  for (int q = 0; q < PP; ++q)
#pragma unroll
    for (int e = 0; e < ET; ++e) out[(size_t)e * npix + p + q * half] = acc[q][e];
}

template <int UI, int PP>
static void run_eval(const double* MT, size_t npix, int kn2, const double* W, double* out) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((eval<32, UI, PP>), dim3((unsigned)((npix / PP + 255) / 256)), dim3(256), 0, 0, MT, npix, kn2, W, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  printf("eval UI=%d PP=%d kn2=%d: %.3f ms\n", UI, PP, kn2, best);
}

template <int K, int N, int SL, int U, int BS, int JETS = 2, bool PF = false, bool PAIR = false>
static void run(const double* a0, const double* ts, int nx, int ny, int nz, double* MT, double* W, double* out) {
  const size_t npix = (size_t)nx * nz;
  const size_t shm = (size_t)JETS * K * N * SL * sizeof(double);
  CK(hipFuncSetAttribute((const void*)moments<K, N, SL, U, BS, JETS, PF, PAIR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f, best2 = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((moments<K, N, SL, U, BS, JETS, PF, PAIR>), dim3((unsigned)(npix / SL)), dim3(BS), shm, 0, a0, ts, ny, nz, 0.0, K / 5.0, MT, npix);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((eval<32, 1, 1>), dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, 0, MT, npix, JETS * K * N, W, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best2) best2 = ms;
  }
  CK(hipGetLastError());
  const double gb = (double)nx * ny * nz * 16 / 1e9;
  printf("BS=%d JETS=%d PF=%d PAIR=%d ", BS, JETS, (int)PF, (int)PAIR); printf("K=%d N=%d SL=%d U=%d  LDS %zu KB: moments %.3f ms (%.0f GB/s of a0+ts)  eval(32 epochs) %.3f ms\n",
         K, N, SL, U, shm / 1024, best, gb / best * 1e3, best2);
}


// mixed precision: orders n < NS are accumulated in f64, the higher ones (whose coefficients
// decay geometrically) in f32 -- half the LDS cycles and bytes each
template <int K, int N, int NS, int SL, int U, int BS>
__global__ __launch_bounds__(BS) void moments_mixed(const double* __restrict__ a0, const double* __restrict__ ts,
                                                    int ny, int nz, double s0, double inv_h,
                                                    double* __restrict__ MT, size_t npix) {
  extern __shared__ double lds[];          // [2][K][NS][SL] doubles, then [2][K][N-NS][SL] floats
  constexpr int TOTD = 2 * K * NS * SL, TOTF = 2 * K * (N - NS) * SL;
  float* ldsf = reinterpret_cast<float*>(lds + TOTD);
  for (int i = threadIdx.x; i < TOTD; i += BS) lds[i] = 0.0;
  for (int i = threadIdx.x; i < TOTF; i += BS) ldsf[i] = 0.0f;
  __syncthreads();
  const int sl = threadIdx.x % SL, yr = threadIdx.x / SL;
  constexpr int YR = BS / SL;
  const size_t p = (size_t)blockIdx.x * SL + sl;
  const size_t x = p / nz;
  const int z = (int)(p - x * nz);
  const size_t col = x * (size_t)ny * nz + z;
  auto cell = [&](double av, double tv) __attribute__((always_inline)) {
      const bool red = av < 0.0;
      const double am = fmax(fabs(av), 0.0);
      const double w = (tv - s0) * inv_h;
      double kf = floor(w);
      kf = fmin(fmax(kf, 0.0), (double)(K - 1));
      const double xi = 2.0 * (w - kf) - 1.0;
      const int bin = (red ? 0 : K) + (int)kf;
      double* based = lds + (bin * NS) * SL + sl;
      float* basef = ldsf + (bin * (N - NS)) * SL + sl;
      double tm = 1.0, tc = xi;
      atomicAdd(based, am);
      atomicAdd(based + SL, am * tc);
      const double x2 = 2.0 * xi;
#pragma unroll
      for (int n = 2; n < N; ++n) {
        const double tn = __builtin_fma(x2, tc, -tm);
        tm = tc; tc = tn;
        if (n < NS) atomicAdd(based + n * SL, am * tn);
        else atomicAdd(basef + (n - NS) * SL, (float)(am * tn));
      }
  };
  double a[U], t[U], an[U], tn_[U];
  auto fetch = [&](double (&aa)[U], double (&tt)[U], int yb) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int y = yb + u * YR;
      const int yc = y < ny ? y : ny - 1;
      aa[u] = __builtin_nontemporal_load(a0 + col + (size_t)yc * nz);
      tt[u] = __builtin_nontemporal_load(ts + col + (size_t)yc * nz);
    }
  };
  fetch(a, t, yr);
  for (int y0 = yr; y0 < ny; y0 += 2 * YR * U) {
    fetch(an, tn_, y0 + YR * U);
#pragma unroll
    for (int u = 0; u < U; ++u) cell((y0 + u * YR < ny) ? a[u] : 0.0, t[u]);
    fetch(a, t, y0 + 2 * YR * U);
#pragma unroll
    for (int u = 0; u < U; ++u) cell((y0 + YR * U + u * YR < ny) ? an[u] : 0.0, tn_[u]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < TOTD; i += BS) {
    const int idx = i / SL, s = i % SL;
    MT[(size_t)idx * npix + (size_t)blockIdx.x * SL + s] = lds[i];
  }
  float* MTF = reinterpret_cast<float*>(MT + (size_t)2 * K * NS * npix);
  for (int i = threadIdx.x; i < TOTF; i += BS) {
    const int idx = i / SL, s = i % SL;
    MTF[(size_t)idx * npix + (size_t)blockIdx.x * SL + s] = ldsf[i];
  }
}

template <int K, int N, int NS, int SL, int U, int BS>
static void run_mixed(const double* a0, const double* ts, int nx, int ny, int nz, double* MT) {
  const size_t npix = (size_t)nx * nz;
  const size_t shm = (size_t)2 * K * NS * SL * 8 + (size_t)2 * K * (N - NS) * SL * 4;
  CK(hipFuncSetAttribute((const void*)moments_mixed<K, N, NS, SL, U, BS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((moments_mixed<K, N, NS, SL, U, BS>), dim3((unsigned)(npix / SL)), dim3(BS), shm, 0, a0, ts, ny, nz, 0.0, K / 5.0, MT, npix);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  CK(hipGetLastError());
  printf("mixed K=%d N=%d (f64 orders %d, f32 orders %d) SL=%d U=%d BS=%d LDS %zu KB: moments %.3f ms\n", K, N, NS, N - NS, SL, U, BS, shm / 1024, best);
}

int main() {
  const int nx = 512, ny = 4096, nz = 512;
  const size_t n = (size_t)nx * ny * nz, npix = (size_t)nx * nz;
  double *a0, *ts, *MT, *W, *out;
  CK(hipMalloc(&a0, n * 8)); CK(hipMalloc(&ts, n * 8));
  CK(hipMalloc(&MT, (size_t)2 * 32 * 20 * npix * 8));
  CK(hipMalloc(&W, (size_t)2 * 32 * 20 * 32 * 8)); CK(hipMemset(W, 0, (size_t)2 * 32 * 20 * 32 * 8));
  CK(hipMalloc(&out, 32 * npix * 8));
  hipLaunchKernelGGL(fill, dim3(8192), dim3(256), 0, 0, a0, ts, n, nz);
  CK(hipDeviceSynchronize());
  run<53, 12, 16, 4, 1024, 2, true>(a0, ts, nx, ny, nz, MT, W, out);
  run<53, 12, 8, 4, 512, 2, true>(a0, ts, nx, ny, nz, MT, W, out);
  run<53, 12, 8, 4, 512, 2, true, true>(a0, ts, nx, ny, nz, MT, W, out);
  run<53, 12, 8, 8, 512, 2, true, true>(a0, ts, nx, ny, nz, MT, W, out);
  run<53, 12, 8, 4, 256, 2, true, true>(a0, ts, nx, ny, nz, MT, W, out);
  run<39, 16, 8, 4, 512, 2, true, true>(a0, ts, nx, ny, nz, MT, W, out);
  run_mixed<53, 12, 12, 16, 4, 1024>(a0, ts, nx, ny, nz, MT);
  run_mixed<53, 12, 6, 16, 4, 1024>(a0, ts, nx, ny, nz, MT);
  run_mixed<53, 12, 4, 16, 4, 1024>(a0, ts, nx, ny, nz, MT);
  run_mixed<64, 12, 5, 16, 4, 1024>(a0, ts, nx, ny, nz, MT);
  run_mixed<53, 12, 5, 16, 4, 512>(a0, ts, nx, ny, nz, MT);
  // sanity: total of the zeroth moments == sum |a0|
  std::vector<double> h(npix);
  CK(hipMemcpy(h.data(), MT, npix * 8, hipMemcpyDeviceToHost));
  printf("M[0][0][0][p=0] = %.6f\n", h[0]);
  return 0;
}
