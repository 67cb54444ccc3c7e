// Feasibility probe (round 5): the launch-time moment pass WITHOUT f64
// LDS atomics.  The shipped pass (ff_moments.hip) adds N moments per cell into LDS accumulators
// (12.6 LDS instructions per cell, LDS-bound: 4.3 ms for 1.07e9 cells).  Here a workgroup of
// 1024 threads owns 16 z-adjacent sightlines as before, but per chunk of R rows it
//   1. counts the chunk's cells per list (sightline, jet, bin) -- ONE u32 LDS atomic per cell,
//      its return value is the cell's rank in its list;
//   2. prefix-sums the 16 x 2K counters into list offsets;
//   3. scatters (|a0|, xi) into a packed LDS array ordered by list -- one 16-byte LDS write;
//   4. every thread OWNS two fixed lists for the whole pass and adds their cells' Chebyshev
//      terms into REGISTER accumulators (2N + 1 FP64 instructions per cell, one 16-byte LDS read).
// 3 LDS operations per cell instead of 12.6 -- but the lists of a chunk are short (R = 512:
// 4.8 cells on average) and Poisson-distributed, and a wave runs as long as its longest list.
//   hipcc --offload-arch=gfx950 -O3 -o moments_queue_probe moments_queue_probe.hip
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

// reference: the shipped scheme (f64 LDS atomics), for the same data -- timing and checksum
template <int K, int N>
__global__ __launch_bounds__(1024) void moments_atomic(const double* __restrict__ a0, const double* __restrict__ ts,
                                                       int ny, int nz, double s0, double inv_h,
                                                       double* __restrict__ MT, size_t npix) {
  constexpr int SL = 16, BS = 1024, U = 4, YR = BS / SL;
  extern __shared__ double lds[];
  constexpr int TOT = 2 * K * N * SL;
  for (int i = threadIdx.x; i < TOT; i += BS) lds[i] = 0.0;
  __syncthreads();
  const int sl = threadIdx.x % SL, yr = threadIdx.x / SL;
  const unsigned per = gridDim.x / 8;
  const unsigned tile = blockIdx.x < 8 * per ? (blockIdx.x % 8) * per + blockIdx.x / 8 : blockIdx.x;
  const size_t p = (size_t)tile * SL + sl;
  const size_t x = p / nz;
  const int z = (int)(p - x * nz);
  const size_t col = x * (size_t)ny * nz + z;
  auto cell = [&](double av, double tv) __attribute__((always_inline)) {
    const bool red = av < 0.0;
    const double am = fabs(av);
    const double w = (tv - s0) * inv_h;
    double kf = fmin(fmax(floor(w), 0.0), (double)(K - 1));
    const double xi = 2.0 * (w - kf) - 1.0;
    double* base = lds + (((red ? 0 : K) + (int)kf) * N) * SL + sl;
    double tm = am, tc = am * xi;
    atomicAdd(base, tm);
    atomicAdd(base + SL, tc);
    const double x2 = 2.0 * xi;
#pragma unroll
    for (int n = 2; n < N; ++n) {
      const double tn = __builtin_fma(x2, tc, -tm);
      tm = tc; tc = tn;
      atomicAdd(base + n * SL, tn);
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
  for (int i = threadIdx.x; i < TOT; i += BS) {
    const int idx = i / SL, s = i % SL;
    MT[(size_t)idx * npix + (size_t)tile * SL + s] = lds[i];
  }
}

struct alignas(16) d2 { double x, y; };

// the queue scheme.  R rows per chunk, CPT = R / 64 cells per thread and chunk.
template <int K, int N, int R>
__global__ __launch_bounds__(1024) void moments_queue(const double* __restrict__ a0, const double* __restrict__ ts,
                                                      int ny, int nz, double s0, double inv_h,
                                                      double* __restrict__ MT, size_t npix) {
  constexpr int SL = 16, BS = 1024, YR = BS / SL, CPT = R / YR;
  constexpr int L = 2 * K * SL;                    // lists of a workgroup
  constexpr int LP = 2048;                         // two lists per thread (L <= 2048)
  static_assert(L <= LP, "more than two lists per thread");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  d2* cells = reinterpret_cast<d2*>(smem);                          // [SL * R]
  unsigned* cnt = reinterpret_cast<unsigned*>(smem + (size_t)SL * R * 16);   // [LP]
  unsigned* off = cnt + LP;                                          // [LP]
  unsigned* wtot = off + LP;                                         // [2][16]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int sl = tid % SL, yr = tid / SL;
  const unsigned per = gridDim.x / 8;
  const unsigned tile = blockIdx.x < 8 * per ? (blockIdx.x % 8) * per + blockIdx.x / 8 : blockIdx.x;
  const size_t p = (size_t)tile * SL + sl;
  const size_t x = p / nz;
  const int z = (int)(p - x * nz);
  const size_t col = x * (size_t)ny * nz + z;
  // the two lists this thread owns for the whole pass: l0 = tid, l1 = tid + 1024
  double M0[N], M1[N];
#pragma unroll
  for (int n = 0; n < N; ++n) { M0[n] = 0.0; M1[n] = 0.0; }
  double a[CPT], t[CPT];
  auto fetch = [&](int ybase) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < CPT; ++u) {
      const int y = ybase + yr + u * YR;
      const int yc = y < ny ? y : ny - 1;
      a[u] = __builtin_nontemporal_load(a0 + col + (size_t)yc * nz);
      t[u] = __builtin_nontemporal_load(ts + col + (size_t)yc * nz);
    }
  };
  fetch(0);
  for (int yb = 0; yb < ny; yb += R) {
    cnt[tid] = 0; cnt[tid + BS] = 0;
    __syncthreads();
    // 1. bin the chunk's cells: list id and rank inside the list ((a, t) become (|a0|, xi) in place)
    int lid[CPT]; unsigned pos[CPT];
#pragma unroll
    for (int u = 0; u < CPT; ++u) {
      const bool in = yb + yr + u * YR < ny;
      const bool red = a[u] < 0.0;
      a[u] = fabs(a[u]);
      const double w = (t[u] - s0) * inv_h;
      const double kf = fmin(fmax(floor(w), 0.0), (double)(K - 1));
      t[u] = 2.0 * (w - kf) - 1.0;
      lid[u] = (((red ? 0 : K) + (int)kf) * SL) + sl;
      pos[u] = (in && a[u] > 0.0) ? atomicAdd(&cnt[lid[u]], 1u) : 0xffffffffu;
    }
    __syncthreads();
    // 2. exclusive prefix over the lists in the order [0, 1024) then [1024, 2048)
    const unsigned c0 = cnt[tid], c1 = cnt[tid + BS];
    unsigned s0i = c0, s1i = c1;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned v0 = __shfl_up(s0i, d, 64), v1 = __shfl_up(s1i, d, 64);
      if (lane >= d) { s0i += v0; s1i += v1; }
    }
    if (lane == 63) { wtot[wv] = s0i; wtot[16 + wv] = s1i; }
    __syncthreads();
    unsigned b0 = 0, b1 = 0, tot0 = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const unsigned w0 = wtot[k], w1 = wtot[16 + k];
      if (k < wv) { b0 += w0; b1 += w1; }
      tot0 += w0;
    }
    const unsigned o0 = b0 + s0i - c0, o1 = tot0 + b1 + s1i - c1;
    off[tid] = o0; off[tid + BS] = o1;
    __syncthreads();
    // 3. scatter into the packed array
#pragma unroll
    for (int u = 0; u < CPT; ++u)
      if (pos[u] != 0xffffffffu) { d2 c; c.x = a[u]; c.y = t[u]; cells[off[lid[u]] + pos[u]] = c; }
    // next chunk's rows on their way while the owners work (the registers are free again)
    fetch(yb + R);
    __syncthreads();
    // 4. the owners: register moments of their two lists
    auto eat = [&](double (&M)[N], unsigned o, unsigned c) __attribute__((always_inline)) {
      for (unsigned i = 0; __any(i < c); ++i) {
        if (i < c) {
          const d2 q = cells[o + i];
          double tm = q.x, tc = q.x * q.y;
          const double x2 = q.y + q.y;
          M[0] += tm; M[1] += tc;
#pragma unroll
          for (int n = 2; n < N; ++n) {
            const double tn = __builtin_fma(x2, tc, -tm);
            tm = tc; tc = tn;
            M[n] += tn;
          }
        }
      }
    };
    eat(M0, o0, c0);
    eat(M1, o1, c1);
    __syncthreads();
  }
  // flush: list l = key * SL + s  ->  MT[(key * N + n)][tile * SL + s]
  {
    const int l0 = tid, l1 = tid + BS;
    if (l0 < L) {
#pragma unroll
      for (int n = 0; n < N; ++n) MT[(size_t)((l0 / SL) * N + n) * npix + (size_t)tile * SL + (l0 % SL)] = M0[n];
    }
    if (l1 < L) {
#pragma unroll
      for (int n = 0; n < N; ++n) MT[(size_t)((l1 / SL) * N + n) * npix + (size_t)tile * SL + (l1 % SL)] = M1[n];
    }
  }
}

template <typename F>
static float timeit(F&& launch, int reps = 4) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(e0));
    launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  CK(hipGetLastError());
  return best;
}

template <int K, int N>
static void run_atomic(const double* a0, const double* ts, int nx, int ny, int nz, double* MT) {
  const size_t npix = (size_t)nx * nz;
  const size_t shm = (size_t)2 * K * N * 16 * sizeof(double);
  CK(hipFuncSetAttribute((const void*)moments_atomic<K, N>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  const float ms = timeit([&] { hipLaunchKernelGGL((moments_atomic<K, N>), dim3((unsigned)(npix / 16)), dim3(1024), shm, 0, a0, ts, ny, nz, 0.0, K / 5.0, MT, npix); });
  printf("atomic  K=%d N=%d          LDS %3zu KB: %.3f ms (%.0f GB/s of a0+ts)\n", K, N, shm / 1024, ms, (double)nx * ny * nz * 16 / 1e6 / ms);
}

template <int K, int N, int R>
static void run_queue(const double* a0, const double* ts, int nx, int ny, int nz, double* MT) {
  const size_t npix = (size_t)nx * nz;
  const size_t shm = (size_t)16 * R * 16 + (2 * 2048 + 32) * sizeof(unsigned);
  CK(hipFuncSetAttribute((const void*)moments_queue<K, N, R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  const float ms = timeit([&] { hipLaunchKernelGGL((moments_queue<K, N, R>), dim3((unsigned)(npix / 16)), dim3(1024), shm, 0, a0, ts, ny, nz, 0.0, K / 5.0, MT, npix); });
  printf("queue   K=%d N=%d R=%4d   LDS %3zu KB: %.3f ms (%.0f GB/s of a0+ts)\n", K, N, R, shm / 1024, ms, (double)nx * ny * nz * 16 / 1e6 / ms);
}

static double checksum(const double* MT, size_t n) {
  std::vector<double> h(n);
  CK(hipMemcpy(h.data(), MT, n * 8, hipMemcpyDeviceToHost));
  double s = 0.0;
  for (size_t i = 0; i < n; ++i) s += h[i] * (1.0 + (double)(i % 97) * 1e-3);
  return s;
}

int main() {
  const int nx = 512, ny = 4096, nz = 512;
  const size_t n = (size_t)nx * ny * nz, npix = (size_t)nx * nz;
  double *a0, *ts, *MT;
  CK(hipMalloc(&a0, n * 8)); CK(hipMalloc(&ts, n * 8));
  CK(hipMalloc(&MT, (size_t)2 * 64 * 12 * npix * 8));
  hipLaunchKernelGGL(fill, dim3(8192), dim3(256), 0, 0, a0, ts, n, nz);
  CK(hipDeviceSynchronize());
  const size_t rows5312 = (size_t)2 * 53 * 12;
  run_atomic<53, 12>(a0, ts, nx, ny, nz, MT);
  const double ref = checksum(MT, rows5312 * 4096);
  CK(hipMemset(MT, 0, rows5312 * npix * 8));
  run_queue<53, 12, 512>(a0, ts, nx, ny, nz, MT);
  const double got = checksum(MT, rows5312 * 4096);
  printf("checksum (first 4096 values of every moment row... flat prefix): atomic %.10e queue %.10e rel %.2e\n",
         ref, got, (got - ref) / ref);
  run_queue<53, 12, 256>(a0, ts, nx, ny, nz, MT);
  run_queue<64, 12, 512>(a0, ts, nx, ny, nz, MT);
  run_queue<64, 10, 512>(a0, ts, nx, ny, nz, MT);
  run_queue<39, 16, 512>(a0, ts, nx, ny, nz, MT);
  return 0;
}
