// Pure streaming-read probe: what does HBM deliver to the simplest possible kernels on this
// device?  hipcc --offload-arch=gfx950 -O3 -o stream_probe stream_probe.hip ; ./stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2 __attribute__((ext_vector_type(2)));

// (a) grid-stride over ONE contiguous array, 16 B per lane per load, UNR loads in flight
template <int UNR, bool NT>
__global__ __launch_bounds__(256) void lin(const d2* __restrict__ a, size_t n2, double* out) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t step = (size_t)gridDim.x * 256;
  double acc = 0.0;
  for (; i + (UNR - 1) * step < n2; i += UNR * step) {
    d2 v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u)
      v[u] = NT ? __builtin_nontemporal_load(a + i + u * step) : a[i + u * step];
#pragma unroll
    for (int u = 0; u < UNR; ++u) acc += v[u].x + v[u].y;
  }
  if (acc == 123.456) out[0] = acc;
}

// (b) block-contiguous: each block owns one contiguous chunk and walks it in 4-KiB rows
template <int UNR>
__global__ __launch_bounds__(256) void chunked(const d2* __restrict__ a, size_t rows_per_block,
                                               double* out) {
  const d2* p = a + (size_t)blockIdx.x * rows_per_block * 256 + threadIdx.x;
  double acc = 0.0;
  for (size_t r = 0; r + UNR <= rows_per_block; r += UNR) {
    d2 v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) v[u] = __builtin_nontemporal_load(p + (r + u) * 256);
#pragma unroll
    for (int u = 0; u < UNR; ++u) acc += v[u].x + v[u].y;
  }
  if (acc == 123.456) out[0] = acc;
}

__global__ void fill(double* a, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t step = (size_t)gridDim.x * 256;
  for (; i < n; i += step) {
    unsigned long long x = i * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull;
    a[i] = 1.0 + (double)(x >> 11) * (1.0 / 9007199254740992.0);
  }
}

// (c) three arrays walked together, block-contiguous rows (the access pattern of K1)
template <int UNR>
__global__ __launch_bounds__(256) void three(const d2* __restrict__ a, const d2* __restrict__ b,
                                             const d2* __restrict__ c, size_t rows_per_block,
                                             double* out) {
  const size_t off = (size_t)blockIdx.x * rows_per_block * 256 + threadIdx.x;
  double acc = 0.0;
  for (size_t r = 0; r + UNR <= rows_per_block; r += UNR) {
    d2 va[UNR], vb[UNR], vc[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      va[u] = __builtin_nontemporal_load(a + off + (r + u) * 256);
      vb[u] = __builtin_nontemporal_load(b + off + (r + u) * 256);
      vc[u] = __builtin_nontemporal_load(c + off + (r + u) * 256);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) acc += va[u].x + va[u].y + vb[u].x * vb[u].y + vc[u].x - vc[u].y;
  }
  if (acc == 123.456) out[0] = acc;
}

// (d) as (c) plus NALU dependent FP64 FMAs per loaded element pair (two cells): how much
// vector-ALU work can ride along with the three streams before the read rate drops?
template <int UNR, int NALU>
__global__ __launch_bounds__(256) void three_alu(const d2* __restrict__ a, const d2* __restrict__ b,
                                                 const d2* __restrict__ c, size_t rows_per_block,
                                                 double* out) {
  const size_t off = (size_t)blockIdx.x * rows_per_block * 256 + threadIdx.x;
  double acc = 0.0;
  for (size_t r = 0; r + UNR <= rows_per_block; r += UNR) {
    d2 va[UNR], vb[UNR], vc[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      va[u] = __builtin_nontemporal_load(a + off + (r + u) * 256);
      vb[u] = __builtin_nontemporal_load(b + off + (r + u) * 256);
      vc[u] = __builtin_nontemporal_load(c + off + (r + u) * 256);
    }
    // NALU FMAs per CELL (a lane holds two cells per row), chains of the 2*UNR cells interleave
    double p[2 * UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) { p[2 * u] = vc[u].x; p[2 * u + 1] = vc[u].y; }
#pragma unroll
    for (int k = 0; k < NALU; ++k)
#pragma unroll
      for (int q = 0; q < 2 * UNR; ++q) p[q] = __builtin_fma(p[q], 0.999999 + 1e-9 * k, 1e-7);
#pragma unroll
    for (int u = 0; u < UNR; ++u)
      acc += va[u].x * p[2 * u] + va[u].y * p[2 * u + 1] + vb[u].x * vb[u].y;
  }
  if (acc == 123.456) out[0] = acc;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
  const size_t nbytes = (size_t)24 << 30;                    // 24 GiB
  const size_t n2 = nbytes / 16;
  d2* a; double* out;
  CK(hipMalloc(&a, nbytes)); CK(hipMalloc(&out, 8));
  const char* zero = getenv("PROBE_ZERO");
  if (zero) CK(hipMemset(a, 0, nbytes));
  else hipLaunchKernelGGL(fill, dim3(8192), dim3(256), 0, 0, (double*)a, nbytes / 8);
  CK(hipDeviceSynchronize());
  printf("data: %s\n", zero ? "zeros" : "pseudo-random doubles in [1,2)");
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](const char* name, auto launch) {
    launch(); CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("%-44s %7.3f ms  %6.0f GB/s\n", name, best, nbytes / best / 1e6);
    return 0;
  };
  for (int blocks : {2048, 8192, 32768}) {
    char nm[96];
    snprintf(nm, 96, "linear grid-stride, unroll 4, nt, %d blocks", blocks);
    time(nm, [&] { hipLaunchKernelGGL((lin<4, true>), dim3(blocks), dim3(256), 0, 0, a, n2, out); });
    snprintf(nm, 96, "linear grid-stride, unroll 8, nt, %d blocks", blocks);
    time(nm, [&] { hipLaunchKernelGGL((lin<8, true>), dim3(blocks), dim3(256), 0, 0, a, n2, out); });
    snprintf(nm, 96, "linear grid-stride, unroll 4, cached, %d blocks", blocks);
    time(nm, [&] { hipLaunchKernelGGL((lin<4, false>), dim3(blocks), dim3(256), 0, 0, a, n2, out); });
  }
  for (int blocks : {4096, 16384, 49152}) {
    const size_t rows = n2 / 256 / blocks;
    char nm[96];
    snprintf(nm, 96, "block-contiguous chunks, unroll 4, %d blocks", blocks);
    time(nm, [&] { hipLaunchKernelGGL((chunked<4>), dim3(blocks), dim3(256), 0, 0, a, rows, out); });
    snprintf(nm, 96, "block-contiguous chunks, unroll 12, %d blocks", blocks);
    time(nm, [&] { hipLaunchKernelGGL((chunked<12>), dim3(blocks), dim3(256), 0, 0, a, rows, out); });
  }
  for (int blocks : {4096, 16384, 49152}) {
    const size_t n2f = n2 / 3;                                // three 8-GiB arrays
    const size_t rows = n2f / 256 / blocks;
    char nm[96];
    snprintf(nm, 96, "three arrays together, unroll 4, %d blocks", blocks);
    time(nm, [&] { hipLaunchKernelGGL((three<4>), dim3(blocks), dim3(256), 0, 0, a, a + n2f, a + 2 * n2f, rows, out); });
    snprintf(nm, 96, "three arrays together, unroll 2, %d blocks", blocks);
    time(nm, [&] { hipLaunchKernelGGL((three<2>), dim3(blocks), dim3(256), 0, 0, a, a + n2f, a + 2 * n2f, rows, out); });
  }
  {
    const int blocks = 4096;
    const size_t n2f = n2 / 3;
    const size_t rows = n2f / 256 / blocks;
    time("three arrays + 10 FMA/cell, unroll 4, 4096 blocks", [&] { hipLaunchKernelGGL((three_alu<4, 10>), dim3(blocks), dim3(256), 0, 0, a, a + n2f, a + 2 * n2f, rows, out); });
    time("three arrays + 20 FMA/cell, unroll 4, 4096 blocks", [&] { hipLaunchKernelGGL((three_alu<4, 20>), dim3(blocks), dim3(256), 0, 0, a, a + n2f, a + 2 * n2f, rows, out); });
    time("three arrays + 40 FMA/cell, unroll 4, 4096 blocks", [&] { hipLaunchKernelGGL((three_alu<4, 40>), dim3(blocks), dim3(256), 0, 0, a, a + n2f, a + 2 * n2f, rows, out); });
    time("three arrays + 75 FMA/cell, unroll 4, 4096 blocks", [&] { hipLaunchKernelGGL((three_alu<4, 75>), dim3(blocks), dim3(256), 0, 0, a, a + n2f, a + 2 * n2f, rows, out); });
    time("three arrays + 110 FMA/cell, unroll 4, 4096 blocks", [&] { hipLaunchKernelGGL((three_alu<4, 110>), dim3(blocks), dim3(256), 0, 0, a, a + n2f, a + 2 * n2f, rows, out); });
  }
  return 0;
}
