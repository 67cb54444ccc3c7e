// Device-side helpers shared by the gfx950 kernels of librjprt.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rjprt.h"

#define RJP_WAVE 64

// ---- vector loads: 16 B per lane -------------------------------------------------------
template <typename T> struct VecOf;
template <> struct VecOf<double> { using type = double2; static constexpr int N = 2; };
template <> struct VecOf<float> { using type = float4; static constexpr int N = 4; };

template <typename T, int V> struct Pack { double v[V]; };

__device__ __forceinline__ void load_vec(const double* __restrict__ p, double (&out)[2]) {
  double2 t = *reinterpret_cast<const double2*>(p);
  out[0] = t.x; out[1] = t.y;
}
__device__ __forceinline__ void load_vec(const float* __restrict__ p, double (&out)[4]) {
  float4 t = *reinterpret_cast<const float4*>(p);
  out[0] = (double)t.x; out[1] = (double)t.y; out[2] = (double)t.z; out[3] = (double)t.w;
}
__device__ __forceinline__ void load_vec(const double* __restrict__ p, double (&out)[1]) {
  out[0] = *p;
}
__device__ __forceinline__ void load_vec(const float* __restrict__ p, double (&out)[1]) {
  out[0] = (double)*p;
}

// ---- burst factor chi(t) (classes.py:442-448, 866-868) ---------------------------------
// Kernel-argument copy of rjp_bursts (lives in SGPRs / scalar cache).
struct BurstsDev {
  int n[2];
  double t0[2][RJP_MAX_BURSTS];
  double amp_rel[2][RJP_MAX_BURSTS];
  double inv2s2[2][RJP_MAX_BURSTS];
};

// exp(x) for x <= 0, relative error < 1e-14 on [-708, 0] (clamped below: ~1e-308).
// Cody-Waite reduction + degree-11 Taylor polynomial of exp(r), |r| <= ln2/2.  About 19 DP
// instructions, no denormal/overflow paths (the argument is a Gaussian exponent).
__device__ __forceinline__ double exp_nonpos(double x) {
  const double L2E = 1.4426950408889634074;
  const double LN2_HI = 6.93147180369123816490e-01;
  const double LN2_LO = 1.90821492927058770002e-10;
  x = fmax(x, -708.0);
  double kd = __builtin_rint(x * L2E);
  double r = __builtin_fma(-kd, LN2_HI, x);
  r = __builtin_fma(-kd, LN2_LO, r);
  double p = 2.505210838544172e-08;                // 1/11!
  p = __builtin_fma(p, r, 2.755731922398589e-07);  // 1/10!
  p = __builtin_fma(p, r, 2.7557319223985893e-06); // 1/9!
  p = __builtin_fma(p, r, 2.48015873015873e-05);   // 1/8!
  p = __builtin_fma(p, r, 1.984126984126984e-04);  // 1/7!
  p = __builtin_fma(p, r, 1.388888888888889e-03);  // 1/6!
  p = __builtin_fma(p, r, 8.333333333333333e-03);  // 1/5!
  p = __builtin_fma(p, r, 4.1666666666666664e-02); // 1/4!
  p = __builtin_fma(p, r, 1.6666666666666666e-01); // 1/3!
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  return __builtin_ldexp(p, (int)kd);
}

// chi for one jet (wave-uniform loop count; parameters come from SGPRs)
__device__ __forceinline__ double chi_jet(const BurstsDev& b, int jet, double tl) {
  double chi = 1.0;
  const int nb = b.n[jet];
  for (int i = 0; i < nb; ++i) {
    double d = tl - b.t0[jet][i];
    double arg = -(d * d) * b.inv2s2[jet][i];
    chi = __builtin_fma(b.amp_rel[jet][i], exp_nonpos(arg), chi);
  }
  return chi;
}

__device__ __forceinline__ double chi_cell(const BurstsDev& b, bool red, double tl) {
  // Both jets' loops are wave-uniform in trip count; a wave that straddles the red/blue
  // plane executes both, every other wave exactly one.
  double chi;
  if (red) chi = chi_jet(b, 0, tl); else chi = chi_jet(b, 1, tl);
  return chi;
}

// T^-1.5 with an f32 rsqrt seed + one fp64 Newton step (rel. err < 1e-13); exact-ish slow
// path outside the f32 exponent range and for T == 0 / inf / negative.
__device__ __forceinline__ double pow_m1p5(double T) {
  if (T > 1e-30 && T < 1e30) {
    double y = (double)__builtin_amdgcn_rsqf((float)T);
    double h = 0.5 * T;
    y = y * __builtin_fma(-h * y, y, 1.5);
    return y * y * y;
  }
  return 1.0 / (T * __builtin_sqrt(T));   // NaN for T<0 or NaN, inf for 0, 0 for inf
}

__device__ __forceinline__ bool signbit_d(double v) {
  return (__double_as_longlong(v) < 0);
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
