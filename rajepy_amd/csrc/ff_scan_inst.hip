// One slice of the K1 kernel family per translation unit (compiled RJP_INST = 0..4 by the
// Makefile, in parallel): the slices share no device code, and the family is what makes the
// library's build time.
//   0  f64 storage, tau layout (a0, ts [+ em0])          -- the Gaunt mode is baked into a0
//   1  f64 storage, compact layout, scalar Gaunt factor
//   2  f64 storage, compact layout, power-law Gaunt factor
//   3  f64 storage, wide layout, both Gaunt modes
//   4  f32 storage, compact and wide layouts, both Gaunt modes
#include "ff_scan_kernels.h"

#ifndef RJP_INST
#error "compile with -DRJP_INST=0..4"
#endif

namespace rjp {

#define RJP_TILE_ARGS fl, b, bursts, t, un, et, nsplit, ylen, ws, want_em, st

#if RJP_INST == 0
hipError_t scan_f64_tau(int vec, const rjp_fields* fl, const BurstsDev& b, bool bursts,
                        const double* t, const UnifDev& un, int et, int nsplit, int ylen,
                        double* ws, bool want_em, hipStream_t st) {
  return vec == 2 ? dispatch_et<double, 2, RJP_GFF_SCALAR, LAY_TAU>(RJP_TILE_ARGS)
                  : dispatch_et<double, 1, RJP_GFF_SCALAR, LAY_TAU>(RJP_TILE_ARGS);
}
#elif RJP_INST == 1
hipError_t scan_f64_cmp_scalar(int vec, const rjp_fields* fl, const BurstsDev& b, bool bursts,
                               const double* t, const UnifDev& un, int et, int nsplit, int ylen,
                               double* ws, bool want_em, hipStream_t st) {
  return vec == 2 ? dispatch_et<double, 2, RJP_GFF_SCALAR, LAY_CMP>(RJP_TILE_ARGS)
                  : dispatch_et<double, 1, RJP_GFF_SCALAR, LAY_CMP>(RJP_TILE_ARGS);
}
#elif RJP_INST == 2
hipError_t scan_f64_cmp_plaw(int vec, const rjp_fields* fl, const BurstsDev& b, bool bursts,
                             const double* t, const UnifDev& un, int et, int nsplit, int ylen,
                             double* ws, bool want_em, hipStream_t st) {
  return vec == 2 ? dispatch_et<double, 2, RJP_GFF_POWERLAW, LAY_CMP>(RJP_TILE_ARGS)
                  : dispatch_et<double, 1, RJP_GFF_POWERLAW, LAY_CMP>(RJP_TILE_ARGS);
}
#elif RJP_INST == 3
hipError_t scan_f64_wide(int vec, const rjp_fields* fl, const BurstsDev& b, bool bursts, int mode,
                         const double* t, const UnifDev& un, int et, int nsplit, int ylen,
                         double* ws, bool want_em, hipStream_t st) {
  if (mode == RJP_GFF_SCALAR)
    return vec == 2 ? dispatch_et<double, 2, RJP_GFF_SCALAR, LAY_WIDE>(RJP_TILE_ARGS)
                    : dispatch_et<double, 1, RJP_GFF_SCALAR, LAY_WIDE>(RJP_TILE_ARGS);
  return vec == 2 ? dispatch_et<double, 2, RJP_GFF_POWERLAW, LAY_WIDE>(RJP_TILE_ARGS)
                  : dispatch_et<double, 1, RJP_GFF_POWERLAW, LAY_WIDE>(RJP_TILE_ARGS);
}
#elif RJP_INST == 4
template <int VEC>
static hipError_t f32_slice(int lay, int mode, const rjp_fields* fl, const BurstsDev& b,
                            bool bursts, const double* t, const UnifDev& un, int et, int nsplit,
                            int ylen, double* ws, bool want_em, hipStream_t st) {
  if (lay == LAY_CMP)
    return mode == RJP_GFF_SCALAR ? dispatch_et<float, VEC, RJP_GFF_SCALAR, LAY_CMP>(RJP_TILE_ARGS)
                                  : dispatch_et<float, VEC, RJP_GFF_POWERLAW, LAY_CMP>(RJP_TILE_ARGS);
  return mode == RJP_GFF_SCALAR ? dispatch_et<float, VEC, RJP_GFF_SCALAR, LAY_WIDE>(RJP_TILE_ARGS)
                                : dispatch_et<float, VEC, RJP_GFF_POWERLAW, LAY_WIDE>(RJP_TILE_ARGS);
}
hipError_t scan_f32(int vec, int lay, const rjp_fields* fl, const BurstsDev& b, bool bursts,
                    int mode, const double* t, const UnifDev& un, int et, int nsplit, int ylen,
                    double* ws, bool want_em, hipStream_t st) {
  return vec == 4 ? f32_slice<4>(lay, mode, RJP_TILE_ARGS) : f32_slice<1>(lay, mode, RJP_TILE_ARGS);
}
#else
#error "RJP_INST out of range"
#endif

}  // namespace rjp
