// Host-side interface between the translation units of librjprt (rjprt.hip = the C-ABI,
// ff_scan.hip, ff_scan_inst.hip x 5, fields.hip, rrl_scan.hip): launch wrappers and the small
// structs they exchange.  Nothing here is exported; the library's surface is include/rjprt.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <vector>

#include "rjp_device.h"

namespace rjp {

// ---- ff_scan.hip --------------------------------------------------------------------------
// The epoch tiling of one scan, decided on the host before anything is enqueued, so that the
// constants of bursts beyond RJP_SGPR_BURSTS (parameters + one q per tile) and the step tables
// of the uniform-epoch tiles can travel to the device in ONE small table:
// ext = [params: 6 * next][tile 0: q (2 * next), step table (2 * nbt * 16 or 0)][tile 1 ...] ...
struct ScanTile {
  int e0, et;
  UnifDev un;          // un.qext / un.atab are patched to the device table by ff_scan_run
  size_t q_off, a_off; // offsets of the tile's q and step table in ScanPlan::ext
  int nsplit, ylen;    // y-ranges of this tile's launch
};
struct ScanPlan {
  bool bursts = false;
  int vec = 1, next = 0;
  std::vector<ScanTile> tiles;
  std::vector<double> ext;       // host image of the overflow table (empty without overflow)
};

// which fields a scan kernel streams (scan_layout)
enum : int {
  LAY_WIDE = 0,            // nd, xi, temp, pf, ts
  LAY_CMP = 1,             // em0, temp, ts
  LAY_TAU = 2              // a0, ts (+ em0 with emission-measure maps); no T_avg sums
};

int ff_scan_vec(const rjp_fields* fl);
int scan_layout(const rjp_fields* fl, int mode, bool want_em);
bool tile_dma_ok(const rjp_fields* fl);
size_t ff_scan_workspace_bytes(int nx, int ny, int nz, int n_epochs);
void ff_scan_plan(const rjp_fields* fl, const rjp_bursts* hb, const double* epochs, int n_epochs,
                  int mode, bool want_em, ScanPlan& pl);
hipError_t ff_scan_run(const rjp_fields* fl, const rjp_bursts* hb, const ScanPlan& pl,
                       const double* d_ext, const double* epochs, int n_epochs, int mode,
                       double* sumA, double* em, double* tavg, double* ws, hipStream_t st);
hipError_t tavg_launch(const rjp_fields* fl, double* tavg, double* ws, hipStream_t st);
hipError_t y_bounds_launch(const rjp_fields* fl, int32_t* ylo, int32_t* yhi, hipStream_t st);
hipError_t occupied_launch(const int32_t* ylo, const int32_t* yhi, int64_t npix,
                           unsigned long long* d_count, hipStream_t st);
hipError_t ff_cells_launch(const rjp_fields* fl, const rjp_bursts* hb, const double* d_ext,
                           double time_s, int mode, const double* d_ctau, int nchan, double* out,
                           hipStream_t st);
size_t ff_maps_workspace_bytes(int64_t npix, int n_epochs, int n_chan);
hipError_t ff_maps_launch(const double* sumA, const double* tavg, int64_t npix, int n_epochs,
                          const double* d_ctau, const double* d_cflux, int n_chan, double* tau,
                          double* flux, double* ftot, double* part, hipStream_t st);
// out[row] = sum of part[row * nblk .. + nblk) in a fixed order
hipError_t sum_partials_launch(const double* part, int rows, int nblk, double* out,
                               hipStream_t st);

// fixed-order reduction of the y-range partials of one tile (ff_reduce_kernel)
hipError_t ff_reduce_launch(const double* ws, int nsplit, int et, int64_t npix, int e0,
                            double em_scale, double* sumA, double* em, double* tavg,
                            hipStream_t st);

// ---- ff_scan_tab.hip: single-epoch tau-layout scan with the burst factor from an LDS table -----
struct ChiPlan {
  bool ok = false;
  bool wide = false;              // the scan reads the five model fields, not the tau layout
  int ni = 0, n[2] = {0, 0};
  double lo = 0.0, inv_h = 0.0;
  std::vector<double> stage;      // Vandermonde inverse, nodes, burst parameters
  mutable bool attr_set = false;  // the kernels' dynamic-LDS limit was raised on this context's device
};
size_t chi_table_workspace_bytes(int64_t npix, int ny);
bool chi_table_plan(const rjp_fields* fl, const rjp_bursts* hb, const double* epochs, int n_epochs,
                    int mode, bool want_em, size_t work_bytes, ChiPlan& cp);
hipError_t chi_table_scan(const rjp_fields* fl, const ChiPlan& cp, const double* d_stage,
                          double t_epoch, int mode, double* sumA, double* em, double* tavg,
                          double* ws, size_t work_bytes, int* d_guard, hipStream_t st);

// ---- ff_moments.hip: epoch sweeps by launch-time moments --------------------------------------
#define RJP_MOM_MAX_IDX 1280      /* 2 jets x K bins x N Chebyshev moments <= this (160 KB of LDS
                                    for 16 sightlines); the (K, N) shapes: ff_moments.hip */
#define RJP_MOM_TILE 32          /* epochs per pass of the contraction */
#define RJP_MOM_MIN_EPOCHS 12    /* below: the epoch tiles are faster */
#define RJP_MOM_TOL 1e-11        /* worst relative error of the expansion the host accepts */
#define RJP_MOM_MAX_CAND 12     /* (bins, order) shapes tried in one table build */
#define RJP_MOM_NMAX 32         /* highest Chebyshev order of any shape */
struct MomCand {
  int K, N;
  int path;                      // 1 = LDS moments, 2 = launch-time-ordered layout
  size_t w_off;                  // doubles: this shape's tables in MomPlan::d_W
  size_t tab_off;                // doubles: its nodes x[N] and DCT matrix cs[N][N] in MomPlan::stage
};
struct MomPlan {
  bool ok = false;
  int path = 0;                  // 1 = LDS moments + contraction, 2 = launch-time-ordered layout
  double s0 = 0.0, inv_h = 0.0, worst = 0.0;
  int has_bursts[2] = {0, 0};
  int nchunk = 0;
  int K = 0, N = 0;              // the shape in use
  const double* d_Wsel = nullptr;   // its tables, [chunk][2 * K * N][TILE], on the device
  // device / pinned state of the table builder (owned; moments_release frees it)
  double* d_W = nullptr;
  size_t capW = 0;               // doubles
  unsigned long long* d_err = nullptr;
  unsigned long long* h_err = nullptr;
  // one table build: the staged host table (bursts, epochs, nodes + DCT matrices) and shapes
  std::vector<double> stage;
  std::vector<MomCand> cands;
  size_t w_total = 0, off_bursts[2] = {0, 0}, off_epochs = 0;
  double build_ms = 0.0;         // host wall time of the last table build (incl. its one sync)
  // the request the tables were built for (a sweep repeated with the same epochs reuses them)
  int key_E = -1, key_n[2] = {0, 0}, key_ltK = -1;
  bool key_ok = false;
  double key_lo = 0.0, key_hi = 0.0;
  std::vector<double> key_epochs, key_bursts;
  mutable bool attr_set[3] = {false, false, false};   // dynamic-LDS limit raised, per LDS shape
};
size_t moments_workspace_bytes(int64_t npix);
size_t moments_scan_workspace_bytes(int64_t npix);
// 0 = this scan keeps the epoch tiles, 1 = moment path with the tables already on the device,
// 2 = moment path after moments_build() (mp.stage has to be staged first)
int moments_plan(const rjp_fields* fl, const rjp_bursts* hb, const double* epochs, int n_epochs,
                 int mode, bool want_em, size_t work_bytes, MomPlan& mp);
// builds and checks the tables of every candidate shape on the device (one launch, one 64-byte
// copy back, ONE stream synchronisation) and selects the cheapest shape that passes; mp.ok says
// whether one did.  `d_stage` = device copy of mp.stage.
hipError_t moments_build(MomPlan& mp, const double* d_stage, hipStream_t st);
void moments_release(MomPlan& mp);
hipError_t moments_run(const rjp_fields* fl, const MomPlan& mp, int n_epochs,
                       double* sumA, double* ws, double* part, hipStream_t st,
                       const double* weights, double scale, int* d_guard, bool skip_pass = false);
// launch-time-ordered layout (ff_lt.hip)
#define RJP_LT_MAX_K 80
#define RJP_LT_MAX_EPOCHS 32     /* one fused pass serves one contraction tile */
size_t lt_rowoff_entries(int nx, int nz, int K);
hipError_t lt_count_launch(const rjp_fields* fl, int K, int32_t* d_rowoff, int* d_guard,
                           hipStream_t st);
hipError_t lt_fill_launch(const rjp_fields* fl, int K, const int32_t* d_rowoff, void* d_cells,
                          double* d_aux, hipStream_t st);
hipError_t lt_run(const rjp_fields* fl, const MomPlan& mp, int n_epochs, double* sumA, double* ws,
                  size_t work_bytes, hipStream_t st);
hipError_t field_range_launch(const void* d_field, int64_t n, int dtype, double* d_part,
                              hipStream_t st);
hipError_t range_check_launch(const void* d_field, int64_t n, int dtype, double lo, double hi,
                              int* d_flag, hipStream_t st);

// ---- ff_scan_inst.hip: one slice of the K1 kernel family per translation unit -------------
#define RJP_SCAN_SLICE_ARGS                                                                  \
  const rjp_fields *fl, const BurstsDev &b, bool bursts, const double *t, const UnifDev &un, \
      int et, int nsplit, int ylen, double *ws, bool want_em, hipStream_t st
hipError_t scan_f64_tau(int vec, RJP_SCAN_SLICE_ARGS);
hipError_t scan_f64_cmp_scalar(int vec, RJP_SCAN_SLICE_ARGS);
hipError_t scan_f64_cmp_plaw(int vec, RJP_SCAN_SLICE_ARGS);
hipError_t scan_f64_wide(int vec, const rjp_fields* fl, const BurstsDev& b, bool bursts, int mode,
                         const double* t, const UnifDev& un, int et, int nsplit, int ylen,
                         double* ws, bool want_em, hipStream_t st);
hipError_t scan_f32(int vec, int lay, const rjp_fields* fl, const BurstsDev& b, bool bursts,
                    int mode, const double* t, const UnifDev& un, int et, int nsplit, int ylen,
                    double* ws, bool want_em, hipStream_t st);

// ---- fields.hip ---------------------------------------------------------------------------
struct GeomDev {
  int nx, ny, nz, ccw;
  int ix0, nx_total;              // x-slab: rows [ix0, ix0+nx) of an nx_total-wide grid
  double cs;
  double ca, sa, cb, sb;          // derotation: alpha = inc - 90 (about x), beta = pa (about y)
  double ca2, sa2, cb2, sb2;      // velocity rotation: alpha = 90 - inc, beta = -pa
  double w_0, r_0, mr0, eps, R_1, R_2;
  double gm;                      // G * M_star * MSOL [SI]
  double v_lsr;
  double n_0, x_0, T_0, v_0;
  double q_n, q_x, q_T, q_v, qd_n, qd_x, qd_T, qd_v;
  double rb_frac;
  double ts_const, ts_pow, ts_base;   // ts = ts_const * (rad^ts_pow - ts_base)  [q^d_v == 0]
  int ts_mode;                        // 0 = skip, 1 = closed form (q^d_v = 0), 2 = with 2F1
  // q^d_v != 0 (maths/geometry.py:150-178): a = q^d_v, b = (1 - q_v + eps q^d_v)/eps,
  // hypergeometric connection coefficients K1 = b/(b-a), K2 = Gamma(b+1)Gamma(a-b)/Gamma(a)
  double hy_a, hy_b, hy_k1, hy_k2, hy_axis;
  double r1_m, r2_m, w0_m, mr0_m, r0_m;
};

hipError_t pack_field_launch(const double* src, const double* den, const uint8_t* red, void* dst,
                             int64_t n, int dtype, hipStream_t st);
hipError_t compact_fields_launch(const rjp_fields* fl, void* d_em0, int64_t* d_n_bad,
                                 hipStream_t st);
hipError_t tau_field_launch(const rjp_fields* fl, int gff_mode, void* d_a0, hipStream_t st);
hipError_t unmask_ts_launch(const rjp_fields* fl, int jet, void* d_out, hipStream_t st);
hipError_t synth_launch(uint64_t seed, int temp_mode, int nz, int64_t cell0, int64_t n, int dtype,
                        void* nd, void* xi, void* temp, void* pf, void* ts, void* vy, void* em0,
                        void* a0, int a0_mode, hipStream_t st);
hipError_t build_fields_launch(const GeomDev& g, int dtype, void* nd, void* xi, void* temp,
                               void* pf, void* ts, void* vy, double* ff_raw, double* areas_raw,
                               double* vx_raw, double* vz_raw, void* em0, void* a0, int a0_mode,
                               hipStream_t st);

// ---- rrl_scan.hip -------------------------------------------------------------------------
hipError_t rrl_scan_launch(const rjp_fields* fl, const rjp_bursts* hb, const double* d_ext,
                           double time_s, const rjp_line* line, const double* h_nu,
                           const double* d_nu, int nchan, double* tau, hipStream_t st);
hipError_t rrl_cells_launch(const rjp_fields* fl, const rjp_bursts* hb, const double* d_ext,
                            double time_s, const rjp_line* line, const double* h_nu,
                            const double* d_nu, int nchan, double* out, hipStream_t st);
hipError_t rrl_maps_launch(const double* tau_rrl, const double* tau_ff, const double* tavg,
                           const double* flux_ff, int64_t npix, const double* d_cflux,
                           const double* d_hnu_k, int nchan, double* flux, double* ftot,
                           double* part, hipStream_t st);

}  // namespace rjp
