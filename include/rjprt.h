/* rjprt.h -- C-ABI of librjprt.so, the MI355X (gfx950) line-of-sight radiative-transfer core
 * behind RaJePy's JetModel.
 *
 * The reference (SimonP2207/RaJePy) has no FFI: its hot path is reached only through Python
 * methods of `JetModel` (classes.py:1101-1541).  Each entry point below names the reference
 * method(s) whose NumPy body it replaces; `rajepy_amd/_lib.py` is the ctypes binding and
 * INTEGRATION.md shows the stub a RaJePy maintainer would add.
 *
 * Conventions
 *  - every function returns an int status: 0 = RJP_OK, negative = error; the message for the
 *    last error of a context is `rjp_last_error(ctx)`.  No exceptions, no exit().
 *  - all array arguments named `d_*` are CALLER-OWNED DEVICE pointers (e.g.
 *    torch.Tensor.data_ptr()); the library never frees or retains them.  Arguments named
 *    `h_*` are host pointers to small per-channel / per-epoch tables (copied on the stream).
 *  - work is enqueued on the caller's `stream` (a hipStream_t passed as void*; NULL = the
 *    default stream) and is asynchronous; the caller synchronises.
 *  - a context is bound to one device and is not thread-safe: one context per rank.
 *  - grids are C-contiguous (n_x, n_y, n_z), line of sight = axis 1 (classes.py:46,
 *    363-367); maps are (n_x, n_z) row-major, P = n_x*n_z pixels.
 */
#ifndef RJPRT_H
#define RJPRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RJP_VERSION 109          /* 0.1.9 */
#define RJP_RANGE_BLOCKS 2048    /* partial (min, max) pairs rjp_field_range writes */
#define RJP_MAX_EPOCH_TILE 32    /* most epochs evaluated per grid pass: 32 uniformly spaced ones (with or without d_em), 16 when only 16-31 are left, else tiles of 8, 4, 2, 1 */

enum rjp_status {
  RJP_OK = 0,
  RJP_ERR_ARG = -1,        /* bad argument (null pointer, shape, dtype tag ...) */
  RJP_ERR_HIP = -2,        /* a HIP runtime call failed */
  RJP_ERR_NODEVICE = -3,   /* no usable gfx950 device */
  RJP_ERR_WORKSPACE = -4,  /* workspace too small */
  RJP_ERR_DEGENERATE = -5  /* rjp_build_fields: the launch-time integral's 2F1 is a logarithmic
                              case the device series does not cover; nothing was enqueued --
                              the caller evaluates `ts` itself (scipy hyp2f1, as the reference
                              does, maths/geometry.py:166-171) and uploads it */
};

enum rjp_dtype { RJP_F32 = 4, RJP_F64 = 8 };   /* storage width of the 3-D fields */

enum rjp_gff_mode {
  RJP_GFF_SCALAR = 0,      /* q_T == 0: one Gaunt factor per channel (classes.py:1388-1389) */
  RJP_GFF_POWERLAW = 1     /* 11.95 T^0.15 nu^-0.1 per cell        (classes.py:1393, 1426) */
};

typedef struct rjp_ctx rjp_ctx;

/* Device-resident model state: the five (continuum) / six (RRL) per-cell fields of
 * SURVEY.md 8(d).  Layout produced by rjp_pack_field / rjp_build_fields:
 *   nd   |value| = steady-state number density `_nd` [cm^-3] (classes.py:889-897);
 *        SIGN BIT = 1 where the cell belongs to the red jet (rr < 0, classes.py:866, 895)
 *   xi   ionisation fraction (classes.py:910-936)
 *   temp temperature [K] (classes.py:942-969)
 *   pf   path-length factor fill_factor / areas (classes.py:1118, 1185, 1396-1397)
 *   ts   launch time `_ts` [s] (classes.py:847-853); may be NULL when there are no bursts
 *   vy   line-of-sight velocity vel[1] [km/s] (classes.py:1093); RRL only, else NULL
 * NaN marks "outside the jet" exactly as in the reference. */
typedef struct rjp_fields {
  const void* d_nd;
  const void* d_xi;
  const void* d_temp;
  const void* d_pf;
  const void* d_ts;
  const void* d_vy;
  int32_t nx, ny, nz;
  int32_t dtype;            /* enum rjp_dtype */
  double csize_au;          /* cell size [au] */
  /* Optional per-sightline occupied y-range [d_ylo[p], d_yhi[p]) from rjp_y_bounds() (both
   * NULL = scan every row).  Real jets fill a few per cent of the grid; rows outside the
   * range hold only cells that cannot contribute (NaN density / T <= 0) and are skipped. */
  const int32_t* d_ylo;
  const int32_t* d_yhi;
  /* Optional compact scan layout, written by rjp_compact_fields() in the storage dtype:
   * d_em0[cell] = (|nd| * xi)^2 * pf, the emission-measure density of the steady-state jet
   * [cm^-6] -- the only combination of nd, xi and pf that emission_measure / optical_depth_ff
   * use (classes.py:1116-1118, 1395-1397) -- with the red-jet flag of nd in its SIGN BIT.
   * The free-free scan then streams 3 fields (em0, temp, ts = 24 B/cell in f64) instead of 5
   * (40 B/cell) and, for f64 storage, returns bit-identical maps (f32 storage rounds the
   * product once more, 6e-8).  When non-NULL, rjp_ff_scan and rjp_y_bounds
   * read d_em0 and ignore d_nd / d_xi / d_pf (which may then be NULL for those two calls);
   * the RRL and collapse=False entry points always read the wide fields. */
  const void* d_em0;
  /* Optional tau scan layout (RJP_F64 storage only), written by rjp_tau_field() or by the
   * producers: d_a0[cell] = em0 * T^-1.5 (a0_mode = RJP_GFF_SCALAR) or em0 * T^-1.35
   * (RJP_GFF_POWERLAW) -- every factor of a cell's free-free optical depth that depends on
   * neither frequency nor epoch (classes.py:1395-1397; the temperature grid is static,
   * classes.py:942-969) -- formed exactly as the scan forms it from em0 and temp, red-jet flag
   * in the SIGN BIT.  When non-NULL and a0_mode equals the gff_mode of the call, rjp_ff_scan
   * streams d_a0 and d_ts = 16 B/cell (plus d_em0, 24 B/cell, only when d_em is asked for) and
   * never reads d_temp; maps are bit-identical to the compact and wide layouts'.  T_avg, which
   * depends on neither frequency nor epoch either (classes.py:1471-1472), then comes from
   * rjp_tavg() once per model (a d_tavg passed to rjp_ff_scan is served by that pass). */
  const void* d_a0;
  int32_t a0_mode;          /* enum rjp_gff_mode d_a0 was built for */
  int32_t reserved_;        /* 0 */
  /* Optional: the range of the finite launch times, [ts_lo, ts_hi] in seconds (rjp_field_range;
   * both 0 = not provided).  With it, a tau-layout scan of >= 12 epochs (with EM maps: when
   * d_em0 is attached -- a second pass takes the moments of em0) may take the MOMENT path: sum_y a0 chi(t_e - ts)^2 is a convolution of the sightline's launch-time
   * distribution with chi^2, so ONE pass over the grid accumulates per-sightline Chebyshev
   * moments of a0 over K launch-time bins of order N -- (K, N) one of (80, 8), (53, 12), (39, 16),
   * the cheapest the host's accuracy check accepts -- and any number of epochs, uniformly spaced or
   * not, becomes a small contraction.  The expansion is checked against chi^2 for the call's bursts and
   * epochs (on the device) and the path is used only when every coefficient table is good to 1e-11
   * relative AND a cost model says it is the faster one (long, densely filled sightlines; else
   * the epoch tiles run, as before); sums are reproducible to rounding, not bit
   * for bit (LDS atomics).  The range must contain every finite launch time of d_ts: pass what
   * rjp_field_range returned for it, or zeros.
   * LAUNCH-TIME RANGE GUARD (ABI 109).  The paths below clamp a launch time into their bins /
   * table, so a range that misses some would give silently wrong maps -- the reference has no
   * such failure mode (classes.py:844-845 evaluates every cell).  Therefore: (1) the first
   * rjp_ff_scan / rjp_ff_step that uses a (d_ts, ts_lo, ts_hi) this context has not seen checks
   * it with one pass over d_ts and ONE stream synchronisation (1.3 ms for 1.07e9 cells, once per
   * model) and returns RJP_ERR_ARG with nothing enqueued when a finite launch time lies outside;
   * rjp_lt_count does the same inside its counting pass; (2) every kernel that bins or tabulates
   * by launch time watches the times it reads: a sightline that meets a finite one outside the
   * range gets NaN sums (never a clamped, plausible-looking value) and the context's guard flag is
   * raised -- the next entry point called on the context returns RJP_ERR_ARG once and enqueues
   * nothing; rjp_range_guard() queries and clears the flag after a synchronisation.
   * The same range lets SINGLE-epoch scans of large f64 maps (>= 32768 sightlines, >= 64 rows; the
   * tau layout or the five model fields) take the burst factor chi(t - ts) from a table in LDS
   * instead of evaluating the Gaussians per cell: piecewise polynomials of degree 7 over the
   * times since launch that can occur, built on the device in front of the scan, bound 2e-13 on
   * chi^2 (bursts with a negative amplitude or a table that would not fit 72 KB keep the
   * Gaussians); maps equal to the Gaussian scan's to rounding (rjp_last_scan_path = 3). */
  double ts_lo, ts_hi;
  /* Optional hint for the choice between the epoch tiles and the moment path: the number of
   * cells inside the occupied y-ranges, sum_p max(0, d_yhi[p] - d_ylo[p]) (0 = unknown: all
   * n_x n_y n_z cells are assumed to matter).  The tiles' cost scales with it, the moment
   * path's per-sightline costs (2 K N doubles of moments, <= 10 KiB, written and read back) do not: a sparse jet
   * keeps the tiles.  A negative value skips the cost model (tests: the moment path on grids it
   * would not pay for). */
  int64_t occupied_cells;
  /* Optional launch-time-ordered layout of (a0, ts) for epoch sweeps, built once per model by
   * rjp_lt_count() + rjp_lt_fill() (all three NULL / 0 = absent).  Every group of 64 consecutive
   * sightlines is bucketed by (jet, launch-time bin): d_lt_cells holds (|a0|, ts) pairs,
   * [(d_lt_rowoff[g * 2 lt_K + q] + r) * 64 + lane], the rows of one (group, bin) contiguous and
   * padded to the largest count among the group's sightlines (in chunks of 4 rows).  A wave then
   * walks the bins of its group with every lane in the SAME bin: the Chebyshev moments of a bin
   * live in registers and are contracted with the bin's coefficient rows at its end -- no LDS
   * atomics, no moment maps in HBM, sums in a fixed order (bit-reproducible for one layout).
   * rjp_ff_scan takes it for tau-layout scans of RJP_MOM_MIN.. 32 epochs without EM maps when
   * the coefficient tables pass the accuracy check at some order <= 32.  It belongs to the d_a0,
   * d_ts, ts_lo, ts_hi it was built from: rebuild after any of them changes. */
  const void* d_lt_cells;
  const int32_t* d_lt_rowoff;
  const double* d_lt_aux;
  int32_t lt_K;             /* launch-time bins per jet the layout was built with */
  int32_t reserved2_;       /* 0 */
  /* Optional cache of the launch-time moment maps of a0 (NULL = none): rjp_moment_cache_bytes()
   * bytes of device memory the CALLER keeps with the model.  The moments depend on the fields,
   * on ts_lo / ts_hi, on the (bins, order) shape and on WHICH jets have bursts -- not on the
   * epochs, not on the burst parameters -- so every further sweep of the model (other epochs,
   * other bursts, e.g. a fit of burst parameters) needs only the contraction.  A sweep that
   * takes the LDS moment path without EM maps writes its moment maps here instead of into the
   * workspace; when mom_cache_K / mom_cache_N equal the shape that sweep selects, the pass over
   * the grid is SKIPPED (rjp_last_scan_path returns 4).  The caller sets mom_cache_K / _N to the
   * shape rjp_last_scan_path reported after a call that filled the buffer, and back to 0
   * whenever d_a0, d_ts, the range or the set of jets with bursts changes. */
  double* d_mom_cache;
  int32_t mom_cache_K, mom_cache_N;
} rjp_fields;

/* Ejection bursts (classes.py:399-463): mdot(t)/mdot_ss = 1 + sum_b amp_rel_b *
 * exp(-(t - t0_b)^2 * inv2s2_b); index 0 = red jet, 1 = blue jet.  ANY number of bursts per
 * jet, as in the reference (classes.py:245-264 registers every row of params["ejection"]):
 * the arrays are HOST pointers to n[j] doubles each (may be NULL when n[j] == 0), read
 * during the call.  The first eight bursts of a jet travel to the kernels as scalar
 * arguments, the rest in a small device table the library stages on the stream. */
typedef struct rjp_bursts {
  int32_t n[2];
  const double* t0[2];        /* [s] */
  const double* amp_rel[2];   /* (peak_jml - ss_jml) / ss_jml */
  const double* inv2s2[2];    /* 1 / (2 sigma^2) [s^-2] */
} rjp_bursts;

/* One radio recombination line, host-side scalars of maths/rrls.py (LTE path). */
typedef struct rjp_line {
  double nu_rest;      /* rrl_nu_0 [Hz] (rrls.py:14-29) */
  double kG;           /* deltanu_g / (nu0 sqrt(T)) = sqrt(4 ln2 2k/(m c^2)) (rrls.py:116-118) */
  double kL;           /* deltanu_l / n_e = 8.2 (n/100)^4.5 (1 + 2.25 dn/n) (rrls.py:101) */
  double kappa0;       /* 1.0991132675738456e-17 n^2 f_n1n2 * n_i/n_e (rrls.py:383-389, 73-83) */
  double en_over_k;    /* Z^2 E_n / k_cgs [K] (rrls.py:386) */
  double h_over_k;     /* h_cgs / k_cgs [K/Hz] (rrls.py:387) */
} rjp_line;

int rjp_version(void);
int rjp_device_count(void);
int rjp_ctx_create(int device, rjp_ctx** out);
int rjp_ctx_destroy(rjp_ctx* ctx);
const char* rjp_last_error(const rjp_ctx* ctx);   /* ctx may be NULL: last create error */

/* ---- field upload layout -------------------------------------------------------------- */

/* d_dst[i] = (dtype)d_src[i]; with d_red != NULL additionally sets the sign bit where
 * d_red[i] != 0 (and clears it elsewhere) -> the packed `nd` field; with d_den != NULL
 * stores d_src[i]/d_den[i] -> the `pf` field from fill_factor and areas. */
int rjp_pack_field(rjp_ctx* ctx, const double* d_src, const double* d_den,
                   const uint8_t* d_red, void* d_dst, int64_t n, int dtype, void* stream);

/* Builds the compact scan field (see rjp_fields.d_em0) from the wide fields nd, xi, pf in one
 * pass over all n_x*n_y*n_z cells, in the fields' storage dtype.  *d_n_bad (a device int64,
 * zeroed by the call) receives the number of cells the field cannot represent: a NEGATIVE
 * path factor (its sign would collide with the jet flag; fill_factor / areas never produce
 * one, classes.py:657-669, 763-764) or, for RJP_F32, a product outside the float range.
 * When it is non-zero the field is unusable and the caller keeps scanning the wide layout. */
int rjp_compact_fields(rjp_ctx* ctx, const rjp_fields* fields, void* d_em0,
                       int64_t* d_n_bad, void* stream);

/* Builds the tau scan field (see rjp_fields.d_a0) for `gff_mode` from fields->d_em0 and
 * fields->d_temp in one pass over all cells.  RJP_F64 storage with the compact field attached
 * (a model that has to stay on the wide layout -- negative path factors -- has no tau layout). */
int rjp_tau_field(rjp_ctx* ctx, const rjp_fields* fields, int32_t gff_mode, void* d_a0,
                  void* stream);

/* Launch times for the free-free scan of a model that has bursts in ONE jet only.  The
 * reference's burst Gaussians carry a NaN launch time into the cell's density, which nansum
 * then drops -- but a jet without any registered burst has the constant steady-state mass-loss
 * rate whatever the launch time (classes.py:232-233, 442-448, 866-875): its cells keep chi = 1.
 * rjp_ff_scan masks EVERY cell with a NaN launch time once bursts are present (a per-cell jet
 * test there cost the single-epoch scan 2.7 %), so such a model scans the copy written here:
 * d_ts_out[i] = fields->ts_lo (0 when no range is given: any finite value gives chi = 1 there;
 * this one lies inside the declared range, which the range guard holds the copy to as well) where
 * d_ts[i] is NaN and the cell belongs to `jet` (0 = red, 1 = blue: the one WITHOUT bursts; the flag
 * is read from the sign bit of d_a0, d_em0 or d_nd), d_ts[i] elsewhere.
 * (rjp_rrl_scan and the collapse=False entry points apply the rule themselves.) */
int rjp_unmask_launch_times(rjp_ctx* ctx, const rjp_fields* fields, int32_t jet, void* d_ts_out,
                            void* stream);

/* Per-block (min, max) of the finite entries of a field of n elements (NaN ignored):
 * d_partials[2 b], d_partials[2 b + 1], b < RJP_RANGE_BLOCKS (+inf / -inf for a block without a
 * finite entry); the caller finishes the reduction on the host.  Used once per model for
 * rjp_fields.ts_lo / ts_hi. */
int rjp_field_range(rjp_ctx* ctx, const void* d_field, int64_t n, int dtype, double* d_partials,
                    void* stream);

/* T_avg map of the model: d_tavg[p] = nanmean_y(T where T > 0) [K], NaN on empty sightlines
 * (classes.py:1471-1472, 1484-1485, 1254-1256).  One pass over fields->d_temp (the only field
 * read; d_ylo / d_yhi are honoured), bit-identical to the map a single-epoch rjp_ff_scan
 * derives.  d_work: at least rjp_ff_scan_workspace(nx, ny, nz, 1) bytes. */
int rjp_tavg(rjp_ctx* ctx, const rjp_fields* fields, double* d_tavg, void* d_work,
             size_t work_bytes, void* stream);

/* Per-sightline occupied y-range of a packed field set: d_ylo[p] = first row, d_yhi[p] = one
 * past the last row whose cell can contribute to any product of the path, i.e. T > 0 (counts
 * in the nanmean of intensity_ff, classes.py:1471) or n, x and ff/areas all non-NaN (emission
 * measure, classes.py:1116-1118); empty sightlines get [ny, 0).  One pass over 2 (compact layout) or 4 fields,
 * amortised over every later scan of the same model. */
int rjp_y_bounds(rjp_ctx* ctx, const rjp_fields* fields, int32_t* d_ylo, int32_t* d_yhi,
                 void* stream);

/* sum_p max(0, d_yhi[p] - d_ylo[p]) -> *h_count: the cells inside the occupied y-ranges, i.e.
 * rjp_fields.occupied_cells (one small launch, an 8-byte copy back; synchronises the stream). */
int rjp_occupied_cells(rjp_ctx* ctx, const int32_t* d_ylo, const int32_t* d_yhi, int64_t n_pix,
                       int64_t* h_count, void* stream);

/* ---- K1: free-free / emission-measure scan -------------------------------------------
 * Replaces the y-reductions of JetModel.emission_measure (classes.py:1116-1120),
 * .optical_depth_ff (1375-1432) and the nanmean of .intensity_ff (1471-1472, 1484-1485),
 * with number_density = _nd * chi_xyz (classes.py:861-875) evaluated in registers for
 * each of E epochs.  One pass over the grid per tile of up to RJP_MAX_EPOCH_TILE epochs.
 *   d_sumA [E*P]  sum_y T^-1.5 (n x)^2 pf          (RJP_GFF_SCALAR)
 *                 sum_y T^-1.35 (n x)^2 pf         (RJP_GFF_POWERLAW)   [cm^-6 K^-1.5|-1.35]
 *   d_em   [E*P]  emission measure [pc cm^-6]      (may be NULL)
 *   d_tavg [P]    nanmean_y(T where T > 0) [K], NaN on empty sightlines (may be NULL; on the
 *                 tau layout it costs a separate pass over d_temp: ask once, see rjp_tavg)
 * h_epochs_s: model times [s] (JetModel.time), E >= 1.
 * d_work: scratch of at least rjp_ff_scan_workspace() bytes. */
size_t rjp_ff_scan_workspace(int32_t nx, int32_t ny, int32_t nz, int32_t n_epochs);
int rjp_ff_scan(rjp_ctx* ctx, const rjp_fields* fields, const rjp_bursts* bursts,
                const double* h_epochs_s, int32_t n_epochs, int32_t gff_mode,
                double* d_sumA, double* d_em, double* d_tavg,
                void* d_work, size_t work_bytes, void* stream);

/* Launch-time range guard (see rjp_fields.ts_lo): 1 = a kernel of this context has met a finite
 * launch time outside fields.ts_lo / ts_hi since the last query (the sums of those sightlines are
 * NaN); the flag is cleared.  0 = none.  Meaningful after the caller has synchronised the stream
 * the scans ran on. */
int rjp_range_guard(rjp_ctx* ctx);

/* Which path the last rjp_ff_scan of this context took: 0 = epoch tiles, 1 = launch-time moments
 * in LDS + contraction, 2 = launch-time moments on the launch-time-ordered layout, 3 = the
 * single-epoch tau-layout scan with the burst factor from a table in LDS, 4 = contraction of the
 * caller's cached moment maps (rjp_fields.d_mom_cache: no pass over the grid) (and, if
 * non-NULL: in *worst_rel_err the worst relative error of the moment expansion measured for
 * that call, in moment_shape[0..1] the (bins, order) shape it chose; zeros for the tiles).
 * For tests and the bench line. */
int rjp_last_scan_path(const rjp_ctx* ctx, double* worst_rel_err, int32_t* moment_shape);

/* Host wall time [ms] the last table build of this context took (the coefficient tables of the
 * moment paths are built and checked on the device when the bursts or epochs of a sweep change:
 * one small launch and ONE stream synchronisation inside that rjp_ff_scan; a repeated request
 * reuses them).  0 before the first build. */
double rjp_last_table_build_ms(const rjp_ctx* ctx);

/* Bytes of rjp_fields.d_mom_cache for an n_x x n_z map (1280 doubles per sightline). */
size_t rjp_moment_cache_bytes(int32_t nx, int32_t nz);

/* ---- launch-time-ordered layout (rjp_fields.d_lt_*) -------------------------------------
 * Two calls, because the size of the padded layout is known only after counting:
 *   rjp_lt_rowoff_entries(nx, nz, K)  entries (int32) of d_rowoff: ceil(nx nz / 64) * 2 K + 1
 *   rjp_lt_count   counts the cells of every (group, jet, bin), pads each to the group's largest
 *                  count, writes the exclusive prefix d_rowoff and returns the total number of
 *                  rows in *h_total_rows (synchronises the stream);
 *   rjp_lt_fill    writes d_cells (total_rows * 64 * 16 bytes) and d_aux (3 * nx * nz doubles:
 *                  per sightline the sums of |a0| over red / blue cells with a NaN launch time --
 *                  added when that jet has no burst, classes.py:232-233 -- and an "infinite
 *                  term" flag).
 * Needs RJP_F64 fields with d_a0, d_ts and ts_lo / ts_hi; 1 <= K <= 80, n_y < 65536.  Cells with
 * a0 == 0 or NaN are dropped (nansum).  Cost: two passes over a0 and ts plus 16-byte scattered
 * writes (~50 ms for 1.07e9 cells): worth it for a model that is swept many times. */
size_t rjp_lt_rowoff_entries(int32_t nx, int32_t nz, int32_t K);
int rjp_lt_count(rjp_ctx* ctx, const rjp_fields* fields, int32_t K, int32_t* d_rowoff,
                 int64_t* h_total_rows, void* stream);
int rjp_lt_fill(rjp_ctx* ctx, const rjp_fields* fields, int32_t K, const int32_t* d_rowoff,
                void* d_cells, double* d_aux, void* stream);

/* ---- K2: per-channel map stage --------------------------------------------------------
 * Replaces the map-level arithmetic of optical_depth_ff / intensity_ff / flux_ff
 * (classes.py:1395-1397, 1473-1475, 1519-1521):
 *   tau[e,f,p]  = h_ctau[f] * sumA[e,p]
 *   flux[e,f,p] = h_cflux[f] * tavg[p] * (1 - exp(-tau))      [Jy/pixel]
 *   ftot[e,f]   = nansum_p flux[e,f,p]
 * h_ctau[f]  = 0.018 nu^-2 gff(nu,T_0) csize au 100      (scalar mode)
 *            = 0.018 nu^-2 11.95 nu^-0.1 csize au 100    (power-law mode)
 * h_cflux[f] = 2 nu^2 k / c^2 * arctan(csize au / (dist pc))^2 / 1e-26
 * Any of d_tau / d_flux / d_ftot may be NULL.  d_work >= rjp_ff_maps_workspace(). */
size_t rjp_ff_maps_workspace(int64_t n_pix, int32_t n_epochs, int32_t n_chan);
int rjp_ff_maps(rjp_ctx* ctx, const double* d_sumA, const double* d_tavg, int64_t n_pix,
                int32_t n_epochs, const double* h_ctau, const double* h_cflux, int32_t n_chan,
                double* d_tau, double* d_flux, double* d_ftot,
                void* d_work, size_t work_bytes, void* stream);

/* ---- K1 + K2 from ONE call ----------------------------------------------------------------
 * rjp_ff_scan followed by rjp_ff_maps on the same stream, for callers whose step is shorter than
 * two round trips through their FFI (an x-slab of a sharded grid: 0.3 ms per step at 64 x 4096 x
 * 512; a 256 x 1024 x 256 model: 0.2 ms): what JetModel.optical_depth_ff / flux_ff ask per epoch
 * (classes.py:1375-1432, 1466-1541).  Arguments as for the two calls; d_tavg is the model's T_avg
 * map (rjp_tavg: INPUT here, the scan derives none); d_em, d_tau, d_flux, d_ftot may be NULL.
 * Every argument of both stages is validated before anything is enqueued.  d_work >=
 * rjp_ff_scan_workspace(), d_work_maps >= rjp_ff_maps_workspace() (two buffers: the scan of the
 * next step may overlap this step's map stage on another stream). */
int rjp_ff_step(rjp_ctx* ctx, const rjp_fields* fields, const rjp_bursts* bursts,
                const double* h_epochs_s, int32_t n_epochs, int32_t gff_mode,
                const double* d_tavg, const double* h_ctau, const double* h_cflux, int32_t n_chan,
                double* d_sumA, double* d_em, double* d_tau, double* d_flux, double* d_ftot,
                void* d_work, size_t work_bytes, void* d_work_maps, size_t work_maps_bytes,
                void* stream);

/* ---- K3: recombination-line scan ------------------------------------------------------
 * Replaces JetModel.optical_depth_rrl (classes.py:1159-1214): per cell Doppler-shifted rest
 * frequency, thermal + Stark widths, Voigt profile Re w(z) (rrls.py:350-354), LTE kappa_L
 * (rrls.py:383-389), summed along y for every channel.
 *   d_tau_rrl [F*P]
 * One epoch per call (time_s). */
int rjp_rrl_scan(rjp_ctx* ctx, const rjp_fields* fields, const rjp_bursts* bursts,
                 double time_s, const rjp_line* line, const double* h_nu, int32_t n_chan,
                 double* d_tau_rrl, void* stream);

/* collapse=False forms (classes.py:1382-1383, 1176-1177): the 3-D per-cell optical depths,
 * d_tau_cells[f * N + cell], NaN outside the jet as in the reference.  h_ctau as for
 * rjp_ff_maps.  Not used by Pipeline; N*F*8 bytes of output. */
int rjp_ff_cells(rjp_ctx* ctx, const rjp_fields* fields, const rjp_bursts* bursts,
                 double time_s, int32_t gff_mode, const double* h_ctau, int32_t n_chan,
                 double* d_tau_cells, void* stream);
int rjp_rrl_cells(rjp_ctx* ctx, const rjp_fields* fields, const rjp_bursts* bursts,
                  double time_s, const rjp_line* line, const double* h_nu, int32_t n_chan,
                  double* d_tau_cells, void* stream);

/* Map stage of intensity_rrl / flux_rrl (classes.py:1280-1282, 1339-1343;
 * rrls.py:444-449; physics.py:571-574):
 *   I_L = B_nu(tavg) exp(-tau_ff) (1 - exp(-tau_rrl)) 1e-3 ; S = I_L * omega / 1e-26
 *   flux[f,p] = S (+ flux_ff[f,p] when d_flux_ff != NULL, i.e. contsub=False)
 * h_cflux_rrl[f] = omega/1e-26 * 1e-3 * 2 h_cgs nu^3 / c_cgs^2 ; h_hnu_k[f] = h nu / k. */
int rjp_rrl_maps(rjp_ctx* ctx, const double* d_tau_rrl, const double* d_tau_ff,
                 const double* d_tavg, const double* d_flux_ff, int64_t n_pix,
                 const double* h_cflux_rrl, const double* h_hnu_k, int32_t n_chan,
                 double* d_flux, double* d_ftot, void* d_work, size_t work_bytes,
                 void* stream);

/* ---- K4: geometry -> fields on the device --------------------------------------------
 * Replaces the lazily cached grids of JetModel (fill_factor/areas classes.py:657-669,
 * 763-764; grid_rwp 521-525; rreff 549-555; _nd 877-897; ion_fraction 915-934;
 * temperature 950-967; ts 847-853 with maths/geometry.py:150-178; vel[1] 1042-1093).
 * Writes the packed device layout directly. */
typedef struct rjp_geometry {
  int32_t nx, ny, nz;
  int32_t rotation_ccw;       /* 1 = "CCW", 0 = "CW" */
  double csize;               /* [au] */
  double inc, pa;             /* [deg] */
  double w_0, r_0, mod_r_0, epsilon;
  double R_1, R_2;            /* [au] */
  double M_star;              /* [Msol] */
  double v_lsr;               /* [km/s] */
  double n_0, x_0, T_0, v_0;  /* axis values at the base */
  double q_n, q_x, q_T, q_v;          /* power laws along r */
  double qd_n, qd_x, qd_T, qd_v;      /* power laws across the jet */
  double rb_frac;             /* mlr_rj / mlr_bj (classes.py:228-229, 895) */
  /* x-slab sharding: build rows [ix0, ix0 + nx) of a grid that is nx_total rows wide
   * (cell x-coordinate = csize * (ix0 + i_x - nx_total / 2)); nx_total = 0 means nx. */
  int32_t ix0, nx_total;
} rjp_geometry;

/* Any output may be NULL to skip it (d_vy / d_ts typically); d_ff_raw / d_areas_raw (float64, optional) receive
 * the un-packed fill factors / areas that JetModel.save pickles (classes.py:1704-1709);
 * d_vx_raw / d_vz_raw (float64, optional) the transverse components of JetModel.vel.
 * Launch times: closed form for q^d_v = 0, otherwise Gauss' 2F1(a, b; b+1; -A) of
 * maths/geometry.py:166-171 evaluated on the device (Pfaff + 1/z connection formula);
 * RJP_ERR_DEGENERATE if a-b or b is a non-positive integer (logarithmic cases) and
 * d_ts != NULL.
 * d_em0 (optional, RJP_F64 only): the compact scan field of rjp_fields.d_em0 written in the
 * same pass, bit-identical to what rjp_compact_fields derives from nd, xi, pf -- with d_nd,
 * d_xi, d_pf NULL a continuum-only model occupies 24 B/cell (12e9 cells per 288 GB GPU).
 * d_a0 (optional, RJP_F64 only): the tau scan field of rjp_fields.d_a0 for gff mode `a0_mode`,
 * bit-identical to rjp_tau_field's. */
int rjp_build_fields(rjp_ctx* ctx, const rjp_geometry* geom, int dtype,
                     void* d_nd, void* d_xi, void* d_temp, void* d_pf, void* d_ts,
                     void* d_vy, double* d_ff_raw, double* d_areas_raw,
                     double* d_vx_raw, double* d_vz_raw, void* d_em0, void* d_a0,
                     int32_t a0_mode, void* stream);

/* ---- measurement harness: synthetic dense fields ---------------------
 * Counter-based: u = splitmix64(seed ^ field_id<<60 ^ linear_cell_index) -> [0,1).
 *   n = 10^(5+2.5u), x = 0.05+0.45u, T = 1e4 (temp_mode 0) or 5e3+1.5e4u (temp_mode 1),
 *   pf = 0.5 w.p. 0.25 else 1, ts = 5u yr, red = i_z < n_z/2, vy = 6.2+60(u-0.5) km/s.
 * Generates cells [cell0, cell0+n) of the flattened grid so a host restatement can
 * regenerate any sub-block.  Any output may be NULL; d_em0 / d_a0 (optional, RJP_F64 only)
 * receive the compact and tau scan fields of the same cells, as for rjp_build_fields. */
int rjp_synth_fields(rjp_ctx* ctx, uint64_t seed, int32_t temp_mode, int32_t nz,
                     int64_t cell0, int64_t n, int dtype, void* d_nd, void* d_xi,
                     void* d_temp, void* d_pf, void* d_ts, void* d_vy, void* d_em0,
                     void* d_a0, int32_t a0_mode, void* stream);

/* Device-time probe used by bench.py: average duration [ms] of `reps` back-to-back
 * rjp_ff_scan launches measured with HIP events on `stream`. */
int rjp_time_ff_scan(rjp_ctx* ctx, const rjp_fields* fields, const rjp_bursts* bursts,
                     const double* h_epochs_s, int32_t n_epochs, int32_t gff_mode,
                     double* d_sumA, double* d_em, double* d_tavg, void* d_work,
                     size_t work_bytes, void* stream, int32_t reps, double* ms_avg);

#ifdef __cplusplus
}
#endif
#endif /* RJPRT_H */
