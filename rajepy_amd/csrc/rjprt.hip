// librjprt.so -- C-ABI entry points (include/rjprt.h) over the gfx950 kernels.
// The kernels live in their own translation units (ff_scan*.hip, fields.hip, rrl_scan.hip;
// csrc/Makefile builds them in parallel); rjp_host.h declares their launch wrappers.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rjp_host.h"

struct rjp_ctx {
  int device = -1;
  std::string err;
  // small per-call tables (channel coefficients, frequencies): a ring of pinned host
  // staging buffers + device copies, grown on demand; the copies are ordered on the
  // caller's stream and a slot is overwritten only after its readers have finished.  A call
  // whose tables equal a slot's content (a sweep over epochs at a fixed channel list) reuses
  // the device copy: no host-to-device copy between the scan and the map stage.
  rjp::MomPlan mom;               // last moment-path request (its tables are reused)
  rjp::ChiPlan chi;               // the burst-factor table of the last single-epoch scan
  int last_path = 0;              // 0 = epoch tiles, 1 = LDS moments, 2 = launch-time-ordered layout
  static constexpr int kSlots = 8;
  struct Slot {
    double* h = nullptr;
    double* d = nullptr;
    size_t cap = 0;               // doubles
    size_t used = 0;              // doubles of valid content (0 = none)
    hipStream_t up_stream = nullptr;   // stream the content was uploaded on
    hipEvent_t free_ev = nullptr; // recorded after the last kernel that reads `d`
  } slot[kSlots];
  int next_slot = 0;
  int cur_slot = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // Launch-time range guard (include/rjprt.h): one int in pinned, device-visible host memory.
  // Kernels that bin or tabulate by launch time store 1 into it when they meet a finite launch
  // time outside rjp_fields.ts_lo / ts_hi (and poison that sightline's sums with NaN); the next
  // entry point of the context reports it.  `range_ok`: the ranges rjp_ff_scan has already
  // checked against their launch-time field with a pass of its own (most recent first).
  int* guard = nullptr;
  unsigned long long* d_count = nullptr;      // device word of rjp_occupied_cells
  struct RangeKey {
    const void* d_ts = nullptr;
    int64_t n = 0;
    int dtype = 0;
    double lo = 0.0, hi = 0.0;
  } range_ok[4];
};

static std::string g_create_err;

static int fail(rjp_ctx* ctx, int code, const char* what, hipError_t e = hipSuccess) {
  std::string m = what;
  if (e != hipSuccess) {
    m += ": ";
    m += hipGetErrorString(e);
  }
  if (ctx) ctx->err = m; else g_create_err = m;
  return code;
}

#define RJP_HIP(ctx, call)                                              \
  do {                                                                  \
    hipError_t _e = (call);                                             \
    if (_e != hipSuccess) return fail((ctx), RJP_ERR_HIP, #call, _e);   \
  } while (0)

static const char* const kGuardMsg =
    "an earlier scan of this context met finite launch times outside fields.ts_lo / ts_hi: the "
    "sums of those sightlines were set to NaN (pass what rjp_field_range returns for d_ts, or "
    "zeros)";

static int bind(rjp_ctx* ctx) {
  if (!ctx) return fail(nullptr, RJP_ERR_ARG, "null context");
  hipError_t e = hipSetDevice(ctx->device);
  if (e != hipSuccess) return fail(ctx, RJP_ERR_HIP, "hipSetDevice", e);
  // the range guard is sticky until reported once (a kernel of an earlier, asynchronous call
  // raised it; whatever this call would enqueue is refused)
  if (ctx->guard && *(volatile int*)ctx->guard != 0) {
    *(volatile int*)ctx->guard = 0;
    for (auto& k : ctx->range_ok) k = rjp_ctx::RangeKey();
    return fail(ctx, RJP_ERR_ARG, kGuardMsg);
  }
  return RJP_OK;
}

// A scan is about to bin / tabulate by launch time on the strength of fields.ts_lo / ts_hi: a
// range this context has not seen with this launch-time field is CHECKED first -- one pass over
// d_ts (8 B/cell: 1.3 ms for 1.07e9 cells) and one stream synchronisation, once per model -- so
// that a wrong range is refused by the very call that passes it, before anything else is
// enqueued.  (Later content changes under the same pointer are caught by the kernels' guard.)
static int check_ts_range(rjp_ctx* ctx, const rjp_fields* f, hipStream_t st) {
  const int64_t n = (int64_t)f->nx * f->ny * f->nz;
  for (const auto& k : ctx->range_ok)
    if (k.d_ts == f->d_ts && k.n == n && k.dtype == f->dtype && k.lo == f->ts_lo && k.hi == f->ts_hi)
      return RJP_OK;
  RJP_HIP(ctx, rjp::range_check_launch(f->d_ts, n, f->dtype, f->ts_lo, f->ts_hi, ctx->guard, st));
  RJP_HIP(ctx, hipStreamSynchronize(st));
  if (*(volatile int*)ctx->guard != 0) {
    *(volatile int*)ctx->guard = 0;
    return fail(ctx, RJP_ERR_ARG,
                "fields.ts_lo / ts_hi do not contain every finite launch time of fields.d_ts "
                "(pass what rjp_field_range returns for it, or zeros); nothing was enqueued");
  }
  for (int i = 3; i > 0; --i) ctx->range_ok[i] = ctx->range_ok[i - 1];
  ctx->range_ok[0].d_ts = f->d_ts; ctx->range_ok[0].n = n; ctx->range_ok[0].dtype = f->dtype;
  ctx->range_ok[0].lo = f->ts_lo; ctx->range_ok[0].hi = f->ts_hi;
  return RJP_OK;
}

// Stage up to four small host tables into one slot of the context's table ring and return
// device pointers.  The caller records `release_tables()` after enqueueing the readers.
static int stage_tables(rjp_ctx* ctx, hipStream_t st, const double* const* src,
                        const size_t* len, int ntab, double** dev_out) {
  size_t tot = 0;
  for (int i = 0; i < ntab; ++i) tot += len[i];
  for (int k = 0; k < rjp_ctx::kSlots; ++k) {
    rjp_ctx::Slot& c = ctx->slot[k];
    if (c.used != tot || c.up_stream != st || tot == 0) continue;
    size_t off = 0;
    bool same = true;
    for (int i = 0; i < ntab && same; ++i) {
      same = memcmp(c.h + off, src[i], len[i] * sizeof(double)) == 0;
      off += len[i];
    }
    if (!same) continue;
    off = 0;
    for (int i = 0; i < ntab; ++i) { dev_out[i] = c.d + off; off += len[i]; }
    ctx->cur_slot = k;
    return RJP_OK;
  }
  rjp_ctx::Slot& sl = ctx->slot[ctx->next_slot];
  ctx->cur_slot = ctx->next_slot;
  ctx->next_slot = (ctx->next_slot + 1) % rjp_ctx::kSlots;
  RJP_HIP(ctx, hipEventSynchronize(sl.free_ev));     // no-op unless its readers still run
  sl.used = 0;
  if (tot > sl.cap) {
    if (sl.h) RJP_HIP(ctx, hipHostFree(sl.h));
    if (sl.d) RJP_HIP(ctx, hipFree(sl.d));
    sl.h = nullptr; sl.d = nullptr; sl.cap = 0;
    const size_t cap = tot < 4096 ? 4096 : tot * 2;
    RJP_HIP(ctx, hipHostMalloc((void**)&sl.h, cap * sizeof(double), hipHostMallocDefault));
    RJP_HIP(ctx, hipMalloc((void**)&sl.d, cap * sizeof(double)));
    sl.cap = cap;
  }
  size_t off = 0;
  for (int i = 0; i < ntab; ++i) {
    if (len[i]) memcpy(sl.h + off, src[i], len[i] * sizeof(double));
    dev_out[i] = sl.d + off;
    off += len[i];
  }
  RJP_HIP(ctx, hipMemcpyAsync(sl.d, sl.h, tot * sizeof(double), hipMemcpyHostToDevice, st));
  sl.used = tot;
  sl.up_stream = st;
  return RJP_OK;
}

static int release_tables(rjp_ctx* ctx, hipStream_t st) {
  RJP_HIP(ctx, hipEventRecord(ctx->slot[ctx->cur_slot].free_ev, st));
  return RJP_OK;
}

// End of a call that staged tables: the slot's release event is recorded on EVERY path -- after
// a failed launch too, since the slot's upload (and anything enqueued before the failure) may
// still be in flight when the ring comes round to it again.
static int finish_staged(rjp_ctx* ctx, hipStream_t st, hipError_t e, const char* what) {
  const int r = release_tables(ctx, st);
  if (e != hipSuccess) return fail(ctx, RJP_ERR_HIP, what, e);
  return r;
}

static bool mode_ok(int m) { return m == RJP_GFF_SCALAR || m == RJP_GFF_POWERLAW; }

// `compact_ok`: the entry point can run from the compact layout alone (d_em0 + d_temp);
// `tau_mode` >= 0: ... or from the tau layout alone (d_a0 built for that mode)
static int check_fields(rjp_ctx* ctx, const rjp_fields* f, bool need_vy, bool compact_ok = false,
                        int tau_mode = -1) {
  if (!f) return fail(ctx, RJP_ERR_ARG, "fields is NULL");
  if (f->dtype != RJP_F32 && f->dtype != RJP_F64)
    return fail(ctx, RJP_ERR_ARG, "fields.dtype must be RJP_F32 (4) or RJP_F64 (8)");
  if (f->nx <= 0 || f->ny <= 0 || f->nz <= 0)
    return fail(ctx, RJP_ERR_ARG, "grid dimensions must be positive");
  if (f->d_a0 && (f->dtype != RJP_F64 || !mode_ok(f->a0_mode)))
    return fail(ctx, RJP_ERR_ARG, "fields.d_a0 needs RJP_F64 storage and a valid a0_mode");
  if (tau_mode >= 0 && f->d_a0 && f->a0_mode == tau_mode) {
    // nothing else is needed for the scan itself (d_em0 / d_temp are checked where asked for)
  } else if (compact_ok && f->d_em0) {
    if (!f->d_temp) return fail(ctx, RJP_ERR_ARG, "fields.d_temp must be a device pointer");
  } else if (!f->d_nd || !f->d_xi || !f->d_temp || !f->d_pf) {
    return fail(ctx, RJP_ERR_ARG, "fields nd/xi/temp/pf must be device pointers");
  }
  if (need_vy && !f->d_vy) return fail(ctx, RJP_ERR_ARG, "fields.d_vy required for RRL");
  if (!(f->csize_au > 0.0)) return fail(ctx, RJP_ERR_ARG, "fields.csize_au must be > 0");
  if ((f->d_ylo == nullptr) != (f->d_yhi == nullptr))
    return fail(ctx, RJP_ERR_ARG, "fields.d_ylo and d_yhi must both be set or both be NULL");
  return RJP_OK;
}

static int check_bursts(rjp_ctx* ctx, const rjp_bursts* b, const rjp_fields* f) {
  if (!b) return RJP_OK;
  for (int j = 0; j < 2; ++j) {
    if (b->n[j] < 0 || b->n[j] > (1 << 20))
      return fail(ctx, RJP_ERR_ARG, "bursts.n must be >= 0");
    if (b->n[j] > 0 && (!b->t0[j] || !b->amp_rel[j] || !b->inv2s2[j]))
      return fail(ctx, RJP_ERR_ARG, "bursts: NULL parameter array for a jet with n > 0");
    if (b->n[j] > 0 && !f->d_ts)
      return fail(ctx, RJP_ERR_ARG, "fields.d_ts required when bursts are present");
  }
  return RJP_OK;
}

// host image of the overflow-burst table of a single-epoch call (parameters only)
static std::vector<double> burst_ext_table(const rjp_bursts* b) {
  std::vector<double> t(bursts_ext_doubles(b), 0.0);
  if (!t.empty()) bursts_fill_ext(b, t.data());
  return t;
}

extern "C" {

int rjp_version(void) { return RJP_VERSION; }

int rjp_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int rjp_ctx_create(int device, rjp_ctx** out) {
  if (!out) return fail(nullptr, RJP_ERR_ARG, "out is NULL");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(nullptr, RJP_ERR_NODEVICE, "no HIP device visible", e);
  if (device < 0 || device >= n) return fail(nullptr, RJP_ERR_ARG, "device index out of range");
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) return fail(nullptr, RJP_ERR_HIP, "hipGetDeviceProperties", e);
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    std::string m = std::string("librjprt is built for gfx950 only; device is ") + prop.gcnArchName;
    return fail(nullptr, RJP_ERR_NODEVICE, m.c_str());
  }
  e = hipSetDevice(device);
  if (e != hipSuccess) return fail(nullptr, RJP_ERR_HIP, "hipSetDevice", e);
  rjp_ctx* c = new rjp_ctx();
  c->device = device;
  bool ok = hipEventCreate(&c->ev0) == hipSuccess && hipEventCreate(&c->ev1) == hipSuccess;
  for (int i = 0; ok && i < rjp_ctx::kSlots; ++i)
    ok = hipEventCreateWithFlags(&c->slot[i].free_ev, hipEventDisableTiming) == hipSuccess;
  ok = ok && hipHostMalloc((void**)&c->guard, sizeof(int), hipHostMallocDefault) == hipSuccess;
  if (!ok) {
    if (c->guard) (void)hipHostFree(c->guard);
    delete c;
    return fail(nullptr, RJP_ERR_HIP, "hipEventCreate / hipHostMalloc failed");
  }
  *c->guard = 0;
  *out = c;
  return RJP_OK;
}

int rjp_ctx_destroy(rjp_ctx* ctx) {
  if (!ctx) return RJP_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipDeviceSynchronize();
  for (auto& sl : ctx->slot) {
    if (sl.h) (void)hipHostFree(sl.h);
    if (sl.d) (void)hipFree(sl.d);
    if (sl.free_ev) (void)hipEventDestroy(sl.free_ev);
  }
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  if (ctx->guard) (void)hipHostFree(ctx->guard);
  if (ctx->d_count) (void)hipFree(ctx->d_count);
  rjp::moments_release(ctx->mom);
  delete ctx;
  return RJP_OK;
}

const char* rjp_last_error(const rjp_ctx* ctx) {
  return ctx ? ctx->err.c_str() : g_create_err.c_str();
}

int rjp_pack_field(rjp_ctx* ctx, const double* d_src, const double* d_den,
                   const uint8_t* d_red, void* d_dst, int64_t n, int dtype, void* stream) {
  if (int r = bind(ctx)) return r;
  if (!d_src || !d_dst || n <= 0) return fail(ctx, RJP_ERR_ARG, "rjp_pack_field: bad pointers/size");
  if (dtype != RJP_F32 && dtype != RJP_F64) return fail(ctx, RJP_ERR_ARG, "bad dtype tag");
  RJP_HIP(ctx, rjp::pack_field_launch(d_src, d_den, d_red, d_dst, n, dtype, (hipStream_t)stream));
  return RJP_OK;
}

int rjp_compact_fields(rjp_ctx* ctx, const rjp_fields* fields, void* d_em0, int64_t* d_n_bad,
                       void* stream) {
  if (int r = bind(ctx)) return r;
  if (int r = check_fields(ctx, fields, false)) return r;
  if (!d_em0 || !d_n_bad) return fail(ctx, RJP_ERR_ARG, "rjp_compact_fields: NULL output");
  RJP_HIP(ctx, rjp::compact_fields_launch(fields, d_em0, d_n_bad, (hipStream_t)stream));
  return RJP_OK;
}

int rjp_tau_field(rjp_ctx* ctx, const rjp_fields* fields, int32_t gff_mode, void* d_a0,
                  void* stream) {
  if (int r = bind(ctx)) return r;
  if (int r = check_fields(ctx, fields, false, true)) return r;
  if (!mode_ok(gff_mode)) return fail(ctx, RJP_ERR_ARG, "bad gff_mode");
  if (fields->dtype != RJP_F64 || !fields->d_em0)
    return fail(ctx, RJP_ERR_ARG, "rjp_tau_field: needs RJP_F64 storage with the compact field "
                                  "d_em0 attached");
  if (!d_a0) return fail(ctx, RJP_ERR_ARG, "rjp_tau_field: NULL output");
  RJP_HIP(ctx, rjp::tau_field_launch(fields, gff_mode, d_a0, (hipStream_t)stream));
  return RJP_OK;
}

int rjp_unmask_launch_times(rjp_ctx* ctx, const rjp_fields* fields, int32_t jet, void* d_ts_out,
                            void* stream) {
  if (int r = bind(ctx)) return r;
  if (!fields || !fields->d_ts || !d_ts_out)
    return fail(ctx, RJP_ERR_ARG, "rjp_unmask_launch_times: fields.d_ts / d_ts_out is NULL");
  if (fields->dtype != RJP_F32 && fields->dtype != RJP_F64)
    return fail(ctx, RJP_ERR_ARG, "fields.dtype must be RJP_F32 (4) or RJP_F64 (8)");
  if (fields->nx <= 0 || fields->ny <= 0 || fields->nz <= 0)
    return fail(ctx, RJP_ERR_ARG, "grid dimensions must be positive");
  if (jet != 0 && jet != 1) return fail(ctx, RJP_ERR_ARG, "jet must be 0 (red) or 1 (blue)");
  if (!fields->d_a0 && !fields->d_em0 && !fields->d_nd)
    return fail(ctx, RJP_ERR_ARG, "rjp_unmask_launch_times: needs a field that carries the jet "
                                  "flag (d_a0, d_em0 or d_nd)");
  RJP_HIP(ctx, rjp::unmask_ts_launch(fields, jet, d_ts_out, (hipStream_t)stream));
  return RJP_OK;
}

int rjp_field_range(rjp_ctx* ctx, const void* d_field, int64_t n, int dtype, double* d_partials,
                    void* stream) {
  if (int r = bind(ctx)) return r;
  if (!d_field || !d_partials || n <= 0) return fail(ctx, RJP_ERR_ARG, "rjp_field_range: bad pointers/size");
  if (dtype != RJP_F32 && dtype != RJP_F64) return fail(ctx, RJP_ERR_ARG, "bad dtype tag");
  RJP_HIP(ctx, rjp::field_range_launch(d_field, n, dtype, d_partials, (hipStream_t)stream));
  return RJP_OK;
}

int rjp_tavg(rjp_ctx* ctx, const rjp_fields* fields, double* d_tavg, void* d_work,
             size_t work_bytes, void* stream) {
  if (int r = bind(ctx)) return r;
  if (!fields || !fields->d_temp) return fail(ctx, RJP_ERR_ARG, "rjp_tavg: fields.d_temp is NULL");
  if (fields->dtype != RJP_F32 && fields->dtype != RJP_F64)
    return fail(ctx, RJP_ERR_ARG, "fields.dtype must be RJP_F32 (4) or RJP_F64 (8)");
  if (fields->nx <= 0 || fields->ny <= 0 || fields->nz <= 0)
    return fail(ctx, RJP_ERR_ARG, "grid dimensions must be positive");
  if ((fields->d_ylo == nullptr) != (fields->d_yhi == nullptr))
    return fail(ctx, RJP_ERR_ARG, "fields.d_ylo and d_yhi must both be set or both be NULL");
  if (!d_tavg || !d_work) return fail(ctx, RJP_ERR_ARG, "rjp_tavg: d_tavg / d_work is NULL");
  if (work_bytes < rjp::ff_scan_workspace_bytes(fields->nx, fields->ny, fields->nz, 1))
    return fail(ctx, RJP_ERR_WORKSPACE, "rjp_tavg: workspace smaller than rjp_ff_scan_workspace(.., 1)");
  RJP_HIP(ctx, rjp::tavg_launch(fields, d_tavg, (double*)d_work, (hipStream_t)stream));
  return RJP_OK;
}

int rjp_y_bounds(rjp_ctx* ctx, const rjp_fields* fields, int32_t* d_ylo, int32_t* d_yhi,
                 void* stream) {
  if (int r = bind(ctx)) return r;
  if (int r = check_fields(ctx, fields, false, true)) return r;
  if (!d_ylo || !d_yhi) return fail(ctx, RJP_ERR_ARG, "rjp_y_bounds: NULL output");
  RJP_HIP(ctx, rjp::y_bounds_launch(fields, d_ylo, d_yhi, (hipStream_t)stream));
  return RJP_OK;
}

size_t rjp_ff_scan_workspace(int32_t nx, int32_t ny, int32_t nz, int32_t n_epochs) {
  if (nx <= 0 || ny <= 0 || nz <= 0) return 0;
  return rjp::ff_scan_workspace_bytes(nx, ny, nz, n_epochs);
}

int rjp_occupied_cells(rjp_ctx* ctx, const int32_t* d_ylo, const int32_t* d_yhi, int64_t n_pix,
                       int64_t* h_count, void* stream) {
  if (int r = bind(ctx)) return r;
  if (!d_ylo || !d_yhi || !h_count || n_pix <= 0)
    return fail(ctx, RJP_ERR_ARG, "rjp_occupied_cells: NULL argument or n_pix <= 0");
  hipStream_t st = (hipStream_t)stream;
  if (!ctx->d_count) RJP_HIP(ctx, hipMalloc((void**)&ctx->d_count, sizeof(unsigned long long)));
  RJP_HIP(ctx, hipMemsetAsync(ctx->d_count, 0, sizeof(unsigned long long), st));
  RJP_HIP(ctx, rjp::occupied_launch(d_ylo, d_yhi, n_pix, ctx->d_count, st));
  unsigned long long v = 0;
  RJP_HIP(ctx, hipMemcpyAsync(&v, ctx->d_count, sizeof(v), hipMemcpyDeviceToHost, st));
  RJP_HIP(ctx, hipStreamSynchronize(st));
  *h_count = (int64_t)v;
  return RJP_OK;
}

// (the body of rjp_ff_scan; also the first half of rjp_ff_step)
static int ff_scan_impl(rjp_ctx* ctx, const rjp_fields* fields, const rjp_bursts* bursts,
                        const double* h_epochs_s, int32_t n_epochs, int32_t gff_mode,
                        double* d_sumA, double* d_em, double* d_tavg, void* d_work,
                        size_t work_bytes, void* stream) {
  if (!mode_ok(gff_mode)) return fail(ctx, RJP_ERR_ARG, "bad gff_mode");
  if (int r = check_fields(ctx, fields, false, true, gff_mode)) return r;
  if (int r = check_bursts(ctx, bursts, fields)) return r;
  if (!h_epochs_s || n_epochs < 1) return fail(ctx, RJP_ERR_ARG, "need >= 1 epoch");
  if (!d_sumA || !d_work) return fail(ctx, RJP_ERR_ARG, "d_sumA / d_work is NULL");
  // check_fields accepted the call on the strength of ONE complete layout; the layout the scan
  // really takes depends on what is asked for (d_em moves a tau-only field set to the compact
  // or wide kernels): every pointer THAT layout dereferences must be there
  switch (rjp::scan_layout(fields, gff_mode, d_em != nullptr)) {
    case rjp::LAY_TAU:
      if (d_tavg && !fields->d_temp)
        return fail(ctx, RJP_ERR_ARG, "rjp_ff_scan: fields hold the tau layout only; d_tavg needs "
                                      "fields.d_temp");
      break;
    case rjp::LAY_CMP:
      if (!fields->d_temp)
        return fail(ctx, RJP_ERR_ARG, "rjp_ff_scan: the compact layout needs fields.d_temp");
      break;
    default:
      if (!fields->d_nd || !fields->d_xi || !fields->d_temp || !fields->d_pf)
        return fail(ctx, RJP_ERR_ARG,
                    d_em ? "rjp_ff_scan: d_em needs fields.d_em0 or the complete wide set "
                           "(nd, xi, temp, pf); these fields hold the tau layout only"
                         : "rjp_ff_scan: fields nd/xi/temp/pf must be device pointers");
  }
  if (work_bytes < rjp::ff_scan_workspace_bytes(fields->nx, fields->ny, fields->nz, n_epochs))
    return fail(ctx, RJP_ERR_WORKSPACE, "rjp_ff_scan: workspace smaller than rjp_ff_scan_workspace()");
  hipStream_t st = (hipStream_t)stream;
  // epoch sweeps by launch-time moments (ff_moments.hip) when the caller provided the launch-time
  // range and the host-side accuracy check of the expansion passes
  const int mr = rjp::moments_plan(fields, bursts, h_epochs_s, n_epochs, gff_mode, d_em != nullptr,
                                   work_bytes, ctx->mom);
  if (mr) {
    if (int r = check_ts_range(ctx, fields, st)) return r;
  }
  if (mr == 2) {
    // a new (bursts, epochs) request: its coefficient tables are built and checked on the device
    const auto t0 = std::chrono::steady_clock::now();
    const double* src[1] = {ctx->mom.stage.data()};
    const size_t len[1] = {ctx->mom.stage.size()};
    double* dev[1];
    if (int r = stage_tables(ctx, st, src, len, 1, dev)) return r;
    const hipError_t e = rjp::moments_build(ctx->mom, dev[0], st);
    if (int r = finish_staged(ctx, st, e, "moments_build")) return r;
    ctx->mom.build_ms =
        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }
  if (mr && ctx->mom.ok) {
    if (d_tavg) {
      if (!fields->d_temp)
        return fail(ctx, RJP_ERR_ARG, "rjp_ff_scan: d_tavg on the tau layout needs fields.d_temp");
      RJP_HIP(ctx, rjp::tavg_launch(fields, d_tavg, (double*)d_work, st));
    }
    ctx->last_path = ctx->mom.path;
    if (ctx->mom.path == 2) {
      RJP_HIP(ctx, rjp::lt_run(fields, ctx->mom, n_epochs, d_sumA, (double*)d_work, work_bytes, st));
      return RJP_OK;
    }
    // the moment maps of a0 are model state (they depend on the launch-time bins and on which
    // jets have bursts, not on the epochs or the burst parameters): a caller-kept cache of the
    // selected shape replaces the pass over the grid; an empty or differently shaped one is
    // filled by this call
    double* mbuf = (double*)d_work;
    bool cached = false;
    if (fields->d_mom_cache && !d_em) {
      mbuf = fields->d_mom_cache;
      cached = fields->mom_cache_K == ctx->mom.K && fields->mom_cache_N == ctx->mom.N;
    }
    if (cached) ctx->last_path = 4;
    // (the chunk sums of a small map's split contraction: behind the moment maps' room in the
    // workspace -- moments_plan has checked that the workspace holds both)
    const int64_t npix_ = (int64_t)fields->nx * fields->nz;
    double* part = nullptr;
    if (rjp::moments_scan_workspace_bytes(npix_) > rjp::moments_workspace_bytes(npix_) &&
        work_bytes >= rjp::moments_scan_workspace_bytes(npix_))
      part = (double*)((char*)d_work + rjp::moments_workspace_bytes(npix_));
    hipError_t e = rjp::moments_run(fields, ctx->mom, n_epochs, d_sumA, mbuf, part, st,
                                    (const double*)fields->d_a0, 1.0, ctx->guard, cached);
    if (e == hipSuccess && d_em) {
      // the emission measure of every epoch: the same pass and tables with em0 as the weight
      // (em = sum (n x)^2 * csize*au/pc * pf, classes.py:1116-1118)
      const double em_scale = fields->csize_au * 149597870700.0 / 3.085677581491367e+16;
      e = rjp::moments_run(fields, ctx->mom, n_epochs, d_em, (double*)d_work, part, st,
                           (const double*)fields->d_em0, em_scale, ctx->guard);
    }
    if (e != hipSuccess) return fail(ctx, RJP_ERR_HIP, "moments_run", e);
    return RJP_OK;
  }
  if (rjp::chi_table_plan(fields, bursts, h_epochs_s, n_epochs, gff_mode, d_em != nullptr,
                          work_bytes, ctx->chi) &&
      (d_tavg == nullptr || ctx->chi.wide)) {      // (T_avg with the scan: the wide kernel only)
    // single epoch on the tau layout: the burst factor from a table in LDS (ff_scan_tab.hip)
    if (int r = check_ts_range(ctx, fields, st)) return r;
    ctx->last_path = 3;
    const double* src[1] = {ctx->chi.stage.data()};
    const size_t len[1] = {ctx->chi.stage.size()};
    double* dev[1];
    if (int r = stage_tables(ctx, st, src, len, 1, dev)) return r;
    return finish_staged(ctx, st, rjp::chi_table_scan(fields, ctx->chi, dev[0], h_epochs_s[0],
                                                      gff_mode, d_sumA, d_em, d_tavg,
                                                      (double*)d_work, work_bytes, ctx->guard, st),
                         "chi_table_scan");
  }
  ctx->last_path = 0;
  rjp::ScanPlan plan;
  rjp::ff_scan_plan(fields, bursts, h_epochs_s, n_epochs, gff_mode, d_em != nullptr, plan);
  double* d_ext = nullptr;
  if (!plan.ext.empty()) {
    const double* src[1] = {plan.ext.data()};
    const size_t len[1] = {plan.ext.size()};
    double* dev[1];
    if (int r = stage_tables(ctx, st, src, len, 1, dev)) return r;
    d_ext = dev[0];
  }
  const hipError_t e = rjp::ff_scan_run(fields, bursts, plan, d_ext, h_epochs_s, n_epochs,
                                        gff_mode, d_sumA, d_em, d_tavg, (double*)d_work, st);
  // the staged slot is marked busy up to here on EVERY path (its upload, and whatever was
  // enqueued before a failure, may still be in flight)
  if (d_ext) return finish_staged(ctx, st, e, "ff_scan_run");
  if (e != hipSuccess) return fail(ctx, RJP_ERR_HIP, "ff_scan_run", e);
  return RJP_OK;
}

int rjp_ff_scan(rjp_ctx* ctx, const rjp_fields* fields, const rjp_bursts* bursts,
                const double* h_epochs_s, int32_t n_epochs, int32_t gff_mode, double* d_sumA,
                double* d_em, double* d_tavg, void* d_work, size_t work_bytes, void* stream) {
  if (int r = bind(ctx)) return r;
  return ff_scan_impl(ctx, fields, bursts, h_epochs_s, n_epochs, gff_mode, d_sumA, d_em, d_tavg,
                      d_work, work_bytes, stream);
}

int rjp_range_guard(rjp_ctx* ctx) {
  if (!ctx || !ctx->guard) return RJP_ERR_ARG;
  if (*(volatile int*)ctx->guard == 0) return 0;
  *(volatile int*)ctx->guard = 0;
  for (auto& k : ctx->range_ok) k = rjp_ctx::RangeKey();
  ctx->err = kGuardMsg;
  return 1;
}

int rjp_last_scan_path(const rjp_ctx* ctx, double* worst_rel_err, int32_t* moment_shape) {
  if (!ctx) return RJP_ERR_ARG;
  const bool mom = ctx->last_path == 1 || ctx->last_path == 2 || ctx->last_path == 4;
  if (worst_rel_err) *worst_rel_err = mom ? ctx->mom.worst : ctx->last_path == 3 ? 2e-13 : 0.0;
  if (moment_shape) {
    // (path 3: the table's intervals per jet and its polynomial degree + 1)
    moment_shape[0] = mom ? ctx->mom.K : ctx->last_path == 3 ? ctx->chi.ni : 0;
    moment_shape[1] = mom ? ctx->mom.N : ctx->last_path == 3 ? 8 : 0;
  }
  return ctx->last_path;
}

double rjp_last_table_build_ms(const rjp_ctx* ctx) { return ctx ? ctx->mom.build_ms : 0.0; }

size_t rjp_moment_cache_bytes(int32_t nx, int32_t nz) {
  if (nx <= 0 || nz <= 0) return 0;
  return rjp::moments_workspace_bytes((int64_t)nx * nz);
}

size_t rjp_lt_rowoff_entries(int32_t nx, int32_t nz, int32_t K) {
  if (nx <= 0 || nz <= 0 || K < 1 || K > RJP_LT_MAX_K) return 0;
  return rjp::lt_rowoff_entries(nx, nz, K);
}

static int check_lt(rjp_ctx* ctx, const rjp_fields* f, int32_t K) {
  if (!f) return fail(ctx, RJP_ERR_ARG, "fields is NULL");
  if (f->dtype != RJP_F64 || !f->d_a0 || !f->d_ts)
    return fail(ctx, RJP_ERR_ARG, "launch-time-ordered layout: needs RJP_F64 fields with d_a0 and d_ts");
  if (f->nx <= 0 || f->ny <= 0 || f->nz <= 0 || f->ny >= 65536)
    return fail(ctx, RJP_ERR_ARG, "launch-time-ordered layout: grid dimensions must be positive, n_y < 65536");
  if (K < 1 || K > RJP_LT_MAX_K)
    return fail(ctx, RJP_ERR_ARG, "launch-time-ordered layout: 1 <= K <= 80");
  if (!(f->ts_hi >= f->ts_lo) || !std::isfinite(f->ts_lo) || !std::isfinite(f->ts_hi) ||
      (f->ts_lo == 0.0 && f->ts_hi == 0.0))
    return fail(ctx, RJP_ERR_ARG, "launch-time-ordered layout: fields.ts_lo / ts_hi (rjp_field_range) required");
  return RJP_OK;
}

int rjp_lt_count(rjp_ctx* ctx, const rjp_fields* fields, int32_t K, int32_t* d_rowoff,
                 int64_t* h_total_rows, void* stream) {
  if (int r = bind(ctx)) return r;
  if (int r = check_lt(ctx, fields, K)) return r;
  if (!d_rowoff || !h_total_rows) return fail(ctx, RJP_ERR_ARG, "rjp_lt_count: NULL output");
  hipStream_t st = (hipStream_t)stream;
  RJP_HIP(ctx, rjp::lt_count_launch(fields, K, d_rowoff, ctx->guard, st));
  int32_t total = 0;
  const size_t n = rjp::lt_rowoff_entries(fields->nx, fields->nz, K);
  RJP_HIP(ctx, hipMemcpyAsync(&total, d_rowoff + (n - 1), sizeof(int32_t), hipMemcpyDeviceToHost, st));
  RJP_HIP(ctx, hipStreamSynchronize(st));
  if (*(volatile int*)ctx->guard != 0) {
    *(volatile int*)ctx->guard = 0;
    return fail(ctx, RJP_ERR_ARG, "rjp_lt_count: fields.ts_lo / ts_hi do not contain every launch "
                                  "time of the cells the layout keeps (rjp_field_range)");
  }
  if (total < 0) return fail(ctx, RJP_ERR_ARG, "rjp_lt_count: more than 2^31 rows");
  *h_total_rows = total;
  return RJP_OK;
}

int rjp_lt_fill(rjp_ctx* ctx, const rjp_fields* fields, int32_t K, const int32_t* d_rowoff,
                void* d_cells, double* d_aux, void* stream) {
  if (int r = bind(ctx)) return r;
  if (int r = check_lt(ctx, fields, K)) return r;
  if (!d_rowoff || !d_cells || !d_aux) return fail(ctx, RJP_ERR_ARG, "rjp_lt_fill: NULL argument");
  RJP_HIP(ctx, rjp::lt_fill_launch(fields, K, d_rowoff, d_cells, d_aux, (hipStream_t)stream));
  return RJP_OK;
}

int rjp_time_ff_scan(rjp_ctx* ctx, const rjp_fields* fields, const rjp_bursts* bursts,
                     const double* h_epochs_s, int32_t n_epochs, int32_t gff_mode,
                     double* d_sumA, double* d_em, double* d_tavg, void* d_work,
                     size_t work_bytes, void* stream, int32_t reps, double* ms_avg) {
  if (!ms_avg || reps < 1) return fail(ctx, RJP_ERR_ARG, "rjp_time_ff_scan: bad reps/ms_avg");
  if (int r = bind(ctx)) return r;
  hipStream_t st = (hipStream_t)stream;
  RJP_HIP(ctx, hipEventRecord(ctx->ev0, st));
  for (int i = 0; i < reps; ++i) {
    int r = ff_scan_impl(ctx, fields, bursts, h_epochs_s, n_epochs, gff_mode, d_sumA, d_em,
                         d_tavg, d_work, work_bytes, stream);
    if (r != RJP_OK) return r;
  }
  RJP_HIP(ctx, hipEventRecord(ctx->ev1, st));
  RJP_HIP(ctx, hipEventSynchronize(ctx->ev1));
  float ms = 0.f;
  RJP_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  *ms_avg = (double)ms / reps;
  return RJP_OK;
}

size_t rjp_ff_maps_workspace(int64_t n_pix, int32_t n_epochs, int32_t n_chan) {
  if (n_pix <= 0 || n_epochs <= 0 || n_chan <= 0) return 0;
  return rjp::ff_maps_workspace_bytes(n_pix, n_epochs, n_chan);
}

static int ff_maps_impl(rjp_ctx* ctx, const double* d_sumA, const double* d_tavg, int64_t n_pix,
                        int32_t n_epochs, const double* h_ctau, const double* h_cflux,
                        int32_t n_chan, double* d_tau, double* d_flux, double* d_ftot,
                        void* d_work, size_t work_bytes, void* stream) {
  if (!d_sumA || !d_tavg || !h_ctau || !h_cflux)
    return fail(ctx, RJP_ERR_ARG, "rjp_ff_maps: NULL input");
  if (n_pix <= 0 || n_epochs <= 0 || n_chan <= 0)
    return fail(ctx, RJP_ERR_ARG, "rjp_ff_maps: sizes must be positive");
  if (d_ftot && (!d_work || work_bytes < rjp::ff_maps_workspace_bytes(n_pix, n_epochs, n_chan)))
    return fail(ctx, RJP_ERR_WORKSPACE, "rjp_ff_maps: workspace smaller than rjp_ff_maps_workspace()");
  hipStream_t st = (hipStream_t)stream;
  const double* src[2] = {h_ctau, h_cflux};
  const size_t len[2] = {(size_t)n_chan, (size_t)n_chan};
  double* dev[2];
  if (int r = stage_tables(ctx, st, src, len, 2, dev)) return r;
  return finish_staged(ctx, st, rjp::ff_maps_launch(d_sumA, d_tavg, n_pix, n_epochs, dev[0], dev[1], n_chan, d_tau,
                                   d_flux, d_ftot, (double*)d_work, st), "ff_maps_launch");
}

int rjp_ff_maps(rjp_ctx* ctx, const double* d_sumA, const double* d_tavg, int64_t n_pix,
                int32_t n_epochs, const double* h_ctau, const double* h_cflux, int32_t n_chan,
                double* d_tau, double* d_flux, double* d_ftot, void* d_work, size_t work_bytes,
                void* stream) {
  if (int r = bind(ctx)) return r;
  return ff_maps_impl(ctx, d_sumA, d_tavg, n_pix, n_epochs, h_ctau, h_cflux, n_chan, d_tau, d_flux,
                      d_ftot, d_work, work_bytes, stream);
}

int rjp_ff_step(rjp_ctx* ctx, const rjp_fields* fields, const rjp_bursts* bursts,
                const double* h_epochs_s, int32_t n_epochs, int32_t gff_mode,
                const double* d_tavg, const double* h_ctau, const double* h_cflux, int32_t n_chan,
                double* d_sumA, double* d_em, double* d_tau, double* d_flux, double* d_ftot,
                void* d_work, size_t work_bytes, void* d_work_maps, size_t work_maps_bytes,
                void* stream) {
  if (int r = bind(ctx)) return r;
  // every argument of the map stage is checked BEFORE the scan is enqueued: a step either runs
  // whole or not at all
  if (!fields) return fail(ctx, RJP_ERR_ARG, "fields is NULL");
  if (!d_tavg || !h_ctau || !h_cflux || n_chan < 1)
    return fail(ctx, RJP_ERR_ARG, "rjp_ff_step: NULL d_tavg / channel table or n_chan < 1");
  if (fields->nx <= 0 || fields->nz <= 0 || n_epochs < 1)
    return fail(ctx, RJP_ERR_ARG, "rjp_ff_step: bad grid or n_epochs < 1");
  const int64_t n_pix = (int64_t)fields->nx * fields->nz;
  if (d_ftot && (!d_work_maps ||
                 work_maps_bytes < rjp::ff_maps_workspace_bytes(n_pix, n_epochs, n_chan)))
    return fail(ctx, RJP_ERR_WORKSPACE, "rjp_ff_step: d_work_maps smaller than rjp_ff_maps_workspace()");
  if (int r = ff_scan_impl(ctx, fields, bursts, h_epochs_s, n_epochs, gff_mode, d_sumA, d_em,
                           nullptr, d_work, work_bytes, stream))
    return r;
  return ff_maps_impl(ctx, d_sumA, d_tavg, n_pix, n_epochs, h_ctau, h_cflux, n_chan, d_tau, d_flux,
                      d_ftot, d_work_maps, work_maps_bytes, stream);
}

int rjp_rrl_scan(rjp_ctx* ctx, const rjp_fields* fields, const rjp_bursts* bursts,
                 double time_s, const rjp_line* line, const double* h_nu, int32_t n_chan,
                 double* d_tau_rrl, void* stream) {
  if (int r = bind(ctx)) return r;
  if (int r = check_fields(ctx, fields, true)) return r;
  if (int r = check_bursts(ctx, bursts, fields)) return r;
  if (!line || !h_nu || n_chan < 1 || !d_tau_rrl)
    return fail(ctx, RJP_ERR_ARG, "rjp_rrl_scan: NULL line / nu / output or n_chan < 1");
  hipStream_t st = (hipStream_t)stream;
  const std::vector<double> ext = burst_ext_table(bursts);
  const double* src[2] = {h_nu, ext.data()};
  const size_t len[2] = {(size_t)n_chan, ext.size()};
  double* dev[2];
  if (int r = stage_tables(ctx, st, src, len, 2, dev)) return r;
  return finish_staged(ctx, st, rjp::rrl_scan_launch(fields, bursts, ext.empty() ? nullptr : dev[1], time_s, line,
                                    h_nu, dev[0], n_chan, d_tau_rrl, st), "rrl_scan_launch");
}

int rjp_ff_cells(rjp_ctx* ctx, const rjp_fields* fields, const rjp_bursts* bursts, double time_s,
                 int32_t gff_mode, const double* h_ctau, int32_t n_chan, double* d_tau_cells,
                 void* stream) {
  if (int r = bind(ctx)) return r;
  if (int r = check_fields(ctx, fields, false)) return r;
  if (int r = check_bursts(ctx, bursts, fields)) return r;
  if (!h_ctau || n_chan < 1 || !d_tau_cells)
    return fail(ctx, RJP_ERR_ARG, "rjp_ff_cells: NULL table / output or n_chan < 1");
  if (gff_mode != RJP_GFF_SCALAR && gff_mode != RJP_GFF_POWERLAW)
    return fail(ctx, RJP_ERR_ARG, "bad gff_mode");
  hipStream_t st = (hipStream_t)stream;
  const std::vector<double> ext = burst_ext_table(bursts);
  const double* src[2] = {h_ctau, ext.data()};
  const size_t len[2] = {(size_t)n_chan, ext.size()};
  double* dev[2];
  if (int r = stage_tables(ctx, st, src, len, 2, dev)) return r;
  return finish_staged(ctx, st, rjp::ff_cells_launch(fields, bursts, ext.empty() ? nullptr : dev[1], time_s,
                                    gff_mode, dev[0], n_chan, d_tau_cells, st), "ff_cells_launch");
}

int rjp_rrl_cells(rjp_ctx* ctx, const rjp_fields* fields, const rjp_bursts* bursts,
                  double time_s, const rjp_line* line, const double* h_nu, int32_t n_chan,
                  double* d_tau_cells, void* stream) {
  if (int r = bind(ctx)) return r;
  if (int r = check_fields(ctx, fields, true)) return r;
  if (int r = check_bursts(ctx, bursts, fields)) return r;
  if (!line || !h_nu || n_chan < 1 || !d_tau_cells)
    return fail(ctx, RJP_ERR_ARG, "rjp_rrl_cells: NULL line / nu / output or n_chan < 1");
  hipStream_t st = (hipStream_t)stream;
  const std::vector<double> ext = burst_ext_table(bursts);
  const double* src[2] = {h_nu, ext.data()};
  const size_t len[2] = {(size_t)n_chan, ext.size()};
  double* dev[2];
  if (int r = stage_tables(ctx, st, src, len, 2, dev)) return r;
  return finish_staged(ctx, st, rjp::rrl_cells_launch(fields, bursts, ext.empty() ? nullptr : dev[1], time_s,
                                     line, h_nu, dev[0], n_chan, d_tau_cells, st), "rrl_cells_launch");
}

int rjp_rrl_maps(rjp_ctx* ctx, const double* d_tau_rrl, const double* d_tau_ff,
                 const double* d_tavg, const double* d_flux_ff, int64_t n_pix,
                 const double* h_cflux_rrl, const double* h_hnu_k, int32_t n_chan,
                 double* d_flux, double* d_ftot, void* d_work, size_t work_bytes, void* stream) {
  if (int r = bind(ctx)) return r;
  if (!d_tau_rrl || !d_tau_ff || !d_tavg || !h_cflux_rrl || !h_hnu_k)
    return fail(ctx, RJP_ERR_ARG, "rjp_rrl_maps: NULL input");
  if (n_pix <= 0 || n_chan <= 0) return fail(ctx, RJP_ERR_ARG, "rjp_rrl_maps: sizes must be positive");
  if (d_ftot && (!d_work || work_bytes < rjp::ff_maps_workspace_bytes(n_pix, 1, n_chan)))
    return fail(ctx, RJP_ERR_WORKSPACE, "rjp_rrl_maps: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const double* src[2] = {h_cflux_rrl, h_hnu_k};
  const size_t len[2] = {(size_t)n_chan, (size_t)n_chan};
  double* dev[2];
  if (int r = stage_tables(ctx, st, src, len, 2, dev)) return r;
  return finish_staged(ctx, st, rjp::rrl_maps_launch(d_tau_rrl, d_tau_ff, d_tavg, d_flux_ff, n_pix, dev[0], dev[1],
                                    n_chan, d_flux, d_ftot, (double*)d_work, st), "rrl_maps_launch");
}

int rjp_build_fields(rjp_ctx* ctx, const rjp_geometry* gm, int dtype, void* d_nd, void* d_xi,
                     void* d_temp, void* d_pf, void* d_ts, void* d_vy, double* d_ff_raw,
                     double* d_areas_raw, double* d_vx_raw, double* d_vz_raw, void* d_em0,
                     void* d_a0, int32_t a0_mode, void* stream) {
  if (int r = bind(ctx)) return r;
  if (!gm) return fail(ctx, RJP_ERR_ARG, "geometry is NULL");
  if (dtype != RJP_F32 && dtype != RJP_F64) return fail(ctx, RJP_ERR_ARG, "bad dtype tag");
  if (gm->nx <= 0 || gm->ny <= 0 || gm->nz <= 0 || !(gm->csize > 0))
    return fail(ctx, RJP_ERR_ARG, "bad grid in geometry");
  if ((d_em0 || d_a0) && dtype != RJP_F64)
    return fail(ctx, RJP_ERR_ARG, "rjp_build_fields: d_em0 / d_a0 are written for RJP_F64 storage "
                                  "only (float storage goes through rjp_compact_fields' range check)");
  if (d_a0 && !mode_ok(a0_mode)) return fail(ctx, RJP_ERR_ARG, "rjp_build_fields: bad a0_mode");
  const double au = 149597870700.0, d2r = M_PI / 180.0;
  rjp::GeomDev g;
  g.nx = gm->nx; g.ny = gm->ny; g.nz = gm->nz; g.ccw = gm->rotation_ccw;
  g.nx_total = gm->nx_total > 0 ? gm->nx_total : gm->nx;
  g.ix0 = gm->ix0;
  if (g.ix0 < 0 || g.ix0 + g.nx > g.nx_total)
    return fail(ctx, RJP_ERR_ARG, "rjp_build_fields: x-slab [ix0, ix0+nx) outside [0, nx_total)");
  g.cs = gm->csize;
  // numpy.radians(x) = x * (pi/180); cos/sin in libm double, as maths/geometry.py:249-253
  const double a = (gm->inc - 90.) * d2r, b = gm->pa * d2r;
  g.ca = cos(a); g.sa = sin(a); g.cb = cos(b); g.sb = sin(b);
  const double a2 = (90. - gm->inc) * d2r, b2 = -gm->pa * d2r;
  g.ca2 = cos(a2); g.sa2 = sin(a2); g.cb2 = cos(b2); g.sb2 = sin(b2);
  g.w_0 = gm->w_0; g.r_0 = gm->r_0; g.mr0 = gm->mod_r_0; g.eps = gm->epsilon;
  g.R_1 = gm->R_1; g.R_2 = gm->R_2;
  g.gm = 6.6743e-11 * gm->M_star * 1.98847e30;
  g.v_lsr = gm->v_lsr;
  g.n_0 = gm->n_0; g.x_0 = gm->x_0; g.T_0 = gm->T_0; g.v_0 = gm->v_0;
  g.q_n = gm->q_n; g.q_x = gm->q_x; g.q_T = gm->q_T; g.q_v = gm->q_v;
  g.qd_n = gm->qd_n; g.qd_x = gm->qd_x; g.qd_T = gm->qd_T; g.qd_v = gm->qd_v;
  g.rb_frac = gm->rb_frac;
  {
    // maths/geometry.py:150-156
    const double mr0 = gm->mod_r_0 * au, r0 = gm->r_0 * au, v0 = gm->v_0 * 1e3;
    const double a = gm->qd_v, b = (1. - gm->q_v + gm->epsilon * gm->qd_v) / gm->epsilon;
    g.ts_pow = 1. - gm->q_v;
    g.ts_const = pow(mr0, gm->q_v) / (v0 * (1. - gm->q_v + gm->epsilon * gm->qd_v));
    g.ts_base = g.ts_const * pow(r0 + mr0 - r0, g.ts_pow);
    g.mr0_m = mr0; g.r0_m = r0;
    g.r1_m = gm->R_1 * au; g.r2_m = gm->R_2 * au; g.w0_m = gm->w_0 * au;
    g.hy_a = a; g.hy_b = b;
    g.hy_axis = 1. + gm->qd_v / (1. - gm->q_v);
    g.hy_k1 = g.hy_k2 = 0.0;
    g.ts_mode = d_ts ? 1 : 0;
    if (d_ts && a != 0.0) {
      // connection coefficients need Gamma(a-b) and 1/Gamma(a): degenerate when a - b or b
      // is an integer <= 0 (logarithmic cases) -> refuse, the caller evaluates ts itself
      const double amb = a - b;
      auto nonpos_int = [](double v) { return v < 0.5 && fabs(v - nearbyint(v)) < 1e-6; };
      if (nonpos_int(amb) || nonpos_int(b) || nonpos_int(b + 1.) || b == a)
        return fail(ctx, RJP_ERR_DEGENERATE,
                    "rjp_build_fields: degenerate 2F1 parameters (a-b or b a non-positive "
                    "integer); pass d_ts = NULL and upload host-computed launch times");
      g.hy_k1 = b / (b - a);
      g.hy_k2 = nonpos_int(a) ? 0.0 : tgamma(b + 1.) * tgamma(amb) / tgamma(a);
      g.ts_mode = 2;
    }
  }
  RJP_HIP(ctx, rjp::build_fields_launch(g, dtype, d_nd, d_xi, d_temp, d_pf, d_ts, d_vy, d_ff_raw,
                                        d_areas_raw, d_vx_raw, d_vz_raw, d_em0, d_a0, (int)a0_mode,
                                        (hipStream_t)stream));
  return RJP_OK;
}

int rjp_synth_fields(rjp_ctx* ctx, uint64_t seed, int32_t temp_mode, int32_t nz, int64_t cell0,
                     int64_t n, int dtype, void* d_nd, void* d_xi, void* d_temp, void* d_pf,
                     void* d_ts, void* d_vy, void* d_em0, void* d_a0, int32_t a0_mode,
                     void* stream) {
  if (int r = bind(ctx)) return r;
  if (n <= 0 || nz <= 0 || cell0 < 0) return fail(ctx, RJP_ERR_ARG, "rjp_synth_fields: bad range");
  if (dtype != RJP_F32 && dtype != RJP_F64) return fail(ctx, RJP_ERR_ARG, "bad dtype tag");
  if ((d_em0 || d_a0) && dtype != RJP_F64)
    return fail(ctx, RJP_ERR_ARG, "rjp_synth_fields: d_em0 / d_a0 are written for RJP_F64 storage only");
  if (d_a0 && !mode_ok(a0_mode)) return fail(ctx, RJP_ERR_ARG, "rjp_synth_fields: bad a0_mode");
  RJP_HIP(ctx, rjp::synth_launch(seed, temp_mode, nz, cell0, n, dtype, d_nd, d_xi, d_temp, d_pf,
                                 d_ts, d_vy, d_em0, d_a0, (int)a0_mode, (hipStream_t)stream));
  return RJP_OK;
}

}  // extern "C"
