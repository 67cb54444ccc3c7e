"""Device engine: owns the librjprt context of one GPU and the device-resident model fields.

PyTorch (ROCm) is used for device memory, streams and (elsewhere) torch.distributed only; all
arithmetic on the RT path happens inside librjprt's HIP kernels.  Nothing here falls back to
the CPU: without a GPU or without the built library, construction raises.
"""
import ctypes as C
import os

import numpy as np

from . import _constants as con
from . import _lib
from ._lib import RJP_F32, RJP_F64, RJP_GFF_POWERLAW, RJP_GFF_SCALAR  # noqa: F401


def _torch():
    import torch
    return torch


class DeviceFields:
    """The packed per-cell state in HBM (include/rjprt.h `rjp_fields`)."""

    def __init__(self, shape, dtype, csize_au, nd, xi, temp, pf, ts=None, vy=None,
                 ff_raw=None, areas_raw=None):
        self.shape = tuple(int(s) for s in shape)
        self.dtype = int(dtype)
        self.csize_au = float(csize_au)
        self.nd, self.xi, self.temp, self.pf, self.ts, self.vy = nd, xi, temp, pf, ts, vy
        self.ff_raw, self.areas_raw = ff_raw, areas_raw
        self.ylo = self.yhi = None          # optional occupied y-range per sightline (int32)
        self.em0 = None                     # optional compact scan field (rjp_fields.d_em0)
        self.a0 = None                      # optional tau scan field (rjp_fields.d_a0) ...
        self.a0_mode = 0                    # ... and the Gaunt mode it was built for
        self.occupied_cells = 0             # cells inside the occupied y-ranges (0 = unknown)
        self.ts_range = None                # optional (ts_lo, ts_hi) of the finite launch times
        self._ts_range_of = None            # ... and the `ts` tensor it was measured on
        self.lt = None                      # optional launch-time-ordered layout (RTEngine.build_lt)
        self.mom_cache = None               # optional cache of the launch-time moment maps of a0

    @property
    def ncells(self):
        return self.shape[0] * self.shape[1] * self.shape[2]

    @property
    def npix(self):
        return self.shape[0] * self.shape[2]

    def struct(self):
        f = _lib.Fields()
        f.d_nd, f.d_xi, f.d_temp, f.d_pf = (t.data_ptr() if t is not None else None for t in
                                            (self.nd, self.xi, self.temp, self.pf))
        f.d_em0 = self.em0.data_ptr() if self.em0 is not None else None
        f.d_a0 = self.a0.data_ptr() if self.a0 is not None else None
        f.a0_mode = int(self.a0_mode)
        if (self.ts_range is not None and self.ts is not None and
                self._ts_range_of == self.ts.data_ptr()):
            f.ts_lo, f.ts_hi = self.ts_range
        f.d_ts = self.ts.data_ptr() if self.ts is not None else None
        f.d_vy = self.vy.data_ptr() if self.vy is not None else None
        f.nx, f.ny, f.nz = self.shape
        f.dtype = self.dtype
        f.csize_au = self.csize_au
        f.d_ylo = self.ylo.data_ptr() if self.ylo is not None else None
        f.d_yhi = self.yhi.data_ptr() if self.yhi is not None else None
        f.occupied_cells = int(self.occupied_cells) if self.ylo is not None else 0
        lt = self.lt
        if (lt is not None and self.a0 is not None and self.ts is not None and
                lt["key"] == (self.a0.data_ptr(), self.ts.data_ptr(), f.ts_lo, f.ts_hi)):
            # (the layout belongs to the a0 / ts / launch-time range it was built from)
            f.d_lt_cells = lt["cells"].data_ptr()
            f.d_lt_rowoff = lt["rowoff"].data_ptr()
            f.d_lt_aux = lt["aux"].data_ptr()
            f.lt_K = int(lt["K"])
        return f

    def scan_fields(self, gff_mode, want_em=True):
        """Fields per cell one continuum grid pass streams: 2 on the tau layout (a0, ts; em0 as
        a third only with emission-measure maps), 3 on the compact one (em0, temp, ts), else the
        5 wide ones."""
        if (self.a0 is not None and self.a0_mode == gff_mode and self.dtype == RJP_F64 and
                (not want_em or self.em0 is not None)):
            return 3 if want_em else 2
        return 3 if self.em0 is not None else 5

    def nbytes(self, rrl=False, gff_mode=None, want_em=True):
        """Bytes one grid pass streams: the RRL scan reads the six wide fields; the continuum
        scan what `scan_fields` says (without `gff_mode`: the compact / wide figure)."""
        if rrl:
            n = 6
        elif gff_mode is None:
            n = 3 if self.em0 is not None else 5
        else:
            n = self.scan_fields(gff_mode, want_em)
        return n * self.ncells * self.dtype

    def drop_wide(self):
        """Free nd / xi / pf once the compact field is attached (continuum-only sweeps of
        grids that would not otherwise fit; the RRL and collapse=False calls need them)."""
        if self.em0 is None:
            raise ValueError("no compact layout attached")
        self.nd = self.xi = self.pf = None

    def drop_tau(self):
        """Detach the tau layout (scans fall back to the compact / wide one)."""
        self.a0 = None


def make_bursts(red, blue):
    """Build an rjp_bursts from per-jet lists of (t0_s, amp_rel, sigma_s)
    (classes.py:442-448: sigma = half_life * 2 / (2 sqrt(2 ln 2))).  Any number per jet, as
    the reference (classes.py:245-264)."""
    b = _lib.Bursts()
    b._keep = []
    for j, lst in enumerate((red, blue)):
        n = len(lst)
        b.n[j] = n
        cols = ([float(t0) for t0, _, _ in lst], [float(a) for _, a, _ in lst],
                [1.0 / (2.0 * float(sg) ** 2.0) for _, _, sg in lst])
        for name, col in zip(("t0", "amp_rel", "inv2s2"), cols):
            arr = (C.c_double * max(n, 1))(*col)
            b._keep.append(arr)
            getattr(b, name)[j] = C.cast(arr, C.POINTER(C.c_double))
    return b


class RTEngine:
    """One per process/rank: binds to `cuda:<device>`."""

    def __init__(self, device=0):
        torch = _torch()
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.RjprtError("no GPU visible to PyTorch: rajepy_amd needs an MI355X "
                                  "(there is no CPU fallback)")
        self.device_index = int(device)
        self.device = torch.device("cuda", self.device_index)
        torch.cuda.set_device(self.device)
        ctx = C.c_void_p()
        _lib.check(self.lib.rjp_ctx_create(self.device_index, C.byref(ctx)), None,
                   "rjp_ctx_create")
        self.ctx = ctx
        self._work = None
        self.use_compact = not (_lib.DEBUG and os.environ.get("RJP_NO_COMPACT"))
        self.use_tau = not (_lib.DEBUG and os.environ.get("RJP_NO_TAU"))
        # epoch sweeps by launch-time moments (rjp_fields.ts_lo / ts_hi): off = the epoch tiles
        self.use_moments = not (_lib.DEBUG and os.environ.get("RJP_NO_MOMENTS"))
        self.force_moments = False     # tests: skip the library's tiles-or-moments cost model
        self.use_lt = True             # False: ignore an attached launch-time-ordered layout
        # single-epoch scans of large maps on the tau layout take the burst factor from a table in
        # LDS (ff_scan_tab.hip; needs the launch-time range); False: always the Gaussians
        self.use_chi_table = True
        # keep the launch-time moment maps of a model that is swept repeatedly (2.7 GB at
        # 512 x 512 sightlines): from the second long sweep on only the contraction runs
        self.cache_moments = True
        self.last_moment_shape = (0, 0)

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.rjp_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers ---------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(_torch().cuda.current_stream(self.device).cuda_stream)

    def _empty(self, n, dtype):
        torch = _torch()
        td = {RJP_F32: torch.float32, RJP_F64: torch.float64}[dtype]
        return torch.empty(int(n), dtype=td, device=self.device)

    def _f64(self, *shape):
        torch = _torch()
        return torch.empty(*shape, dtype=torch.float64, device=self.device)

    def _workspace(self, nbytes):
        torch = _torch()
        if self._work is None or self._work.numel() < nbytes:
            self._work = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        return self._work

    def _workspace_maps(self, nbytes):
        """The map stages' own scratch (per-wave flux partials): a scan and a map stage of two
        consecutive epochs may then run side by side on two streams."""
        torch = _torch()
        if getattr(self, "_work_m", None) is None or self._work_m.numel() < nbytes:
            self._work_m = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        return self._work_m

    def synchronize(self):
        _torch().cuda.synchronize(self.device)

    # -- field producers ---------------------------------------------------------------------
    def upload_fields(self, nd, xi, temp, ff, areas, ts, red, vy=None, csize_au=1.0,
                      dtype=RJP_F64):
        """Host float64 grids (reference layout, classes.py:216-227) -> packed device state.
        `red` is a boolean grid (rr < 0)."""
        torch = _torch()
        shape = np.shape(nd)
        n = int(np.prod(shape))
        st = self._stream()

        def up(a):
            return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).ravel()).to(
                self.device)

        def pack(src, den=None, redt=None):
            dst = self._empty(n, dtype)
            _lib.check(self.lib.rjp_pack_field(
                self.ctx, src.data_ptr(), den.data_ptr() if den is not None else None,
                redt.data_ptr() if redt is not None else None, dst.data_ptr(), n, dtype, st),
                self.ctx, "rjp_pack_field")
            return dst

        red_t = torch.from_numpy(np.ascontiguousarray(red, dtype=np.uint8).ravel()).to(
            self.device)
        d_nd = pack(up(nd), None, red_t)
        d_xi = pack(up(xi))
        d_t = pack(up(temp))
        d_pf = pack(up(ff), up(areas))
        d_ts = pack(up(ts)) if ts is not None else None
        d_vy = pack(up(vy)) if vy is not None else None
        self.synchronize()
        return self.compact(DeviceFields(shape, dtype, csize_au, d_nd, d_xi, d_t, d_pf, d_ts,
                                         d_vy))

    def compact(self, fields):
        """Attach the compact scan layout (rjp_compact_fields): K1 then streams 3 fields
        (em0, temp, ts) instead of 5 -- bit-identical maps for f64 storage.  Fields with a
        negative path factor (nothing the reference's fill_factor / areas can produce) or,
        in f32 storage, a product outside the float range keep the wide layout.
        `RTEngine.use_compact = False` keeps every model on the wide layout (A/B runs; with
        RJP_DEBUG=1 the variable RJP_NO_COMPACT=1 sets it)."""
        torch = _torch()
        fields.em0 = None
        self._drop_derived_state(fields)
        if not self.use_compact:
            return fields
        em0 = self._empty(fields.ncells, fields.dtype)
        bad = torch.empty(1, dtype=torch.int64, device=self.device)
        fs = fields.struct()
        _lib.check(self.lib.rjp_compact_fields(self.ctx, C.byref(fs), em0.data_ptr(),
                                               bad.data_ptr(), self._stream()), self.ctx,
                   "rjp_compact_fields")
        if int(bad.item()) == 0:
            fields.em0 = em0
        return fields

    def tau_layout(self, fields, gff_mode):
        """Attach the tau scan layout for `gff_mode` (rjp_tau_field): a0 = em0 T^-1.5|-1.35,
        everything of a cell's optical depth that depends on neither frequency nor epoch.  K1
        then streams a0 and ts (16 B/cell; em0 as well only when EM maps are asked for) and
        returns bit-identical maps.  f64 storage with the compact field attached; anything else
        keeps its layout."""
        fields.a0 = None
        self._drop_derived_state(fields)
        if not self.use_tau or fields.dtype != RJP_F64 or fields.em0 is None:
            return fields
        a0 = self._f64(fields.ncells)
        fs = fields.struct()
        _lib.check(self.lib.rjp_tau_field(self.ctx, C.byref(fs), int(gff_mode), a0.data_ptr(),
                                          self._stream()), self.ctx, "rjp_tau_field")
        fields.a0, fields.a0_mode = a0, int(gff_mode)
        return fields

    @staticmethod
    def _drop_derived_state(fields):
        """What was built FROM a0 / em0 / ts goes whenever one of them is rebuilt: the
        launch-time-ordered layout, the cached moment maps, the unmasked launch-time copy.  (The
        caching allocator hands a rebuilt field the old one's address more often than not: these
        are never validated by pointer alone.)"""
        fields.lt = None
        fields.mom_cache = None
        fields._ts_unmasked = None

    def tavg(self, fields):
        """T_avg map of the model, nanmean_y(T where T > 0) -> device tensor [P] (rjp_tavg):
        depends on neither frequency nor epoch, so a model asks once."""
        nx, ny, nz = fields.shape
        out = self._f64(fields.npix)
        wb = self.lib.rjp_ff_scan_workspace(nx, ny, nz, 1)
        work = self._workspace(wb)
        fs = fields.struct()
        _lib.check(self.lib.rjp_tavg(self.ctx, C.byref(fs), out.data_ptr(), work.data_ptr(),
                                     work.numel(), self._stream()), self.ctx, "rjp_tavg")
        return out

    def launch_time_range(self, fields):
        """(min, max) of the finite launch times of `fields` (rjp_field_range), measured once
        per `ts` tensor: with it, scans of >= 12 epochs on the tau layout (EM maps: with em0) may
        take the moment path (include/rjprt.h `rjp_fields.ts_lo`)."""
        if fields.ts is None:
            return None
        if fields.ts_range is not None and fields._ts_range_of == fields.ts.data_ptr():
            return fields.ts_range
        part = self._f64(2 * _lib.RJP_RANGE_BLOCKS)
        _lib.check(self.lib.rjp_field_range(self.ctx, fields.ts.data_ptr(), fields.ncells,
                                            fields.dtype, part.data_ptr(), self._stream()),
                   self.ctx, "rjp_field_range")
        h = part.cpu().numpy().reshape(-1, 2)
        lo, hi = float(h[:, 0].min()), float(h[:, 1].max())
        fields.ts_range = (lo, hi) if np.isfinite(lo) and np.isfinite(hi) and hi >= lo else None
        fields._ts_range_of = fields.ts.data_ptr()
        return fields.ts_range

    def build_lt(self, fields, K=20):
        """Attach the launch-time-ordered layout of (a0, ts) to `fields` (rjp_lt_count +
        rjp_lt_fill; include/rjprt.h `rjp_fields.d_lt_cells`): every group of 64 sightlines
        bucketed by (jet, launch-time bin), so that epoch sweeps of 12-32 epochs accumulate their
        Chebyshev moments in registers -- no LDS atomics, no moment maps in HBM.  A one-off per
        model (two passes + scattered 16-byte writes: ~50 ms for 1.07e9 cells, ~1.2 x the bytes of
        a0 + ts resident): worth it for a model that is swept many times.  Rebuild after `a0` or
        `ts` change (a stale layout is ignored by `struct()`).  K bins per jet: fewer, wider bins
        pad less (rows are as long as the fullest of 64 lanes: 1.15 x at K = 16, 1.17 x at 20,
        1.22 x at 32 on the dense benchmark grid) but need a higher order for the same bursts
        (24 / 20 / 16 for the example's); 20 measured fastest there."""
        torch = _torch()
        fields.lt = None
        if fields.a0 is None or fields.ts is None or fields.dtype != RJP_F64:
            raise ValueError("the launch-time-ordered layout needs f64 fields with the tau "
                             "layout (a0) and launch times")
        if self.launch_time_range(fields) is None:
            raise ValueError("no finite launch time in the model")
        fs = fields.struct()
        nx, ny, nz = fields.shape
        n_off = self.lib.rjp_lt_rowoff_entries(nx, nz, int(K))
        if n_off == 0:
            raise ValueError("K must be 1..80")
        rowoff = torch.empty(n_off, dtype=torch.int32, device=self.device)
        total = C.c_int64()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev[0].record()
        _lib.check(self.lib.rjp_lt_count(self.ctx, C.byref(fs), int(K), rowoff.data_ptr(),
                                         C.byref(total), self._stream()), self.ctx, "rjp_lt_count")
        ev[1].record()
        # (allocating ~1.2 x the bytes of a0 + ts is the slow part of a first build: hipMalloc)
        cells = torch.empty(max(1, total.value) * 64 * 2, dtype=torch.float64, device=self.device)
        aux = torch.empty(3 * fields.npix, dtype=torch.float64, device=self.device)
        ev[2].record()
        _lib.check(self.lib.rjp_lt_fill(self.ctx, C.byref(fs), int(K), rowoff.data_ptr(),
                                        cells.data_ptr(), aux.data_ptr(), self._stream()),
                   self.ctx, "rjp_lt_fill")
        ev[3].record()
        torch.cuda.synchronize(self.device)
        kernels_ms = ev[0].elapsed_time(ev[1]) + ev[2].elapsed_time(ev[3])
        fields.lt = {"cells": cells, "rowoff": rowoff, "aux": aux, "K": int(K),
                     "rows": int(total.value), "build_ms": kernels_ms,
                     "build_with_allocation_ms": ev[0].elapsed_time(ev[3]),
                     "bytes": cells.numel() * 8,
                     "key": (fields.a0.data_ptr(), fields.ts.data_ptr(), fs.ts_lo, fs.ts_hi)}
        return fields.lt

    def compute_y_bounds(self, fields):
        """Attach the per-sightline occupied y-range to `fields` (rjp_y_bounds): later scans
        skip the rows no cell of which can contribute.  Recompute after changing a field."""
        torch = _torch()
        fields.ylo = fields.yhi = None
        fields.occupied_cells = 0
        lo = torch.empty(fields.npix, dtype=torch.int32, device=self.device)
        hi = torch.empty(fields.npix, dtype=torch.int32, device=self.device)
        fs = fields.struct()
        _lib.check(self.lib.rjp_y_bounds(self.ctx, C.byref(fs), lo.data_ptr(), hi.data_ptr(),
                                         self._stream()), self.ctx, "rjp_y_bounds")
        fields.ylo, fields.yhi = lo, hi
        # (hint for the tiles-or-moments choice of long epoch sweeps, include/rjprt.h; summed by
        # the library: the first use of the equivalent torch ops costs ~0.2 s of lazy loading)
        n = C.c_int64()
        _lib.check(self.lib.rjp_occupied_cells(self.ctx, lo.data_ptr(), hi.data_ptr(),
                                               fields.npix, C.byref(n), self._stream()),
                   self.ctx, "rjp_occupied_cells")
        fields.occupied_cells = int(n.value)
        return lo, hi

    def replace_field(self, fields, name, host_array):
        """Re-upload one plain field (the reference's public setters: ts, ion_fraction,
        temperature; classes.py:857-859, 938-940, 998-1000)."""
        torch = _torch()
        assert name in ("xi", "temp", "ts", "vy")
        src = torch.from_numpy(np.ascontiguousarray(host_array, dtype=np.float64).ravel()).to(
            self.device)
        if src.numel() != fields.ncells:
            raise ValueError("grid shape mismatch")
        dst = self._empty(fields.ncells, fields.dtype)
        _lib.check(self.lib.rjp_pack_field(self.ctx, src.data_ptr(), None, None,
                                           dst.data_ptr(), fields.ncells, fields.dtype,
                                           self._stream()), self.ctx, "rjp_pack_field")
        self.synchronize()
        setattr(fields, name, dst)
        if name in ("ts", "xi", "temp"):
            fields.lt = None                    # the launch-time-ordered layout holds (a0, ts)
            fields.mom_cache = None             # ... and the moment maps are sums over them
        if name == "ts":
            # what was measured on / derived from the old launch times (the new tensor may well
            # sit at the old one's address: never key these on the pointer alone)
            fields.ts_range = fields._ts_range_of = None
            fields._ts_unmasked = None
        had_tau = fields.a0 is not None
        if name == "xi" and fields.em0 is not None:
            self.compact(fields)                # em0 holds (nd xi)^2 pf
        if had_tau and name in ("xi", "temp"):
            self.tau_layout(fields, fields.a0_mode)     # a0 holds em0 T^-1.5|-1.35
        if fields.ylo is not None and name in ("xi", "temp"):
            self.compute_y_bounds(fields)       # the occupied range depends on these fields

    def _direct_em0(self, n, dtype):
        """f64 producers write the compact scan field in their own pass (f32 goes through
        rjp_compact_fields and its range check)."""
        if dtype != RJP_F64 or not self.use_compact:
            return None
        return self._f64(n)

    def _direct_a0(self, n, dtype, tau_mode):
        """... and the tau scan field when a Gaunt mode is named."""
        if tau_mode is None or dtype != RJP_F64 or not (self.use_compact and self.use_tau):
            return None
        return self._f64(n)

    def build_fields(self, geom, dtype=RJP_F64, want_ts=True, want_vy=True, want_raw=True,
                     want_vxz=False, want_wide=True, tau_mode=None):
        """K4: geometry -> packed fields on the device (`geom` is a _lib.Geometry).
        `want_wide=False` (f64, continuum only) skips nd / xi / pf: 24 B/cell resident.
        `tau_mode` = the model's Gaunt mode: K4 writes the tau scan field a0 in the same pass."""
        n = geom.nx * geom.ny * geom.nz
        em0 = self._direct_em0(n, dtype)
        a0 = self._direct_a0(n, dtype, tau_mode)
        if not want_wide and em0 is None:
            raise ValueError("want_wide=False needs the compact layout (f64 storage)")
        nd, xi, pf = ((self._empty(n, dtype) for _ in range(3)) if want_wide
                      else (None, None, None))
        temp = self._empty(n, dtype)
        ts = self._empty(n, dtype) if want_ts else None
        vy = self._empty(n, dtype) if want_vy else None
        ffr = self._f64(n) if want_raw else None
        arr = self._f64(n) if want_raw else None
        vxr = self._f64(n) if want_vxz else None
        vzr = self._f64(n) if want_vxz else None
        ptr = lambda t: t.data_ptr() if t is not None else None
        _lib.check(self.lib.rjp_build_fields(
            self.ctx, C.byref(geom), dtype, ptr(nd), ptr(xi), temp.data_ptr(), ptr(pf), ptr(ts),
            ptr(vy), ptr(ffr), ptr(arr), ptr(vxr), ptr(vzr), ptr(em0), ptr(a0),
            int(tau_mode or 0), self._stream()), self.ctx, "rjp_build_fields")
        out = DeviceFields((geom.nx, geom.ny, geom.nz), dtype, geom.csize, nd, xi, temp, pf,
                           ts, vy, ffr, arr)
        out.vx_raw, out.vz_raw = vxr, vzr
        if em0 is not None:
            out.em0 = em0
            if a0 is not None:
                out.a0, out.a0_mode = a0, int(tau_mode)
            return out
        out = self.compact(out)
        return self.tau_layout(out, tau_mode) if tau_mode is not None else out

    def build_wide(self, fields, geom):
        """Attach the wide fields a lean model left out (K4 again, writing nd / xi / pf / vy only):
        what the RRL scan, the collapse=False kernels, the grid accessors and the `ion_fraction` /
        `vel` setters need -- a continuum pipeline never asks (classes.JetModel._wide_fields)."""
        n = fields.ncells
        new = {k: self._empty(n, fields.dtype) for k in ("nd", "xi", "pf", "vy")
               if getattr(fields, k) is None}
        if not new:
            return fields
        ptr = lambda k: new[k].data_ptr() if k in new else None
        _lib.check(self.lib.rjp_build_fields(
            self.ctx, C.byref(geom), fields.dtype, ptr("nd"), ptr("xi"), None, ptr("pf"), None,
            ptr("vy"), None, None, None, None, None, None, 0, self._stream()), self.ctx,
            "rjp_build_fields")
        for k, t in new.items():
            setattr(fields, k, t)
        return fields

    def synth_fields(self, shape, seed, temp_mode=0, dtype=RJP_F64, csize_au=0.5,
                     with_vy=False, cell0=0, wide=True, tau_mode=None, with_em0=True):
        """Measurement harness: dense synthetic fields generated on the device
        (SURVEY.md 8(d)); `shape` may be a sub-block starting at flat cell `cell0` of a
        grid whose z-extent is shape[2].  `wide=False` (f64) generates only the compact scan
        layout (em0, temp, ts): 24 B/cell.  `tau_mode`: also the tau scan field a0 for that
        Gaunt mode, in the same pass; with `with_em0=False` as well the generator leaves a0,
        temp, ts (scans without emission-measure maps: 16 B/cell streamed)."""
        nx, ny, nz = shape
        n = nx * ny * nz
        em0 = self._direct_em0(n, dtype)
        a0 = self._direct_a0(n, dtype, tau_mode)
        if not wide and em0 is None:
            raise ValueError("wide=False needs the compact layout (f64 storage)")
        if not with_em0:
            if wide or a0 is None:
                raise ValueError("with_em0=False is for wide=False with a tau_mode (f64 storage)")
            em0 = None
        nd, xi, pf = ((self._empty(n, dtype) for _ in range(3)) if wide else (None, None, None))
        temp, ts = self._empty(n, dtype), self._empty(n, dtype)
        vy = self._empty(n, dtype) if with_vy else None
        ptr = lambda t: t.data_ptr() if t is not None else None
        _lib.check(self.lib.rjp_synth_fields(
            self.ctx, int(seed), int(temp_mode), int(nz), int(cell0), int(n), dtype,
            ptr(nd), ptr(xi), temp.data_ptr(), ptr(pf), ts.data_ptr(), ptr(vy), ptr(em0),
            ptr(a0), int(tau_mode or 0), self._stream()), self.ctx, "rjp_synth_fields")
        out = DeviceFields(shape, dtype, csize_au, nd, xi, temp, pf, ts, vy)
        if a0 is not None:
            out.a0, out.a0_mode = a0, int(tau_mode)
        if em0 is not None:
            out.em0 = em0
        if em0 is not None or not with_em0:
            return out
        out = self.compact(out)
        return self.tau_layout(out, tau_mode) if tau_mode is not None else out

    # -- K1 / K2 -----------------------------------------------------------------------------
    def _scan_struct(self, fields, bursts, n_epochs=1):
        """`rjp_fields` for a free-free scan.  A model with bursts in ONE jet only scans a copy
        of the launch times in which the NaNs of the other jet's cells are cleared
        (rjp_unmask_launch_times): the reference's burst-less jet has a constant mass-loss rate,
        so a NaN launch time does not drop its cells (classes.py:232-233, 442-448).  The copy
        is kept with the fields and rebuilt when `ts` or the flag-carrying field changes."""
        if bursts is not None and (
                (self.use_moments and n_epochs >= 12 and fields.a0 is not None) or
                (self.use_chi_table and n_epochs == 1 and fields.dtype == RJP_F64 and
                 fields.ts is not None)):
            self.launch_time_range(fields)
        fs = fields.struct()
        if n_epochs == 1 and not self.use_chi_table:
            fs.ts_lo = fs.ts_hi = 0.0
        if not (self.use_lt and self.use_moments):
            fs.d_lt_cells = fs.d_lt_rowoff = fs.d_lt_aux = None
            fs.lt_K = 0
        if not self.use_moments and n_epochs != 1:
            fs.ts_lo = fs.ts_hi = 0.0
        elif self.force_moments:
            fs.occupied_cells = -1
        if bursts is None or fields.ts is None:
            return fs
        n_r, n_b = int(bursts.n[0]), int(bursts.n[1])
        if (n_r == 0) == (n_b == 0):
            return fs
        jet = 0 if n_r == 0 else 1
        flag = fields.a0 if fields.a0 is not None else (fields.em0 if fields.em0 is not None
                                                        else fields.nd)
        # (the copy's replacement value is the lower end of the launch-time range, so that the
        # copy obeys the range it is scanned under: measure the range first, key the copy on it)
        rng = self.launch_time_range(fields)
        full = fields.struct()
        key = (jet, fields.ts.data_ptr(), flag.data_ptr(), rng)
        cached = getattr(fields, "_ts_unmasked", None)
        if cached is None or cached[0] != key:
            out = self._empty(fields.ncells, fields.dtype)
            _lib.check(self.lib.rjp_unmask_launch_times(self.ctx, C.byref(full), jet,
                                                        out.data_ptr(), self._stream()),
                       self.ctx, "rjp_unmask_launch_times")
            cached = fields._ts_unmasked = (key, out)
        fs.d_ts = cached[1].data_ptr()
        return fs

    def ff_scan(self, fields, bursts, epochs_s, gff_mode, want_em=True, out=None,
                want_tavg=True):
        """-> (sumA[E,P], em[E,P] or None, tavg[P] or None) device tensors (float64).
        `want_tavg=False`: no T_avg map (a caller that keeps the model's map from `tavg()`; on
        the tau layout the scan then never reads the temperature field)."""
        E = len(epochs_s)
        P = fields.npix
        nx, ny, nz = fields.shape
        if out is None:
            sumA = self._f64(E, P)
            em = self._f64(E, P) if want_em else None
            tavg = self._f64(P) if want_tavg else None
        else:
            sumA, em, tavg = out
        wb = self.lib.rjp_ff_scan_workspace(nx, ny, nz, E)
        work = self._workspace(wb)
        fs = self._scan_struct(fields, bursts, E)
        mkey = self._attach_moment_cache(fields, bursts, fs, E, em is not None)
        ep = _lib.dbl_array(epochs_s)
        _lib.check(self.lib.rjp_ff_scan(
            self.ctx, C.byref(fs), C.byref(bursts) if bursts is not None else None, ep, E,
            int(gff_mode), sumA.data_ptr(), em.data_ptr() if em is not None else None,
            tavg.data_ptr() if tavg is not None else None, work.data_ptr(), work.numel(),
            self._stream()), self.ctx, "rjp_ff_scan")
        if mkey is not None:
            self._note_moment_sweep(fields, mkey)
        return sumA, em, tavg

    def ff_step(self, fields, bursts, epochs_s, gff_mode, tavg, ctau, cflux, out):
        """K1 + K2 from ONE call into the library (rjp_ff_step): the scan of `epochs_s` and the map
        stage for the channels of (ctau, cflux), for callers whose step is shorter than two trips
        through ctypes (x-slabs of a sharded grid, small models).  `tavg`: the model's T_avg map;
        `out` = (sumA[E,P], em[E,P] | None, tau[E,F,P] | None, flux[E,F,P] | None,
        ftot[E,F] | None) device tensors."""
        sumA, em, tau, flux, ftot = out
        E, P, F = len(epochs_s), fields.npix, len(ctau)
        nx, ny, nz = fields.shape
        work = self._workspace(self.lib.rjp_ff_scan_workspace(nx, ny, nz, E))
        wm = self._workspace_maps(self.lib.rjp_ff_maps_workspace(P, E, F)) \
            if ftot is not None else None
        fs = self._scan_struct(fields, bursts, E)
        mkey = self._attach_moment_cache(fields, bursts, fs, E, em is not None)
        ptr = lambda t: t.data_ptr() if t is not None else None
        _lib.check(self.lib.rjp_ff_step(
            self.ctx, C.byref(fs), C.byref(bursts) if bursts is not None else None,
            _lib.dbl_array(epochs_s), E, int(gff_mode), tavg.data_ptr(), _lib.dbl_array(ctau),
            _lib.dbl_array(cflux), F, sumA.data_ptr(), ptr(em), ptr(tau), ptr(flux), ptr(ftot),
            work.data_ptr(), work.numel(), ptr(wm), wm.numel() if wm is not None else 0,
            self._stream()), self.ctx, "rjp_ff_step")
        if mkey is not None:
            self._note_moment_sweep(fields, mkey)
        return out

    def range_guard(self):
        """True when a scan of this engine met finite launch times outside the range its fields
        declared (rjp_range_guard; the sums of those sightlines are NaN).  Synchronises."""
        self.synchronize()
        return self.lib.rjp_range_guard(self.ctx) == 1

    def _attach_moment_cache(self, fields, bursts, fs, n_epochs, want_em):
        """The caller-kept moment maps of include/rjprt.h `rjp_fields.d_mom_cache`.  A long sweep
        of a densely filled model gets the buffer at once: its own pass fills it, and every later
        sweep of the same fields (same launch times, same set of jets with bursts) is the
        contraction alone; a sparse model -- whose sweeps the library's cost model keeps on the
        epoch tiles -- reserves it only after a sweep HAS taken the moment path.  Returns the
        key the cache would be valid for (None: this scan cannot use one)."""
        if not (self.cache_moments and self.use_moments and n_epochs >= 12 and not want_em and
                bursts is not None and fields.a0 is not None and fields.ts is not None and
                fs.ts_lo != fs.ts_hi):
            return None
        key = (fields.a0.data_ptr(), fs.d_ts, fs.ts_lo, fs.ts_hi, int(bursts.n[0]) > 0,
               int(bursts.n[1]) > 0)
        mc = fields.mom_cache
        if mc is None and (fields.ylo is None or
                           2 * int(fields.occupied_cells) >= fields.ncells):
            mc = self._reserve_moment_cache(fields, key)
        if mc is not None and mc.get("buf") is not None:
            if mc["key"] != key:
                mc["K"] = mc["N"] = 0
                mc["key"] = key
            fs.d_mom_cache = mc["buf"].data_ptr()
            fs.mom_cache_K, fs.mom_cache_N = mc["K"], mc["N"]
            # the call may start rewriting the buffer in another shape and then fail: the recorded
            # shape is void until `_note_moment_sweep` says what the buffer holds afterwards
            mc["held"] = (mc["K"], mc["N"])
            mc["K"] = mc["N"] = 0
        return key

    def _reserve_moment_cache(self, fields, key):
        nx, _, nz = fields.shape
        nbytes = self.lib.rjp_moment_cache_bytes(nx, nz)
        try:
            buf = _torch().empty(nbytes // 8, dtype=_torch().float64, device=self.device)
        except RuntimeError:                      # no room for it: sweeps keep their pass
            return None
        fields.mom_cache = {"buf": buf, "K": 0, "N": 0, "key": key}
        return fields.mom_cache

    def _note_moment_sweep(self, fields, key):
        path = self.last_scan_path()[0]
        mc = fields.mom_cache
        if path in ("moments", "cached"):
            if mc is None or mc.get("buf") is None:
                self._reserve_moment_cache(fields, key)       # the next sweep fills it
            else:
                mc["K"], mc["N"] = self.last_moment_shape     # this sweep's pass filled it
                mc["key"] = key
        elif mc is not None:
            # another path ran (tiles, the launch-time-ordered layout): nothing was written -- the
            # buffer holds what it held; one that has never held moments is given back
            mc["K"], mc["N"] = mc.pop("held", (0, 0))
            if mc["K"] == 0:
                fields.mom_cache = None

    def last_scan_path(self):
        """('tiles' | 'moments' | 'lt' | 'table', worst relative error of the expansion) of the
        last rjp_ff_scan of this engine ('lt' = moments on the launch-time-ordered layout,
        'table' = single-epoch scan with the burst factor from a table in LDS)."""
        err = C.c_double()
        shape = (C.c_int32 * 2)()
        path = self.lib.rjp_last_scan_path(self.ctx, C.byref(err), shape)
        self.last_moment_shape = (int(shape[0]), int(shape[1]))      # (bins, order); (0, 0) = tiles
        return {0: "tiles", 1: "moments", 2: "lt", 3: "table", 4: "cached"}[path], err.value

    def last_table_build_ms(self):
        """Host wall time of the last coefficient-table build (a new bursts / epochs request)."""
        return float(self.lib.rjp_last_table_build_ms(self.ctx))

    def time_ff_scan(self, fields, bursts, epochs_s, gff_mode, reps=5, want_em=True,
                     want_tavg=True):
        """Average device time [ms] of one rjp_ff_scan (HIP events on the launch stream)."""
        E = len(epochs_s)
        P = fields.npix
        nx, ny, nz = fields.shape
        sumA, em = self._f64(E, P), (self._f64(E, P) if want_em else None)
        tavg = self._f64(P) if want_tavg else None
        wb = self.lib.rjp_ff_scan_workspace(nx, ny, nz, E)
        work = self._workspace(wb)
        fs = self._scan_struct(fields, bursts, E)
        ep = _lib.dbl_array(epochs_s)
        ms = C.c_double()
        _lib.check(self.lib.rjp_time_ff_scan(
            self.ctx, C.byref(fs), C.byref(bursts) if bursts is not None else None, ep, E,
            int(gff_mode), sumA.data_ptr(), em.data_ptr() if want_em else None,
            tavg.data_ptr() if want_tavg else None, work.data_ptr(), work.numel(),
            self._stream(), int(reps), C.byref(ms)), self.ctx, "rjp_time_ff_scan")
        return ms.value

    def ff_maps(self, sumA, tavg, ctau, cflux, want_tau=True, want_flux=True,
                want_ftot=True, out=None):
        """-> (tau[E,F,P], flux[E,F,P], ftot[E,F]) device tensors (None where not wanted)."""
        E, P = sumA.shape
        F = len(ctau)
        if out is None:
            tau = self._f64(E, F, P) if want_tau else None
            flux = self._f64(E, F, P) if want_flux else None
            ftot = self._f64(E, F) if want_ftot else None
        else:
            tau, flux, ftot = out
        wb = self.lib.rjp_ff_maps_workspace(P, E, F)
        work = self._workspace_maps(wb) if ftot is not None else None
        a, b = _lib.dbl_array(ctau), _lib.dbl_array(cflux)
        ptr = lambda t: t.data_ptr() if t is not None else None
        _lib.check(self.lib.rjp_ff_maps(
            self.ctx, sumA.data_ptr(), tavg.data_ptr(), P, E, a, b, F, ptr(tau), ptr(flux),
            ptr(ftot), ptr(work), work.numel() if work is not None else 0, self._stream()),
            self.ctx, "rjp_ff_maps")
        return tau, flux, ftot

    def ff_cells(self, fields, bursts, time_s, gff_mode, ctau):
        """collapse=False: per-cell free-free optical depths -> device tensor [F, N]."""
        F = len(ctau)
        out = self._f64(F, fields.ncells)
        fs = fields.struct()
        _lib.check(self.lib.rjp_ff_cells(
            self.ctx, C.byref(fs), C.byref(bursts) if bursts is not None else None,
            float(time_s), int(gff_mode), _lib.dbl_array(ctau), F, out.data_ptr(),
            self._stream()), self.ctx, "rjp_ff_cells")
        return out

    def rrl_cells(self, fields, bursts, time_s, line, nus):
        """collapse=False: per-cell RRL optical depths -> device tensor [F, N]."""
        F = len(nus)
        out = self._f64(F, fields.ncells)
        fs = fields.struct()
        _lib.check(self.lib.rjp_rrl_cells(
            self.ctx, C.byref(fs), C.byref(bursts) if bursts is not None else None,
            float(time_s), C.byref(line), _lib.dbl_array(nus), F, out.data_ptr(),
            self._stream()), self.ctx, "rjp_rrl_cells")
        return out

    # -- K3 ------------------------------------------------------------------------------------
    def rrl_scan(self, fields, bursts, time_s, line, nus):
        """-> tau_rrl[F,P] device tensor."""
        F = len(nus)
        tau = self._f64(F, fields.npix)
        fs = fields.struct()
        nu = _lib.dbl_array(nus)
        _lib.check(self.lib.rjp_rrl_scan(
            self.ctx, C.byref(fs), C.byref(bursts) if bursts is not None else None,
            float(time_s), C.byref(line), nu, F, tau.data_ptr(), self._stream()), self.ctx,
            "rjp_rrl_scan")
        return tau

    def rrl_maps(self, tau_rrl, tau_ff, tavg, flux_ff, cflux_rrl, hnu_k, want_ftot=True):
        """-> (flux[F,P], ftot[F])."""
        F, P = tau_rrl.shape
        flux = self._f64(F, P)
        ftot = self._f64(F) if want_ftot else None
        wb = self.lib.rjp_ff_maps_workspace(P, 1, F)
        work = self._workspace_maps(wb) if want_ftot else None
        a, b = _lib.dbl_array(cflux_rrl), _lib.dbl_array(hnu_k)
        ptr = lambda t: t.data_ptr() if t is not None else None
        _lib.check(self.lib.rjp_rrl_maps(
            self.ctx, tau_rrl.data_ptr(), tau_ff.data_ptr(), tavg.data_ptr(), ptr(flux_ff), P,
            a, b, F, flux.data_ptr(), ptr(ftot), ptr(work),
            work.numel() if work is not None else 0, self._stream()), self.ctx,
            "rjp_rrl_maps")
        return flux, ftot


# -- host-side per-channel coefficients (scalars; classes.py:1395-1397, 1473-1475, 1519-1521) --
def solid_angle(csize_au, dist_pc):
    return np.arctan((csize_au * con.au) / (dist_pc * con.parsec)) ** 2.


def ff_channel_coeffs(freqs, csize_au, dist_pc, gff_mode, gff_values=None):
    """ctau[f], cflux[f] of include/rjprt.h `rjp_ff_maps`."""
    freqs = np.atleast_1d(np.asarray(freqs, dtype=np.float64))
    path0 = csize_au * con.au * 1e2
    if gff_mode == RJP_GFF_SCALAR:
        g = np.asarray(gff_values, dtype=np.float64)
        ctau = 0.018 * freqs ** -2. * path0 * g
    else:
        ctau = 0.018 * freqs ** -2. * path0 * (11.95 * freqs ** -0.1)
    cflux = 2. * freqs ** 2. * con.k / con.c ** 2. * solid_angle(csize_au, dist_pc) / 1e-26
    return ctau, cflux


def rrl_channel_coeffs(freqs, csize_au, dist_pc):
    """cflux_rrl[f], hnu_k[f] of `rjp_rrl_maps` (physics.py:571-574; rrls.py:444-449)."""
    freqs = np.atleast_1d(np.asarray(freqs, dtype=np.float64))
    p1 = 2. * con.h * 1e7 * freqs ** 3. / (con.c * 1e2) ** 2.
    cflux = p1 * 1e-7 * 1e4 * solid_angle(csize_au, dist_pc) / 1e-26
    hnu_k = con.h * 1e7 * freqs / (con.k * 1e7)
    return cflux, hnu_k
