"""Host-side mirror of RaJePy's `JetModel` / `ContinuumRun` / `RRLRun` / `Pipeline`
(reference: classes.py) for the radiative-transfer path: same constructor arguments,
method names, argument meaning, return shapes/dtypes, file names and error behaviour.

What is different underneath: the 3-D grids live in HBM as five/six packed fields plus two
derived scan fields (`engine.DeviceFields`), are BUILT on the GPU (`rjp_build_fields`) and
every line-of-sight reduction runs in librjprt's HIP kernels through the C-ABI of
include/rjprt.h.  One grid pass serves every continuum channel of an epoch (the reference
re-streams the grid per channel), up to 32 uniformly spaced epochs share a pass (16 when
fewer are left, else tiles of 8 / 4 / 2 / 1), and everything that depends on neither
frequency nor epoch -- the temperature factor of the optical depth, the T_avg map -- is
evaluated once per model.  There is no CPU fallback for any of it.
"""
import collections
import os
import pickle
import runpy
import time as _time

import numpy as np

from . import _constants as con
from . import _lib
from . import fits as _fits
from . import logger
from .maths import geometry as mgeom
from .maths import physics as mphys
from .maths import rrls as mrrl
from .miscellaneous import functions as miscf

_STORAGE = {'f64': _lib.RJP_F64, 'f32': _lib.RJP_F32, 8: _lib.RJP_F64, 4: _lib.RJP_F32}


def geometry_struct(params, nx, ny, nz, ix0=0, nx_total=0):
    """`rjp_geometry` (include/rjprt.h) from a model params dict with derived keys; with
    `ix0`/`nx_total` only rows [ix0, ix0+nx) of an nx_total-wide grid are described."""
    g, t, pl, pr = (params['geometry'], params['target'], params['power_laws'],
                    params['properties'])
    s = _lib.Geometry()
    s.nx, s.ny, s.nz = int(nx), int(ny), int(nz)
    s.rotation_ccw = 1 if g["rotation"].lower() == 'ccw' else 0
    s.csize = params['grid']['c_size']
    s.inc, s.pa = g['inc'], g['pa']
    s.w_0, s.r_0, s.mod_r_0, s.epsilon = g['w_0'], g['r_0'], g['mod_r_0'], g['epsilon']
    s.R_1, s.R_2, s.M_star, s.v_lsr = t['R_1'], t['R_2'], t['M_star'], t['v_lsr']
    s.n_0, s.x_0, s.T_0, s.v_0 = pr['n_0'], pr['x_0'], pr['T_0'], pr['v_0']
    s.q_n, s.q_x, s.q_T, s.q_v = pl['q_n'], pl['q_x'], pl['q_T'], pl['q_v']
    s.qd_n, s.qd_x, s.qd_T, s.qd_v = pl['q^d_n'], pl['q^d_x'], pl['q^d_T'], pl['q^d_v']
    s.rb_frac = pr['mlr_rj'] / pr['mlr_bj']
    s.ix0, s.nx_total = int(ix0), int(nx_total)
    return s


def build_model_fields(model, geom, **kw):
    """K4 for `geom` (the whole grid of `model`, or an x-slab of it).  Only the library's
    dedicated refusal of a logarithmic 2F1 case (RJP_ERR_DEGENERATE: q^d_v and the
    launch-time exponent differ by an integer) is answered with the host evaluation of the
    launch times; every other failure (bad geometry, HIP error, ABI mismatch) propagates."""
    eng = model.engine
    # fill factor and areas as separate arrays are redundant for a model (areas is 1 wherever
    # it is not NaN, so the path factor ff/areas IS the fill factor: the accessors read `pf`):
    # two grid-sized f64 arrays less to allocate and to write
    kw.setdefault("want_raw", False)
    # the tau scan field for the model's Gaunt branch is written in K4's own pass
    kw.setdefault("tau_mode", model.gff_mode)
    try:
        return eng.build_fields(geom, model._dtype, want_ts=True, **kw)
    except _lib.RjprtError as exc:
        if exc.status != _lib.RJP_ERR_DEGENERATE:
            raise
    dev = eng.build_fields(geom, model._dtype, want_ts=False, **kw)
    ts = model._host_launch_times(geom.ix0, geom.ix0 + geom.nx)
    eng.replace_field(dev, "ts", ts)
    return dev


def _dist_info():
    """(rank, world, torch.distributed or None) -- world > 1 only inside an initialised
    process group (one process per GPU, e.g. under torchrun)."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size(), dist
    except ImportError:
        pass
    return 0, 1, None


def _to_host(tensor):
    """Device tensor -> NumPy array.  Large products go through page-locked memory (6x the
    pageable rate on MI355X: 19 ms instead of 121 ms for the 1.07 GB cfg4 cubes); the array
    returned is backed by that buffer, which PyTorch's host allocator recycles once the array
    is dropped."""
    import torch
    if tensor.is_cuda and tensor.numel() * tensor.element_size() >= (8 << 20):
        try:
            host = torch.empty(tensor.shape, dtype=tensor.dtype, pin_memory=True)
            host.copy_(tensor)
            return host.numpy()
        except RuntimeError:
            pass                                   # page-locked allocation refused: plain copy
    return tensor.cpu().numpy()


def _load_params_file(py_file, checker):
    if not os.path.exists(py_file):
        raise FileNotFoundError(py_file + " does not exist")
    params = runpy.run_path(py_file)["params"]
    err = checker(params)
    if err is not None:
        raise err
    return params


class JetModel:
    """Physical model of an ionised jet + its line-of-sight radiative transfer."""
    _arr_indexing = 'ij'

    # ------------------------------------------------------------------ construction ----
    @classmethod
    def load_model(cls, model_file, engine=None):
        """Restore a model saved by `JetModel.save` (classes.py:48-88)."""
        with open(os.path.expanduser(model_file), 'rb') as f:
            loaded = pickle.load(f)
        log = loaded.get('log')
        if log is None:
            log = logger.Log(os.path.expanduser('~') + os.sep + 'temp.log')
        new = cls(loaded["params"], log=log, engine=engine,
                  storage=loaded.get('storage', 'f64'))
        new.time = loaded['time']
        return new

    @staticmethod
    def lz_to_grid_dims(params):
        """Grid dimensions from the requested jet length l_z [arcsec] (classes.py:90-122)."""
        cs = params["grid"]["c_size"]
        g = params["geometry"]
        i_rads, pa_rads = np.radians(g["inc"]), np.radians(g["pa"])
        l_xz = params['grid']['l_z'] * params['target']['dist']
        xmax, ymax, zmax = (l_xz * np.sin(pa_rads), l_xz * np.tan(1.571 - i_rads),
                            l_xz * np.cos(pa_rads))
        rmax = mgeom.xyz_to_rwp(xmax, ymax, zmax, g["inc"], g["pa"])[0]
        wmax = mgeom.w_r(rmax, g["w_0"], g["mod_r_0"], g["r_0"], g["epsilon"])
        pad = 2 * int(np.ceil(np.abs(wmax / cs)))
        dims = [int(np.ceil(np.abs(v / cs))) + pad for v in (xmax, ymax, zmax)]
        return tuple(n if n % 2 == 0 else n + 1 for n in dims)

    @staticmethod
    def py_to_dict(py_file):
        """Model parameter file -> dict (classes.py:124-142)."""
        return _load_params_file(py_file, miscf.check_model_params)

    def __init__(self, params, log=None, engine=None, storage='f64'):
        if isinstance(params, dict):
            self._params = params
        elif isinstance(params, str):
            self._params = JetModel.py_to_dict(params)
        else:
            raise TypeError("Supplied arg params must be dict or file path (str)")
        if storage not in _STORAGE:
            raise ValueError("storage must be 'f64' or 'f32'")
        self._dtype = _STORAGE[storage]
        p = self._params
        self._name = p['target']['name']
        self._csize = p['grid']['c_size']

        # derived parameters, written back into the dict as the reference does
        # (classes.py:169-180)
        g, pl = p['geometry'], p['power_laws']
        g["mod_r_0"] = mgeom.mod_r_0(g['opang'], g['epsilon'], g['w_0'])
        pl["q_n"] = mphys.q_n(g["epsilon"], pl["q_v"])
        pl["q_tau"] = mphys.q_tau(g["epsilon"], pl["q_x"], pl["q_n"], pl["q_T"])

        if log is not None:
            self._log = log
        else:
            self._log = logger.Log(os.path.expanduser('~') + os.sep + 'temp.log',
                                   verbose=True)

        if p['grid'].get('l_z') is not None:
            nx, ny, nz = JetModel.lz_to_grid_dims(p)
            self.log.add_entry(
                "INFO", 'For a (bipolar) jet length of {:.1f}", cell size of {:.2f}au and '
                        'distance of {:.0f}pc, a grid size of (n_x, n_y, n_z) = ({}, {}, {}) '
                        'voxels is calculated'.format(p['grid']['l_z'], p["grid"]["c_size"],
                                                      p["target"]["dist"], nx, ny, nz))
        else:
            nx, ny, nz = ((p['grid'][k] + 1) // 2 * 2 for k in ('n_x', 'n_y', 'n_z'))
        p['grid']['n_x'], p['grid']['n_y'], p['grid']['n_z'] = nx, ny, nz
        self._nx, self._ny, self._nz = int(nx), int(ny), int(nz)

        pr = p["properties"]
        self._ss_jml_rb_frac = pr["mlr_rj"] / pr["mlr_bj"]
        self._ss_jml_bj = pr["mlr_bj"] * 1.989e30 / con.year          # classes.py:230-231
        self._ss_jml_rj = self._ss_jml_bj * self._ss_jml_rb_frac
        pr["n_0"] = mphys.n_0_from_mlr(pr["mlr_bj"], pr["v_0"], g["w_0"], pr["mu"],
                                       pl["q^d_n"], pl["q^d_v"], p["target"]["R_1"],
                                       p["target"]["R_2"])

        self._ejections = {}
        self._bursts = {'R': [], 'B': []}           # (t0 [s], amp_rel, sigma [s])
        for idx, t0 in enumerate(p['ejection']['t_0']):
            which = str(p['ejection']['which'][idx])
            for jet, ss in (('R', self._ss_jml_rj), ('B', self._ss_jml_bj)):
                if jet in which:
                    self.add_ejection_event(t0 * con.year, ss * p['ejection']['chi'][idx],
                                            p['ejection']['hl'][idx] * con.year, which=jet)
        self._time = 0. * con.year

        # device state (created lazily: constructing a model needs no GPU, using it does)
        self._engine = engine
        self._dev = None
        self._version = 0            # bumped whenever a field or the burst list changes
        self._scan_cache = collections.OrderedDict()   # time -> (sumA[1,P], em[1,P]) tensors
        self._tavg = None
        self._vxz = None
        self._rrl_cache = None
        self._dev_product = None     # device cube of the last product, for _save_cube

    # ------------------------------------------------------------------ bookkeeping ----
    def __str__(self):
        """The parameter table the reference prints and embeds in every FITS header
        (classes.py:268-361)."""
        p = self.params
        g, pl, pr, tg = p['geometry'], p['power_laws'], p['properties'], p['target']
        rows = [('epsilon', format(g['epsilon'], '+.3f')),
                ('opang', format(g['opang'], '+.0f') + ' deg'),
                ('q_v', format(pl['q_v'], '+.3f')), ('q_T', format(pl['q_T'], '+.3f')),
                ('q_x', format(pl['q_x'], '+.3f')), ('q_n', format(pl['q_n'], '+.3f')),
                ('q^d_v', format(pl['q^d_v'], '+.3f')), ('q^d_T', format(pl['q^d_T'], '+.3f')),
                ('q^d_x', format(pl['q^d_x'], '+.3f')), ('q^d_n', format(pl['q^d_n'], '+.3f')),
                ('q_tau', format(pl['q_tau'], '+.3f')),
                ('cell', format(p['grid']['c_size'], '.1f') + ' au'),
                ('w_0', format(g['w_0'], '.2f') + ' au'),
                ('r_0', format(g['r_0'], '.2f') + ' au'),
                ('v_0', format(pr['v_0'], '.0f') + ' km/s'),
                ('x_0', format(pr['x_0'], '.3f')),
                ('n_0', format(pr['n_0'], '.3e') + ' cm^-3'),
                ('T_0', format(pr['T_0'], '.0e') + ' K'),
                ('f_R2B', format(self._ss_jml_rb_frac, '.2e')),
                ('i', format(g['inc'], '+.1f') + ' deg'),
                ('theta', format(g['pa'], '+.1f') + ' deg'),
                ('D', format(tg['dist'], '+.0f') + ' pc'),
                ('M*', format(tg['M_star'], '+.1f') + ' Msol'),
                ('R_1', format(tg['R_1'], '+.1f') + ' au'),
                ('R_2', format(tg['R_2'], '+.1f') + ' au')]
        if len(p['ejection']['t_0']) > 0:
            rows.append(('t_now', format(self.time / con.year, '+.3f') + ' yr'))
        head = ('Parameter', 'Value')
        w1 = max(len(head[0]), *(len(r[0]) for r in rows)) + 2
        w2 = max(len(head[1]), *(len(r[1]) for r in rows)) + 2
        width = w1 + w2 + 3
        rule = '-' * width

        def line(cells, widths):
            return '|' + '|'.join(format(c, '^' + str(w)) for c, w in zip(cells, widths)) + '|\n'

        s = rule + '\n' + '/' + format('JET MODEL', '^' + str(width - 2)) + '/\n' + rule + '\n'
        s += line(head, (w1, w2)) + rule + '\n'
        for r in rows:
            s += line(r, (w1, w2))
        s += rule + '\n'
        s += '/' + format('BURSTS', '^' + str(width - 2)) + '/\n' + rule + '\n'
        ej = p["ejection"]
        if len(ej["t_0"]) == 0:
            return s + '|' + format(' None ', '-^' + str(width - 2)) + '|\n' + rule + '\n'
        base, extra = divmod(width - 4, 3)
        bw = [base + (1 if extra > 0 else 0), base + (1 if extra == 2 else 0), base]
        s += line(('t_0', 'FWHM', 'chi'), bw) + line(('[yr]', '[yr]', ''), bw) + rule + '\n'
        for i, t in enumerate(ej["t_0"]):
            s += line((format(t, '.2f'), format(ej["hl"][i], '.2f'),
                       format(ej["chi"][i], '.2f')), bw)
        return s + rule + '\n'

    @property
    def los_axis(self):
        return 1 if self._arr_indexing == 'ij' else 0

    @property
    def time(self):
        """Model time [s]."""
        return self._time

    @time.setter
    def time(self, new_time):
        self._time = new_time

    @property
    def log(self):
        return self._log

    @log.setter
    def log(self, new_log):
        self._log = new_log

    csize = property(lambda self: self._csize)
    nx = property(lambda self: self._nx)
    ny = property(lambda self: self._ny)
    nz = property(lambda self: self._nz)
    params = property(lambda self: self._params)
    name = property(lambda self: self._name)
    ejections = property(lambda self: self._ejections)

    def ss_jml(self, which):
        if which == 'R':
            return self._ss_jml_rj
        if which == 'B':
            return self._ss_jml_bj
        if 'R' in which and 'B' in which:
            return self._ss_jml_rj + self._ss_jml_bj
        raise ValueError("which must be one of 'R', 'B', or 'RB'")

    def add_ejection_event(self, t_0, peak_jml, half_life, which):
        """Gaussian mass-loss burst (classes.py:399-463): t_0, half_life [s], peak [kg/s]."""
        which = which.upper()                                    # classes.py:450, 455
        if which not in ('R', 'B'):
            raise ValueError("which must be 'R' or 'B'")
        ss = self._ss_jml_bj if which == 'B' else self._ss_jml_rj
        sigma = half_life * 2. / (2. * np.sqrt(2. * np.log(2.)))
        self._bursts[which].append((t_0, (peak_jml - ss) / ss, sigma))
        self._ejections[str(len(self._ejections) + 1)] = {
            't_0': t_0, 'peak_jml': peak_jml, 'half_life': half_life, 'which': which}
        self._invalidate()

    def jml_t(self, which):
        """Callable mdot(t) [kg/s] of the red and/or blue jet (classes.py:383-397)."""
        def mdot(t):
            tot = 0.
            for jet, ss in (('R', self._ss_jml_rj), ('B', self._ss_jml_bj)):
                if jet in which:
                    tot = tot + ss * self._chi_host(jet, t)
            return tot
        return mdot

    def _chi_host(self, jet, t):
        chi = 1.0
        for t0, amp_rel, sigma in self._bursts[jet]:
            chi = chi + amp_rel * np.exp(-(t - t0) ** 2. / (2. * sigma ** 2.))
        return chi

    # ------------------------------------------------------------------ device state ----
    @property
    def engine(self):
        if self._engine is None:
            from .engine import RTEngine
            # one process per GPU: LOCAL_RANK picks the device (RJP_DEVICE overrides it,
            # e.g. to rehearse several ranks on one GPU)
            self._engine = RTEngine(int(os.environ.get("RJP_DEVICE",
                                                       os.environ.get("LOCAL_RANK", "0"))))
        return self._engine

    def _invalidate(self):
        self._version = getattr(self, "_version", 0) + 1
        self._scan_cache = collections.OrderedDict()
        self._tavg = None
        self._rrl_cache = None

    @property
    def gff_mode(self):
        """classes.py:1388-1393: one Gaunt factor per channel iff q_T == 0."""
        return _lib.RJP_GFF_SCALAR if self.params['power_laws']['q_T'] == 0. \
            else _lib.RJP_GFF_POWERLAW

    @property
    def device_fields(self):
        """The packed HBM state; built on the GPU on first use (K4)."""
        if self._dev is None:
            if self.log:
                self.log.add_entry("INFO", "Calculating cells' fill factors/projected areas")
            then = _time.time()
            geom = geometry_struct(self.params, self.nx, self.ny, self.nz)
            # LEAN first (f64 storage): the continuum path reads a0, em0, temp, ts -- four
            # grid-sized arrays; nd / xi / pf / vy are built by the first call that needs them
            # (`_wide_fields`: RRL, collapse=False, grid accessors, the xi / vel setters), as
            # the reference's lazy properties do (classes.py:571-1000)
            eng = self.engine
            lean = self._dtype == _lib.RJP_F64 and eng.use_compact and eng.use_tau
            self._dev = build_model_fields(self, geom, want_wide=not lean, want_vy=not lean)
            # a jet fills a few per cent of its grid: record each sightline's occupied rows
            # once so that every later scan touches only those
            self.engine.compute_y_bounds(self._dev)
            self.engine.synchronize()
            if self.log:
                self.log.add_entry("INFO", _time.strftime(
                    'Finished in %Hh%Mm%Ss', _time.gmtime(_time.time() - then)))
        return self._dev

    def _wide_fields(self):
        """`device_fields` with nd, xi, pf and vy resident (built on first need)."""
        dev = self.device_fields
        if dev.nd is None or dev.xi is None or dev.pf is None or dev.vy is None:
            self.engine.build_wide(dev, geometry_struct(self.params, self.nx, self.ny, self.nz))
        return dev

    def _host_launch_times(self, x0=0, x1=None):
        """Launch times [s] of rows [x0, x1) of the grid evaluated on the host with scipy's
        hyp2f1, exactly as the reference does (maths/geometry.py:150-178) -- only for the
        logarithmic 2F1 cases the device series refuses (RJP_ERR_DEGENERATE)."""
        g = self.params['geometry']
        x1 = self.nx if x1 is None else x1
        ix, iy, iz = np.meshgrid(np.arange(x0, x1), np.arange(self.ny), np.arange(self.nz),
                                 indexing='ij')
        c = self.csize
        rr, ww, _ = mgeom.xyz_to_rwp(c * (ix - self.nx // 2) + c / 2., c * (iy - self.ny // 2) +
                                     c / 2., c * (iz - self.nz // 2) + c / 2., g["inc"], g["pa"])
        r = np.abs(rr)
        r = np.where((r < g['r_0']) & ((r + c / 2.) >= g['r_0']), (g['r_0'] + r + c / 2.) / 2., r)
        with np.errstate(all="ignore"):
            return mgeom.t_rw(r, ww, self.params) * con.year

    def _rjp_bursts(self):
        from .engine import make_bursts
        return make_bursts(self._bursts['R'], self._bursts['B'])

    def _grid(self, tensor):
        return tensor.cpu().numpy().astype(np.float64).reshape(self.nx, self.ny, self.nz)

    # ---- geometry accessors (host NumPy; plotting / inspection only, never on the RT path) --
    @property
    def indices(self):
        """Cell index grids (classes.py:465-474)."""
        return tuple(np.meshgrid(np.arange(self.nx), np.arange(self.ny), np.arange(self.nz),
                                 indexing=self._arr_indexing))

    ix = property(lambda self: self.indices[0])
    iy = property(lambda self: self.indices[1])
    iz = property(lambda self: self.indices[2])

    @property
    def grid(self):
        """Bottom-left-front cell corners [au] (classes.py:488-501)."""
        ix, iy, iz = self.indices
        return (self.csize * (ix - self.nx // 2), self.csize * (iy - self.ny // 2),
                self.csize * (iz - self.nz // 2))

    xx = property(lambda self: self.grid[0])
    yy = property(lambda self: self.grid[1])
    zz = property(lambda self: self.grid[2])
    xs = property(lambda self: self.csize * (np.arange(self.nx) - self.nx // 2))
    ys = property(lambda self: self.csize * (np.arange(self.ny) - self.ny // 2))
    zs = property(lambda self: self.csize * (np.arange(self.nz) - self.nz // 2))

    @property
    def grid_rwp(self):
        """Jet coordinates (r, w, phi) of the cell centroids (classes.py:515-526)."""
        xx, yy, zz = self.grid
        h = self.csize / 2.
        g = self.params["geometry"]
        return mgeom.xyz_to_rwp(xx + h, yy + h, zz + h, g["inc"], g["pa"])

    rr = property(lambda self: self.grid_rwp[0])
    ww = property(lambda self: self.grid_rwp[1])
    pp = property(lambda self: self.grid_rwp[2])

    @property
    def rreff(self):
        """Effective launching radius in the disc [au] (classes.py:543-557)."""
        g, t = self.params["geometry"], self.params["target"]
        r, w, _ = self.grid_rwp
        return mgeom.r_eff(w, t["R_1"], t["R_2"], g['w_0'], np.abs(r), g['mod_r_0'], g['r_0'],
                           g["epsilon"])

    @property
    def mass_density(self):
        """[g cm^-3] (classes.py:901-908)."""
        return self.params['properties']['mu'] * mphys.atomic_mass("H") * 1e3 * \
            self.number_density

    @property
    def pressure(self):
        """[dyn cm^-2] (classes.py:1002-1007)."""
        return self.number_density * self.temperature * con.k * 1e7

    # grids as host arrays (inspection / plotting / pickling; not on the RT path)
    @property
    def fill_factor(self):
        return self._grid(self.device_fields.ff_raw) if self.device_fields.ff_raw is not None \
            else self._grid(self._wide_fields().pf)

    @property
    def areas(self):
        d = self.device_fields
        return self._grid(d.areas_raw) if d.areas_raw is not None else \
            np.where(np.isnan(self._grid(self._wide_fields().pf)), np.nan, 1.0)

    @property
    def ts(self):
        """Time since launch of the material in each cell [s] (classes.py:838-855)."""
        return self.time - self._grid(self.device_fields.ts)

    @ts.setter
    def ts(self, new_ts):
        self.engine.replace_field(self.device_fields, "ts", new_ts)
        self._invalidate()

    @property
    def ion_fraction(self):
        return self._grid(self._wide_fields().xi)

    @ion_fraction.setter
    def ion_fraction(self, new_xis):
        self.engine.replace_field(self._wide_fields(), "xi", new_xis)
        self._invalidate()

    @property
    def temperature(self):
        return self._grid(self.device_fields.temp)

    @temperature.setter
    def temperature(self, new_ts):
        self.engine.replace_field(self.device_fields, "temp", new_ts)
        self._invalidate()

    @property
    def chi_xyz(self):
        """Burst factor per cell (classes.py:861-870)."""
        red = np.signbit(self._grid(self._wide_fields().nd))
        tl = self.ts
        return np.where(red, self._chi_host('R', tl), self._chi_host('B', tl))

    @property
    def number_density(self):
        return np.abs(self._grid(self._wide_fields().nd)) * self.chi_xyz

    @property
    def vel(self):
        """(v_x, v_y + v_lsr, v_z) [km/s] (classes.py:1009-1095).  Only v_y feeds the RT path
        and stays resident; the transverse components are built on demand."""
        if self._vxz is None:
            geom = geometry_struct(self.params, self.nx, self.ny, self.nz)
            tmp = self.engine.build_fields(geom, _lib.RJP_F64, want_ts=False, want_vy=False,
                                           want_raw=False, want_vxz=True)
            self._vxz = (self._grid(tmp.vx_raw), self._grid(tmp.vz_raw))
        return self._vxz[0], self._grid(self._wide_fields().vy), self._vxz[1]

    @vel.setter
    def vel(self, new_vs):
        """Install caller-supplied velocity grids (v_x, v_y, v_z) [km/s] (classes.py:1097-1099).
        The line-of-sight component feeds `optical_depth_rrl` (classes.py:1160-1161) and goes to
        the device; the transverse ones are kept for the getter."""
        vx, vy, vz = new_vs
        shape = (self.nx, self.ny, self.nz)
        vx, vy, vz = (np.asarray(v, dtype=np.float64) for v in (vx, vy, vz))
        if not (vx.shape == vy.shape == vz.shape == shape):
            raise ValueError("velocity grids must have the model's shape {}".format(shape))
        self.engine.replace_field(self._wide_fields(), "vy", vy)
        self._vxz = (vx.copy(), vz.copy())
        self._invalidate()

    # ------------------------------------------------------------------ K1 cache ----
    SCAN_CACHE_EPOCHS = 64       # base-map pairs kept on the device (2 x P x 8 B each)

    def _model_tavg(self):
        """T_avg = nanmean_y(T where T > 0) (classes.py:1471-1472, 1254-1256): depends on
        neither frequency nor epoch, so it is evaluated once per model (and again after the
        temperature grid is replaced through its setter)."""
        if self._tavg is None:
            self._tavg = self.engine.tavg(self.device_fields)
        return self._tavg

    def prefetch_epochs(self, times_s):
        """Scan the grid for several model times at once (8-32 epochs share one pass over
        HBM); later RT calls at those times reuse the base maps."""
        todo = [t for t in dict.fromkeys(float(t) for t in times_s)
                if t not in self._scan_cache]
        if not todo:
            return
        # the cache is bounded (a long sweep must not pin an [E, P] pair per epoch for ever):
        # at most SCAN_CACHE_EPOCHS epochs stay resident, oldest first out; a longer request
        # keeps its FIRST epochs, the ones a caller walking the list in order needs next
        todo = todo[:self.SCAN_CACHE_EPOCHS]
        dev = self.device_fields
        self._model_tavg()
        sumA, em, _ = self.engine.ff_scan(dev, self._rjp_bursts(), todo, self.gff_mode,
                                          want_tavg=False)
        for i, t in enumerate(todo):
            # own copies: a slice would keep the whole [E, P] result alive
            self._scan_cache[t] = (sumA[i:i + 1].clone(), em[i:i + 1].clone())
        while len(self._scan_cache) > self.SCAN_CACHE_EPOCHS:
            old = next(iter(self._scan_cache))
            if old in todo:
                break
            del self._scan_cache[old]

    def _base_maps(self):
        t = float(self.time)
        if t not in self._scan_cache:
            self.prefetch_epochs([t])
        return self._scan_cache[t] + (self._model_tavg(),)

    def _map(self, tensor, lead=()):
        return _to_host(tensor).reshape(*lead, self.nx, self.nz)

    FITS_DEVICE_MIN_BYTES = 8 << 20      # products from this size on are laid out for FITS on the GPU

    def _ff_products(self, freq, tau=False, flux=False, intensity=False, device=False,
                     keep=False):
        """`keep`: remember the device cube so that a following _save_cube can build the FITS
        payload (axis order + byte order) on the GPU instead of in three host passes."""
        from . import engine as E
        scalar = np.isscalar(freq)
        freqs = np.atleast_1d(np.asarray(freq, dtype=np.float64))
        sumA, _, tavg = self._base_maps()
        gv = None
        if self.gff_mode == _lib.RJP_GFF_SCALAR:
            gv = [mphys.gff(nu, self.params['properties']['T_0']) for nu in freqs]
        ctau, cflux = E.ff_channel_coeffs(freqs, self.csize, self.params["target"]["dist"],
                                          self.gff_mode, gv)
        if intensity:        # W m^-2 Hz^-1 sr^-1 (classes.py:1475, 1488)
            cflux = 2. * freqs ** 2. * con.k / con.c ** 2.
        t, s, _ = self.engine.ff_maps(sumA, tavg, ctau, cflux, want_tau=tau,
                                      want_flux=flux or intensity, want_ftot=False)
        out = t if tau else s
        if device:
            return out.reshape(len(freqs), self.nx, self.nz)
        arr = self._map(out, (len(freqs),))
        self._dev_product = out.reshape(len(freqs), self.nx, self.nz) if keep else None
        return arr[0] if scalar else arr

    def _cells(self, freq, savefits, rrl=None):
        """collapse=False: un-summed per-cell optical depths, (n_x,n_y,n_z) for a scalar
        frequency, (F,n_x,n_y,n_z) for an array (classes.py:1176-1177, 1382-1383)."""
        from . import engine as E
        if savefits:
            # the reference's writer accepts 2-D / 3-D (freq, dec, ra) data only
            raise ValueError("Unexpected number of data dimensions (4)")
        scalar = np.isscalar(freq)
        freqs = np.atleast_1d(np.asarray(freq, dtype=np.float64))
        dev = self._wide_fields()
        if rrl is None:
            gv = None
            if self.gff_mode == _lib.RJP_GFF_SCALAR:
                gv = [mphys.gff(nu, self.params['properties']['T_0']) for nu in freqs]
            ctau, _ = E.ff_channel_coeffs(freqs, self.csize, self.params["target"]["dist"],
                                          self.gff_mode, gv)
            out = self.engine.ff_cells(dev, self._rjp_bursts(), float(self.time),
                                       self.gff_mode, ctau)
        else:
            out = self.engine.rrl_cells(dev, self._rjp_bursts(), float(self.time),
                                        _lib.Line(**mrrl.line_constants(rrl)), freqs)
        arr = _to_host(out).reshape(len(freqs), self.nx, self.ny, self.nz)
        return arr[0] if scalar else arr

    def flux_vs_time(self, times_s, freq):
        """Light curves: total flux density [Jy] of the whole map at every (model time,
        frequency) -> array (len(times), len(freq)).  The reference gets these numbers by
        looping `time` and summing `flux_ff` maps (Pipeline results, classes.py:2461-2467);
        here the maps are reduced on the device and up to 32 epochs share one pass over HBM.
        Inside a torch.distributed group the epochs are shared out over the ranks.
        Memory: a DENSELY filled model keeps its launch-time moment maps after the first sweep of
        >= 12 epochs (1280 doubles per sightline: 2.7 GB at 512 x 512 sightlines) so that later
        sweeps -- other epochs, other burst parameters -- are contractions only; a pipeline that
        holds many such models sets `model.engine.cache_moments = False` (every sweep then makes
        its own pass over the grid), and a setter that replaces a field drops the maps."""
        from . import parallel
        rank, world, _ = _dist_info()
        return parallel.sweep_flux_vs_time(self, np.atleast_1d(np.asarray(times_s, float)),
                                           freq, rank=rank, world=world)

    def prepare_epoch_sweeps(self, bins=20):
        """Optional, for a model whose light curves are computed MANY times (e.g. while fitting
        burst parameters): bucket the cells of every sightline by (jet, launch-time bin) once
        (`RTEngine.build_lt`, ~50 ms and ~1.2 x the bytes of two fields for 1e9 cells).  Sweeps of
        12-32 epochs (`flux_vs_time`) then keep their launch-time moments in registers: about a
        quarter faster than the first-sweep path on dense grids.  Returns the layout record; a
        setter that changes the fields (`ts`, `ion_fraction`, `temperature`) invalidates it."""
        return self.engine.build_lt(self.device_fields, int(bins))

    # ------------------------------------------------------------------ RT methods ----
    def emission_measure(self, savefits=False):
        """Emission measure along y [pc cm^-6] (classes.py:1101-1128)."""
        _, em, _ = self._base_maps()
        ems = self._map(em)
        if savefits:
            self.save_fits(np.transpose(ems, (1, 0)), savefits, 'em')
        return ems

    def optical_depth_ff(self, freq, savefits=False, collapse=True):
        """Free-free optical depth along y (classes.py:1353-1447)."""
        if not collapse:
            return self._cells(freq, savefits)
        tff = self._ff_products(freq, tau=True, keep=bool(savefits))
        self._save_cube(tff, savefits, 'tau', freq)
        return tff

    def intensity_ff(self, freq, savefits=False):
        """Radio intensity [W m^-2 Hz^-1 sr^-1] (classes.py:1449-1496)."""
        ints = self._ff_products(freq, intensity=True, keep=bool(savefits))
        self._save_cube(ints, savefits, 'intensity', freq)
        return ints

    def flux_ff(self, freq, savefits=False):
        """Flux density [Jy/pixel] (classes.py:1498-1541)."""
        fluxes = self._ff_products(freq, flux=True, keep=bool(savefits))
        self._save_cube(fluxes, savefits, 'flux', freq)
        return fluxes

    def _rrl_tau_device(self, rrl, freqs):
        """tau_rrl[F, P] on the device.  The last cube is kept: `Pipeline` asks for the optical
        depths and then for the fluxes of the same line, channels and epoch
        (classes.py:2437, 2450), and the Voigt scan is the expensive kernel."""
        key = (rrl, tuple(float(f) for f in freqs), float(self.time), self._version)
        if self._rrl_cache is not None and self._rrl_cache[0] == key:
            return self._rrl_cache[1]
        line = _lib.Line(**mrrl.line_constants(rrl))
        tau = self.engine.rrl_scan(self._wide_fields(), self._rjp_bursts(), float(self.time),
                                   line, freqs)
        self._rrl_cache = (key, tau)
        return tau

    def optical_depth_rrl(self, rrl, freq, lte=True, savefits=False, collapse=True):
        """RRL optical depth along y (classes.py:1130-1229)."""
        if not collapse:
            return self._cells(freq, savefits, rrl=rrl)
        scalar = np.isscalar(freq)
        freqs = np.atleast_1d(np.asarray(freq, dtype=np.float64))
        dev = self._rrl_tau_device(rrl, freqs)
        tau = self._map(dev, (len(freqs),))
        tau = tau[0] if scalar else tau
        self._dev_product = dev.reshape(len(freqs), self.nx, self.nz) if savefits else None
        self._save_cube(tau, savefits, 'tau', freq)
        return tau

    def _rrl_flux(self, rrl, freq, lte, contsub, intensity=False):
        from . import engine as E
        if not lte:
            raise ValueError("Non-LTE RRL calculations not yet supported")   # classes.py:1261
        scalar = np.isscalar(freq)
        freqs = np.atleast_1d(np.asarray(freq, dtype=np.float64))
        F, P = len(freqs), self.nx * self.nz
        tau_rrl = self._rrl_tau_device(rrl, freqs)
        tau_ff = self._ff_products(freqs, tau=True, device=True).reshape(F, P)
        flux_ff = None
        if not contsub:
            flux_ff = self._ff_products(freqs, flux=True, device=True).reshape(F, P)
        cfl, hnu = E.rrl_channel_coeffs(freqs, self.csize, self.params["target"]["dist"])
        if intensity:
            cfl = cfl / (E.solid_angle(self.csize, self.params["target"]["dist"]) / 1e-26)
        flux, _ = self.engine.rrl_maps(tau_rrl, tau_ff, self._model_tavg(), flux_ff, cfl, hnu,
                                       want_ftot=False)
        out = self._map(flux, (F,))
        self._dev_product = flux.reshape(F, self.nx, self.nz)
        return out[0] if scalar else out

    def intensity_rrl(self, rrl, freq, lte=True, savefits=False):
        """RRL intensity [W m^-2 Hz^-1 sr^-1] (classes.py:1231-1290).  Arrays of
        frequencies are handled channel by channel; the reference's own array branch raises
        (it passes the whole array where a scalar is meant, classes.py:1266-1271)."""
        ints = self._rrl_flux(rrl, freq, lte, contsub=True, intensity=True)
        self._save_cube(ints, savefits, 'intensity', freq)
        return ints

    def flux_rrl(self, rrl, freq, lte=True, contsub=True, savefits=False):
        """RRL flux [Jy/pixel], continuum-subtracted unless contsub=False
        (classes.py:1292-1351)."""
        fluxes = self._rrl_flux(rrl, freq, lte, contsub)
        self._save_cube(fluxes, savefits, 'flux', freq)
        return fluxes

    # ------------------------------------------------------------------ products ----
    def _save_cube(self, data, savefits, image_type, freq):
        dev, self._dev_product = getattr(self, "_dev_product", None), None
        if not savefits:
            return
        if (dev is not None and np.ndim(data) == 3 and tuple(dev.shape) == np.shape(data) and
                dev.numel() * 8 >= self.FITS_DEVICE_MIN_BYTES):
            # (F, n_x, n_z) -> FITS order (F, dec = z, ra = x), big-endian, on the GPU: the
            # host only receives the bytes and writes them
            import torch
            t = dev.transpose(1, 2).contiguous()
            be = t.view(torch.uint8).reshape(-1, 8).flip(1).contiguous()
            arr = _fits.BigEndian(t.shape, _to_host(be.reshape(-1)))
        elif np.ndim(data) == 3:
            # views, not copies: the writer lays the bytes out in one pass
            arr = np.transpose(data, (0, 2, 1))
        else:
            arr = np.transpose(data, (1, 0))
        self.save_fits(arr, savefits, image_type, freq)

    def save_fits(self, data, filename, image_type, freq=None):
        """Write a map/cube with the reference's header (classes.py:1543-1652)."""
        if image_type not in ('flux', 'tau', 'em', 'intensity'):
            raise ValueError("arg image_type must be one of 'flux', 'tau' or 'em'")
        ndims = len(data.shape if isinstance(data, _fits.BigEndian) else np.shape(data))
        if ndims not in (2, 3):
            raise ValueError(f"Unexpected number of data dimensions ({ndims})")
        tg = self.params['target']
        _, dec_deg = miscf.sexagesimal_to_deg(tg['ra'], tg['dec'])
        # astropy's hourangle -> degree scale factor is (15 pi/180)/(pi/180), not a literal 15
        ra_deg = miscf.parse_sexagesimal(tg['ra']) * ((15. * (np.pi / 180.)) / (np.pi / 180.))
        csize_deg = np.degrees(np.arctan(self.csize * con.au / (tg['dist'] * con.parsec)))
        h = _fits.Header()
        h.set('AUTHOR', 'S.J.D.Purser')
        h.set('OBJECT', tg['name'])
        h.set('CTYPE1', 'RA---TAN', 'x-coord type is RA Tan Gnomonic projection')
        h.set('CTYPE2', 'DEC--TAN', 'y-coord type is DEC Tan Gnomonic projection')
        h.set('EQUINOX', 2000., 'Equinox of coordinates')
        h.set('CRPIX1', self.nx / 2 + 0.5, 'Reference pixel in RA')
        h.set('CRPIX2', self.nz / 2 + 0.5, 'Reference pixel in DEC')
        h.set('CRVAL1', ra_deg, 'Reference pixel value in RA (deg)')
        h.set('CRVAL2', dec_deg, 'Reference pixel value in DEC (deg)')
        h.set('CDELT1', -csize_deg, 'Pixel increment in RA (deg)')
        h.set('CDELT2', csize_deg, 'Pixel size in DEC (deg)')
        if image_type in ('flux', 'tau', 'intensity'):
            if ndims == 3:
                nchan = len(freq)
                chan_width = float(freq[1] - freq[0]) if nchan != 1 else 1.
                h.set('CTYPE3', 'FREQ', 'Spectral axis (frequency)')
                h.set('CRPIX3', nchan / 2. + 0.5, 'Reference frequency (channel number)')
                h.set('CRVAL3', float(freq[len(freq) // 2 - 1] + chan_width / 2),
                      'Reference frequency (Hz)')
                h.set('CDELT3', chan_width, 'Frequency increment (Hz)')
            else:
                f0 = float(freq[0]) if not np.isscalar(freq) else float(freq)
                h.set('CDELT3', 1., 'Frequency increment (Hz)')
                h.set('CRPIX3', 0.5, 'Reference frequency (channel number)')
                h.set('CRVAL3', f0, 'Reference frequency (Hz)')
        h.set('BUNIT', {'flux': 'Jy pixel^-1', 'intensity': 'W m^-2 Hz^-1 sr^-1',
                        'em': 'pc cm^-6', 'tau': 'dimensionless'}[image_type])
        lines = self.__str__().split('\n')
        h.add_history((' ' * (72 - len(lines[0]))).join(lines))
        _fits.writeto(filename, data, h)

    def save(self, filename):
        """Pickle the model state (classes.py:1704-1713).  The grids themselves are not
        stored: they are rebuilt on the GPU in milliseconds."""
        ps = {'params': self._params, 'areas': None, 'ffs': None, 'time': self.time,
              'log': self.log, 'storage': 'f64' if self._dtype == _lib.RJP_F64 else 'f32'}
        self.log.add_entry("INFO", "Saving physical model to {}".format(filename))
        with open(filename, "wb") as f:
            pickle.dump(ps, f)


class ContinuumRun:
    """One (epoch, frequency band) radiative-transfer run (classes.py:1716-1900)."""

    def __init__(self, dcy, year, freq=None, bandwidth=None, chanwidth=None, t_obs=None,
                 t_int=None, tscop=None):
        self._year, self._dcy, self._obs_type = year, dcy, 'continuum'
        self._freq, self._t_obs, self._t_int, self._tscop = freq, t_obs, t_int, tscop
        self._products, self._results = {}, {}
        self._bandwidth = bandwidth if bandwidth is not None else 1.
        self._chanwidth = chanwidth if chanwidth is not None else 1.
        self.completed = False
        self.radiative_transfer = freq is not None
        self.simobserve = all(v is not None for v in (tscop, bandwidth, chanwidth, t_obs,
                                                      t_int))

    line = None

    def _row(self):
        val = [self._year, self._obs_type.capitalize(), self._tscop, self._t_obs, self._t_int,
               self.line, self._freq, self._bandwidth, self._chanwidth,
               self.radiative_transfer, self.simobserve, self.completed]
        return ['-' if v is None else v for v in val]

    def __str__(self):
        return _run_table([self._row()], "grid")

    @property
    def results(self):
        return self._results

    @results.setter
    def results(self, new_results):
        if not isinstance(new_results, dict):
            raise TypeError("setter method for results attribute requires dict")
        self._results = new_results

    @property
    def products(self):
        return self._products

    @products.setter
    def products(self, new_products):
        if not isinstance(new_products, dict):
            raise TypeError("setter method for products attribute requires dict")
        self._products = new_products

    obs_type = property(lambda self: self._obs_type)
    year = property(lambda self: self._year)
    freq = property(lambda self: self._freq)
    bandwidth = property(lambda self: self._bandwidth)
    chanwidth = property(lambda self: self._chanwidth)
    t_obs = property(lambda self: self._t_obs)
    t_int = property(lambda self: self._t_int)
    tscop = property(lambda self: self._tscop)

    @property
    def dcy(self):
        return self._dcy

    @dcy.setter
    def dcy(self, path):
        self._dcy = path

    @property
    def day(self):
        return int(self.year * 365.)

    @property
    def model_dcy(self):
        return os.sep.join([self.dcy, f'Day{self.day}'])

    def _tag(self):
        return miscf.freq_str(self.freq)

    @property
    def rt_dcy(self):
        if not self.radiative_transfer:
            return None
        return os.sep.join([self.model_dcy, self._tag()])

    def _fits(self, kind):
        return self.rt_dcy + os.sep + '_'.join([kind, 'Day' + str(self.day),
                                                self._tag()]) + '.fits'

    fits_flux = property(lambda self: self._fits('Flux'))
    fits_tau = property(lambda self: self._fits('Tau'))
    fits_em = property(lambda self: self._fits('EM'))

    @property
    def nchan(self):
        return int(self.bandwidth / self.chanwidth)

    @property
    def chan_freqs(self):
        chan1 = self.freq - self.bandwidth / 2. + self.chanwidth / 2.
        return chan1 + np.arange(self.nchan) * self.chanwidth


class RRLRun(ContinuumRun):
    """One (epoch, recombination line) run (classes.py:1903-1967)."""

    def __init__(self, dcy, year, line=None, bandwidth=None, chanwidth=None, t_obs=None,
                 t_int=None, tscp=None):
        freq = mrrl.rrl_nu_0(*mrrl.rrl_parser(line))
        super().__init__(dcy, year, freq, bandwidth, chanwidth, t_obs, t_int, tscp)
        self.line = line
        self._obs_type = 'rrl'

    def _tag(self):
        return self.line


_RUN_HDR = ['Year', 'Type', 'Telescope', 't_obs', 't_int', 'Line', 'Frequency', 'Bandwidth',
            'Channel width', 'Radiative Transfer?', 'Synthetic Obs.?', 'Completed?']
_RUN_UNITS = ['yr', '', '', 's', 's', '', 'Hz', 'Hz', 'Hz', '', '', '']
_RUN_FMT = ['.2f', '', '', '.0f', '.0f', '', '.3e', '.3e', '.3e', '', '', '']


def _run_table(rows, tablefmt, **kw):
    import tabulate
    head = [h + ('\n[' + u + ']' if u else '') for h, u in zip(_RUN_HDR, _RUN_UNITS)]
    return tabulate.tabulate(rows, head, tablefmt=tablefmt, floatfmt=_RUN_FMT, **kw)


class Pipeline:
    """Runs the radiative transfer of every (epoch x band/line) of a pipeline-params file
    and writes the reference's products (classes.py:1970-2868, RT section 2386-2479).
    The CASA synthetic-observation section is outside this core."""

    @classmethod
    def load_pipeline(cls, load_file):
        home = os.path.expanduser('~')
        with open(os.path.expanduser(load_file), 'rb') as f:
            loaded = pickle.load(f)
        for run in loaded['runs']:
            run.dcy = run.dcy.replace('~', home)
        loaded['model_file'] = loaded['model_file'].replace('~', home)
        loaded['params']['dcys']['model_dcy'] = \
            loaded['params']['dcys']['model_dcy'].replace('~', home)
        jm = JetModel.load_model(loaded["model_file"])
        new = cls(jm, loaded["params"], log=loaded.get('log'))
        new.runs = loaded["runs"]
        return new

    @staticmethod
    def py_to_dict(py_file):
        return _load_params_file(py_file, miscf.check_pline_params)

    def __init__(self, jetmodel, params, log=None):
        if not isinstance(jetmodel, JetModel):
            raise TypeError("Supplied arg jetmodel must be JetModel instance not {}"
                            "".format(type(jetmodel)))
        self.model = jetmodel
        if isinstance(params, dict):
            err = miscf.check_pline_params(params)
            if err:
                raise err
            self._params = params
        elif isinstance(params, str):
            self._params = Pipeline.py_to_dict(params)
        else:
            raise TypeError("Supplied arg params must be dict or full path (str)")

        self.dcy = self.params['dcys']['model_dcy'].rstrip(os.sep)
        self.model_file = self.dcy + os.sep + "jetmodel.save"
        self.save_file = self.dcy + os.sep + "pipeline.save"
        log_name = "Pipeline_{}.log".format(_time.strftime("%Y%m%d%H-%M-%S",
                                                           _time.localtime()))
        created = not os.path.exists(self.dcy)
        if created:
            os.makedirs(self.dcy, exist_ok=True)      # several ranks may get here together
        self._log = log if log is not None else logger.Log(os.sep.join([self.dcy, log_name]))
        if created:
            self.log.add_entry("INFO", f"Creating pipeline directory, {self.dcy}")
        if self.model.log is None:
            self.model.log = self.log
        elif self.model.log is not self.log:
            merged = logger.Log.combine_logs(self.log, self.model.log, self.log.filename,
                                             delete_old_logs=True)
            self.log = self.model.log = merged

        for band in ('continuum', 'rrls'):
            if self.params[band]['times'] is not None:
                self.params[band]['times'].sort()
            else:
                self.params[band]['times'] = np.array([])

        def pick(v, i):
            return v[i] if miscf.is_iter(v) else v

        runs = []
        c = self.params['continuum']
        self.log.add_entry("INFO", "Reading continuum runs into pipeline")
        for t in c['times']:
            for i, freq in enumerate(c['freqs']):
                runs.append(ContinuumRun(self.dcy, t, freq, pick(c['bws'], i),
                                         pick(c['chanws'], i), pick(c['t_obs'], i),
                                         pick(c['t_ints'], i), pick(c['tscps'], i)))
        if not runs:
            self.log.add_entry("WARNING", "No continuum runs found", timestamp=True)
        r = self.params['rrls']
        self.log.add_entry("INFO", "Reading radio recombination line runs into pipeline")
        n0 = len(runs)
        for t in r['times']:
            for i, line in enumerate(r['lines']):
                runs.append(RRLRun(self.dcy, t, str(line), pick(r['bws'], i),
                                   pick(r['chanws'], i), pick(r['t_obs'], i),
                                   pick(r['t_ints'], i), pick(r['tscps'], i)))
        if len(runs) == n0:
            self.log.add_entry("WARNING", "No RRL runs found", timestamp=True)
        self._runs = runs
        self.log.add_entry("INFO", self.__str__(), timestamp=True)

    def __str__(self):
        return _run_table([run._row() for run in self.runs], "psql", numalign='center',
                          stralign='center')

    params = property(lambda self: self._params)

    @property
    def runs(self):
        return self._runs

    @runs.setter
    def runs(self, new_runs):
        self._runs = new_runs

    @property
    def log(self):
        return self._log

    @log.setter
    def log(self, new_log):
        self._log = new_log

    def save(self, save_file, absolute_directories=False):
        """Pickle the pipeline state (classes.py:2215-2258)."""
        home = os.path.expanduser('~')
        mf = self.model_file
        if not absolute_directories:
            for run in self.runs:
                run.dcy = run.dcy.replace(home, '~')
            self._params['dcys']['model_dcy'] = \
                self._params['dcys']['model_dcy'].replace(home, '~')
            mf = mf.replace(home, '~')
        self.log.add_entry("INFO", "Saving pipeline to " + save_file)
        with open(save_file, 'wb') as f:
            pickle.dump({"runs": self.runs, "params": self._params, "model_file": mf,
                         'log': self.log}, f)

    def execute(self, simobserve=True, verbose=True, dryrun=False, resume=True, clobber=False):
        """Radiative transfer for every run; writes EM/Tau/Flux FITS products, records
        `results['flux']`, pickles model and pipeline state (classes.py:2296-2479)."""
        self.log.add_entry("INFO", "Beginning pipeline execution")
        if verbose != self.log.verbose:
            self.log.verbose = verbose
        if simobserve:
            self.log.add_entry("WARNING", "CASA synthetic observations are outside the "
                                          "radiative-transfer core and are skipped")
        if resume and os.path.exists(self.model_file):
            self.model = JetModel.load_model(self.model_file, engine=self.model._engine)

        # Multi-GPU (one process per GPU inside an initialised torch.distributed group): the
        # EPOCHS of the run table are dealt round-robin to the ranks -- every epoch is a pass
        # over the grid, each rank holds its own copy of the model on its GPU and writes the
        # products of its runs; only the run results are exchanged at the end.
        rank, world, dist = _dist_info()
        years = sorted({float(r.year) for r in self.runs})
        owner = {y: i % world for i, y in enumerate(years)}
        mine = [i for i, r in enumerate(self.runs) if owner[float(r.year)] == rank]
        self._multi_rank = world > 1

        pending = []
        if not dryrun:
            pending = [self.runs[i].year * con.year for i in mine
                       if self.runs[i].radiative_transfer and
                       not (self.runs[i].completed and resume and not clobber)]
            pending = list(dict.fromkeys(pending))

        # A rank that fails must still reach the gather and the barrier below, or its peers
        # would wait in the collective until the RCCL timeout: per-run failures are recorded
        # and re-raised on EVERY rank after the exchange.
        failure = None
        for idx in mine:
            run = self.runs[idx]
            self.model.time = run.year * con.year
            self.log.add_entry("INFO", "Executing run #{} -> Details:\n{}"
                                       "".format(idx + 1, run.__str__()))
            if run.completed and resume and not clobber:
                self.log.add_entry("INFO", "Run #{} previously completed, skipping"
                                           "".format(idx + 1), timestamp=False)
                continue
            try:
                if not os.path.exists(run.rt_dcy):
                    self.log.add_entry("INFO", "{} doesn't exist, creating"
                                               "".format(run.rt_dcy), timestamp=False)
                    os.makedirs(run.rt_dcy, exist_ok=True)
                if not dryrun and run.radiative_transfer:
                    t_now = float(self.model.time)
                    if t_now in pending:
                        # one pass over HBM serves up to 32 of the epochs still to come; the
                        # whole chunk leaves the list, so the next pass starts where this one
                        # ended (ceil(N / 32) scans for N epochs)
                        k = pending.index(t_now)
                        self.model.prefetch_epochs(pending[k:k + 32])
                        del pending[:k + 32]
                    self._radiative_transfer(idx, run, clobber)
            except KeyboardInterrupt:
                self.log.add_entry("ERROR", "Pipeline interrupted by user, saving state")
                failure = (idx, "KeyboardInterrupt", "Pipeline interrupted by user")
                break
            except Exception as exc:                      # noqa: BLE001 -- re-raised below
                if world == 1:
                    raise
                self.log.add_entry("ERROR", "Run #{} failed on rank {}: {}: {}".format(
                    idx + 1, rank, type(exc).__name__, exc))
                failure = (idx, type(exc).__name__, str(exc))
                break
            run.completed = True                                   # classes.py:2853

        failures = [failure] if failure else []
        if dist is not None:
            # (inside an initialised group of ANY size, so that a one-rank RCCL group exercises
            # exactly the collectives an eight-rank one does)
            # one small object gather: {run index: (results, completed)} + the failure (if
            # any) of every rank
            local = {"runs": {i: (self.runs[i].results, self.runs[i].completed) for i in mine},
                     "failure": failure, "rank": rank}
            parts = [None] * world
            dist.all_gather_object(parts, local)
            failures = []
            for part in parts:
                for i, (res, done) in part["runs"].items():
                    self.runs[i].results = res
                    self.runs[i].completed = done
                if part["failure"]:
                    failures.append(part["failure"] + (part["rank"],))
        if rank == 0:
            self.save(self.save_file)
            self.model.save(self.model_file)
        if dist is not None:
            dist.barrier()
        if failures:
            f = failures[0]
            if f[1] == "KeyboardInterrupt":
                raise KeyboardInterrupt(f[2])
            where = " on rank %d" % f[3] if len(f) > 3 else ""
            raise RuntimeError("Pipeline run #%d failed%s: %s: %s (state saved; completed "
                               "runs are skipped on resume)" % (f[0] + 1, where, f[1], f[2]))

    def radio_plot(self, run, percentile=5., savefig=False):
        """The reference draws flux / optical depth / emission measure panels with matplotlib
        (`classes.py:3015-3183`); plotting is outside this package: the
        FITS products of `run` hold the same maps."""
        raise NotImplementedError("rajepy_amd produces no plots; read the run's FITS products "
                                  "(run.products) instead")

    def _radiative_transfer(self, idx, run, clobber):
        m = self.model
        self.log.add_entry("INFO", "Conducting radiative transfer at "
                                   f"{run.freq / 1e9:.1f}GHz for a model time of "
                                   f"{run.year:.1f}yr")
        if not os.path.exists(run.fits_em) or clobber:
            self.log.add_entry("INFO", f"Emission measures saved to {run.fits_em}")
            m.emission_measure(savefits=run.fits_em)
        else:
            self.log.add_entry("INFO", f"Emission measures already exist -> {run.fits_em}",
                               timestamp=False)
        rrl = run.obs_type != 'continuum'
        if not os.path.exists(run.fits_tau) or clobber:
            self.log.add_entry("INFO", f"Computing optical depths and saving to {run.fits_tau}")
            if rrl:
                m.optical_depth_rrl(run.line, run.chan_freqs, savefits=run.fits_tau)
            else:
                m.optical_depth_ff(run.chan_freqs, savefits=run.fits_tau)
        else:
            self.log.add_entry("INFO", f"Optical depths already exist -> {run.fits_tau}",
                               timestamp=False)
        if not os.path.exists(run.fits_flux) or clobber:
            self.log.add_entry("INFO", f"Calculating fluxes and saving to {run.fits_flux}")
            if rrl:
                fluxes = m.flux_rrl(run.line, run.chan_freqs, contsub=False,
                                    savefits=run.fits_flux)
            else:
                fluxes = m.flux_ff(run.chan_freqs, savefits=run.fits_flux)
        else:
            self.log.add_entry("INFO", f"Fluxes already exist -> {run.fits_flux}",
                               timestamp=False)
            fluxes = _fits.read(run.fits_flux)[0]
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            if not rrl:
                # classes.py:2461-2467: total of the channel-averaged map
                flux = np.nansum(np.nanmean(fluxes, axis=0))
                self.log.add_entry("INFO", f"Total, average, channel flux of {flux:.2e}Jy "
                                           "calculated")
            else:
                flux = np.nansum(np.nansum(fluxes, axis=1), axis=1)     # classes.py:2471
        self.runs[idx].results['flux'] = flux
        if not getattr(self, "_multi_rank", False):
            # single process: checkpoint after every run as the reference does
            # (classes.py:2475-2479); with several ranks rank 0 saves once at the end
            if not os.path.exists(self.model_file):
                m.save(self.model_file)
            self.save(self.save_file, absolute_directories=True)
