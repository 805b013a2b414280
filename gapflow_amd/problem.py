"""Problem driver with the reference's API surface, running its hot path on an MI355X.

Mirrors GaPFlow/problem.py (reference): ``Problem.from_yaml / from_string / _from_dict``,
``run``, ``_pre_run``, ``update``, the scalar properties and the ``pressure`` /
``wall_stress_xz`` / ``wall_stress_yz`` / ``bulk_stress`` / ``topo`` members the reference's
tests touch.  All per-step arithmetic happens in libgapflow_hip.so (HIP kernels, fp64); this
module only keeps the host-side bookkeeping of problem.py:368-451 and 616-637.

Host <-> device coherence of ``q``: the reference hands out a writable NumPy view
(problem.py:314-317) and its tests assign into it (tests/test_wave_decay.py:101).  Here ``q``
is a host mirror: reading the property downloads the device field if a step ran since the
last read; before the next device operation the mirror is compared with what was downloaded
and uploaded again if the user changed it.
"""
import ctypes as C
import io as _io
import os
import signal
from collections import deque
from datetime import datetime

import numpy as np

from . import _lib
from . import __version__
from .io import read_yaml_input, write_yaml, create_output_directory, history_to_csv
from .topography import Topography
from .stress import Pressure, WallStress, BulkStress


def _termination_signals():
    # utils.py:80-95 of the reference
    names = ('SIGINT', 'SIGTERM', 'SIGHUP', 'SIGUSR1')
    return [getattr(signal, n) for n in names if hasattr(signal, n)]


class Problem:
    """Gap-averaged lubrication problem advanced by the fused MacCormack HIP kernel."""

    def __init__(self, options, grid, numerics, prop, geo, gp=None, database=None, extra_field=None, device=0):
        if gp is not None and database is None:
            raise IOError("GP closures need a training database (`db` section)")
        if database is not None and not getattr(database, 'has_mock_md', True):
            prop['shear'] = 0.                       # problem.py:110-113
            prop['bulk'] = 0.
        self.options, self.grid, self.numerics, self.geo, self.prop = options, grid, numerics, geo, prop
        self.has_gp_model = gp is not None
        self._lib = _lib.require_device()
        Nx, Ny = grid['Nx'], grid['Ny']
        self._shape = (Nx + 2, Ny + 2)

        # device problem
        self._cfg = self._make_config(device)
        self._h = C.c_void_p()
        _lib.check(self._lib.gpf_create(C.byref(self._cfg), C.byref(self._h)))

        # uniform initial state (problem.py:662-670)
        self.step = None
        self._q_host = np.empty((3,) + self._shape)
        self._q_host[0] = prop['rho0']
        self._q_host[1] = prop['rho0'] * geo['U'] / 2.0
        self._q_host[2] = prop['rho0'] * geo['V'] / 2.0
        self._q_snapshot = None
        self._device_newer = False
        self._upload(_lib.FIELD_Q, self._q_host)
        self._q_snapshot = self._q_host.copy()

        # extra field = slip length by default (problem.py:132-135)
        self._extra = np.zeros((1,) + self._shape)
        if extra_field is not None:
            self._extra[...] = extra_field
            self._upload(_lib.FIELD_EXTRA, self._extra)

        self.topo = Topography(grid, geo, prop, on_change=self._upload_topo)
        self._upload_topo()
        self._elastic = None
        if self.topo.elastic:                   # topography.py:236-249
            from .elastic import ElasticDeformation
            el = prop['elastic']
            self._elastic = ElasticDeformation(el['E'], el['v'], el['alpha_underrelax'], grid, el['n_images'])
            self._elastic.attach(self)
            self.topo.refresh = self._download_topo

        self._closures_stale = True
        self.database = database
        self._gp_models = {}
        if gp is not None:
            from .gp import attach_surrogates
            self._gp_models = attach_surrogates(self, gp, database)
        self.pressure = Pressure(self, self._gp_models.get('zz'))
        self.bulk_stress = BulkStress(self)
        self.wall_stress_xz = WallStress(self, 'x', self._gp_models.get('xz'))
        self.wall_stress_yz = WallStress(self, 'y', self._gp_models.get('yz'))

        sc = self._scalars()
        self._kinetic_energy_old = sc.ekin         # problem.py:670
        self._ekin_old_user = None
        self._stop = False
        self.history = {k: [] for k in ('step', 'time', 'ekin', 'residual', 'vsound')}

        if not options['silent']:
            self.outdir = create_output_directory(options['output'], options['use_tstamp'])
            if database is not None:            # problem.py:158-164: MD datasets go below the run's output directory
                database.set_training_path(os.path.join(self.outdir, 'train'), check_temporary=True)
                database.output_path = self.outdir
                options['output'] = self.outdir
            full = {'version': __version__}
            for k, v in zip(['options', 'grid', 'numerics', 'geo', 'prop'], [options, grid, numerics, geo, prop]):
                full[k] = v
            if database is not None:            # problem.py:175-178
                full['gp'], full['db'], full['md'] = gp, database.config, database.md_config
            write_yaml(full, os.path.join(self.outdir, 'config.yml'))
            from .output import FieldWriter
            self._writer = FieldWriter(self)

    def __del__(self):
        h = getattr(self, '_h', None)
        if h is not None and h.value:
            self._lib.gpf_destroy(h)
            h.value = None

    # -------------------------------------------------------------------------------------
    # constructors (problem.py:211-308)
    # -------------------------------------------------------------------------------------
    @classmethod
    def from_yaml(cls, fname, device=0):
        print(f"Reading input file: {fname}")
        with open(fname, "r") as f:
            return cls._from_dict(read_yaml_input(f), device=device)

    @classmethod
    def from_string(cls, ymlstring, device=0):
        with _io.StringIO(ymlstring) as f:
            return cls._from_dict(read_yaml_input(f), device=device)

    @classmethod
    def _from_dict(cls, input_dict, device=0):
        gp = input_dict.get('gp', None)
        db = input_dict.get('db', None)
        database = None
        if db is not None:
            from .gp import make_database
            database = make_database(input_dict, device)
        return cls(input_dict['options'], input_dict['grid'], input_dict['numerics'], input_dict['properties'],
                   input_dict['geometry'], gp=gp, database=database, extra_field=None, device=device)

    # -------------------------------------------------------------------------------------
    # configuration -> C struct
    # -------------------------------------------------------------------------------------
    def _edge_rules(self):
        """Resolve problem.py:676-768 into (rule per component, Dirichlet target) for the four ghost edges.

        Reference quirks kept: the low-x ghost takes the xW value and the high-x ghost the xE value,
        but the low-y ghost takes the *yN* value and the high-y ghost the *yS* value (problem.py:746-754).
        """
        g = self.grid
        # (edge, side whose masks select the assigned rows, side that supplies masks+value of the data)
        table = [(0, 'xE', 'xW'), (1, 'xW', 'xE'), (2, 'yS', 'yN'), (3, 'yN', 'yS')]
        rules, values = [], []
        for _, assign, data in table:
            if all(g[f'bc_{assign}_P']):
                rules.append([_lib.BC_P] * 3)
                values.append(0.0)
                continue
            for t in 'DN':
                if list(g[f'bc_{assign}_{t}']) != list(g[f'bc_{data}_{t}']):
                    raise NotImplementedError("opposite edges must declare the same D/N types per component "
                                              "(the reference's mask/value pairing, problem.py:685-707, breaks otherwise)")
            r = []
            for c in range(3):
                if g[f'bc_{data}_D'][c]:
                    r.append(_lib.BC_D)
                elif g[f'bc_{data}_N'][c]:
                    r.append(_lib.BC_N)
                else:
                    raise NotImplementedError("an edge must be periodic for all components or for none")
            rules.append(r)
            values.append(float(g.get(f'bc_{data}_D_val', 0.0)) if any(g[f'bc_{data}_D']) else 0.0)
        return rules, values

    def _make_config(self, device):
        g, n, p, geo = self.grid, self.numerics, self.prop, self.geo
        cfg = _lib.GpfConfig()
        cfg.Nx, cfg.Ny, cfg.dx, cfg.dy = g['Nx'], g['Ny'], g['dx'], g['dy']
        cfg.U, cfg.V = geo['U'], geo['V']
        cfg.eta, cfg.zeta = p['shear'], p['bulk']
        if p['EOS'] not in _lib.EOS_IDS:
            raise NotImplementedError(f"EOS '{p['EOS']}' needs a surrogate model (gp/db sections)")
        cfg.eos = _lib.EOS_IDS[p['EOS']]
        for i, k in enumerate(_lib.EOS_KEYS[p['EOS']]):
            if k not in p and k not in _lib.EOS_DEFAULTS[p['EOS']]:
                raise TypeError(f"EOS '{p['EOS']}' needs the property '{k}'")
            cfg.eos_par[i] = p.get(k, _lib.EOS_DEFAULTS[p['EOS']].get(k))
        cfg.piezo = 0
        if 'piezo' in p and p['piezo']['name'] in _lib.PIEZO_IDS:
            name = p['piezo']['name']
            cfg.piezo = _lib.PIEZO_IDS[name]
            for i, k in enumerate(_lib.PIEZO_KEYS[name]):
                cfg.piezo_par[i] = p['piezo'][k]
        cfg.thinning = 0
        if 'thinning' in p and p['thinning']['name'] in _lib.THINNING_IDS:
            name = p['thinning']['name']
            cfg.thinning = _lib.THINNING_IDS[name]
            for i, k in enumerate(_lib.THINNING_KEYS[name]):
                cfg.thinning_par[i] = p['thinning'][k]
        rules, values = self._edge_rules()
        for e in range(4):
            for c in range(3):
                cfg.bc_rule[e][c] = rules[e][c]
            cfg.bc_value[e] = values[e]
        cfg.halo_lo = cfg.halo_hi = 0
        cfg.adaptive = int(bool(n['adaptive']))
        cfg.CFL, cfg.dt_fixed, cfg.tol = n['CFL'], n['dt'], n['tol']
        cfg.max_it = n['max_it']
        cfg.mc_order = n['MC_order']
        cfg.device = device
        return cfg

    # -------------------------------------------------------------------------------------
    # host <-> device
    # -------------------------------------------------------------------------------------
    def _upload(self, field, arr):
        a = _lib.f64c(arr)
        _lib.check(self._lib.gpf_upload(self._h, field, _lib.as_dp(a), a.size))

    def _download(self, field, ncomp):
        out = np.empty((ncomp,) + self._shape)
        _lib.check(self._lib.gpf_download(self._h, field, _lib.as_dp(out), out.size))
        return out

    def _upload_topo(self):
        self._upload(_lib.FIELD_TOPO, self.topo.full[:3])
        self._closures_stale = True

    def _download_topo(self, field):
        """Host mirror of the deformed gap: h, dh/dx, dh/dy and the displacement (topography.py:283-305)."""
        field[:3] = self._download(_lib.FIELD_TOPO, 3)
        field[3] = self._download(_lib.FIELD_DEFORMATION, 1)[0]

    def _sync_to_device(self):
        """Push user edits of ``q`` before any device operation."""
        if self._q_snapshot is not None and not self._device_newer:
            if not np.array_equal(self._q_host, self._q_snapshot, equal_nan=True):
                self._upload(_lib.FIELD_Q, self._q_host)
                self._q_snapshot = self._q_host.copy()
                self._closures_stale = True

    def _mark_device_advanced(self):
        """The device field has moved on: the host mirror is stale until `q` is read again.  In the reference `q` is the
        live field, so `q = p.q; p.update(); q[0] *= 1.01` edits the NEW state; here that array still holds the old one
        and the edit could only be dropped or misapplied.  The stale mirror is therefore made read-only -- such an edit
        raises at once -- and the next read of `p.q` hands out a NEW array holding the current state.  (NumPy views keep the
        flag they were created with: a view taken earlier, `rho = p.q[0]`, stays writable, but it aliases the retired array
        and can no longer reach the mirror in use; an edit through it goes nowhere, like an edit of any copy.)"""
        self._device_newer = True
        self._closures_stale = True
        self._q_host.flags.writeable = False

    @property
    def q(self):
        """Full density field (3, Nx+2, Ny+2): rho, jx, jy -- a writable host mirror."""
        if self._device_newer:
            self._q_host = np.empty_like(self._q_host)      # the retired array stays read-only; its old views cannot alias this one
            _lib.check(self._lib.gpf_download(self._h, _lib.FIELD_Q, _lib.as_dp(self._q_host), self._q_host.size))
            self._q_snapshot = self._q_host.copy()
            self._device_newer = False
        return self._q_host

    def _derived(self, field):
        self._sync_to_device()
        if self._closures_stale:
            _lib.check(self._lib.gpf_update_closures(self._h))
            self._closures_stale = False
        return self._download(field, _lib.FIELD_NCOMP[field])

    def _scalars(self):
        self._sync_to_device()
        sc = _lib.GpfScalars()
        _lib.check(self._lib.gpf_scalars(self._h, C.byref(sc)))
        return sc

    # -------------------------------------------------------------------------------------
    # scalar properties (problem.py:319-362)
    # -------------------------------------------------------------------------------------
    @property
    def q_has_nan(self):
        return bool(self._scalars().invalid == 1) or bool(np.any(np.isnan(self.q)))

    @property
    def q_has_negative_density(self):
        return bool(np.any(self.q[0] < 0.))

    @property
    def q_is_valid(self):
        return not self.q_has_nan and not self.q_has_negative_density

    @property
    def mass(self):
        return np.float64(self._scalars().mass)

    @property
    def kinetic_energy(self):
        return np.float64(self._scalars().ekin)

    @property
    def kinetic_energy_old(self):
        if self.step is None:
            return np.float64(self._kinetic_energy_old)
        return np.float64(self._scalars().ekin_old)

    @kinetic_energy_old.setter
    def kinetic_energy_old(self, value):
        self._kinetic_energy_old = float(value)
        if self.step is not None:
            _lib.check(self._lib.gpf_set_ekin_old(self._h, float(value)))

    @property
    def v_max(self):
        return np.float64(self._scalars().v_max)

    @property
    def dt_crit(self):
        sc = self._scalars()
        return min(self.grid['dx'], self.grid['dy']) / (sc.v_max + sc.v_sound)

    @property
    def cfl(self):
        return self.dt / self.dt_crit

    @property
    def converged(self):
        return bool(np.all(np.array(self.residual_buffer) < self.tol))

    # -------------------------------------------------------------------------------------
    # run loop (problem.py:368-503)
    # -------------------------------------------------------------------------------------
    def _features(self):
        """(ncell, 7) GP feature matrix of the current state, all cells incl. ghosts (gp.py:223-232)."""
        return np.vstack([self.q, self.topo.full[:3], self._extra]).reshape(7, -1).T

    def _pre_run(self):
        self._sync_to_device()
        if self._gp_models:
            # init_database + init of every surrogate (problem.py:418-424, stress.py:278-287, 586-598)
            self.database.initialize(self._features(), self.grid['dim'])
            for m in self._gp_models.values():
                m.init()
        _lib.check(self._lib.gpf_pre_run(self._h))
        if self._kinetic_energy_old is not None:
            _lib.check(self._lib.gpf_set_ekin_old(self._h, float(self._kinetic_energy_old)))
        sc = self._scalars()
        self.step = 0
        self.simtime = 0.
        self.residual = 1.
        self.residual_buffer = deque([self.residual], 5)
        self.dt = sc.dt
        self.tol = self.numerics['tol']
        self.max_it = self.numerics['max_it']

    def _absorb(self, entries):
        """Fold the per-step records of a batch into the host-side mirror of the run state."""
        for e in entries:
            if e.invalid and e.step == self.step:
                continue
            self.residual = e.residual
            self.residual_buffer.append(e.residual)
            self.step = int(e.step)
            self.simtime = e.simtime
            self.dt = e.dt

    def _advance(self, n, honor_stop):
        """Enqueue up to n updates on the device; returns the per-step records that actually ran."""
        self._sync_to_device()
        log = (_lib.GpfScalars * n)()
        nexec = C.c_int64(0)
        before = self.step
        _lib.check(self._lib.gpf_step(self._h, n, int(honor_stop), log, n, C.byref(nexec)))
        ran = int(nexec.value) - before
        entries = [log[i] for i in range(ran)]
        self._absorb(entries)
        if ran > 0:
            self._mark_device_advanced()
        failed = ran < n and log[ran].invalid != 0 if ran < n else False
        if failed:
            self._finalize(log[ran].invalid)
        return entries

    def update(self):
        """One MacCormack predictor-corrector time step (problem.py:509-569), on the device."""
        if self.step is None:
            raise RuntimeError("call _pre_run() (or run()) before update()")
        if self._gp_models or self._cfg.thinning or self._elastic:
            self._update_with_surrogates()      # stage-wise pipeline (host between stages / grad p for thinning / elastic gap)
        else:
            self._advance(1, honor_stop=False)

    def _update_with_surrogates(self):
        """The same step through the stage-wise pipeline: the host sits between the stages because a
        surrogate may retrain or extend its database there (gp.py:435-506, problem.py:532-560)."""
        self._sync_to_device()
        lib, h = self._lib, self._h
        one_step_before_output = (self.step + 1) % self.options['write_freq'] == 0       # problem.py:530
        feats = None

        def features_of_cell(i):
            nonlocal feats
            if feats is None:
                feats = self._features()        # active learning runs in the predictor only: working field == q
            return feats[i]

        _lib.check(lib.gpf_open_step(h))
        for i in range(2):
            for m in self._gp_models.values():
                m.sync_scales()             # the database may have grown through another model since this one was fitted
            _lib.check(lib.gpf_stage_closures(h))
            changed = False
            for name in ('zz', 'xz', 'yz'):
                m = self._gp_models.get(name)
                if m is not None:
                    changed |= m.stage(i == 0, one_step_before_output, features_of_cell)
            if changed:
                _lib.check(lib.gpf_stage_closures(h))
            _lib.check(lib.gpf_stage_advance(h, i))
        for m in self._gp_models.values():
            m.sync_scales()                 # the sound speed that closes the step sees the current scales too
        sc = _lib.GpfScalars()
        _lib.check(lib.gpf_close_step(h, C.byref(sc)))
        if sc.invalid:
            self._finalize(sc.invalid)
            return
        if self._elastic:
            # Topography.update (problem.py:566): the gap deforms under the pressure of the last closure evaluation
            # (the corrector's), on the device; the host mirror of the topography is refreshed on access
            _lib.check(lib.gpf_elastic_update(h))
            self.topo.mark_stale()
        self._absorb([sc])
        self._mark_device_advanced()
        # the derived fields on the device ARE what the reference's field objects hold now: the closures of the corrector stage
        # (problem.py:531-560; neither the averaging nor Topography.update re-evaluates them)
        self._closures_stale = False

    def _finalize(self, reason):
        # problem.py:588-610: the device kept the pre-step field; closures refresh lazily
        print('NaN detected.' if reason == 1 else 'Negative density detected.', end=' ')
        print('Writing previous step and aborting simulation.')
        self._closures_stale = True
        self._stop = True

    def _receive_signal(self, signum, frame):
        if signum in _termination_signals():
            self._stop = True

    def run(self, keep_open=False):
        if self.step is None:
            self._pre_run()
        self._stop = False
        self.history = {k: [] for k in ('step', 'time', 'ekin', 'residual', 'vsound')}
        silent = self.options['silent']
        if not silent:
            print(61 * '-')
            print(f"{'Step':6s} {'Timestep':10s} {'Time':10s} {'CFL':10s} {'Residual':10s}")
            print(61 * '-')
            self.write(params=False)
        old = {s: signal.signal(s, self._receive_signal) for s in _termination_signals()} \
            if _in_main_thread() else {}
        self._tic = datetime.now()
        wf = self.options['write_freq']
        try:
            while (self._gp_models or self._cfg.thinning or self._elastic) and not self.converged and self.step < self.max_it and not self._stop:
                self.update()                   # surrogates: one host-driven step at a time
                if self.step % wf == 0 and not silent and not self._stop:
                    self.write()
            while not self.converged and self.step < self.max_it and not self._stop:
                # steps until the next frame (problem.py:404) or max_it, whichever comes first; the
                # device stops by itself at convergence, so a batch never overshoots the reference's loop
                n = min(wf - self.step % wf, self.max_it - self.step, 4096)
                self._advance(n, honor_stop=True)
                if self.step % wf == 0 and not silent and not self._stop:
                    self.write()
        finally:
            for s, hdl in old.items():
                signal.signal(s, hdl)
        if not keep_open:
            self._post_run()

    def _post_run(self):
        walltime = datetime.now() - self._tic
        silent = self.options['silent']
        if self.step % self.options['write_freq'] != 0 and not silent:
            self.write()
        if not silent:
            self._writer.close()
        speed = self.step / max(walltime.total_seconds(), 1e-12)
        print(33 * '=')
        print("Total walltime   : ", str(walltime).split('.')[0])
        print(f"({speed:.2f} steps/s)")
        for name in ('zz', 'xz', 'yz'):             # problem.py:475-483
            m = self._gp_models.get(name)
            if m is not None:
                print(f" - GP train ({name}) : ", str(m.cumtime_train).split('.')[0])
                print(f" - GP infer ({name}) : ", str(m.cumtime_infer).split('.')[0])
        print(33 * '=')
        if not silent:
            history_to_csv(os.path.join(self.outdir, 'history.csv'), self.history)
            for name, m in self._gp_models.items():  # problem.py:490-503
                history_to_csv(os.path.join(self.outdir, f'gp_{name}.csv'), m.history)
                with open(os.path.join(self.outdir, f'gp_{name}.txt'), 'w') as f:
                    print(m, file=f)

    def write(self, scalars=True, fields=True, params=True):
        # problem.py:616-637
        if scalars:
            sc = self._scalars()
            cfl = self.dt / (min(self.grid['dx'], self.grid['dy']) / (sc.v_max + sc.v_sound))
            print(f"{self.step:<6d} {self.dt:.4e} {self.simtime:.4e} {cfl:.4e} {self.residual:.4e}")
            self.history["step"].append(self.step)
            self.history["time"].append(self.simtime)
            self.history["ekin"].append(sc.ekin)
            self.history["residual"].append(self.residual)
            self.history["vsound"].append(sc.v_sound)
        if fields and not self.options['silent']:
            self._writer.append_frame()
        if params:
            for m in self._gp_models.values():
                m.write()


def _in_main_thread():
    import threading
    return threading.current_thread() is threading.main_thread()
