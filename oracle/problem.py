"""MacCormack driver, restated from GaPFlow/problem.py.  Test infrastructure only.

Follows (relative to /root/reference/GaPFlow):
  problem.py:77-151    field set-up, uniform initial state (662-670)
  problem.py:319-362   validity flags and the scalars mass / Ekin / v_max / dt_crit / cfl / converged
  problem.py:368-443   run loop and _pre_run
  problem.py:509-610   update, _post_update, _finalize
  problem.py:676-768   ghost-cell boundary conditions (including the side-swapped
                       Dirichlet values in y and the opposite-side row masks)
  models/stress.py:289-362, 427-459, 600-622   which closure is written into which field

Array layout is the reference's: (component, ix, iy), one ghost cell per side.
The GP closure is in oracle/gp.py and plugged in through ``gp_models``.
"""
from collections import deque
import io
import numpy as np

from . import closures as cl
from .config import read_yaml_input
from .integrate import predictor_corrector, source
from .topography import build_topography, central_gradients


class OracleProblem:

    def __init__(self, options, grid, numerics, prop, geo, gp=None, database=None, extra_field=None):
        self.options, self.grid, self.numerics, self.prop, self.geo = options, grid, numerics, prop, geo
        shape = (grid['Nx'] + 2, grid['Ny'] + 2)
        self.q = np.zeros((3,) + shape)
        self.q[0] = prop['rho0']                    # problem.py:662-670
        self.q[1] = prop['rho0'] * geo['U'] / 2.0
        self.q[2] = prop['rho0'] * geo['V'] / 2.0
        self.extra = np.zeros((1,) + shape)         # problem.py:132-135 (slip length Ls by default)
        if extra_field is not None:
            self.extra[...] = extra_field
        self.topo, self.x, self.y = build_topography(grid, geo)
        self.pressure = np.zeros(shape)
        self.tau_avg = np.zeros((3,) + shape)
        self.wall_lower = np.zeros((6,) + shape)    # = wall_stress_xz.lower + wall_stress_yz.lower
        self.wall_upper = np.zeros((6,) + shape)
        self.gp_models = None                       # set by oracle.gp when a surrogate is attached
        self.elastic = None                         # topography.py:236-249
        el = prop.get('elastic', {})
        if el.get('enabled', False):
            from .elastic import ElasticDeformation
            self.elastic = ElasticDeformation(el['E'], el['v'], el['alpha_underrelax'], grid, el['n_images'])
            self.h_undeformed = self.topo[0].copy()
            self.deformation = np.zeros(shape)
        self.step = None
        self.kinetic_energy_old = self.kinetic_energy
        self._stop = False

    # -- constructors (problem.py:251-308) ---------------------------------
    @classmethod
    def from_dict(cls, d):
        return cls(d['options'], d['grid'], d['numerics'], d['properties'], d['geometry'], gp=d.get('gp'))

    @classmethod
    def from_string(cls, s):
        with io.StringIO(s) as f:
            return cls.from_dict(read_yaml_input(f))

    @classmethod
    def from_yaml(cls, fname):
        with open(fname) as f:
            return cls.from_dict(read_yaml_input(f))

    # -- scalars (problem.py:319-362), all over the full array incl. ghost cells
    @property
    def q_is_valid(self):
        return (not np.any(np.isnan(self.q))) and (not np.any(self.q[0] < 0.))

    @property
    def mass(self):
        return np.sum(self.q[0] * self.topo[0] * self.grid['dx'] * self.grid['dy'])

    @property
    def kinetic_energy(self):
        return np.sum((self.q[1]**2 + self.q[2]**2) / self.q[0] / 2.)

    @property
    def v_max(self):
        return np.sqrt((self.q[1]**2 + self.q[2]**2) / self.q[0]).max()     # sic: /rho once, problem.py:347

    @property
    def v_sound(self):
        if self.gp_models is not None and self.gp_models.get('press') is not None:
            return self.gp_models['press'].v_sound(self)
        return cl.eos_sound_speed(self.q[0], self.prop).max()               # stress.py:539

    @property
    def dt_crit(self):
        return min(self.grid['dx'], self.grid['dy']) / (self.v_max + self.v_sound)

    @property
    def cfl(self):
        return self.dt / self.dt_crit

    @property
    def converged(self):
        return bool(np.all(np.array(self.residual_buffer) < self.tol))

    # -- closures (stress.py) ---------------------------------------------
    def viscosity(self):
        """Shear viscosity: scalar, or a field with piezo/thinning (stress.py:306-326)."""
        prop = self.prop
        mu = prop['shear']
        if 'piezo' in prop:
            mu = cl.piezoviscosity(self.pressure if prop['EOS'] != 'Bayada' else self.q[0], prop['shear'], prop['piezo'])
        if 'thinning' in prop:
            dpx = np.gradient(self.pressure, self.x[:, 0], axis=0) if self.x.shape[0] > 1 else 0.
            dpy = np.gradient(self.pressure, self.y[0, :], axis=1)
            sr = cl.shear_rate_avg(dpx, dpy, self.topo[0], self.geo['U'], self.geo['V'], mu)
            mu = mu * cl.shear_thinning_factor(sr, mu, prop['thinning'])
        return mu

    def update_closures(self, predictor=False, compute_var=False):
        """Pressure.update, WallStress('x'/'y').update, BulkStress.update in the order of problem.py:534-540."""
        q, h, U, V = self.q, self.topo[:3], self.geo['U'], self.geo['V']
        zeta, Ls = self.prop['bulk'], self.extra[0]
        gpm = self.gp_models or {}
        if gpm.get('press') is not None:
            self.pressure[...] = gpm['press'].predict(self, predictor, compute_var)
        else:
            self.pressure[...] = cl.eos_pressure(q[0], self.prop)
        eta = self.viscosity()
        bot = cl.stress_bottom(q, h, U, V, eta, zeta, Ls)
        top = cl.stress_top(q, h, U, V, eta, zeta, Ls)
        # halves stored by the two WallStress objects add up to the full tensor (stress.py:346-350, problem.py:554-555)
        self.wall_lower[...] = bot
        self.wall_upper[...] = top
        for name, idx in (('shear_x', 4), ('shear_y', 3)):
            if gpm.get(name) is not None:
                lo, up = gpm[name].predict(self, predictor, compute_var)
                self.wall_lower[idx], self.wall_upper[idx] = lo, up
        self.tau_avg[...] = cl.stress_avg(q, h, U, V, self.viscosity(), zeta, Ls)

    # -- time stepping ------------------------------------------------------
    def _pre_run(self):
        # problem.py:412-443
        if self.gp_models:
            for m in self.gp_models.values():
                if m is not None:
                    m.init(self)
        self.step = 0
        self.simtime = 0.
        self.residual = 1.
        self.residual_buffer = deque([self.residual], 5)
        self.dt = self.numerics['CFL'] * self.dt_crit if self.numerics['adaptive'] else self.numerics['dt']
        self.tol = self.numerics['tol']
        self.max_it = self.numerics['max_it']

    def stage(self, direction, dt, predictor=False, compute_var=False):
        """One predictor or corrector stage, problem.py:532-560."""
        self.update_closures(predictor, compute_var)
        fX, fY = predictor_corrector(self.q, self.pressure, self.tau_avg, direction)
        src = source(self.q, self.topo, self.tau_avg, self.wall_lower, self.wall_upper)
        self.q[...] = self.q - dt * (fX / self.grid['dx'] + fY / self.grid['dy'] - src)
        self.communicate_ghost_buffers()

    def update(self):
        # problem.py:509-569
        mc = self.numerics['MC_order']
        switch = (self.step % 2 == 0) * 2 - 1 if mc == 0 else mc
        directions = [[-1, 1], [1, -1]][(switch + 1) // 2]
        dt = self.dt
        q0 = self.q.copy()
        one_before_output = (self.step + 1) % self.options['write_freq'] == 0
        for i, d in enumerate(directions):
            self.stage(d, dt, predictor=(i == 0), compute_var=one_before_output)
        self.q[...] = (self.q + q0) / 2.0
        if self.q_is_valid:
            if self.elastic is not None:            # Topography.update, problem.py:566, with the STORED pressure (stage 2's)
                self.deformation = self.elastic.update(self.pressure)
                self.topo[0] = self.h_undeformed + self.deformation
                self.topo[1] = np.gradient(self.topo[0], axis=0) / self.grid['dx']      # topography.py:273-280
                self.topo[2] = np.gradient(self.topo[0], axis=1) / self.grid['dy']
            self._post_update()
        else:                                       # problem.py:588-610
            self.q[...] = q0
            self.update_closures(False, True)
            self._stop = True

    def _post_update(self):
        # problem.py:571-586
        self.communicate_ghost_buffers()
        ekin = self.kinetic_energy
        self.residual = abs(ekin - self.kinetic_energy_old) / self.kinetic_energy_old / self.cfl
        self.residual_buffer.append(self.residual)
        self.kinetic_energy_old = ekin
        self.step += 1
        self.simtime += self.dt
        if self.numerics['adaptive']:
            self.dt = self.numerics['CFL'] * self.dt_crit

    def run(self):
        # problem.py:368-410 (history rows as written at every write_freq-th step, problem.py:616-626)
        if self.step is None:
            self._pre_run()
        self._stop = False
        self.history = {k: [] for k in ('step', 'time', 'ekin', 'residual', 'vsound')}
        self._record()
        while not self.converged and self.step < self.max_it and not self._stop:
            self.update()
            if self.step % self.options['write_freq'] == 0:
                self._record()

    def _record(self):
        for k, v in zip(('step', 'time', 'ekin', 'residual', 'vsound'),
                        (self.step, self.simtime, self.kinetic_energy, self.residual, self.v_sound)):
            self.history[k].append(v)

    # -- ghost cells (problem.py:676-768) ------------------------------------
    def communicate_ghost_buffers(self):
        g, p = self.grid, self.q
        if all(g['bc_xE_P']):
            p[:, 0, :] = p[:, -2, :].copy()
        else:
            p[g['bc_xE_D'], :1, :] = self._ghost('D', 0, -1)
            p[g['bc_xE_N'], :1, :] = self._ghost('N', 0, -1)
        if all(g['bc_xW_P']):
            p[:, -1, :] = p[:, 1, :].copy()
        else:
            p[g['bc_xW_D'], -1:, :] = self._ghost('D', 0, 1)
            p[g['bc_xW_N'], -1:, :] = self._ghost('N', 0, 1)
        if all(g['bc_yS_P']):
            p[:, :, 0] = p[:, :, -2].copy()
        else:
            p[g['bc_yS_D'], :, :1] = self._ghost('D', 1, -1)
            p[g['bc_yS_N'], :, :1] = self._ghost('N', 1, -1)
        if all(g['bc_yN_P']):
            p[:, :, -1] = p[:, :, 1].copy()
        else:
            p[g['bc_yN_D'], :, -1:] = self._ghost('D', 1, 1)
            p[g['bc_yN_N'], :, -1:] = self._ghost('N', 1, 1)

    def _ghost(self, bc_type, axis, direction):
        # problem.py:709-768: value side is xW for the low-x ghost, xE for high-x,
        # but yN for the low-y ghost and yS for high-y (reference quirk).
        g, p = self.grid, self.q
        if axis == 0:
            side = 'xE' if direction > 0 else 'xW'
            mask = g[f'bc_{side}_{bc_type}']
            adj = p[mask, -2:-1, :] if direction > 0 else p[mask, 1:2, :]
        else:
            side = 'yS' if direction > 0 else 'yN'
            mask = g[f'bc_{side}_{bc_type}']
            adj = p[mask, :, -2:-1] if direction > 0 else p[mask, :, 1:2]
        if bc_type == 'D':
            return (g[f'bc_{side}_D_val'] - 0.5 * adj) / 0.5
        return (0.5 * adj) / 0.5
