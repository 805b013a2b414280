"""N > 1 path on CPU: gapflow_amd.slab's partitioning, halo-message pairing and rank-ordered reduction,
driven over gloo (world_size 2 and 3) with a CPU engine built from the oracle.

The engine below obeys the same split-step protocol as the HIP engine (csrc/api.hip: gpf_step_local /
gpf_step_commit): a full MacCormack step on the slab from halo rows one cell deep, local ghost rules,
an 8-double record, ONE row exchanged per neighbour and step.  The assembled slabs must reproduce the
single-process oracle.
"""
import io
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

PERIODIC = """
options: {silent: True}
grid: {Nx: 37, Ny: 12, dx: 2.e-5, dy: 2.e-5}
geometry: {type: journal, CR: 1.e-2, eps: 0.6, U: 0.1, V: 0.03}
numerics: {CFL: 0.4, adaptive: 1, MC_order: 0, tol: 1.e-12, max_it: 1000}
properties: {EOS: DH, shear: 0.0794, bulk: 0.01, rho0: 877.7007, C1: 3.5e9}
"""
DIRICHLET = """
options: {silent: True}
grid: {Nx: 30, Ny: 9, Lx: 0.1, Ly: 0.03, xE: ['D', 'N', 'N'], xW: ['D', 'N', 'N'], xE_D: 877.7007, xW_D: 875.,
       yS: ['D', 'N', 'N'], yN: ['D', 'N', 'N'], yS_D: 877., yN_D: 876.}
geometry: {type: inclined, hmax: 6.6e-5, hmin: 1.e-5, U: 50., V: 4.}
numerics: {CFL: 0.4, adaptive: 1, MC_order: -1, tol: 1.e-12, max_it: 1000}
properties: {EOS: DH, shear: 0.0794, bulk: 0., rho0: 877.7007}
"""


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


class OracleSlabEngine:
    """CPU stand-in for HipSlabEngine (tests only)."""

    def __init__(self, input_dict, layout, torch):
        from oracle.problem import OracleProblem
        from oracle.topography import build_topography
        from oracle import closures as cl
        self.cl, self.torch, self.L = cl, torch, layout
        d = dict(input_dict)
        d['grid'] = layout.local_grid(input_dict['grid'])
        self.p = p = OracleProblem.from_dict(d)
        gtopo, _, _ = build_topography(input_dict['grid'], input_dict['geometry'])
        p.topo = gtopo[:, layout.rows()].copy()
        self.kinds = (layout.kind_lo, layout.kind_hi)
        ny2 = d['grid']['Ny'] + 2
        self.rowlen = 3 * ny2
        self.msg = torch.zeros(2 * self.rowlen + 8, dtype=torch.float64)      # [first row | last row | record]
        full_bc = p.communicate_ghost_buffers

        def bc():   # y rules everywhere, x rules only on physical edges (halo rows belong to the exchange)
            keep = [p.q[:, 0, :].copy(), p.q[:, -1, :].copy()]
            full_bc()
            if self.kinds[0]:
                p.q[:, 0, :] = keep[0]
            if self.kinds[1]:
                p.q[:, -1, :] = keep[1]
        p.communicate_ghost_buffers = bc

    def message(self):
        return self.msg

    def pre_run(self, dt, ekin_old):
        p = self.p
        p._pre_run()
        p.dt, p.kinetic_energy_old = dt, ekin_old

    def local_scalars(self):
        """(ekin, v2max, c2max, flags) over the cells this slab owns, mirroring csrc/aux_kernels.hip."""
        p, cl = self.p, self.cl
        q = p.q
        w = np.ones(q.shape[1])
        w[0] = 1. if self.kinds[0] == 0 else 0.          # outer rows count only when they are physical ghost rows
        w[-1] = 1. if self.kinds[1] == 0 else 0.
        if self.kinds[0] == 2:
            w[1] += 1.                                     # stands in for the far slab's periodic ghost row
        if self.kinds[1] == 2:
            w[-2] += 1.
        v2 = (q[1]**2 + q[2]**2) / q[0]
        own = w > 0
        c2 = cl.eos_sound_speed(q[0][own], p.prop)**2
        flags = (1 if np.isnan(q[:, own]).any() else 0) | (2 if (q[0][own] < 0).any() else 0)
        return float(np.sum(w[:, None] * v2 / 2.)), float(v2[own].max()), float(c2.max()), flags

    def step_local(self, honor_stop):
        p = self.p
        mc = p.numerics['MC_order']
        switch = (p.step % 2 == 0) * 2 - 1 if mc == 0 else mc
        directions = [[-1, 1], [1, -1]][(switch + 1) // 2]
        self.q0 = p.q.copy()
        for i, d in enumerate(directions):
            p.stage(d, p.dt, predictor=(i == 0))
        p.q[...] = (p.q + self.q0) / 2.0
        p.communicate_ghost_buffers()
        e, v2, c2, fl = self.local_scalars()
        n = self.rowlen
        self.msg[:n] = self.torch.from_numpy(p.q[:, 1, :].reshape(-1).copy())
        self.msg[n:2 * n] = self.torch.from_numpy(p.q[:, -2, :].reshape(-1).copy())
        self.msg[2 * n:] = self.torch.tensor([e, v2, c2, fl, 0, 0, 0, 0], dtype=self.torch.float64)

    def commit(self, gathered, honor_stop, rank_lo, rank_hi):
        p = self.p
        ny2, n = p.q.shape[2], self.rowlen
        all_msgs = gathered.numpy().reshape(-1, 2 * n + 8)
        if self.kinds[0] and rank_lo >= 0:
            p.q[:, 0, :] = all_msgs[rank_lo, n:2 * n].reshape(3, ny2)       # the lower neighbour's LAST row
        if self.kinds[1] and rank_hi >= 0:
            p.q[:, -1, :] = all_msgs[rank_hi, :n].reshape(3, ny2)           # the upper neighbour's FIRST row
        g = all_msgs[:, 2 * n:]
        ekin, v2, c2 = 0.0, 0.0, 0.0
        for r in g:                                         # rank order, like k_commit_gathered
            ekin += r[0]
            v2, c2 = max(v2, r[1]), max(c2, r[2])
        assert not g[:, 3].any()
        hmin = min(p.grid['dx'], p.grid['dy'])
        dt_crit = hmin / (np.sqrt(v2) + np.sqrt(c2))
        p.residual = abs(ekin - p.kinetic_energy_old) / p.kinetic_energy_old / (p.dt / dt_crit)
        p.kinetic_energy_old = ekin
        p.step += 1
        p.simtime += p.dt
        if p.numerics['adaptive']:
            p.dt = p.numerics['CFL'] * dt_crit


def _worker(rank, world, port, text, nsteps, out_dir):
    import torch
    import torch.distributed as dist
    from gapflow_amd.slab import SlabLayout, SlabDriver
    from oracle.config import read_yaml_input
    from oracle.problem import OracleProblem
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        d = read_yaml_input(io.StringIO(text))
        layout = SlabLayout(d['grid'], rank, world)
        engine = OracleSlabEngine(d, layout, torch)
        ref = OracleProblem.from_dict(read_yaml_input(io.StringIO(text)))
        ref._pre_run()
        engine.pre_run(ref.dt, ref.kinetic_energy_old)       # domain-wide initial scalars
        driver = SlabDriver(engine, layout, dist, torch)
        driver.advance(nsteps)
        for _ in range(nsteps):
            ref.update()
        np.savez(os.path.join(out_dir, f'rank{rank}.npz'), q=engine.p.q, ref=ref.q[:, layout.rows()], dt=engine.p.dt,
                 ref_dt=ref.dt, res=engine.p.residual, ref_res=ref.residual, lo=layout.lo, hi=layout.hi)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('text,world', [(PERIODIC, 2), (PERIODIC, 3), (DIRICHLET, 2), (DIRICHLET, 3)])
def test_slabs_reproduce_the_serial_run(tmp_path, text, world):
    import torch.multiprocessing as mp
    nsteps = 12
    mp.spawn(_worker, args=(world, _free_port(), text, nsteps, str(tmp_path)), nprocs=world, join=True)
    covered = []
    for r in range(world):
        z = np.load(tmp_path / f'rank{r}.npz')
        for c in range(3):
            scale = np.abs(z['ref'][c]).max()
            assert np.abs(z['q'][c] - z['ref'][c]).max() <= 1e-11 * scale, f'rank {r} component {c}'
        np.testing.assert_allclose(z['dt'], z['ref_dt'], rtol=1e-12)
        np.testing.assert_allclose(z['res'], z['ref_res'], rtol=1e-6, atol=1e-10)
        covered += list(range(int(z['lo']), int(z['hi']) + 1))
    assert covered == list(range(1, 38 if text is PERIODIC else 31))


@pytest.mark.parametrize('text', [PERIODIC, DIRICHLET], ids=['periodic', 'dirichlet'])
def test_eight_ranks_in_one_process_reproduce_the_serial_run(text):
    """gapflow_amd.slab.ThreadWorld (eight ranks as threads of one process: how the GPU tests rehearse the 8-slab runs on a
    box that allows six processes on its card) drives the same SlabDriver / engine protocol as the gloo groups above."""
    import torch
    from gapflow_amd.slab import SlabLayout, SlabDriver, ThreadWorld
    from oracle.config import read_yaml_input
    from oracle.problem import OracleProblem
    nsteps, world = 12, 8
    ref = OracleProblem.from_dict(read_yaml_input(io.StringIO(text)))
    ref._pre_run()
    dt0, ekin0 = ref.dt, ref.kinetic_energy_old
    for _ in range(nsteps):
        ref.update()

    def rank_body(group):
        d = read_yaml_input(io.StringIO(text))
        layout = SlabLayout(d['grid'], group.get_rank(), world)
        engine = OracleSlabEngine(d, layout, torch)
        engine.pre_run(dt0, ekin0)
        SlabDriver(engine, layout, group, torch).advance(nsteps)
        t = torch.tensor([float(group.get_rank())], dtype=torch.float64)
        group.all_reduce(t, op=group.ReduceOp.MAX)
        assert float(t) == world - 1
        return layout, engine.p.q.copy(), engine.p.dt, engine.p.residual

    covered = []
    for layout, q, dt, res in ThreadWorld(world, torch).run(rank_body):
        for c in range(3):
            assert np.abs(q[c] - ref.q[c, layout.rows()]).max() <= 1e-11 * np.abs(ref.q[c]).max(), f'rank {layout.rank} component {c}'
        np.testing.assert_allclose(dt, ref.dt, rtol=1e-12)
        np.testing.assert_allclose(res, ref.residual, rtol=1e-6, atol=1e-10)
        covered += list(range(layout.lo, layout.hi + 1))
    assert covered == list(range(1, ref.grid['Nx'] + 1))


def test_thread_world_reports_the_failing_rank():
    import torch
    from gapflow_amd.slab import ThreadWorld

    def body(group):
        if group.get_rank() == 2:
            raise ValueError('boom')
        group.barrier()

    with pytest.raises(RuntimeError, match='rank 2 .* boom'):
        ThreadWorld(4, torch).run(body)


def test_partition_and_layout():
    from gapflow_amd.slab import partition, SlabLayout
    from gapflow_amd.io import sanitize_grid
    assert partition(10, 3) == [(1, 4), (5, 7), (8, 10)]
    assert partition(4096, 8)[-1] == (3585, 4096)
    with pytest.raises(ValueError):
        partition(3, 4)
    per = sanitize_grid({'Nx': 16, 'dx': 1., 'Ny': 4, 'dy': 1.})
    lay = [SlabLayout(per, r, 4) for r in range(4)]
    assert [(l.kind_lo, l.kind_hi) for l in lay] == [(2, 1), (1, 1), (1, 1), (1, 2)]
    assert [(l.lower, l.upper) for l in lay] == [(3, 1), (0, 2), (1, 3), (2, 0)]
    dn = sanitize_grid({'Nx': 16, 'dx': 1., 'Ny': 4, 'dy': 1., 'xE': ['D', 'N', 'N'], 'xW': ['D', 'N', 'N']})
    lay = [SlabLayout(dn, r, 2) for r in range(2)]
    assert [(l.kind_lo, l.kind_hi, l.lower, l.upper) for l in lay] == [(0, 1, None, 1), (1, 0, 0, None)]
    one = SlabLayout(per, 0, 1)
    assert (one.kind_lo, one.kind_hi, one.lower, one.upper) == (0, 0, None, None)
