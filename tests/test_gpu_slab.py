"""Slab engine on the GPU.  A 1-GPU box can only host a one-rank group over RCCL, which still runs the whole
split-step plumbing (library-owned buffers wrapped as torch tensors, record all-gather, commit)."""
import io
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SIM = """
options: {silent: True}
grid: {Nx: 150, Ny: 70, dx: 1.e-5, dy: 1.e-5}
geometry: {type: journal, CR: 1.e-2, eps: 0.7, U: 0.1, V: 0.02}
numerics: {CFL: 0.5, adaptive: 1, MC_order: 0, tol: 1.e-12, max_it: 1000}
properties: {EOS: DH, shear: 0.0794, bulk: 0., rho0: 877.7007}
"""


@pytest.mark.parametrize('graph', ['0', '1'])
def test_one_rank_group_equals_serial_problem(hiplib, graph, monkeypatch):
    monkeypatch.setenv('GPF_SLAB_GRAPH', graph)      # '1': pairs of steps replayed from a captured hipGraph
    import torch
    import torch.distributed as dist
    from gapflow_amd import Problem
    from gapflow_amd.slab import SlabProblem
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        slab = SlabProblem.from_string(SIM)
        slab.pre_run()
        slab.advance(13)
        slab.advance(7)
        st = slab.state()
        assert (slab.driver.graph is not None) == (graph == '1')
        serial = Problem.from_string(SIM)
        serial._pre_run()
        serial._advance(20, honor_stop=False)
        assert st.step == 20 and st.invalid == 0
        assert st.dt == serial.dt
        # Ekin is summed over a different block partition (the slab launches extra row-shipping blocks)
        np.testing.assert_allclose(st.residual, serial.residual, rtol=1e-9)
        np.testing.assert_array_equal(slab.local_q(), serial.q)
    finally:
        dist.destroy_process_group()


# RCCL refuses two ranks on one device ("Duplicate GPU detected"): on a 1-GPU box the multi-rank behaviour of the HIP slab
# engine is exercised with 2-3 processes sharing the GPU and collectives staged through host memory over gloo.
from gapflow_amd.slab import HostStagedGroup as StagedGloo      # noqa: E402


def _slab_worker(rank, world, port, text, nsteps, out_dir, p2p=False):
    import torch
    import torch.distributed as dist
    from gapflow_amd.slab import SlabProblem
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        slab = SlabProblem.from_string(text, device=0, dist=StagedGloo(dist, torch))
        if p2p:
            assert slab.connect_p2p(), 'HIP IPC mapping of the peers\' mailboxes failed'
        slab.pre_run()
        slab.advance(nsteps)
        st = slab.state()
        np.savez(os.path.join(out_dir, f'rank{rank}.npz'), q=slab.local_q(), lo=slab.layout.lo, hi=slab.layout.hi,
                 dt=st.dt, residual=st.residual, step=st.step, ekin=st.ekin, invalid=st.invalid,
                 db_size=0 if slab.database is None else slab.database.size)
    finally:
        dist.destroy_process_group()


DIRICHLET = """
options: {silent: True}
grid: {Nx: 150, Ny: 70, Lx: 0.1, Ly: 0.05, xE: ['D', 'N', 'N'], xW: ['D', 'N', 'N'], xE_D: 877.7007, xW_D: 875.,
       yS: ['D', 'N', 'N'], yN: ['D', 'N', 'N'], yS_D: 877., yN_D: 876.}
geometry: {type: inclined, hmax: 6.6e-5, hmin: 1.e-5, U: 50., V: 4.}
numerics: {CFL: 0.4, adaptive: 1, MC_order: -1, tol: 1.e-12, max_it: 1000}
properties: {EOS: DH, shear: 0.0794, bulk: 0., rho0: 877.7007}
"""


@pytest.mark.parametrize('p2p', [False, True], ids=['allgather', 'p2p'])
@pytest.mark.parametrize('text,world', [(SIM, 2), (SIM, 3), (DIRICHLET, 2), (DIRICHLET, 3)],
                         ids=['periodic-2', 'periodic-3', 'dirichlet-2', 'dirichlet-3'])
def test_multi_rank_engine_matches_serial(hiplib, tmp_path, text, world, p2p):
    """2 and 3 processes sharing this GPU, each owning an x-slab: assembled result == the one-handle run.
    `p2p`: rows and records travel through IPC-mapped mailboxes written and polled by the step's own kernels
    (the all-gather variant moves them through the process group)."""
    import torch.multiprocessing as mp
    from gapflow_amd import Problem
    nsteps = 20
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_slab_worker, args=(world, port, text, nsteps, str(tmp_path), p2p), nprocs=world, join=True)
    serial = Problem.from_string(text)
    serial._pre_run()
    serial._advance(nsteps, honor_stop=False)
    ref = serial.q
    for r in range(world):
        z = np.load(tmp_path / f'rank{r}.npz')
        lo, hi = int(z['lo']), int(z['hi'])
        assert int(z['step']) == nsteps and int(z['invalid']) == 0
        for c in range(3):
            scale = np.abs(ref[c]).max() or 1.
            # all rows of the slab including its outer rows (halo / seam / physical ghost)
            assert np.abs(z['q'][c] - ref[c, lo - 1:hi + 2]).max() <= 1e-11 * scale, f'rank {r} comp {c}'
        np.testing.assert_allclose(z['dt'], serial.dt, rtol=1e-12)
        np.testing.assert_allclose(z['ekin'], serial.kinetic_energy, rtol=1e-12)
        np.testing.assert_allclose(z['residual'], serial.residual, rtol=1e-6, atol=1e-10)


GP_SIM = """
options: {silent: True, write_freq: 1000}
grid: {Nx: 41, Ny: 24, Lx: 0.05, Ly: 0.03, xE: ['D', 'N', 'N'], xW: ['D', 'N', 'N'], xE_D: 877.7007, xW_D: 877.7007,
       yS: ['P', 'P', 'P'], yN: ['P', 'P', 'P']}
geometry: {type: inclined, hmax: 6.6e-5, hmin: 2.e-5, U: 20., V: 3.}
numerics: {CFL: 0.3, adaptive: 1, tol: 1.e-10, max_it: 100}
properties: {EOS: DH, shear: 0.0794, bulk: 0., rho0: 877.7007, C1: 3.5e9}
gp:
    press: {atol: 1., rtol: 0.1, obs_stddev: 1.e5, active_learning: AL_PRESS, max_steps: 3, pause_steps: 4}
    shear: {atol: 1., rtol: 0.1, obs_stddev: 500., active_learning: False}
db: {init_size: 24, init_method: lhc, init_width: 0.001}
"""

GP_PERIODIC = GP_SIM.replace("xE: ['D', 'N', 'N'], xW: ['D', 'N', 'N'], xE_D: 877.7007, xW_D: 877.7007",
                             "xE: ['P', 'P', 'P'], xW: ['P', 'P', 'P']").replace(
    'type: inclined, hmax: 6.6e-5, hmin: 2.e-5', 'type: journal, CR: 1.e-2, eps: 0.6')


@pytest.mark.parametrize('text,world', [(GP_SIM.replace('AL_PRESS', 'False'), 2), (GP_SIM.replace('AL_PRESS', 'True'), 3),
                                        (GP_PERIODIC.replace('AL_PRESS', 'False'), 2)],
                         ids=['dirichlet-2', 'dirichlet-active-learning-3', 'periodic-2'])
def test_multi_rank_gp_closures_match_serial(hiplib, tmp_path, text, world):
    """Surrogate closures across slabs (stage-wise step, rows exchanged after each stage, replicated database,
    domain-wide active learning) against the one-handle Problem.update() on the same input."""
    import torch.multiprocessing as mp
    from gapflow_amd import Problem
    nsteps = 6
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_slab_worker, args=(world, port, text, nsteps, str(tmp_path)), nprocs=world, join=True)
    serial = Problem.from_string(text)
    serial._pre_run()
    for _ in range(nsteps):
        serial.update()
    ref = serial.q
    for r in range(world):
        z = np.load(tmp_path / f'rank{r}.npz')
        lo, hi = int(z['lo']), int(z['hi'])
        assert int(z['step']) == nsteps and int(z['invalid']) == 0
        assert int(z['db_size']) == serial.database.size
        for c in range(3):
            scale = np.abs(ref[c]).max() or 1.
            assert np.abs(z['q'][c] - ref[c, lo - 1:hi + 2]).max() <= 1e-10 * scale, f'rank {r} comp {c}'
        np.testing.assert_allclose(z['dt'], serial.dt, rtol=1e-10)
        np.testing.assert_allclose(z['ekin'], serial.kinetic_energy, rtol=1e-10)
        np.testing.assert_allclose(z['residual'], serial.residual, rtol=1e-5, atol=1e-10)


def _slab_run_worker(rank, world, port, text, p2p):
    import torch
    import torch.distributed as dist
    from gapflow_amd.slab import SlabProblem
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        slab = SlabProblem.from_string(text, device=0, dist=StagedGloo(dist, torch))
        if p2p:
            assert slab.connect_p2p()
        st = slab.run()
        assert st.step == 45 and st.invalid == 0
    finally:
        dist.destroy_process_group()


RUN_SIM = """
options: {{output: {out}, write_freq: 20, use_tstamp: False, silent: False}}
grid: {{Nx: 150, Ny: 70, dx: 1.e-5, dy: 1.e-5}}
geometry: {{type: journal, CR: 1.e-2, eps: 0.7, U: 0.1, V: 0.02}}
numerics: {{CFL: 0.5, adaptive: 1, MC_order: 0, tol: 1.e-12, max_it: 45}}
properties: {{EOS: DH, shear: 0.0794, bulk: 0., rho0: 877.7007}}
"""


@pytest.mark.parametrize('p2p', [False, True], ids=['allgather', 'p2p'])
def test_slab_run_writes_the_same_files_as_the_serial_run(hiplib, tmp_path, p2p):
    """SlabProblem.run(): frames at write_freq and at the end, history.csv, config.yml -- by rank 0, equal to
    Problem.run() on the same input."""
    import csv
    import torch.multiprocessing as mp
    from scipy.io import netcdf_file
    from gapflow_amd import Problem
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_slab_run_worker, args=(3, port, RUN_SIM.format(out=str(tmp_path / 'slab')), p2p), nprocs=3, join=True)
    Problem.from_string(RUN_SIM.format(out=str(tmp_path / 'serial'))).run()
    for f in ('config.yml', 'history.csv', 'sol.nc', 'topo.nc'):
        assert (tmp_path / 'slab' / f).exists(), f
    rows = [list(csv.reader(open(tmp_path / d / 'history.csv'))) for d in ('slab', 'serial')]
    assert rows[0][0] == rows[1][0] and len(rows[0]) == len(rows[1]) == 5          # header + steps 0, 20, 40, 45
    a, b = (np.array(r[1:], float) for r in rows)
    np.testing.assert_array_equal(a[:, 0], [0, 20, 40, 45])
    np.testing.assert_allclose(a[:, 1:3], b[:, 1:3], rtol=1e-11)                  # time, ekin
    np.testing.assert_allclose(a[:, 3], b[:, 3], rtol=1e-5, atol=1e-12)            # residual (difference of near-equal sums)
    np.testing.assert_allclose(a[:, 4], b[:, 4], rtol=1e-11)                      # vsound
    with netcdf_file(str(tmp_path / 'slab' / 'sol.nc'), mmap=False) as fa, netcdf_file(str(tmp_path / 'serial' / 'sol.nc'), mmap=False) as fb:
        va, vb = fa.variables['solution'][:], fb.variables['solution'][:]
        assert va.shape == vb.shape and va.shape[0] == 4
        assert np.abs(va - vb).max() <= 1e-11 * np.abs(vb).max()
        # Pressure frames.  The serial run writes what the reference's field holds -- the corrector stage's closure, on the
        # predictor's field.  Rank 0's writer sees the gathered STATE only (the predictor's field of a slab would need its
        # neighbours' predictor rows): its frames hold the closures of the frame's own state.
        from oracle import closures as ocl
        from oracle.config import read_yaml_input as oracle_reader
        prop = oracle_reader(io.StringIO(RUN_SIM.format(out='x')))['properties']
        pa = fa.variables['pressure'][:]
        for k in range(pa.shape[0]):
            ref = ocl.eos_pressure(va[k, 0, 0], prop)
            # (dp/drho = c^2 ~ 1e8 for this law: one ulp of the density is 1e-5 Pa, 1e-10 of the pressure scale)
            assert np.abs(pa[k] - ref).max() <= 1e-9 * np.abs(ref).max(), f'frame {k}'


def _slab_gp_run_worker(rank, world, port, text):
    import torch
    import torch.distributed as dist
    from gapflow_amd.slab import SlabProblem
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        st = SlabProblem.from_string(text, device=0, dist=StagedGloo(dist, torch)).run()
        assert st.step == 6 and st.invalid == 0
    finally:
        dist.destroy_process_group()


def test_slab_run_with_surrogates_writes_the_same_frames(hiplib, tmp_path):
    """SlabProblem.run() with GP closures: the frames written by rank 0 (through a whole-domain Problem that mirrors the
    slab's models) equal those of the serial run."""
    import torch.multiprocessing as mp
    from scipy.io import netcdf_file
    from gapflow_amd import Problem
    base = GP_SIM.replace('AL_PRESS', 'False').replace('max_it: 100', 'max_it: 6')
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    text = lambda d: base.replace('options: {silent: True, write_freq: 1000}', f'options: {{output: {tmp_path / d}, write_freq: 3, use_tstamp: False}}')
    mp.spawn(_slab_gp_run_worker, args=(2, port, text('slab')), nprocs=2, join=True)
    Problem.from_string(text('serial')).run()
    with netcdf_file(str(tmp_path / 'slab' / 'sol.nc'), mmap=False) as fa, netcdf_file(str(tmp_path / 'serial' / 'sol.nc'), mmap=False) as fb:
        for name in ('solution', 'pressure', 'wall_stress_xz'):
            va, vb = fa.variables[name][:], fb.variables[name][:]
            assert va.shape == vb.shape and va.shape[0] == 3, name          # frames at steps 0, 3, 6
            if name == 'solution':
                assert np.abs(va - vb).max() <= 1e-9 * np.abs(vb).max(), name
            else:
                # closures of the frame's own state (slab writer) against the corrector stage's (serial run, the reference's
                # semantics): the same surrogate evaluated on fields one predictor stage apart
                assert np.abs(va[0] - vb[0]).max() <= 1e-9 * np.abs(vb[0]).max(), name      # frame 0: the initial state in both
                assert np.abs(va - vb).max() <= 0.2 * np.abs(vb).max(), name


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize('p2p', [False, True], ids=['allgather', 'p2p'])
def test_n_slabs_are_bitwise_the_one_slab_run(hiplib, tmp_path, p2p):
    """Determinism across decompositions on an all-periodic problem: the field after 20 steps on 2 and on 3 slabs is
    bit for bit the field of the 1-slab run.  Every stage-1 value -- whether the stencil forms it or the ghost kernels
    on the far side of a seam -- goes through the same spelled-out FMA sequence (predictor_value, step_kernel.hip), dt
    comes from exact maxima reduced in rank order, and only the kinetic-energy SUM (hence the residual) depends on the
    partition."""
    import torch.multiprocessing as mp
    nsteps = 20
    runs = {}
    for world in (1, 2, 3):
        out = tmp_path / f'w{world}'
        out.mkdir()
        mp.spawn(_slab_worker, args=(world, _free_port(), SIM, nsteps, str(out), p2p and world > 1), nprocs=world, join=True)
        parts = [np.load(out / f'rank{r}.npz') for r in range(world)]
        assert all(int(z['step']) == nsteps and int(z['invalid']) == 0 for z in parts)
        runs[world] = (np.concatenate([z['q'][:, 1:-1] for z in parts], axis=1), [float(z['dt']) for z in parts])
    q1, dt1 = runs[1]
    for world in (2, 3):
        qn, dtn = runs[world]
        assert qn.shape == q1.shape
        assert np.array_equal(qn, q1), f'{world} slabs: {np.abs(qn - q1).max():.3e} away from the 1-slab field'
        assert all(d == dt1[0] for d in dtn)


WIDE = SIM.replace('Nx: 150, Ny: 70', 'Nx: 90, Ny: 300')         # three strips of the step kernel, x-only gap
ASPERITY = """
options: {silent: True}
grid: {Nx: 60, Ny: 270, Lx: 6.e-4, Ly: 2.7e-3}
geometry: {type: asperity, hmin: 2.e-6, hmax: 1.e-5, num: 1, U: 10., V: 5.}
numerics: {CFL: 0.4, adaptive: 1, MC_order: 0, tol: 1.e-12, max_it: 1000}
properties: {EOS: DH, shear: 0.0794, bulk: 0., rho0: 877.7007}
"""


# the general closure (piezo-viscosity) and another equation of state across the periodic seam
PIEZO = SIM.replace('rho0: 877.7007}', 'rho0: 877.7007, piezo: {name: Barus, aB: 2.e-8}}')
MURNAGHAN = SIM.replace('EOS: DH', 'EOS: MT').replace('rho0: 877.7007}', 'rho0: 700., P0: 0.101e6, K: 0.557e9, n: 7.33}')


@pytest.mark.parametrize('p2p', [False, True], ids=['allgather', 'p2p'])
@pytest.mark.parametrize('text', [SIM, DIRICHLET, WIDE, ASPERITY, PIEZO, MURNAGHAN],
                         ids=['periodic', 'dirichlet', 'wide', 'asperity-planes', 'piezo', 'murnaghan-tait'])
def test_fused_slab_step_is_bitwise_the_split_step(hiplib, tmp_path, monkeypatch, text, p2p):
    """The slab step with its edge work inside k_step2 (message rows, stage-1 field across the seam, record; one launch
    plus the commit) against the older split form (GPF_STEP_UNFUSED_EDGES=1: k_ghost_stage1 / k_ghost_fill around the
    stencil): same field bit for bit -- outer rows included -- and the same dt on 3 slabs (Dowson-Higginson; a few ulps with
    an equation of state that calls pow()); the kinetic energy is summed in a different order."""
    import torch.multiprocessing as mp
    nsteps, world = 21, 3
    runs = {}
    for mode in ('fused', 'split'):
        if mode == 'split':
            monkeypatch.setenv('GPF_STEP_UNFUSED_EDGES', '1')
        else:
            monkeypatch.delenv('GPF_STEP_UNFUSED_EDGES', raising=False)
        out = tmp_path / mode
        out.mkdir()
        mp.spawn(_slab_worker, args=(world, _free_port(), text, nsteps, str(out), p2p), nprocs=world, join=True)
        runs[mode] = [np.load(out / f'rank{r}.npz') for r in range(world)]
    for a, b in zip(runs['fused'], runs['split']):
        assert int(a['step']) == nsteps == int(b['step']) and int(a['invalid']) == 0
        if text is MURNAGHAN:
            # the Murnaghan-Tait pressure goes through pow(): its call sites in the two forms are contracted differently by
            # the compiler (only the Dowson-Higginson closures spell out every FMA), a few ulps apart
            for c in range(3):
                assert np.abs(a['q'][c] - b['q'][c]).max() <= 1e-13 * np.abs(b['q'][c]).max()
            np.testing.assert_allclose(float(a['dt']), float(b['dt']), rtol=1e-13)
            continue
        assert np.array_equal(a['q'], b['q']), f"{np.abs(a['q'] - b['q']).max():.3e}"
        assert float(a['dt']) == float(b['dt'])
        np.testing.assert_allclose(a['ekin'], b['ekin'], rtol=1e-13)


def _timeout_worker(rank, world, port, text, out_dir):
    """Rank 0 steps once; rank 1 connects its mailbox and then never sends: rank 0's launch must stop the handle."""
    import time
    import torch
    import torch.distributed as dist
    from gapflow_amd import _lib
    from gapflow_amd.slab import SlabProblem
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        slab = SlabProblem.from_string(text, device=0, dist=StagedGloo(dist, torch))
        assert slab.connect_p2p(), 'HIP IPC mapping of the peers\' mailboxes failed'
        slab.pre_run()
        _lib.check(slab.lib.gpf_p2p_set_timeout(slab._h, 0.25))
        if rank == 0:
            t0 = time.perf_counter()
            slab.advance(3)                     # three steps enqueued: the first waits and times out, the others are skipped
            outcome = 'completed'
            try:
                slab.state()
            except RuntimeError as e:
                outcome = 'timeout' if 'did not deliver' in str(e) else f'other: {e}'
            waited = time.perf_counter() - t0
            sc = _lib.GpfScalars()
            _lib.check(slab.lib.gpf_state(slab._h, C_byref(sc)))
            # a second attempt on the stopped handle returns at once and changes nothing
            slab.advance(2)
            sc2 = _lib.GpfScalars()
            _lib.check(slab.lib.gpf_state(slab._h, C_byref(sc2)))
            np.savez(os.path.join(out_dir, 'timeout.npz'), outcome=outcome, waited=waited, step=sc.step, invalid=sc.invalid,
                     step2=sc2.step, invalid2=sc2.invalid)
        dist.barrier()                          # rank 1 keeps its mailbox mapped until rank 0 is done
    finally:
        dist.destroy_process_group()


def C_byref(x):
    import ctypes
    return ctypes.byref(x)


def test_silent_peer_stops_the_handle_once(hiplib, tmp_path):
    """gpf_step_p2p against a peer that never delivers: the wait is bounded, EVERY block of the waiting
    launch goes through the arrival counter, the last one records the time-out -- no commit, no sequence advance, the
    arrival counters are back at zero -- and later launches see a stopped handle."""
    import torch.multiprocessing as mp
    mp.spawn(_timeout_worker, args=(2, _free_port(), SIM, str(tmp_path)), nprocs=2, join=True)
    z = np.load(tmp_path / 'timeout.npz')
    assert str(z['outcome']) == 'timeout', str(z['outcome'])
    assert int(z['invalid']) == 3 and int(z['step']) == 0            # nothing was committed
    assert int(z['invalid2']) == 3 and int(z['step2']) == 0
    assert 0.2 < float(z['waited']) < 20.0, float(z['waited'])       # bounded by the configured 0.25 s, not by 30 s


def _nccl_worker(rank, world, port, text, nsteps, out_dir, p2p):
    import torch
    import torch.distributed as dist
    from gapflow_amd.slab import SlabProblem
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    torch.cuda.set_device(rank)
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', rank))
    try:
        slab = SlabProblem.from_string(text, device=rank)
        if p2p:
            assert slab.connect_p2p(), 'HIP IPC mapping of the peers\' mailboxes failed'
        slab.pre_run()
        slab.advance(nsteps)
        st = slab.state()
        np.savez(os.path.join(out_dir, f'rank{rank}.npz'), q=slab.local_q(), lo=slab.layout.lo, hi=slab.layout.hi,
                 dt=st.dt, step=st.step, invalid=st.invalid)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('p2p', [False, True], ids=['allgather', 'p2p'])
def test_two_gpus_over_rccl_match_serial(hiplib, tmp_path, p2p):
    """The real thing where the box has it: one rank per GPU, RCCL (`nccl` backend) over xGMI, both transports.  Skipped
    on the one-GPU development and test boxes -- nothing with more than one GPU has run on hardware so far."""
    if hiplib.gpf_device_count() < 2:
        pytest.skip('needs two GPUs')
    import torch.multiprocessing as mp
    from gapflow_amd import Problem
    nsteps = 20
    mp.spawn(_nccl_worker, args=(2, _free_port(), SIM, nsteps, str(tmp_path), p2p), nprocs=2, join=True)
    serial = Problem.from_string(SIM)
    serial._pre_run()
    serial._advance(nsteps, honor_stop=False)
    for r in range(2):
        z = np.load(tmp_path / f'rank{r}.npz')
        lo, hi = int(z['lo']), int(z['hi'])
        assert int(z['step']) == nsteps and int(z['invalid']) == 0
        for c in range(3):
            scale = np.abs(serial.q[c]).max() or 1.
            assert np.abs(z['q'][c] - serial.q[c, lo - 1:hi + 2]).max() <= 1e-11 * scale
        np.testing.assert_allclose(z['dt'], serial.dt, rtol=1e-12)


THINNING_SLAB = """
options: {{silent: True}}
grid: {{Nx: 48, Ny: 10, Lx: 0.05, Ly: 0.01{bc}}}
geometry: {{type: {geo}, U: 10., V: 1.}}
numerics: {{CFL: 0.4, adaptive: 1, max_it: 100}}
properties:
    EOS: DH
    shear: 0.05
    bulk: 0.
    rho0: 877.7007
    thinning: {{name: {law}}}
"""


@pytest.mark.parametrize('law,bc,geo,world', [
    ('Eyring, tauE: 5.e5', ", xE: ['D', 'N', 'N'], xW: ['D', 'N', 'N'], xE_D: 877.7007, xW_D: 877.7007", 'parabolic, hmin: 1.e-5, hmax: 4.e-5', 2),
    ('Carreau, mu_inf: 1.e-3, lam: 1.e-5, a: 2., N: 0.6', ", xE: ['D', 'N', 'N'], xW: ['D', 'N', 'N'], xE_D: 877.7007, xW_D: 877.7007",
     'parabolic, hmin: 1.e-5, hmax: 4.e-5', 3),
    ('Eyring, tauE: 5.e5', '', 'journal, CR: 1.e-2, eps: 0.6', 3)], ids=['eyring-dirichlet-2', 'carreau-dirichlet-3', 'eyring-periodic-3'])
def test_shear_thinning_across_slabs(hiplib, tmp_path, law, bc, geo, world):
    """Shear thinning (stress.py:170-192, 314-326; viscosity.py:110-141): the viscosity needs np.gradient(p), so a slab's
    halo row needs the neighbour's pressure one row further in -- a two-row halo, carried by the stage messages.  2 and 3
    slabs against the one-handle run AND the oracle, 12 steps."""
    import torch.multiprocessing as mp
    from gapflow_amd import Problem
    from oracle.problem import OracleProblem
    text = THINNING_SLAB.format(law=law, bc=bc, geo=geo)
    nsteps = 12
    mp.spawn(_slab_worker, args=(world, _free_port(), text, nsteps, str(tmp_path)), nprocs=world, join=True)
    serial, cpu = Problem.from_string(text), OracleProblem.from_string(text)
    serial._pre_run()
    cpu._pre_run()
    for _ in range(nsteps):
        serial.update()
        cpu.update()
    for r in range(world):
        z = np.load(tmp_path / f'rank{r}.npz')
        lo, hi = int(z['lo']), int(z['hi'])
        assert int(z['step']) == nsteps and int(z['invalid']) == 0
        for c in range(3):
            scale = np.abs(serial.q[c]).max() or 1.
            assert np.abs(z['q'][c] - serial.q[c, lo - 1:hi + 2]).max() <= 1e-12 * scale, f'rank {r} comp {c} vs serial'
            assert np.abs(z['q'][c] - cpu.q[c, lo - 1:hi + 2]).max() <= 1e-9 * scale, f'rank {r} comp {c} vs oracle'
        np.testing.assert_allclose(z['dt'], serial.dt, rtol=1e-12)
