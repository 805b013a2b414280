"""Rows next to the hot path: piezo-viscosity closures and the output files the reference's tools read."""
import io
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PIEZO = """
options: {{silent: True}}
grid: {{Nx: 64, Ny: {ny}, Lx: 0.05, Ly: {ly}, xE: ['D', 'N', 'N'], xW: ['D', 'N', 'N'], xE_D: {rho0}, xW_D: {rho0}}}
geometry: {{type: parabolic, hmin: 1.e-5, hmax: 4.e-5, U: 10., V: {v}}}
numerics: {{CFL: 0.4, adaptive: 1, max_it: 100}}
properties:
    EOS: {eos}
    shear: 0.05
    bulk: 0.01
    rho0: {rho0}
    piezo: {{name: {name}}}
"""


@pytest.mark.parametrize('eos,name,rho0,ny', [('DH', 'Barus', 877.7007, 1), ('DH', 'Roelands', 877.7007, 12),
                                                ('Bayada', 'Dukler', 850., 1), ('Bayada', 'McAdams', 850., 12)])
def test_piezoviscosity_steps_match_oracle(hiplib, eos, name, rho0, ny):
    """models/viscosity.py:34-66 inside the stress closures (stress.py:306-312): 20 steps against the oracle."""
    from gapflow_amd import Problem
    from oracle.problem import OracleProblem
    text = PIEZO.format(eos=eos, name=name, rho0=rho0, ny=ny, ly=0.01 if ny > 1 else 1., v=1. if ny > 1 else 0.)
    gpu, cpu = Problem.from_string(text), OracleProblem.from_string(text)
    gpu._pre_run()
    cpu._pre_run()
    for _ in range(20):
        gpu.update()
        cpu.update()
    for c in range(3):
        scale = np.abs(cpu.q[c]).max() or 1.
        assert np.abs(gpu.q[c] - cpu.q[c]).max() <= 1e-9 * scale
    np.testing.assert_allclose(gpu.dt, cpu.dt, rtol=1e-10)
    # the stress FIELD as update() leaves it: the corrector stage's closure, on the predictor's field (problem.py:531-560) -- the
    # device re-runs that predictor from the state it retained (gpf_update_closures)
    np.testing.assert_allclose(gpu.bulk_stress.stress, cpu.tau_avg, rtol=1e-9, atol=1e-9 * np.abs(cpu.tau_avg).max())


THINNING = """
options: {{silent: True}}
grid: {{Nx: 48, Ny: {ny}, Lx: 0.05, Ly: {ly}, xE: ['D', 'N', 'N'], xW: ['D', 'N', 'N'], xE_D: 877.7007, xW_D: 877.7007}}
geometry: {{type: parabolic, hmin: 1.e-5, hmax: 4.e-5, U: 10., V: {v}}}
numerics: {{CFL: 0.4, adaptive: 1, max_it: 100}}
properties:
    EOS: DH
    shear: 0.05
    bulk: 0.
    rho0: 877.7007
    thinning: {{name: {name}{extra}}}
{piezo}
"""


@pytest.mark.parametrize('name,extra,ny,piezo', [('Eyring', ', tauE: 5.e5', 1, ''), ('Carreau', ', mu_inf: 1.e-3, lam: 1.e-5, a: 2., N: 0.6', 10, ''),
                                                 ('Eyring', '', 10, '    piezo: {name: Barus}')])
def test_shear_thinning_steps_match_oracle(hiplib, name, extra, ny, piezo):
    """viscosity.py:69-141 with np.gradient(p) over the ghosted array (stress.py:170-192, 314-326): the stage-wise
    pipeline (the fused step cannot see grad p) against the oracle, 15 steps."""
    from gapflow_amd import Problem
    from oracle.problem import OracleProblem
    text = THINNING.format(name=name, extra=extra, ny=ny, ly=0.01 if ny > 1 else 1., v=1. if ny > 1 else 0., piezo=piezo)
    gpu, cpu = Problem.from_string(text), OracleProblem.from_string(text)
    gpu._pre_run()
    cpu._pre_run()
    for _ in range(15):
        gpu.update()
        cpu.update()
    assert gpu.step == 15
    for c in range(3):
        scale = np.abs(cpu.q[c]).max() or 1.
        assert np.abs(gpu.q[c] - cpu.q[c]).max() <= 1e-9 * scale
    np.testing.assert_allclose(gpu.dt, cpu.dt, rtol=1e-10)
    # (the stage-wise pipeline leaves the corrector stage's closures in the derived fields, like the reference's update())
    np.testing.assert_allclose(gpu.bulk_stress.stress, cpu.tau_avg, rtol=1e-8, atol=1e-9 * np.abs(cpu.tau_avg).max())
    # the fused entry point refuses such a problem instead of ignoring the thinning law
    from gapflow_amd import _lib
    import ctypes as C
    assert gpu._lib.gpf_step(gpu._h, 1, 0, None, 0, C.byref(C.c_int64())) == -5


def test_output_files_have_the_reference_layout(hiplib, tmp_path):
    """sol.nc / topo.nc / history.csv / config.yml as the reference's viz tools read them
    (viz/animations.py:173-182, viz/plotting.py:331-348; problem.py:385-391)."""
    from scipy.io import netcdf_file
    import yaml
    from gapflow_amd import Problem
    sim = f"""
options: {{output: {tmp_path}/run, write_freq: 10, use_tstamp: False, silent: False}}
grid: {{Nx: 50, Ny: 1, dx: 1.e-5, dy: 1., xE: ['D', 'N', 'N'], xW: ['D', 'N', 'N'], xE_D: 877.7007, xW_D: 877.7007}}
geometry: {{type: journal, CR: 1.e-2, eps: 0.7, U: 0.1, V: 0.}}
numerics: {{tol: 1e-9, dt: 1e-10, max_it: 35}}
properties: {{shear: 0.0794, bulk: 0., EOS: DH, rho0: 877.7007}}
"""
    prob = Problem.from_string(sim)
    prob.run()
    out = os.path.join(tmp_path, 'run')
    assert sorted(os.listdir(out)) == ['config.yml', 'history.csv', 'sol.nc', 'topo.nc']
    rows = open(os.path.join(out, 'history.csv')).read().strip().split('\n')
    assert rows[0] == 'step,time,ekin,residual,vsound'
    assert [int(float(r.split(',')[0])) for r in rows[1:]] == [0, 10, 20, 30, 35]      # frames at write_freq + the final one
    with netcdf_file(os.path.join(out, 'sol.nc'), 'r', mmap=False) as f:
        assert f.variables['solution'].shape == (5, 3, 1, 52, 3)
        assert f.variables['pressure'].shape == (5, 52, 3)
        assert f.variables['wall_stress_xz'].shape == (5, 12, 1, 52, 3)
        assert f.variables['wall_stress_yz'].shape == (5, 12, 1, 52, 3)
        np.testing.assert_array_equal(f.variables['solution'][-1][:, 0], prob.q)
        # halves of the shared components + own shear component (stress.py:346-362)
        np.testing.assert_allclose(f.variables['wall_stress_xz'][-1][4, 0], prob.wall_stress_xz.lower[4])
    with netcdf_file(os.path.join(out, 'topo.nc'), 'r', mmap=False) as f:
        assert f.variables['topography'].shape == (1, 4, 1, 52, 3)
        np.testing.assert_array_equal(f.variables['topography'][0][:, 0], prob.topo.full)
    cfg = yaml.safe_load(open(os.path.join(out, 'config.yml')))
    assert cfg['grid']['Nx'] == 50 and cfg['prop']['EOS'] == 'DH'


def test_small_grid_kernel_equals_launch_per_kernel_path(hiplib, tmp_path):
    """k_small_steps (one workgroup, many steps per launch) against the three-launch path on the same problem: two child
    processes, because the switch (GPF_SMALL_GRID) is read once per process."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = '''
import sys, io, contextlib, numpy as np
sys.path.insert(0, %r)
from gapflow_amd import Problem
text = """
options: {silent: True}
grid: {Nx: 33, Ny: 17, Lx: 0.01, Ly: 0.005, xE: ['D', 'N', 'N'], xW: ['D', 'N', 'N'], xE_D: 877.7007, xW_D: 876.,
       yS: ['P', 'P', 'P'], yN: ['P', 'P', 'P']}
geometry: {type: asperity, hmin: 2.e-6, hmax: 1.e-5, num: 1, U: 0.5, V: 0.1}
numerics: {CFL: 0.4, adaptive: 1, MC_order: 0, tol: 1.e-14, max_it: 100000}
properties: {EOS: DH, shear: 0.0794, bulk: 0., rho0: 877.7007, piezo: {name: Barus, aB: 2.e-8}}
"""
with contextlib.redirect_stdout(io.StringIO()):
    p = Problem.from_string(text)
    p._pre_run()
    p._advance(37, honor_stop=False)
    p._advance(40, honor_stop=False)
np.savez(sys.argv[1], q=p.q, dt=p.dt, ekin=p.kinetic_energy, residual=p.residual, step=p.step, simtime=p.simtime)
''' % root
    out = {}
    for mode in ('0', '1'):
        fn = str(tmp_path / f'm{mode}.npz')
        subprocess.run([sys.executable, '-c', code, fn], check=True, env=dict(os.environ, GPF_SMALL_GRID=mode), timeout=300)
        out[mode] = np.load(fn)
    a, b = out['0'], out['1']
    assert int(a['step']) == int(b['step']) == 77
    for c in range(3):
        # two correct evaluation orders of a stiff problem (dp/drho ~ 1e8): rounding differences grow to ~1e-12 in 77 steps
        assert np.abs(a['q'][c] - b['q'][c]).max() <= 1e-10 * np.abs(a['q'][c]).max(), c
    np.testing.assert_allclose(b['dt'], a['dt'], rtol=1e-13)
    np.testing.assert_allclose(b['simtime'], a['simtime'], rtol=1e-13)
    np.testing.assert_allclose(b['ekin'], a['ekin'], rtol=1e-12)


def test_strict_atomics_build_gives_identical_results(hiplib, tmp_path):
    """the in-launch hand-offs (per-block records -> last block) are ordered by write-through stores, a drained
    arrival add and sc1 loads, not by the HIP memory model.  `__graft_entry__.build()` (gapflow_amd.build.build_strict_variant)
    also builds the same kernels with release / acquire orders (-DGPF_STRICT_ATOMICS; Dowson-Higginson instantiations only, which
    is what these fixtures use); both libraries must produce the same fields, bit for bit, on a fused run (k_step2's in-kernel
    commit) -- run in two child processes because the library path is read at import."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from gapflow_amd.build import STRICT_LIB as strict, build_strict_variant, is_stale
    if is_stale(strict):        # older than a source (or missing): its ABI may no longer match -- 30 s with hipcc
        if not os.path.exists('/opt/rocm/bin/hipcc'):
            pytest.skip('gapflow_amd/lib/variants/strict.so is stale and there is no hipcc to rebuild it')
        build_strict_variant()
    code = f"""
import sys, numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, {os.path.join(root, 'tests')!r})
from test_gpu_parity import make_problem
out = {{}}
for name in ('slider2d_dn', 'seam2d_asperity', 'journal2d_periodic50_u10'):
    prob, fx, meta = make_problem(name)
    prob._advance(40, honor_stop=False)
    out[name] = prob.q.copy(); out[name + '_dt'] = prob.dt; out[name + '_res'] = prob.residual
np.savez(sys.argv[1], **out)
"""
    outs = []
    for tag, lib in (('fast', None), ('strict', strict)):
        env = dict(os.environ)
        if lib:
            env['GPF_LIB_PATH'] = lib
        res = subprocess.run([sys.executable, '-c', code, str(tmp_path / f'{tag}.npz')], env=env, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-2000:]
        outs.append(np.load(tmp_path / f'{tag}.npz'))
    for k in outs[0].files:
        assert np.array_equal(outs[0][k], outs[1][k]), k


def test_large_fields_come_back_whole_after_other_handles_are_destroyed(hiplib):
    """Large fields are ranges of virtual addresses backed by shuffled 16-MiB pieces of device memory (csrc/api_fields.inc: field_malloc),
    and the placement search allocates and returns a dozen of them per handle.  Handles are created, stepped and destroyed in
    turn: every one of a set of identical problems must produce the same bits -- a range returned with a live mapping inside
    would hand the next field another field's pages -- and the device memory in use must return to where it started."""
    import gc
    import torch
    from gapflow_amd import Problem
    import reference_suite as rs
    text = rs.JOURNAL_2D.format(dx='1.e-5', dy='1.e-5', n=2048)

    def run(nsteps):
        p = Problem.from_string(text)
        p._pre_run()
        p._advance(nsteps, honor_stop=False)
        return p

    first = run(6)
    ref = first.q.copy()
    assert np.isfinite(ref).all() and np.abs(ref[0] - 877.7007).max() > 0.0          # the field has moved
    del first
    gc.collect()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    keep = []
    for k in range(4):
        a, b = run(6), run(6)                   # two alive at once, their searches interleaved with each other's fields
        np.testing.assert_array_equal(a.q, ref)
        np.testing.assert_array_equal(b.q, ref)
        keep.append(a if k % 2 == 0 else b)     # some survive the others' destruction ...
        del a, b
        gc.collect()
    for p in keep:                              # ... and still hold and advance their own state
        np.testing.assert_array_equal(p.q, ref)
    del keep, p
    gc.collect()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert abs(free1 - free0) < (64 << 20), f'device memory in use changed by {(free0 - free1) / 2**20:.0f} MiB'


def test_tuner_and_scattered_fields_change_no_bit_of_the_result(hiplib, monkeypatch):
    """What a large handle decides by measurement before its first step -- waves per SIMD, cache hints on loads and stores, which
    buffers its state lives in, and whether those buffers are scattered over shuffled pieces of device memory -- must not show in
    the fields: the same problem with the tuner off and plain hipMalloc gives the same bits (the kinetic energy is summed in
    another order: last bits only)."""
    from gapflow_amd import Problem
    import reference_suite as rs
    text = rs.JOURNAL_2D.format(dx='1.e-5', dy='1.e-5', n=1536).replace('V: 0.', 'V: 0.03')

    def run():
        p = Problem.from_string(text)
        p._pre_run()
        p._advance(7, honor_stop=False)
        return p.q.copy(), p.dt, p.kinetic_energy, p._lib.gpf_plan_note(p._h).decode()

    q_tuned, dt_tuned, e_tuned, note = run()
    assert 'timed' in note and 'placement' in note, note
    monkeypatch.setenv('GPF_PLAN_TUNE', '0')
    monkeypatch.setenv('GPF_SCATTER_MB', '0')
    q_plain, dt_plain, e_plain, note_plain = run()
    assert 'rule of thumb' in note_plain, note_plain
    np.testing.assert_array_equal(q_tuned, q_plain)
    assert dt_tuned == dt_plain
    np.testing.assert_allclose(e_tuned, e_plain, rtol=1e-13)
    print(f'\n[tuner] {note[:120]} ... | fields bitwise the untuned run\'s, Ekin rel. difference {abs(e_tuned - e_plain) / e_plain:.1e}')
