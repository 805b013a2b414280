"""Elastic deformation of the gap on the GPU (gpf_elastic_setup / gpf_elastic_update, hipFFT) against oracle/elastic.py.

PARITY UNPINNED with respect to the reference's ContactMechanics dependency (absent here; the reference has no test or
fixture on this path): the oracle restates the published half-space responses and is pinned to analytic solutions in
tests/test_oracle_elastic.py; these tests pin the device path to that oracle."""
import io

import numpy as np
import pytest

from oracle.problem import OracleProblem

pytestmark = pytest.mark.gpu

BASE = """
options: {{silent: True}}
grid: {{{grid}}}
geometry: {{type: parabolic, hmin: 2.54e-5, hmax: 5.08e-5, U: 4.57, V: {V}}}
numerics: {{adaptive: 1, CFL: 0.45, tol: 1e-8, dt: 1.e-10, max_it: 60}}
properties:
    EOS: Bayada
    rho0: 850.
    shear: 0.039
    bulk: 0.
    cl: 1600.
    cv: 352.
    elastic: {{E: 50e09, v: 0.3, alpha_underrelax: {alpha}{images}}}
    piezo: {{name: Dukler, shearv: 3.9e-5, rhol: 850., rhov: 0.019}}
"""
DN = "xE: ['D', 'N', 'N'], xW: ['D', 'N', 'N'], xE_D: 850., xW_D: 850."
CASES = {
    # examples/config/parabolic_1d_elastic.yaml: 1-D line contact, treated as non-periodic in both directions
    'example_1d': dict(grid=f"Lx: 0.0762, Ly: 1., Nx: 100, Ny: 1, {DN}, yS: ['P', 'P', 'P'], yN: ['P', 'P', 'P']", V=0., alpha=1e-3, images=''),
    'free_2d': dict(grid=f"Lx: 0.0762, Ly: 0.04, Nx: 48, Ny: 30, {DN}, yS: ['D', 'N', 'N'], yN: ['D', 'N', 'N'], yS_D: 850., yN_D: 850.",
                    V=0.3, alpha=0.05, images=''),
    'semi_periodic_2d': dict(grid=f"Lx: 0.0762, Ly: 0.04, Nx: 48, Ny: 30, {DN}, yS: ['P', 'P', 'P'], yN: ['P', 'P', 'P']",
                             V=0.3, alpha=0.05, images=', n_images: 3'),
    'periodic_2d': dict(grid="Lx: 0.0762, Ly: 0.04, Nx: 48, Ny: 30", V=0.3, alpha=0.05, images=''),
}


@pytest.mark.parametrize('name', sorted(CASES))
def test_elastic_steps_match_oracle(hiplib, name):
    """A run with the gap deforming every step: fields, gap height, slopes and displacement after 25 steps."""
    from gapflow_amd import Problem
    text = BASE.format(**CASES[name])
    gpu = Problem.from_string(text)
    cpu = OracleProblem.from_string(text)
    gpu._pre_run()
    cpu._pre_run()
    assert gpu._elastic.periodicity == cpu.elastic.periodicity
    for _ in range(25):
        gpu.update()
        cpu.update()
    assert gpu.step == cpu.step == 25
    for c in range(3):
        scale = np.abs(cpu.q[c]).max() or 1.
        assert np.abs(gpu.q[c] - cpu.q[c]).max() <= 1e-9 * scale, c
    np.testing.assert_allclose(gpu.topo.deformation, cpu.deformation, rtol=1e-9, atol=1e-12 * np.abs(cpu.deformation).max())
    np.testing.assert_allclose(gpu.topo.h, cpu.topo[0], rtol=1e-12)
    for k in (1, 2):
        np.testing.assert_allclose(gpu.topo.full[k], cpu.topo[k], rtol=1e-9, atol=1e-9 * np.abs(cpu.topo[k]).max())
    assert np.abs(cpu.deformation).max() > 0
    np.testing.assert_allclose(gpu.dt, cpu.dt, rtol=1e-10)
    np.testing.assert_allclose(gpu.mass, cpu.mass, rtol=1e-11)


def test_elastic_example_runs_and_writes_topography_frames(hiplib, tmp_path):
    """examples/config/parabolic_1d_elastic.yaml end to end (shortened): topo.nc gets one frame per solution frame."""
    from scipy.io import netcdf_file
    from gapflow_amd import Problem
    text = BASE.format(**CASES['example_1d']).replace("options: {silent: True}",
                                                      f"options: {{output: {tmp_path / 'el'}, write_freq: 20, use_tstamp: False}}")
    with pytest.warns(UserWarning, match='semi-periodic 1D'):
        prob = Problem.from_string(text)
    prob.run()
    assert prob.step == 60 and np.isfinite(prob.q).all()
    with netcdf_file(str(tmp_path / 'el' / 'topo.nc'), mmap=False) as f, netcdf_file(str(tmp_path / 'el' / 'sol.nc'), mmap=False) as g:
        topo = f.variables['topography'][:]
        assert topo.shape[0] == g.variables['solution'][:].shape[0] + 1 == 5          # initial + frames at 0, 20, 40, 60
        assert np.abs(topo[-1, 3]).max() > 0 and np.all(topo[0, 3] == 0)
        np.testing.assert_allclose(topo[-1, 0], prob.topo.h[None], rtol=1e-14)
