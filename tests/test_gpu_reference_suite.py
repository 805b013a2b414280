"""The reference's own solver tests (see reference_suite.py), run through gapflow_amd.Problem on the GPU."""
import pytest

import reference_suite as rs

pytestmark = pytest.mark.gpu


def make(d):
    from gapflow_amd import Problem
    return Problem._from_dict(d)


def reader(f):
    from gapflow_amd.io import read_yaml_input
    return read_yaml_input(f)


@pytest.mark.parametrize('eps', [0.5, 0.7, 0.9])
def test_sommerfeld(hiplib, eps):
    rs.check_sommerfeld(make, reader, eps)


@pytest.mark.parametrize('n', [1, 2, 4, 8])
def test_shear_wave_decay(hiplib, n):
    rs.check_shear_wave_decay(make, reader, n)


@pytest.mark.parametrize('n', [1, 2, 3, 4])
def test_sound_wave_decay(hiplib, n):
    rs.check_sound_wave_decay(make, reader, n)


def test_mass_conservation(hiplib):
    rs.check_mass_conservation(make, reader)


def test_flip_axes(hiplib):
    rs.check_flip_axes(make, reader, n=100)


def test_run_matches_oracle_step_count_and_history(hiplib):
    """Problem.run(): same number of steps to convergence and same residual history as the CPU oracle."""
    import io
    import numpy as np
    from oracle.problem import OracleProblem
    from oracle.config import read_yaml_input as oracle_reader
    text = rs.JOURNAL_1D.replace('write_freq: 1000', 'write_freq: 50').replace('C1: 3.5e12', 'C1: 3.5e10').replace('tol: 1e-8', 'tol: 1e-6')
    gpu = make(reader(io.StringIO(text)))
    cpu = OracleProblem.from_dict(oracle_reader(io.StringIO(text)))
    gpu.run()
    cpu.run()
    assert gpu.step == cpu.step
    np.testing.assert_allclose(gpu.simtime, cpu.simtime, rtol=1e-10)
    np.testing.assert_allclose(gpu.kinetic_energy, cpu.kinetic_energy, rtol=1e-10)
    np.testing.assert_allclose(gpu.residual, cpu.residual, rtol=1e-5, atol=1e-9)
    for c in range(3):
        scale = np.abs(cpu.q[c]).max() or 1.
        assert np.abs(gpu.q[c] - cpu.q[c]).max() / scale < 1e-9
