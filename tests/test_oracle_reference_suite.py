"""The reference's own solver tests, run against the oracle (CPU).

These are the analytic / invariant pins the reference holds for this path:
  tests/test_sommerfeld.py:116-141        steady journal-bearing pressure vs Sommerfeld, < 2 %
  tests/test_wave_decay.py:86-146         shear / sound wave decay vs analytic, per step
  tests/test_mass_conservation.py:67-77   mass after 50 steps of a periodic 2-D run
  tests/test_flip_axes.py:68-97           x-problem == transposed y-problem
The same functions are re-used with the GPU problem class in test_gpu_reference_suite.py.
"""
import numpy as np
import pytest

import reference_suite as rs
from oracle.problem import OracleProblem
from oracle.config import read_yaml_input


def make(d):
    return OracleProblem.from_dict(d)


@pytest.mark.parametrize('eps', [0.5, 0.7, 0.9])
def test_sommerfeld(eps):
    rs.check_sommerfeld(make, read_yaml_input, eps)


@pytest.mark.parametrize('n', [1, 2, 4, 8])
def test_shear_wave_decay(n):
    rs.check_shear_wave_decay(make, read_yaml_input, n)


@pytest.mark.parametrize('n', [1, 2, 3, 4])
def test_sound_wave_decay(n):
    rs.check_sound_wave_decay(make, read_yaml_input, n)


def test_mass_conservation():
    rs.check_mass_conservation(make, read_yaml_input)


def test_flip_axes():
    rs.check_flip_axes(make, read_yaml_input, n=60)
