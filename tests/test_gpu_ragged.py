"""Ragged shapes: grid sizes around the kernel's tiling constants (62 output columns per wavefront, row chunks of
>= 8 rows, 4 strips per workgroup), tiny grids, all boundary-rule combinations and both sweep orders -- six steps each
against the oracle."""
import io

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SIM = """
options: {{silent: True}}
grid: {{Nx: {nx}, Ny: {ny}, dx: 2.e-5, dy: 2.5e-5{bc}}}
geometry: {{type: {geo}, {geopar}, U: 0.2, V: {v}}}
numerics: {{CFL: 0.4, adaptive: {adaptive}, dt: 2.e-10, MC_order: {mc}, max_it: 100}}
properties: {{EOS: DH, shear: 0.0794, bulk: 0.02, rho0: 877.7007, C1: 3.5e9}}
"""
BC = {
    'pp': "",
    'dp': ", xE: ['D', 'N', 'N'], xW: ['D', 'N', 'N'], xE_D: 877.7007, xW_D: 876.",
    'pd': ", yS: ['D', 'N', 'N'], yN: ['D', 'N', 'N'], yS_D: 877.7007, yN_D: 878.",
    'dd': ", xE: ['D', 'N', 'N'], xW: ['D', 'N', 'N'], xE_D: 877.7007, xW_D: 876., yS: ['D', 'N', 'N'], yN: ['D', 'N', 'N'], yS_D: 877., yN_D: 878.5",
}
GEO = {'journal': "CR: 1.e-2, eps: 0.6", 'asperity': "hmin: 2.e-6, hmax: 1.e-5, num: 1", 'inclined': "hmax: 8.e-6, hmin: 3.e-6"}

CASES = [
    # nx, ny, bc, geo, mc, adaptive, v
    (1, 1, 'pp', 'inclined', 1, 1, 0.0),
    (2, 3, 'dd', 'inclined', -1, 1, 0.1),
    (3, 1, 'dp', 'inclined', 0, 0, 0.0),
    (1, 70, 'pd', 'inclined', 0, 1, 0.1),
    (7, 61, 'pp', 'journal', 1, 1, 0.05),
    (9, 62, 'pd', 'journal', -1, 1, 0.05),
    (17, 63, 'dp', 'inclined', 0, 1, 0.05),
    (33, 124, 'pp', 'asperity', 0, 1, 0.1),
    (16, 125, 'dd', 'asperity', 1, 0, 0.1),
    (130, 7, 'pp', 'journal', -1, 1, 0.0),
    (131, 249, 'dp', 'asperity', 0, 1, 0.1),
    (64, 310, 'pd', 'journal', 1, 1, 0.05),
]


@pytest.mark.parametrize('nx,ny,bc,geo,mc,adaptive,v', CASES)
def test_ragged_shapes_match_oracle(hiplib, nx, ny, bc, geo, mc, adaptive, v):
    from gapflow_amd import Problem
    from oracle.problem import OracleProblem
    text = SIM.format(nx=nx, ny=ny, bc=BC[bc], geo=geo, geopar=GEO[geo], mc=mc, adaptive=adaptive, v=v)
    gpu, cpu = Problem.from_string(text), OracleProblem.from_string(text)
    np.testing.assert_array_equal(gpu.topo.full, cpu.topo)
    gpu._pre_run()
    cpu._pre_run()
    np.testing.assert_allclose(gpu.dt, cpu.dt, rtol=1e-12)
    for _ in range(6):
        gpu.update()
        cpu.update()
    assert gpu.step == cpu.step == 6
    for c in range(3):
        scale = np.abs(cpu.q[c]).max() or 1.
        assert np.abs(gpu.q[c] - cpu.q[c]).max() <= 1e-9 * scale, f'component {c}'     # ghost cells and corners included
    np.testing.assert_allclose(gpu.dt, cpu.dt, rtol=1e-9)
    np.testing.assert_allclose(gpu.kinetic_energy, cpu.kinetic_energy, rtol=1e-9)
    np.testing.assert_allclose(gpu.mass, cpu.mass, rtol=1e-12)
