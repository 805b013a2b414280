"""Ragged shapes: grid sizes around the tiling constants of the step kernels -- k_step2 (csrc/step2_kernel.hip): a wavefront
owns a window of 128 columns, two per lane, and produces 126 of them (one strip: Ny <= 126; two: <= 252; the ghost column of
the last strip needs two idle lanes behind it, strip2_geom), row chunks of >= 4 rows, four waves per workgroup; problems of up
to ~1200 cells run in k_small_steps (one workgroup, csrc/small_kernel.hip) -- tiny grids, all boundary-rule combinations and
both sweep orders, six steps each against the oracle.  The last cases force long marches (GPF_CHUNKS: few row chunks of
>= 100 rows, the plan the 4096^2 benchmark runs with one wave per SIMD)."""
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
    # around one and two full strips of k_step2 (126 output columns each), both sweep orders and all edge kinds
    (12, 126, 'pp', 'asperity', 0, 1, 0.1),
    (11, 126, 'dd', 'journal', -1, 1, 0.05),
    (13, 127, 'pd', 'asperity', 0, 1, 0.1),
    (10, 127, 'dp', 'journal', 1, 1, 0.05),
    (9, 252, 'pp', 'journal', 0, 1, 0.05),
    (14, 252, 'dd', 'asperity', 1, 1, 0.1),
    (8, 253, 'dp', 'asperity', 0, 1, 0.1),
    (15, 253, 'pd', 'journal', -1, 1, 0.05),
]
# long marches: (nx, ny, bc, geo, mc, adaptive, v, GPF_CHUNKS) -- >= 100 rows per wave
LONG = [
    (420, 130, 'pp', 'journal', 0, 1, 0.05, 4),
    (405, 127, 'dd', 'asperity', 0, 1, 0.1, 3),
    (640, 64, 'dp', 'inclined', -1, 1, 0.05, 2),
]


def run_case(nx, ny, bc, geo, mc, adaptive, v, nsteps=6):
    from gapflow_amd import Problem
    from oracle.problem import OracleProblem
    text = SIM.format(nx=nx, ny=ny, bc=BC[bc], geo=geo, geopar=GEO[geo], mc=mc, adaptive=adaptive, v=v)
    gpu, cpu = Problem.from_string(text), OracleProblem.from_string(text)
    np.testing.assert_array_equal(gpu.topo.full, cpu.topo)
    gpu._pre_run()
    cpu._pre_run()
    np.testing.assert_allclose(gpu.dt, cpu.dt, rtol=1e-12)
    for _ in range(nsteps):
        gpu.update()
        cpu.update()
    assert gpu.step == cpu.step == nsteps
    errs = []
    for c in range(3):
        scale = np.abs(cpu.q[c]).max() or 1.
        errs.append(np.abs(gpu.q[c] - cpu.q[c]).max() / scale)
        assert errs[-1] <= 1e-9, f'component {c}: {errs[-1]:.3e}'       # ghost cells and corners included
    np.testing.assert_allclose(gpu.dt, cpu.dt, rtol=1e-9)
    np.testing.assert_allclose(gpu.kinetic_energy, cpu.kinetic_energy, rtol=1e-9)
    np.testing.assert_allclose(gpu.mass, cpu.mass, rtol=1e-12)
    print(f'\n[{nx} x {ny} {bc} {geo} MC_order {mc}] max error of (rho, jx, jy) after {nsteps} steps: '
          + ', '.join(f'{e:.1e}' for e in errs) + ' of scale (tolerance 1e-9)')


@pytest.mark.parametrize('nx,ny,bc,geo,mc,adaptive,v', CASES)
def test_ragged_shapes_match_oracle(hiplib, nx, ny, bc, geo, mc, adaptive, v):
    run_case(nx, ny, bc, geo, mc, adaptive, v)


@pytest.mark.parametrize('nx,ny,bc,geo,mc,adaptive,v,chunks', LONG)
def test_long_marches_match_oracle(hiplib, monkeypatch, nx, ny, bc, geo, mc, adaptive, v, chunks):
    """k_step2 with few, long row chunks (>= 100 rows per wave: the pipelined row loads run through many iterations of the
    march loop, dummy requests only at the very end), the arrangement plan_step2 picks for the 4096^2 benchmark."""
    monkeypatch.setenv('GPF_CHUNKS', str(chunks))
    run_case(nx, ny, bc, geo, mc, adaptive, v)
