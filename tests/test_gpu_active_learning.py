"""Active learning on the device path against the oracle's restatement of the reference's loop
(oracle/gp.py: OracleGP / OracleDatabase / OracleMock; GaPFlow/models/gp.py:290-335, 390-430, 435-506, db.py:278-369,
md/mock.py:81-107): WHICH cell is added, at WHICH step, by WHICH model, with what tolerance, and what the run looks like
afterwards.  PARITY UNPINNED with respect to tinygp/jax like every GP test here; the control flow is the reference's line by
line, the arithmetic the published formulas."""
import io

import numpy as np
import pytest

from oracle import gp as ogp
from oracle.config import read_yaml_input as oracle_reader
from oracle.problem import OracleProblem

pytestmark = pytest.mark.gpu

# the reference's inference test set-up (tests/test_inference.py:30-72): 1-D parabolic slider, BWR, Mock MD, active learning
PARABOLIC_1D = """
options: {silent: True, write_freq: 100}
grid: {Lx: 1470., Ly: 1., Nx: 200, Ny: 1, xE: ['D', 'N', 'N'], xW: ['D', 'N', 'N'], yS: ['P', 'P', 'P'], yN: ['P', 'P', 'P'],
       xE_D: 0.8, xW_D: 0.8}
geometry: {type: parabolic, hmin: 12., hmax: 60., U: 0.12, V: 0.}
numerics: {CFL: 0.5, adaptive: 1, tol: 1e-8, dt: 0.05, max_it: 5000}
properties: {shear: 2.15, bulk: 0., EOS: BWR, T: 1.0, rho0: 0.8}
gp:
    press: {fix_noise: True, atol: .7, rtol: 0., obs_stddev: 2.e-2, max_steps: MAXS, pause_steps: 2, active_learning: True}
    shear: {fix_noise: True, atol: .9, rtol: 0., obs_stddev: 4.e-3, max_steps: MAXS, pause_steps: 2, active_learning: True}
db: {init_size: 3, init_method: rand, init_width: 0.01}
"""
INCLINED_1D = PARABOLIC_1D.replace('type: parabolic', 'type: inclined')
# write_freq 3: the variance is also recomputed one step before an output step (problem.py:530) by the model without
# active learning; pause_steps 1: the pressure model runs out of additions, pauses, and resumes within the run
SLIDER_2D = """
options: {silent: True, write_freq: 3}
grid: {Nx: 40, Ny: 24, Lx: 0.05, Ly: 0.03, xE: ['D', 'N', 'N'], xW: ['D', 'N', 'N'], xE_D: 877.7007, xW_D: 877.7007,
       yS: ['P', 'P', 'P'], yN: ['P', 'P', 'P']}
geometry: {type: inclined, hmax: 6.6e-5, hmin: 2.e-5, U: 20., V: 3.}
numerics: {CFL: 0.3, adaptive: 1, tol: 1.e-10, max_it: 100}
properties: {EOS: DH, shear: 0.0794, bulk: 0., rho0: 877.7007, C1: 3.5e9}
gp:
    press: {atol: 1., rtol: 0.002, obs_stddev: 1.e5, active_learning: True, max_steps: MAXS, pause_steps: 1}
    shear: {atol: 1., rtol: 0.1, obs_stddev: 500., active_learning: False}
db: {init_size: 24, init_method: lhc, init_width: 0.001}
"""
NAMES = {'press': 'zz', 'shear_x': 'xz', 'shear_y': 'yz'}


def run_pair(sim, nsteps, optimise):
    """The same input through gapflow_amd.Problem (device) and the oracle, step by step."""
    from gapflow_amd import Problem
    gpu = Problem.from_string(sim)
    for m in gpu._gp_models.values():
        m.optimise = optimise
    d = oracle_reader(io.StringIO(sim))
    cpu = OracleProblem.from_dict(d)
    odb, om = ogp.attach(cpu, d, optimise=optimise)
    gpu._pre_run()
    cpu._pre_run()
    per_step = []
    for _ in range(nsteps):
        gpu.update()
        cpu.update()
        per_step.append((gpu.database.size, odb.size))
    return gpu, cpu, odb, om, per_step


def decisions(events):
    return [(e[0], e[1]) + ((e[2], e[3]) if e[0] == 'train' else ()) for e in events]


def compare(gpu, cpu, odb, om, per_step, field_tol, row_tol, theta_tol, mirror=False, same_cells=True):
    """mirror: the gap is symmetric about the middle of the domain (parabolic slider) and the models do not see dh/dx, so a cell
    and its mirror image have the same variance up to the rounding of h -- which of the two is the arg max is decided in the last
    bit, differently by any two implementations, and changes nothing downstream: rows are then compared up to the sign of dh/dx and
    the outputs on the columns the models read (pressure, wall shear xz)."""
    report = []
    # the database grew by the same points at the same steps
    assert [a for a, _ in per_step] == [b for _, b in per_step], f'database sizes per step {per_step}'
    assert gpu.database.size == odb.size
    fold = (lambda X: np.column_stack([X[:, :4], np.abs(X[:, 4]), X[:, 5:]])) if mirror else (lambda X: X)
    Xg, Xo = fold(gpu.database._Xtrain), fold(odb._Xtrain)
    scale = np.abs(Xo).max(axis=0) + 1e-300
    if same_cells:
        assert np.abs((Xg - Xo) / scale).max() <= row_tol, f'database rows differ by {np.abs((Xg - Xo) / scale).max():.3e} of scale'
        ycols = [0, 5, 11] if mirror else list(range(13))
        np.testing.assert_allclose(gpu.database._Ytrain[:, ycols], odb._Ytrain[:, ycols], rtol=max(row_tol, 1e-9), atol=1e-9 * np.abs(odb._Ytrain).max())
    else:
        report.append(f'{int((np.abs((Xg - Xo) / scale).max(axis=1) > 1e-9).sum())} of {len(Xo)} database rows are other cells')
    for kind, m in om.items():
        g = gpu._gp_models[NAMES[kind]]
        # same decisions in the same order: (train, step, reason, size) / (add, step)
        assert decisions(g.events) == decisions(m.events), f'{kind}: {decisions(g.events)} vs {decisions(m.events)}'
        adds_g = [e[2] for e in g.events if e[0] == 'add']
        adds_o = [e[3] for e in m.events if e[0] == 'add']
        for a, b in zip(adds_g, adds_o):
            assert not same_cells or np.abs((fold(a[None]) - fold(b[None])) / scale).max() <= row_tol, f'{kind}: a different cell was added: {a} vs {b}'
        assert g._pause == m._pause and g._step == m._step and g.last_fit_train_size == m.last_fit_train_size
        np.testing.assert_allclose(g.variance_tol, m.variance_tol, rtol=1e-12)
        ev = np.abs(g.variance - m.variance).max() / (np.exp(m.theta[0]) * m.Yscale**2)
        if same_cells:
            np.testing.assert_allclose(g.maximum_variance, m.maximum_variance, rtol=0, atol=max(field_tol, 1e-9) * np.exp(m.theta[0]) * m.Yscale**2)
            np.testing.assert_allclose(g.theta, m.theta, rtol=0, atol=theta_tol)
            # the stored variance field, possibly of an earlier state (gp.py:406-414)
            assert ev <= max(field_tol, 1e-9), f'{kind}: stored variance field {ev:.3e} of the prior variance away'
        report.append(f'{kind}: {len(adds_o)} added, var field {ev:.1e}')
    for c in range(3):
        s = max(np.abs(cpu.q[1]).max(), np.abs(cpu.q[2]).max()) if c else np.abs(cpu.q[0]).max()
        e = np.abs(gpu.q[c] - cpu.q[c]).max() / s
        assert e <= field_tol, f'component {c}: {e:.3e} of scale after the run'
        report.append(f'q[{c}] {e:.1e}')
    np.testing.assert_allclose(gpu.dt, cpu.dt, rtol=max(10 * field_tol, 1e-9))
    assert gpu.step == cpu.step
    return ', '.join(report)


@pytest.mark.parametrize('name,sim,nsteps', [('inclined-1d', INCLINED_1D.replace('MAXS', '4'), 8),
                                             ('parabolic-1d', PARABOLIC_1D.replace('MAXS', '4'), 8),
                                             ('slider-2d', SLIDER_2D.replace('MAXS', '3'), 7)], ids=['inclined-1d', 'parabolic-1d', 'slider-2d'])
def test_same_cells_at_the_same_steps_with_fixed_hyperparameters(hiplib, name, sim, nsteps):
    """Hyper-parameters kept at params_init (the optimiser out of the comparison): every decision of the loop -- retrain because the
    database grew through another model, variance criterion, argmax with first-hit tie-break over all cells incl. ghosts,
    max_steps, pause and resume, stale variance between evaluations, variance one step before an output step -- must be the
    oracle's, the database rows equal to 1e-12 and the fields after the run to 1e-9."""
    gpu, cpu, odb, om, per_step = run_pair(sim, nsteps, optimise=False)
    assert odb.size > odb._db['init_size'], 'the set-up must make the loop add points'
    assert any(m._pause >= 0 for m in om.values()) or name != 'slider-2d', 'the 2-D case is meant to run into a pause'
    print(f'\n[active learning, {name}, fixed hyper-parameters] database {odb._db["init_size"]} -> {odb.size}; '
          + compare(gpu, cpu, odb, om, per_step, field_tol=1e-9, row_tol=1e-12, theta_tol=0.0, mirror=name == 'parabolic-1d'))


def test_same_cells_at_the_same_steps_with_trained_hyperparameters(hiplib):
    """The same with the marginal-likelihood fit in the loop (device objective + host BFGS against the oracle's SciPy BFGS on its own
    objective): the two optimisers stop within 1e-4 of each other (tests/test_host_gp.py), so the fields agree to what that leaves
    (1e-5) -- but the DECISIONS must still be the same: same cells, same steps."""
    sim = INCLINED_1D.replace('MAXS', '4')
    gpu, cpu, odb, om, per_step = run_pair(sim, 5, optimise=True)
    print(f'\n[active learning, inclined-1d, trained hyper-parameters] database {odb._db["init_size"]} -> {odb.size}; '
          + compare(gpu, cpu, odb, om, per_step, field_tol=1e-5, row_tol=1e-12, theta_tol=2e-3))


def test_trained_hyperparameters_in_two_dimensions_same_additions_per_step(hiplib):
    """2-D slider with the fit in the loop.  The likelihood of this training set has flat directions (the length scale of the gap
    height ends up at e^17..e^20 wherever rounding takes BFGS), the variance field of two such fits differs in the 4th digit,
    and the arg max over 1092 cells -- most of them near-ties along the periodic direction -- need not be the same cell; what any
    two correct implementations share is the control flow: the same NUMBER of additions at the same steps, the same pauses, and
    solutions that agree as far as two surrogates trained on different -- equally justified -- points do: to their own accuracy
    (the criterion asks for 1e-2 of scale; observed 1e-3)."""
    sim = SLIDER_2D.replace('MAXS', '3').replace('atol: 1., rtol: 0.002', 'atol: 0.3, rtol: 0.')
    gpu, cpu, odb, om, per_step = run_pair(sim, 5, optimise=True)
    assert odb.size > odb._db['init_size']
    print(f'\n[active learning, slider-2d, trained hyper-parameters] database {odb._db["init_size"]} -> {odb.size}; '
          + compare(gpu, cpu, odb, om, per_step, field_tol=1e-2, row_tol=1e-12, theta_tol=np.inf, same_cells=False))
