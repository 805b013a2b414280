"""The two 8-rank workloads of BASELINE.json as EIGHT ranks, on the one GPU a test box has:

  configs[2] on 8 slabs  2-D journal bearing 4096 x 4096, Dowson-Higginson, all-periodic (the strong-scaling run of bench.py)
  configs[4]             2-D journal bearing 8192 x 8192 with the three 512-point surrogates on 8 slabs

Eight ranks = eight threads of this process (gapflow_amd.slab.ThreadWorld): the boxes allow six processes on a card, so one
process per rank stops at five.  Every rank is a real SlabProblem -- own library handle, its 1/8 of the rows, halo kinds
(seam, neighbour) / (neighbour, neighbour) / (neighbour, seam), seam topography between rank 7 and rank 0, eight records
reduced in rank order, three exchanges per surrogate step with eight contributors; the collective itself is a device copy
(what RCCL does on 8 GPUs is torch.distributed's stock path: tests/test_slab_gloo.py, test_two_gpus_over_rccl_match_serial).
Checked against the undivided handle on the same GPU -- bitwise for the fixed-form closures -- and, with surrogates, against
the oracle on sampled cells of ranks 0, 3 and 7.  Reference: none (README.md:60-61, single process)."""
import io

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

WORLD = 8

JOURNAL_4096 = """
options: {silent: True, write_freq: 1000000}
grid: {dx: 1.e-5, dy: 1.e-5, Nx: 4096, Ny: 4096, xE: ['P', 'P', 'P'], xW: ['P', 'P', 'P'], yS: ['P', 'P', 'P'], yN: ['P', 'P', 'P']}
geometry: {type: journal, CR: 1.e-2, eps: 0.7, U: 0.1, V: 0.}
numerics: {CFL: 0.5, adaptive: 1, tol: 1e-12, dt: 1e-10, max_it: 100000000}
properties: {shear: 0.0794, bulk: 0., EOS: DH, P0: 101325., rho0: 877.7007, C1: 3.5e10, C2: 1.23}
"""


@pytest.mark.parametrize('mc_order', [1, 0], ids=['bench-workload', 'alternating-sweeps'])
def test_cfg2_on_eight_slabs_is_bitwise_the_undivided_run(hiplib, mc_order):
    """bench.py's workload (MC_order 1, its default) cut into 8 x-slabs of 512 rows, 10 steps, and the same with the sweep
    order alternating from step to step (MC_order 0: both predictor directions of the kernel): every rank's owned
    rows AND its two outer rows are bit for bit the undivided handle's, dt is the same number on every rank; the kinetic
    energy is summed over another partition (1e-12)."""
    import torch
    from gapflow_amd import Problem
    from gapflow_amd.slab import SlabProblem, ThreadWorld, HALO_SEAM, HALO_NEIGHBOUR
    nsteps = 10
    text = JOURNAL_4096.replace('max_it: 100000000', f'max_it: 100000000, MC_order: {mc_order}')

    def rank_body(group):
        slab = SlabProblem.from_string(text, device=0, dist=group)
        slab.pre_run()
        slab.advance(nsteps)
        st = slab.state()
        return slab.layout, slab.local_q(), st

    runs = ThreadWorld(WORLD, torch).run(rank_body)
    serial = Problem.from_string(text)
    serial._pre_run()
    serial._advance(nsteps, honor_stop=False)
    ref = serial.q
    kinds = [(L.kind_lo, L.kind_hi) for L, _, _ in runs]
    assert kinds == [(HALO_SEAM, HALO_NEIGHBOUR)] + [(HALO_NEIGHBOUR, HALO_NEIGHBOUR)] * 6 + [(HALO_NEIGHBOUR, HALO_SEAM)]
    covered = 0
    for L, q, st in runs:
        assert (L.lo, L.hi) == (1 + 512 * L.rank, 512 * (L.rank + 1))
        assert int(st.step) == nsteps and int(st.invalid) == 0
        assert st.dt == serial.dt, f'rank {L.rank}: dt {st.dt!r} vs {serial.dt!r}'
        same = np.array_equal(q, ref[:, L.lo - 1:L.hi + 2])
        assert same, f'rank {L.rank}: max |difference| {np.abs(q - ref[:, L.lo - 1:L.hi + 2]).max():.3e}'
        np.testing.assert_allclose(st.ekin, serial.kinetic_energy, rtol=1e-12)
        np.testing.assert_allclose(st.residual, serial.residual, rtol=1e-6, atol=1e-12)
        covered += L.nx
    assert covered == 4096
    print(f"\n[8 slabs, 4096^2, {nsteps} steps] fields bitwise; Ekin rel. difference "
          f"{abs(runs[0][2].ekin - serial.kinetic_energy) / serial.kinetic_energy:.2e}")


def test_cfg4_eight_slabs_with_surrogates_match_the_undivided_run_and_the_oracle(hiplib):
    """BASELINE.json configs[4]: 8192 x 8192 journal bearing, pressure + wall-shear surrogates (512 training points, SURVEY
    8(d)'s cfg4 recipe, hyper-parameters fixed), two stage-wise steps on 8 slabs of 1024 rows against the undivided 8192^2
    handle (1e-11 of scale, outer rows included; dt 1e-11), then -- on a state that varies from cell to cell -- the
    surrogates' means on sampled cells of ranks 0, 3 and 7 against the oracle (1e-9 of scale)."""
    import torch
    from gapflow_amd import _lib
    from gapflow_amd.io import read_yaml_input
    from gapflow_amd.slab import SlabProblem, ThreadWorld
    from test_gpu_gp_large import JOURNAL_SLAB, N_TRAIN, make_problem, oracle_models, perturb, sample_cells, features_at, quiet
    N = 8192
    sim = JOURNAL_SLAB.format(nx=N, ny=N)
    serial, d, (X, Y, Ye) = make_problem(sim)
    om = oracle_models(serial, X, Y, Ye)
    thetas = {name: np.array(m.theta) for name, m in serial._gp_models.items()}
    dt0 = serial.dt
    serial.update()
    serial.update()
    ref, ref_dt, ref_ekin = serial.q, serial.dt, serial.kinetic_energy
    assert serial.step == 2
    del serial                                          # its 18 GB of device planes go before the slabs allocate theirs

    inputs = [quiet(read_yaml_input, io.StringIO(sim)) for _ in range(WORLD)]

    def rank_body(group):
        rank = group.get_rank()
        sp = SlabProblem(inputs[rank], device=0, dist=group)      # (no stdout redirection inside threads: it is process-wide)
        sp.database.set_arrays(X, Y, Ye)
        for m in sp._gp_models.values():
            m.optimise = False
        sp.pre_run()
        for name, m in sp._gp_models.items():
            np.testing.assert_array_equal(m.theta, thetas[name])
        dt_start = sp.state().dt
        sp.advance(2)
        st = sp.state()
        q = sp.local_q()
        errs = None
        if rank in (0, 3, 7):
            # the surrogates of this slab on a state that varies from cell to cell, sampled cells against the oracle
            qp = perturb(q)
            sp._upload(_lib.FIELD_Q, qp)
            _lib.check(sp.lib.gpf_update_closures(sp._h))
            cells, _ = sample_cells(qp[0].size, N_TRAIN, nrandom=1500, seed=11 + rank)
            F = features_at(qp, sp._topo_local, cells)
            p = sp._download(_lib.FIELD_PRESSURE, 1)[0].reshape(-1)[cells]
            lower, upper = sp._download(_lib.FIELD_WALL_LOWER, 6), sp._download(_lib.FIELD_WALL_UPPER, 6)
            errs = {}
            mp_ = om['press']
            mean_p = mp_.fit.mean((F / mp_.X_scale)[:, mp_.dims])[:, 0] * mp_.Yscale
            errs['press'] = np.abs(p - mean_p).max() / np.abs(mean_p).max()
            for kind, k in (('shear_x', 4), ('shear_y', 3)):
                m = om[kind]
                mean = m.fit.mean((F / m.X_scale)[:, m.dims]) * m.Yscale
                scale = np.abs(mean).max()
                errs[kind] = max(np.abs(lower[k].reshape(-1)[cells] - mean[:, 0]).max(), np.abs(upper[k].reshape(-1)[cells] - mean[:, 1]).max()) / scale
        return sp.layout, q, st, dt_start, errs

    runs = ThreadWorld(WORLD, torch).run(rank_body)
    worst = 0.0
    for L, q, st, dt_start, errs in runs:
        assert (L.lo, L.hi) == (1 + 1024 * L.rank, 1024 * (L.rank + 1))
        np.testing.assert_allclose(dt_start, dt0, rtol=1e-12)
        assert int(st.step) == 2 and int(st.invalid) == 0
        np.testing.assert_allclose(st.dt, ref_dt, rtol=1e-11)
        np.testing.assert_allclose(st.ekin, ref_ekin, rtol=1e-11)
        for c in range(3):
            s = np.abs(ref[c]).max() or 1.0
            e = np.abs(q[c] - ref[c, L.lo - 1:L.hi + 2]).max() / s
            worst = max(worst, e)
            assert e <= 1e-11, f'rank {L.rank} component {c}: {e:.3e} of scale away from the undivided run'
        if errs is not None:
            for kind, e in errs.items():
                assert e <= 1e-9, f'rank {L.rank} {kind}: {e:.3e} of scale away from the oracle'
    print(f"\n[8 slabs, 8192^2 + 3 surrogates, 2 steps] max field difference to the undivided run {worst:.2e} of scale; "
          f"oracle on sampled cells: " + ", ".join(f"rank {L.rank} {max(e.values()):.1e}" for L, _, _, _, e in runs if e))
