#!/usr/bin/env python3
"""GP-closure timing (BASELINE.json configs[3]: 2-D slider 2048^2, 512 training points, Matern-3/2 + Cholesky).

    python tools/bench_gp.py [--n 2048] [--ntrain 512] [--steps 3]

BASELINE.json configs[4] (2-D journal 8192^2, GP closure, slab-decomposed over the GPUs of a node):

    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/bench_gp.py --journal --n 8192

Hyper-parameters stay at their initial values (timing run); training data are Mock-law samples.
Prints one JSON line: time of one full MacCormack step with all three surrogates, of one variance pass, of
the Cholesky fit, and the implied kernel-evaluation rate of the posterior-mean kernel.
"""
import argparse, contextlib, io, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

SIM = """
options: {{silent: True, write_freq: 1000000}}
grid: {{Nx: {n}, Ny: {n}, Lx: 0.1, Ly: 0.1, xE: ['D', 'N', 'N'], xW: ['D', 'N', 'N'], xE_D: 877.7007, xW_D: 877.7007,
       yS: ['P', 'P', 'P'], yN: ['P', 'P', 'P']}}
geometry: {{type: inclined, hmax: 6.6e-5, hmin: 1.e-5, U: 50., V: 0.}}
numerics: {{CFL: 0.4, adaptive: 1, tol: 1.e-12, max_it: 100000}}
properties: {{EOS: DH, shear: 0.0794, bulk: 0., rho0: 877.7007}}
gp:
    press: {{atol: 1., rtol: 0.1, obs_stddev: 1.e4, active_learning: False}}
    shear: {{atol: 1., rtol: 0.1, obs_stddev: 50., active_learning: False}}
db: {{init_size: {nt}, init_method: lhc, init_width: 0.001}}
"""


JOURNAL = """
options: {{silent: True, write_freq: 1000000}}
grid: {{Nx: {n}, Ny: {n}, dx: 1.e-5, dy: 1.e-5}}
geometry: {{type: journal, CR: 1.e-2, eps: 0.7, U: 0.1, V: 0.}}
numerics: {{CFL: 0.5, adaptive: 1, tol: 1.e-12, max_it: 100000}}
properties: {{EOS: DH, shear: 0.0794, bulk: 0., rho0: 877.7007, C1: 3.5e10}}
gp:
    press: {{atol: 1., rtol: 0.1, obs_stddev: 100., active_learning: False}}
    shear: {{atol: 1., rtol: 0.1, obs_stddev: 1., active_learning: False}}
db: {{init_size: {nt}, init_method: lhc, init_width: 0.01}}
"""


def run_slabs(a):
    """configs[4]: every rank owns an x-slab; the stage-wise step exchanges rows after each stage (gapflow_amd/slab.py)."""
    import torch
    import torch.distributed as dist
    from gapflow_amd.slab import SlabProblem
    rank, world = int(os.environ.get('RANK', '0')), int(os.environ['WORLD_SIZE'])
    local = int(os.environ.get('LOCAL_RANK', '0'))
    real_stdout = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)                           # RCCL's banner goes to stderr
    torch.cuda.set_device(local)
    dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    with contextlib.redirect_stdout(sys.stderr):
        prob = SlabProblem.from_string((JOURNAL if a.journal else SIM).format(n=a.n, nt=a.ntrain), device=local)
        for m in prob._gp_models.values():
            m.optimise = False
        prob.pre_run()
        prob.advance(1)
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        prob.advance(a.steps)
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        w = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device='cuda')
        dist.all_reduce(w, op=dist.ReduceOp.MAX)
        st = prob.state()
        assert st.step == a.steps + 1 and st.invalid == 0
    t_step = float(w.item()) / a.steps
    if rank == 0:
        cells = (a.n + 2)**2
        out = {"workload": f"2D {'journal' if a.journal else 'slider'} {a.n}x{a.n}, GP closures, {a.ntrain} training points, "
                           f"{world} x-slabs", "n_gpus": world, "ms_per_step": t_step * 1e3,
               "Mcell_updates_per_s": a.n * a.n / t_step / 1e6,
               "matern_kernel_evaluations_per_s": cells * a.ntrain * 7 / t_step}
        real_stdout.write(json.dumps(out) + '\n')
        real_stdout.flush()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=2048)
    ap.add_argument('--ntrain', type=int, default=512)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--journal', action='store_true', help='configs[4]: all-periodic journal bearing instead of the slider')
    a = ap.parse_args()
    if 'WORLD_SIZE' in os.environ:
        return run_slabs(a)
    from gapflow_amd import Problem
    with contextlib.redirect_stdout(sys.stderr):
        prob = Problem.from_string((JOURNAL if a.journal else SIM).format(n=a.n, nt=a.ntrain))
        for m in prob._gp_models.values():
            m.optimise = False
        t0 = time.perf_counter()
        prob._pre_run()                     # database (Mock laws) + three device fits
        t_init = time.perf_counter() - t0
        t0 = time.perf_counter()
        for m in prob._gp_models.values():
            m.attach()
        prob._scalars()
        t_fit = (time.perf_counter() - t0) / 3
        prob.update()                       # warm-up
        t0 = time.perf_counter()
        for _ in range(a.steps):
            prob.update()
        prob._scalars()
        t_step = (time.perf_counter() - t0) / a.steps
        t0 = time.perf_counter()
        prob._gp_models['zz'].compute_variance(on_open_step=False)
        t_var = time.perf_counter() - t0
    cells = (a.n + 2)**2
    # per step: 2 stages x (press m=1 + shear_x m=2 + shear_y m=2) mean passes + 1 sound-speed pass on the final field
    kernel_evals = cells * a.ntrain * (2 * 3 + 1)
    out = {"workload": f"2D slider {a.n}x{a.n}, GP closures (press, shear xz, shear yz), {a.ntrain} training points, Matern-3/2 ARD",
           "ms_per_step": t_step * 1e3, "Mcell_updates_per_s": a.n * a.n / t_step / 1e6,
           "matern_kernel_evaluations_per_s": kernel_evals / t_step, "ms_variance_pass_one_model": t_var * 1e3,
           "variance_trsm_TFLOPs": cells * a.ntrain**2 / t_var / 1e12, "ms_fit_one_model": t_fit * 1e3,
           "s_setup_incl_mock_database": t_init}
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
