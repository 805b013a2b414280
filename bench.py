#!/usr/bin/env python3
"""Headline benchmark: fused MacCormack step on the 4096 x 4096 fp64 journal-bearing grid.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one full MacCormack time step (both stages, averaging, ghost cells, per-step
scalars and the device-side dt/residual update) of BASELINE.json configs[2]: 2-D journal bearing,
4096 x 4096, Dowson-Higginson EOS, all-periodic, adaptive CFL 0.5.  The field is
resident in HBM before the timed region; `value` = Nx*Ny*K / t in Mcell-updates/s over all ranks.

For N > 1 one rank runs per GPU -- under the driver's torch.distributed.run, or started by this file itself as a child
process when it is called as plain `python bench.py --gpus N`; the grid is cut into N x-slabs (strong scaling: the 4096^2
problem is fixed) with one all-gather per step over RCCL (2 halo rows + a 64-byte record per rank).

Extra objects on the JSON line:
  roofline     -- dominant kernel (k_step2, the fused step in one launch): bytes COMPULSORY for this workload per launch
                  (48 B x cells: the journal gap varies along x only, so the topography is one triple per row;
                  `frac_survey_8d` prices the same launch at SURVEY.md 8(d)'s 72 B) / its mean launch duration from
                  HIP events on the launch stream, against 8 TB/s HBM3E.
  cpu_baseline -- the NumPy oracle (oracle/, a restatement of the reference's CPU path) timed on
                  this host on a bounded sample of the same workload.
"""
import argparse
import contextlib
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_GRID = int(os.environ.get('GPF_BENCH_N', 4096))
BYTES_PER_CELL = 72.0           # read q (24) + read h, dh/dx, dh/dy (24) + write q (24), SURVEY.md 8d
BYTES_PER_CELL_LINE = 48.0      # x-only gap: the topography is one triple per row, only q moves
HBM_PEAK_GBS = 8000.0           # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md

WORKLOAD_YAML = """
options:
    output: data/bench
    write_freq: 1000000
    silent: True
grid:
    dx: 1.e-5
    dy: 1.e-5
    Nx: {N}
    Ny: {N}
    xE: ['P', 'P', 'P']
    xW: ['P', 'P', 'P']
    yS: ['P', 'P', 'P']
    yN: ['P', 'P', 'P']
geometry:
    type: journal
    CR: 1.e-2
    eps: 0.7
    U: 0.1
    V: 0.
numerics:
    CFL: 0.5
    adaptive: 1
    tol: 1e-12
    dt: 1e-10
    max_it: 100000000
properties:
    shear: 0.0794
    bulk: 0.
    EOS: DH
    P0: 101325.
    rho0: 877.7007
    C1: 3.5e10
    C2: 1.23
"""


def measured_traffic(kind):
    """HBM bytes per step-kernel launch from the PMC passes committed under profiles/ (rocprofv3 cannot run inside
    this process): 2 x FETCH_SIZE + WRITE_SIZE per the gfx950 note of MI355X_MICROARCH.md.  `kind` names the topography
    path the pass was taken on ('line': x-only gap read as a per-row profile, 'planes': three topography planes).
    None if absent or if it was taken on another grid size."""
    best = None
    pdir = os.path.join(ROOT, 'profiles')
    if N_GRID != 4096 or not os.path.isdir(pdir):
        return None, None
    # the newest set: highest round, and within a round the one named *_final (the others are earlier stages of that round)
    for d in sorted(os.listdir(pdir), key=lambda d: (d[:3], 'final' in d, d)):
        f = os.path.join(pdir, d, f'traffic_{kind}.json')
        if os.path.exists(f):
            best = (json.load(open(f))['hbm_bytes_per_launch'], f'profiles/{d}/traffic_{kind}.json')
    return best if best else (None, None)


def cpu_baseline(sample_n=2048, steps=10):
    """The oracle on a bounded sample: same YAML with Nx=Ny=sample_n, `steps` timed steps after one warm-up."""
    from oracle.problem import OracleProblem
    with contextlib.redirect_stdout(sys.stderr):
        p = OracleProblem.from_string(WORKLOAD_YAML.format(N=sample_n))
        p._pre_run()
        p.update()
        t0 = time.perf_counter()
        for _ in range(steps):
            p.update()
        dt = time.perf_counter() - t0
    return {"value": sample_n * sample_n * steps / dt / 1e6, "unit": "Mcell-updates/s", "cores": 1, "kind": "port",
            "sample": f"oracle/ (NumPy restatement of the reference CPU path, single-threaded ufuncs; host has "
                      f"{os.cpu_count()} cores), same YAML at {sample_n}x{sample_n}, {steps} steps, {dt:.1f} s"}


REPEATS = int(os.environ.get('GPF_BENCH_REPEATS', 5))
PLAN_NOTES = []            # gpf_plan_note of every problem time_problem has run, in order


def time_problem(text, steps, warmup):
    """(wall seconds of `steps` steps: the median of REPEATS timed regions, k_step ms per launch, all kernels ms per step,
    the walls of all repeats) for one YAML input."""
    import ctypes as C
    import statistics
    from gapflow_amd import Problem, _lib
    with contextlib.redirect_stdout(sys.stderr):
        prob = Problem.from_string(text)
        prob._pre_run()
        lib = prob._lib
        if warmup > 0:
            prob._advance(warmup, honor_stop=False)
        # --- timed region: EXACTLY K steps enqueued back to back, one host sync at the end; repeated REPEATS times on the running
        #     problem (the boxes of the pool and the clocks within a call vary: one region of a few ms is one sample) ---
        nexec = C.c_int64(0)
        walls = []
        for rep in range(max(1, REPEATS)):
            _lib.check(lib.gpf_scalars(prob._h, C.byref(_lib.GpfScalars())))      # drains the stream
            t0 = time.perf_counter()
            done = 0
            while done < steps:
                n = min(4096, steps - done)
                _lib.check(lib.gpf_step(prob._h, n, 0, None, 0, C.byref(nexec)))    # returns after a stream sync
                done += n
            walls.append(time.perf_counter() - t0)
            assert int(nexec.value) == warmup + (rep + 1) * steps, "steps were skipped inside the timed region"
        t0, t1 = 0.0, statistics.median(walls)
        # --- kernel time of the dominant kernel, HIP events on the launch stream ---
        kt, tt = C.c_double(0), C.c_double(0)
        nk = min(max(steps, 1), 200)
        _lib.check(lib.gpf_step_timed(prob._h, nk, C.byref(kt), C.byref(tt)))
        sc = prob._scalars()
        assert sc.invalid == 0 and sc.ekin == sc.ekin, "state went invalid during the benchmark"
        PLAN_NOTES.append(lib.gpf_plan_note(prob._h).decode())
        del prob
    return t1 - t0, kt.value / nk, tt.value / nk, walls


def stream_ceiling(planes_in, planes_out, cells):
    """GB/s this device streams for the step's byte count with no stencil at all (gpf_stream_probe): the data-sheet 8 TB/s
    is not attainable, and boxes of one pool differ by ~20 % in what they do attain."""
    import ctypes as C
    from gapflow_amd import _lib
    ms = C.c_double(0)
    _lib.check(_lib.require_device().gpf_stream_probe(0, planes_in, planes_out, cells, 20, C.byref(ms)))
    return (planes_in + planes_out) * 8.0 * cells / (ms.value / 1e3) / 1e9


def roofline(kernel_ms, cells, bytes_per_cell, traffic_kind, kernel):
    """HBM roofline of the step kernel: bytes COMPULSORY for this workload per launch / mean launch duration."""
    alg = bytes_per_cell * cells
    achieved = alg / (kernel_ms / 1e3) / 1e9
    traffic, src = measured_traffic(traffic_kind)
    planes_in = 6 if traffic_kind == 'planes' else 3
    stream = stream_ceiling(planes_in, 3, cells)
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": src, "kernel": kernel, "kernel_ms": kernel_ms,
            "algorithmic_bytes_per_cell": bytes_per_cell, "algorithmic_bytes_per_launch": alg,
            "stream_ceiling_GBps": stream, "frac_of_stream_ceiling": achieved / stream,
            "stream_ceiling_note": f"elementwise {planes_in}-in / 3-out fp64 kernel of the same byte count on this device, "
                                   "timed in this run (gpf_stream_probe)"}


def run_single(args):
    wall, kernel_ms, all_ms, walls = time_problem(WORKLOAD_YAML.format(N=N_GRID), args.steps, args.warmup)
    cells = N_GRID * N_GRID

    def spread(ws):
        ms = sorted(w / args.steps * 1e3 for w in ws)
        return {"repeats": len(ms), "ms_per_step_min": ms[0], "ms_per_step_median": ms[len(ms) // 2], "ms_per_step_max": ms[-1],
                "note": f"{len(ms)} timed regions of exactly {args.steps} steps each on the running problem; value / ms_per_step are the median's"}
    # The journal-bearing gap varies along x only: the kernel reads its topography as one (h, hx, hy) triple per row
    # (GPF_TOPO_PLANES=1 disables this), so of SURVEY.md 8(d)'s 72 B per cell-update only 48 B (q read + q write) are
    # compulsory HBM traffic for THIS workload.  `frac` is priced against those 48 B; the 72-B figure of the survey is
    # kept beside it, and the workload where all 72 B are compulsory is the variant below.
    roof = roofline(kernel_ms, cells, BYTES_PER_CELL_LINE, 'line', "k_step2<DH, topography line>")
    roof["frac_survey_8d"] = BYTES_PER_CELL * cells / (kernel_ms / 1e3) / 1e9 / HBM_PEAK_GBS
    roof["all_kernels_ms_per_step"] = all_ms
    roof["plan"] = PLAN_NOTES[-1]
    roof["note"] = ("x-only gap: 48 B per cell-update are compulsory (q read + q write), the topography is one triple per row; "
                    "frac_survey_8d prices the same launch at SURVEY.md 8(d)'s 72 B, which this workload does not move")
    out = {
        "metric": "Mcell-updates/s (fp64), 4096^2 grid", "value": cells * args.steps / wall / 1e6,
        "unit": "Mcell-updates/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"2D journal bearing {N_GRID}x{N_GRID}, fixed DH EOS, all-periodic, adaptive CFL 0.5 "
                               "(BASELINE.json configs[2])", "slabs": 1},
        "roofline": roof, "spread": spread(walls),
    }
    if not args.no_variants:
        # the journal gap is y-invariant, so also time a gap that varies in both directions with a
        # cross flow (all three topography planes are read: the full 72 B per cell-update are compulsory there)
        text = WORKLOAD_YAML.format(N=N_GRID).replace("type: journal\n    CR: 1.e-2\n    eps: 0.7\n    U: 0.1\n    V: 0.",
                                                      "type: asperity\n    hmin: 2.e-6\n    hmax: 1.e-5\n    num: 1\n    U: 0.1\n    V: 0.05")
        assert 'asperity' in text
        w2, k2, a2, walls2 = time_problem(text, args.steps, args.warmup)
        r2 = roofline(k2, cells, BYTES_PER_CELL, 'planes', "k_step2<DH, topography planes>")
        r2["all_kernels_ms_per_step"] = a2
        r2["plan"] = PLAN_NOTES[-1]
        out["variants"] = {"asperity_gap_2d_V0.05": {
            "workload": f"as the headline workload with a gap that varies in x and y (asperity, num 1) and a cross flow V = 0.05",
            "value": cells * args.steps / w2 / 1e6, "unit": "Mcell-updates/s", "ms_per_step": w2 / args.steps * 1e3,
            "kernel_ms": k2, "roofline": r2, "roofline_achieved_GBps": r2["achieved"], "roofline_frac": r2["frac"], "spread": spread(walls2)}}
        # BASELINE.json configs[1]: the 2-D inclined slider at 1024^2 (Dirichlet / Neumann edges in x) -- a million
        # cells, 50 MB per step: the fixed part of a launch (~13 us) is half of it
        # (opt-in, --cfg1: it launches the headline's kernel instantiation on another grid, and the per-kernel tables of a profiled
        # run -- rocprofv3 --stats, the PMC means -- would then mix the two sizes)
        n3 = max(args.steps, 100)
        if args.cfg1:
            w3, k3, a3, walls3 = time_problem(SLIDER_1024_YAML, n3, args.warmup)
            out["variants"]["slider_1024x1024"] = {
                "workload": "2D inclined slider 1024x1024, fixed DH EOS, D/N/N in x, periodic in y, adaptive CFL 0.4 (BASELINE.json configs[1])",
                "value": 1024 * 1024 * n3 / w3 / 1e6, "unit": "Mcell-updates/s", "ms_per_step": w3 / n3 * 1e3, "steps": n3, "kernel_ms": k3,
                "roofline_frac": BYTES_PER_CELL_LINE * 1024 * 1024 / (k3 / 1e3) / 1e9 / HBM_PEAK_GBS, "plan": PLAN_NOTES[-1]}
        if not args.no_gp:
            out["variants"]["gp_2048x2048_512pts"] = gp_variant(max(2, min(args.steps, 10)))
            if not args.no_cpu:
                out["variants"]["gp_2048x2048_512pts"]["cpu_baseline"] = gp_cpu_baseline()
            out["variants"]["gp_slab_1024x8192_rank3of8_512pts"] = gp_slab_variant(max(2, min(args.steps, 6)))
    if not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline()
    return out


SLIDER_1024_YAML = """
options: {silent: True, write_freq: 1000000}
grid: {Nx: 1024, Ny: 1024, Lx: 0.1, Ly: 0.1, xE: ['D', 'N', 'N'], xW: ['D', 'N', 'N'], xE_D: 877.7007, xW_D: 877.7007,
       yS: ['P', 'P', 'P'], yN: ['P', 'P', 'P']}
geometry: {type: inclined, hmax: 6.6e-5, hmin: 1.e-5, U: 50., V: 0.}
numerics: {CFL: 0.4, adaptive: 1, tol: 1.e-12, max_it: 100000000}
properties: {EOS: DH, shear: 0.0794, bulk: 0., rho0: 877.7007, P0: 101325., C1: 3.5e10, C2: 1.23}
"""


GP_YAML = """
options: {{silent: True, write_freq: 1000000}}
grid: {{Nx: {n}, Ny: {n}, Lx: 0.1, Ly: 0.1, xE: ['D', 'N', 'N'], xW: ['D', 'N', 'N'], xE_D: 877.7007, xW_D: 877.7007,
       yS: ['P', 'P', 'P'], yN: ['P', 'P', 'P']}}
geometry: {{type: inclined, hmax: 6.6e-5, hmin: 1.e-5, U: 50., V: 0.}}
numerics: {{CFL: 0.4, adaptive: 1, tol: 1.e-12, max_it: 100000}}
properties: {{EOS: DH, shear: 0.0794, bulk: 0., rho0: 877.7007, P0: 101325., C1: 3.5e10, C2: 1.23}}
gp:
    press: {{atol: 1., rtol: 0.1, obs_stddev: 100., active_learning: False}}
    shear: {{atol: 1., rtol: 0.1, obs_stddev: 1., active_learning: False}}
db: {{init_size: {nt}, init_method: lhc, init_width: 0.01, init_seed: 123}}
"""
FP64_PEAK_TFLOPS = 78.6         # MI355X fp64 vector = fp64 matrix dense peak


def gp_isa_counts():
    """fp64 / other VALU instructions per Matern evaluation in k_gp_mean's inner loop, counted in the gfx950 assembly
    (tools/gp_isa_count.py -> profiles/r03_gp/isa_counts_k_gp_mean.json); None if the file is absent."""
    f = os.path.join(ROOT, 'profiles', 'r03_gp', 'isa_counts_k_gp_mean.json')
    return json.load(open(f)) if os.path.exists(f) else None


def gp_cpu_baseline(n=128, ntrain=512, steps=2):
    """The oracle's stage-wise step with the three surrogates (oracle/gp.py, NumPy / SciPy: the restatement of the reference's
    tinygp path, which cannot run here) on a crop of the same YAML: n x n cells, `ntrain` points, hyper-parameters at their
    initial values.  Bounded sample: ~10-20 s on one core."""
    import io
    from oracle.config import read_yaml_input
    from oracle.problem import OracleProblem
    from oracle import gp as ogp
    try:
        from threadpoolctl import threadpool_limits
        one_thread = threadpool_limits(limits=1, user_api='blas')
    except ImportError:
        one_thread = contextlib.nullcontext()
    with contextlib.redirect_stdout(sys.stderr), one_thread:
        d = read_yaml_input(io.StringIO(GP_YAML.format(n=n, nt=ntrain)))
        p = OracleProblem.from_dict(d)
        ogp.attach(p, d, optimise=False)
        p._pre_run()
        p.update()
        t0 = time.perf_counter()
        for _ in range(steps):
            p.update()
        dt = time.perf_counter() - t0
    return {"value": n * n * steps / dt / 1e6, "unit": "Mcell-updates/s", "cores": 1, "kind": "port",
            "sample": f"oracle/ stage-wise step with three GP closures (NumPy / SciPy restatement of the reference's tinygp path, BLAS held to "
                      f"one thread; host has {os.cpu_count()} cores), same YAML at {n}x{n}, {ntrain} training points, {steps} steps, {dt:.1f} s"}


def gp_variant(steps, n=2048, ntrain=512):
    """BASELINE.json configs[3] (SURVEY.md 8d, cfg4 recipe): 2-D slider 2048^2 with pressure + wall-shear surrogates,
    512 Latin-hypercube training points (seed 123, Mock laws), hyper-parameters fixed at their initial values.
    Times whole MacCormack steps of the stage-wise pipeline (three posterior-mean passes per stage + one sound-speed
    pass) and one predictive-variance pass.  Algorithmic flops per SURVEY 8(d): mean cells*N*(3d+12) per model and
    evaluation (exp / sqrt count 1), variance cells*N^2."""
    from gapflow_amd import Problem
    with contextlib.redirect_stdout(sys.stderr):
        prob = Problem.from_string(GP_YAML.format(n=n, nt=ntrain))
        for m in prob._gp_models.values():
            m.optimise = False
        prob._pre_run()
        prob.update()                                   # warm-up: first-use allocations, rocBLAS handle
        prob._scalars()

        def passes():
            launched, reused = (ctypes.c_int64 * 3)(), ctypes.c_int64()
            assert prob._lib.gpf_gp_pass_counts(prob._h, launched, ctypes.byref(reused)) == 0
            return list(launched), reused.value

        p0, r0 = passes()
        t0 = time.perf_counter()
        for _ in range(steps):
            prob.update()
        sc = prob._scalars()                            # drains the stream
        t_step = (time.perf_counter() - t0) / steps
        p1, r1 = passes()
        factorisation = prob._lib.gpf_gp_factorisation().decode()
        p1[0] -= 1                                      # the pass of the closing _scalars() call is outside a step
        per_step = [(b - a) / steps for a, b in zip(p0, p1)]
        assert sc.invalid == 0 and prob.step == steps + 1, "GP steps were skipped or went invalid"
        prob._gp_models['zz'].compute_variance(on_open_step=False)
        t0 = time.perf_counter()
        prob._gp_models['zz'].compute_variance(on_open_step=False)      # returns after a stream sync
        t_var = time.perf_counter() - t0
        del prob
    cells = (n + 2)**2
    # passes actually launched per step: 2 stages x 3 models + 1 sound speed, minus the pressure evaluation the previous
    # step's sound-speed pass already delivered (gpf_gp_pass_counts)
    evals = cells * ntrain * sum(per_step)
    flops_mean = cells * ntrain * (per_step[0] * (3 * 2 + 12) + (per_step[1] + per_step[2]) * (3 * 3 + 12))
    flops_var = cells * ntrain**2
    # issue-level view: what the vector ALUs were asked to do.  One wave64 VALU instruction occupies a SIMD for 4 cycles whatever
    # its type, so the chip issues 256 CUs x 4 SIMDs x 16 lanes x clock = 39.3e12 lane-instructions/s at 2.4 GHz (= 78.6 TFLOP/s / 2).
    isa = gp_isa_counts()
    issue = None
    if isa:
        per_model = {'pressure': isa['k_gp_mean<2, 1, false>'], 'shear': isa['k_gp_mean<3, 2, false>']}
        ev = [cells * ntrain * per_step[0], cells * ntrain * (per_step[1] + per_step[2])]
        fp64 = ev[0] * per_model['pressure']['fp64_per_evaluation'] + ev[1] * per_model['shear']['fp64_per_evaluation']
        valu = fp64 + ev[0] * per_model['pressure']['other_valu_per_evaluation'] + ev[1] * per_model['shear']['other_valu_per_evaluation']
        peak = FP64_PEAK_TFLOPS / 2 * 1e12
        issue = {"fp64_instructions_per_evaluation": {k: v['fp64_per_evaluation'] for k, v in per_model.items()},
                 "other_valu_instructions_per_evaluation": {k: v['other_valu_per_evaluation'] for k, v in per_model.items()},
                 "fp64_issue_frac": fp64 / t_step / peak, "valu_issue_frac": valu / t_step / peak,
                 "peak_lane_instructions_per_s": peak,
                 "source": "static count in the gfx950 ISA of k_gp_mean's inner loop, profiles/r03_gp/isa_counts_k_gp_mean.json (tools/gp_isa_count.py)"}
    return {
        "workload": f"2D slider {n}x{n}, GP closures (pressure d=2, wall shear xz/yz d=3, Matern-3/2 ARD), {ntrain} training "
                    "points (LHC seed 123), hyper-parameters fixed (BASELINE.json configs[3])",
        "value": n * n / t_step / 1e6, "unit": "Mcell-updates/s", "ms_per_step": t_step * 1e3, "steps": steps,
        "matern_kernel_evaluations_per_s": evals / t_step, "factorisation": factorisation,
        "roofline": {"bound": "fp64_valu", "achieved": flops_mean / t_step / 1e12, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": flops_mean / t_step / 1e12 / FP64_PEAK_TFLOPS, "traffic": None,
                     "kernel": "k_gp_mean (whole stage-wise step timed: ~95 % of it is the posterior-mean passes)",
                     "algorithmic_flops_per_step": flops_mean, "issue_level": issue,
                     "posterior_mean_passes_per_step": {"pressure": per_step[0], "shear_xz": per_step[1], "shear_yz": per_step[2],
                                                        "pressure_reused_from_sound_speed_pass": (r1 - r0) / steps}},
        "variance_pass": {"ms": t_var * 1e3, "model": "pressure",
                          "roofline": {"bound": "mfma_f64", "achieved": flops_var / t_var / 1e12, "peak": FP64_PEAK_TFLOPS,
                                       "unit": "TFLOP/s", "frac": flops_var / t_var / 1e12 / FP64_PEAK_TFLOPS, "traffic": None,
                                       "kernel": "k_gp_var_fused: Matern values through LDS, L^-1 Ks on v_mfma_f64_16x16x4_f64 "
                                                 "(triangular: 52 % of the square product issued), column norms in registers",
                                       "algorithmic_flops_per_pass": flops_var,
                                       "note": "algorithmic flops = cells * N^2 (the triangular product); on gfx950 fp64 VALU work "
                                               "shares the f64 MFMA units, so the Matern evaluations are not hidden"}},
    }


GP_SLAB_YAML = """
options: {{silent: True, write_freq: 1000000}}
grid: {{Nx: {n}, Ny: {n}, dx: 1.e-5, dy: 1.e-5}}
geometry: {{type: journal, CR: 1.e-2, eps: 0.7, U: 0.1, V: 0.}}
numerics: {{CFL: 0.5, adaptive: 1, tol: 1.e-12, max_it: 100000}}
properties: {{EOS: DH, shear: 0.0794, bulk: 0., rho0: 877.7007, P0: 101325., C1: 3.5e10, C2: 1.23}}
gp:
    press: {{atol: 1., rtol: 0.1, obs_stddev: 100., active_learning: False}}
    shear: {{atol: 1., rtol: 0.1, obs_stddev: 1., active_learning: False}}
db: {{init_size: {nt}, init_method: lhc, init_width: 0.01, init_seed: 123}}
"""


def gp_slab_variant(steps, n=8192, ntrain=512, world=8, rank=3):
    """BASELINE.json configs[4] as far as ONE GPU can run it: rank 3's slab (1024 x 8192 cells, neighbours on both sides) of
    the 8192^2 journal bearing with the three surrogates, stage-wise step, three exchanges per step looped back inside the
    process (gapflow_amd.slab.LoopbackGroup: the rank's own rows stand in for its neighbours', so the kernels, message sizes
    and launch sequence are the real ones and the collectives cost a device copy).  Timing only."""
    from gapflow_amd.io import read_yaml_input
    from gapflow_amd.slab import SlabProblem, LoopbackGroup
    import io
    with contextlib.redirect_stdout(sys.stderr):
        d = read_yaml_input(io.StringIO(GP_SLAB_YAML.format(n=n, nt=ntrain)))
        prob = SlabProblem(d, device=0, dist=LoopbackGroup(rank, world))
        for m in prob._gp_models.values():
            m.optimise = False
        prob.pre_run()
        prob.advance(1)
        st0 = prob.state()
        t0 = time.perf_counter()
        prob.advance(steps)
        st = prob.state()                                    # reads the state: drains the stream
        t_step = (time.perf_counter() - t0) / steps
        assert st.invalid == 0 and st.step == st0.step + steps, "GP slab steps were skipped or went invalid"
        rows = prob.layout.nx
        del prob
    return {
        "workload": f"rank {rank} of {world}: {rows} x {n} slab of the 2D journal bearing {n}x{n}, GP closures, {ntrain} training "
                    "points (BASELINE.json configs[4]); neighbour exchange looped back inside one process",
        "value": rows * n / t_step / 1e6, "unit": "Mcell-updates/s (this rank's cells)", "ms_per_step": t_step * 1e3, "steps": steps,
        "whole_job_if_all_ranks_alike_Mcell_per_s": world * rows * n / t_step / 1e6,
    }


def run_slabs(args, rank, world):
    """N x-slabs, one rank per GPU: K steps, host-dispatched, one RCCL all-gather per step (the conservative transport).
    Opt-in (GPF_BENCH_TRY_P2P=1 / GPF_BENCH_TRY_GRAPH=1) the same K steps are timed again with the peer-to-peer mailbox
    transport and with the all-gather steps replayed from a hipGraph; rank 0 then reports the fastest run that completed
    and reproduced the first one, and a failed attempt makes the process exit non-zero."""
    import threading
    import torch
    import torch.distributed as dist
    from gapflow_amd.slab import SlabProblem
    # GPF_BENCH_ONE_GPU_REHEARSAL=1 (development boxes with one GPU; at most 5 ranks there): every rank on device 0, the
    # collectives staged through host memory over gloo -- RCCL refuses two ranks on one device.  Checks the launch path,
    # the partition and the line; its numbers say nothing about xGMI and the line says so.
    rehearsal = os.environ.get('GPF_BENCH_ONE_GPU_REHEARSAL') == '1'
    local = 0 if rehearsal else int(os.environ.get('LOCAL_RANK', rank))
    torch.cuda.set_device(local)
    if rehearsal:
        from gapflow_amd.slab import HostStagedGroup
        dist.init_process_group('gloo')
        dist = HostStagedGroup(dist, torch)
    else:
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    cells = N_GRID * N_GRID

    def timed(prob, k):
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        prob.advance(k)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        w = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device='cuda')
        dist.all_reduce(w, op=dist.ReduceOp.MAX)
        return float(w.item())

    def line(wall, mode):
        per_gpu = BYTES_PER_CELL_LINE * cells * args.steps / wall / 1e9 / world     # x-only gap: 48 B per cell-update
        return {
            "metric": "Mcell-updates/s (fp64), 4096^2 grid", "value": cells * args.steps / wall / 1e6,
            "unit": "Mcell-updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"2D journal bearing {N_GRID}x{N_GRID}, fixed DH EOS, all-periodic, adaptive CFL 0.5 "
                                   "(BASELINE.json configs[2])", "slabs": world, "rccl_ranks": dist.get_world_size(), "dispatch": mode,
                       "parallelism": f"{world} x-slabs, one RCCL all-gather (2 halo rows + 64-B record per rank) per step"},
            "roofline": {"bound": "hbm", "achieved": per_gpu, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": per_gpu / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "whole step per GPU (wall clock incl. the collective)"},
        }

    with contextlib.redirect_stdout(sys.stderr):
        prob = SlabProblem.from_string(WORKLOAD_YAML.format(N=N_GRID), device=local, dist=dist)
        prob.pre_run()
        prob.advance(args.warmup)
        wall = timed(prob, args.steps)
        st = prob.state()
        assert st.step == args.warmup + args.steps and st.invalid == 0, "steps were skipped inside the timed region"
    best = line(wall, "host-dispatched steps, RCCL all-gather" if not rehearsal else
                "REHEARSAL on one GPU: all ranks share device 0, collectives staged through the host over gloo")
    if rehearsal:
        best["rehearsal_one_gpu"] = True
    st_ref = (int(st.step), float(st.dt), float(st.ekin))

    # Two faster ways to run the same K steps can follow (GPF_BENCH_TRY_P2P=1 / GPF_BENCH_TRY_GRAPH=1; both OFF by default:
    # neither can be rehearsed across GPUs on the one-GPU development box).  An attempt counts only if it finishes,
    # reproduces the run above (same step count, dt and kinetic energy) on every rank, and is faster.  An attempt that
    # fails -- watchdog expiry, Python exception, fatal signal from the GPU runtime -- is NOT success: the line already
    # measured is still printed, with the failure recorded in "attempt_failures", and the process exits non-zero.
    failures = []
    FAILED_RC = 3

    def fail_out(name, reason):
        failures.append({"name": name, "reason": reason})
        if rank == 0:
            best["attempt_failures"] = failures
            _emit(best)
        os._exit(FAILED_RC)     # peers may be stuck in a collective or a wedged exchange this rank cannot leave cleanly

    current = {"name": None}

    # a GPU fault inside an attempt aborts the process from a runtime thread (SIGABRT): libc-level handler, the callback
    # only writes the prepared line and exits with the failure status
    import ctypes
    import signal

    @ctypes.CFUNCTYPE(None, ctypes.c_int)
    def on_fatal(signum):
        fail_out(current["name"] or "attempt", f"fatal signal {signum} in the attempt")
    _keep_alive.append(on_fatal)
    libc = ctypes.CDLL(None)
    libc.signal.restype = ctypes.c_void_p
    libc.signal.argtypes = [ctypes.c_int, ctypes.c_void_p]

    def same_run(s2, steps):
        ok = (int(s2.step) == steps and s2.invalid == 0 and abs(s2.dt - st_ref[1]) <= 1e-12 * st_ref[1]
              and abs(s2.ekin - st_ref[2]) <= 1e-10 * st_ref[2])
        t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return float(t.item()) == 1.0

    def attempt(name, fn):
        current["name"] = name
        limit = float(os.environ.get('GPF_BENCH_ATTEMPT_TIMEOUT', 120))
        dog = threading.Timer(limit, fail_out, args=(name, f"watchdog: no result after {limit:.0f} s"))
        dog.daemon = True
        dog.start()
        for sig in (signal.SIGABRT, signal.SIGSEGV, signal.SIGBUS):
            libc.signal(int(sig), ctypes.cast(on_fatal, ctypes.c_void_p))
        try:
            with contextlib.redirect_stdout(sys.stderr):
                fn()
        except Exception as e:      # noqa: BLE001
            dog.cancel()
            print(f"[bench] {name} failed: {type(e).__name__}: {e}", file=sys.stderr)
            fail_out(name, f"{type(e).__name__}: {e}")
        dog.cancel()
        for sig in (signal.SIGABRT, signal.SIGSEGV, signal.SIGBUS):
            libc.signal(int(sig), None)                     # SIG_DFL again
        current["name"] = None

    def try_p2p():
        # the step's own kernels write rows and records into the peers' IPC-mapped mailboxes (gapflow_amd/slab.py)
        nonlocal best
        p2 = SlabProblem.from_string(WORKLOAD_YAML.format(N=N_GRID), device=local, dist=dist)
        if not p2.connect_p2p():
            return
        p2.pre_run()
        p2.advance(args.warmup)
        wall_p = timed(p2, args.steps)
        if same_run(p2.state(), args.warmup + args.steps) and wall_p < wall:
            cand = line(wall_p, "device-driven steps, peer-to-peer mailboxes over xGMI (no collective library)")
            cand["config"]["parallelism"] = (f"{world} x-slabs; per step each GPU stores 2 boundary rows into its "
                                             "neighbours' memory and a 64-B record into everyone's, then polls")
            cand["config"]["rccl_allgather_ms_per_step"] = wall / args.steps * 1e3
            best = cand

    def try_graph():
        nonlocal best
        prob.driver.use_graph = True
        prob.advance(8)                                    # captures
        wall_g = timed(prob, args.steps)
        s2 = prob.state()
        ok = torch.tensor([1.0 if (s2.step == args.warmup + 2 * args.steps + 8 and s2.invalid == 0) else 0.0],
                          dtype=torch.float64, device='cuda')
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) == 1.0 and wall_g * 1e3 / args.steps < best["ms_per_step"]:
            cand = line(wall_g, "hipGraph replay of step pairs, RCCL all-gather")
            cand["config"]["eager_ms_per_step"] = wall / args.steps * 1e3
            best = cand

    if os.environ.get('GPF_BENCH_TRY_P2P', '0') == '1':
        attempt("peer-to-peer transport", try_p2p)
    if os.environ.get('GPF_BENCH_TRY_GRAPH', '0') == '1':
        attempt("graph replay", try_graph)
    best["attempt_failures"] = failures
    dist.barrier()
    dist.destroy_process_group()
    return best if rank == 0 else None


def launch_ranks(args, real_stdout):
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nproc-per-node N bench.py ...`
    as a child on 127.0.0.1 with a free port, pass its stdout (rank 0's JSON line) through, return its exit status."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    for ln in child.stdout:
        real_stdout.write(ln)
        real_stdout.flush()
    return child.wait()


_emit = lambda obj: print(json.dumps(obj), flush=True)
_keep_alive = []


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--no-cpu', action='store_true', help='skip the cpu_baseline leg')
    ap.add_argument('--no-variants', action='store_true', help='skip the 2-D-gap and GP-closure variants of the workload')
    ap.add_argument('--no-gp', action='store_true', help='skip the GP-closure variant (BASELINE.json configs[3])')
    ap.add_argument('--cfg1', action='store_true', help='also time BASELINE.json configs[1] (2-D inclined slider 1024^2) as variants.slider_1024x1024')
    args = ap.parse_args()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    # Native libraries (RCCL prints its version banner) write to file descriptor 1 directly: park the real stdout
    # and point fd 1 at stderr, so that the JSON line is the only thing the driver reads on stdout.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)
    global _emit
    _emit = lambda obj: (real_stdout.write(json.dumps(obj) + '\n'), real_stdout.flush())
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # called as plain `python bench.py --gpus N`: start the N ranks ourselves.  A CHILD process (nothing here has touched
        # the GPU yet, and nothing is re-exec'ed), its JSON line relayed, its exit status passed on.
        raise SystemExit(launch_ranks(args, real_stdout))
    if args.gpus > 1 or world > 1:
        if world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with "
                             f"python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...")
        out = run_slabs(args, rank, world)
    else:
        out = run_single(args)
    if out is not None:
        _emit(out)


if __name__ == '__main__':
    main()
