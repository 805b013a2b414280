import sys, time, io, contextlib
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gapflow_amd import Problem
T = """
options: {silent: True}
grid: {Nx: %d, Ny: 1, dx: 1.e-5, dy: 1.}
geometry: {type: journal, CR: 1.e-2, eps: 0.7, U: 0.1, V: 0.}
numerics: {CFL: 0.25, adaptive: 1, tol: 1.e-30, max_it: 100000000}
properties: {EOS: DH, shear: 0.0794, bulk: 0., rho0: 877.7007}
"""
for nx in (100, 390):
    with contextlib.redirect_stdout(io.StringIO()):
        p = Problem.from_string(T % nx)
        p._pre_run()
        p._advance(2000, honor_stop=False)
        t0 = time.perf_counter()
        p._advance(4000, honor_stop=False); p._advance(4000, honor_stop=False); p._advance(4000, honor_stop=False)
        dt = time.perf_counter() - t0
    print(f"Nx={nx}: {dt / 12000 * 1e6:.2f} us/step, {12000 / dt:.0f} steps/s, ekin={p.kinetic_energy:.12e} step={p.step}")
