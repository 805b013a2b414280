"""Predictive-variance pass of the pressure surrogate at BASELINE configs[3] (2048^2 cells x 512 training points): wall time
of gpf_gp_variance (returns after a stream sync).  Usage: python tools/time_variance.py [n] [ntrain]"""
import contextlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import GP_YAML
from gapflow_amd import Problem

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 512
with contextlib.redirect_stdout(sys.stderr):
    prob = Problem.from_string(GP_YAML.format(n=n, nt=nt))
    for m in prob._gp_models.values():
        m.optimise = False
    prob._pre_run()
    zz = prob._gp_models['zz']
    zz.compute_variance(on_open_step=False)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        zz.compute_variance(on_open_step=False)
        ts.append(time.perf_counter() - t0)
    var = zz.variance
print(f"[{os.path.basename(os.environ.get('GPF_LIB_PATH', 'default'))}] variance pass {n}x{n} x {nt} points: {min(ts) * 1e3:.2f} ms (max var {zz.maximum_variance:.17e}, checksum {float(var.sum()):.17e})"
      , flush=True)
