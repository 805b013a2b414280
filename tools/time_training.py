"""Hyper-parameter training of the surrogates at BASELINE configs[3]'s training-set size (512 points): wall time of
the BFGS run of Surrogate.train with the objective on the device (default) and on the host (GPF_GP_TRAIN=host), and the
time of ONE objective + gradient evaluation each way.  Usage: python tools/time_training.py [ntrain]"""
import contextlib
import io
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from bench import GP_YAML
from gapflow_amd import Problem
from gapflow_amd.gp import NegLogLikelihood, DeviceNegLogLikelihood

nt = int(sys.argv[1]) if len(sys.argv) > 1 else 512
with contextlib.redirect_stdout(io.StringIO()):
    text = GP_YAML.format(n=256, nt=nt)
    if os.environ.get('BENIGN', '1') == '1':        # the reference examples' noise levels instead of bench.py's tiny ones
        text = text.replace('obs_stddev: 100.', 'obs_stddev: 1.e5').replace('obs_stddev: 1.,', 'obs_stddev: 500.,')
    prob = Problem.from_string(text)
    for m in prob._gp_models.values():
        m.optimise = False
    prob._pre_run()
for name in ('zz', 'xz'):
    m = prob._gp_models[name]
    X, Y, s = m.Xtrain, m.Ytrain, m.Yerr
    theta0 = np.concatenate([[0.0], np.log(np.std(X, axis=0))])
    host = NegLogLikelihood(X, Y, s)
    with DeviceNegLogLikelihood(X, Y, s) as dev:
        dev(theta0)
        t0 = time.perf_counter()
        for _ in range(10):
            fd, gd = dev(theta0)
        t_dev = (time.perf_counter() - t0) / 10
    host(theta0)
    t0 = time.perf_counter()
    for _ in range(3):
        fh, gh = host(theta0)
    t_host = (time.perf_counter() - t0) / 3
    print(f"{name}: n = {X.shape[0]}, d = {X.shape[1]}, m = {Y.shape[1]}: one objective + gradient {t_dev * 1e3:.2f} ms on the device, "
          f"{t_host * 1e3:.1f} ms on the host (value {fd:.10g} / {fh:.10g})", flush=True)
    from scipy.optimize import minimize
    f_h, g_h = host(theta0)
    for mode in ('device', 'host'):
        t0 = time.perf_counter()
        if mode == 'device':
            with DeviceNegLogLikelihood(X, Y, s) as dev:
                res = minimize(dev, theta0, jac=True, method='BFGS')
        else:
            res = minimize(host, theta0, jac=True, method='BFGS')
        t = time.perf_counter() - t0
        fh, gh = host(res.x)
        print(f"   BFGS ({mode}): {t:.3f} s, {res.nfev} evaluations, objective {res.fun:.8g}, theta = {np.array2string(res.x, precision=4)}; "
              f"host statement there: {fh:.8g}, max |grad| {np.abs(gh).max():.3g}", flush=True)
