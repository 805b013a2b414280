"""rocSOLVER's dpotrf / dpotrs against the in-library blocked Cholesky, through gpf_gp_fit, one child process per factorisation
(the switch GPF_USE_ROCSOLVER is read once per process): time per fit at the sizes a surrogate has, and the numerically singular
kernel matrix a TRAINED pressure surrogate produces (tests/test_gpu_gp.py: the blocked one must factorise it like LAPACK).

    python tools/rocsolver_vs_inlibrary.py            # prints a table"""
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import contextlib, io
    from scipy.linalg import lapack
    t0 = time.time()
    from gapflow_amd import _lib
    lib = _lib.require_device()
    out = {'load_s': time.time() - t0, 'fits': {}}
    rng = np.random.default_rng(4)
    for n in (128, 256, 512):
        d, m = 3, 2
        X = rng.uniform(0.5, 1.0, (n, d))
        Y = rng.standard_normal((n, m))
        Xc, Yc, sc = _lib.f64c(X), _lib.f64c(Y), _lib.f64c(np.array([2.0, 0.8, 1.5]))
        L, alpha, ld = np.empty((n, n)), np.empty((n, m)), C.c_double(0)
        ts = []
        for rep in range(7):
            t = time.perf_counter()
            _lib.check(lib.gpf_gp_fit(0, n, d, m, _lib.as_dp(Xc), _lib.as_dp(Yc), 1.2, _lib.as_dp(sc), 0.05, _lib.as_dp(L), _lib.as_dp(alpha), C.byref(ld)))
            ts.append(time.perf_counter() - t)
        out['fits'][n] = {'first_ms': ts[0] * 1e3, 'median_ms': sorted(ts[2:])[len(ts[2:]) // 2] * 1e3}
    # the singular trained kernel matrix
    from bench import GP_YAML
    from gapflow_amd import Problem
    text = GP_YAML.format(n=64, nt=256).replace('obs_stddev: 100., active_learning: False', 'obs_stddev: 1.e5, active_learning: False')
    with contextlib.redirect_stdout(io.StringIO()):
        prob = Problem.from_string(text)
        for m_ in prob._gp_models.values():
            m_.optimise = False
        prob._pre_run()
    mz = prob._gp_models['zz']
    X, Y, s = mz.Xtrain, mz.Ytrain, mz.Yerr
    th = np.array([11.70824469, 2.93533174, 14.77329233])
    amp, inv = np.exp(th[0]), np.exp(-th[1:])
    Z = X * inv
    r = np.sqrt(3 * ((Z[:, None, :] - Z[None, :, :])**2).sum(-1))
    K = amp * (1 + r) * np.exp(-r) + s**2 * np.eye(len(X))
    c, info = lapack.dpotrf(K, lower=True)
    L, alpha, ld = np.zeros_like(K), np.zeros((len(X), 1)), C.c_double()
    rc = lib.gpf_gp_fit(0, len(X), X.shape[1], 1, _lib.as_dp(_lib.f64c(X)), _lib.as_dp(_lib.f64c(Y)), amp, _lib.as_dp(_lib.f64c(inv)), s,
                        _lib.as_dp(L), _lib.as_dp(alpha), C.byref(ld))
    out['singular'] = {'rc': rc, 'error': lib.gpf_last_error().decode() if rc else '', 'lapack_info': int(info),
                       'max_abs_L_minus_lapack_over_max_L': float(np.abs(L - np.tril(c)).max() / np.abs(c).max()) if rc == 0 else None,
                       'min_pivot_over_sigma': float(np.diag(L).min() / s) if rc == 0 else None}
    print('RESULT ' + json.dumps(out))


def main():
    if '--child' in sys.argv:
        return child()
    rows = {}
    for name, env in (('in-library blocked Cholesky', {'GPF_USE_ROCSOLVER': '0'}), ('rocSOLVER dpotrf / dpotrs', {'GPF_USE_ROCSOLVER': '1'})):
        res = subprocess.run([sys.executable, os.path.abspath(__file__), '--child'], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        line = [l for l in res.stdout.splitlines() if l.startswith('RESULT ')]
        rows[name] = json.loads(line[0][7:]) if line else {'failed': res.stderr[-800:]}
    for name, r in rows.items():
        print(name)
        if 'failed' in r:
            print('   FAILED', r['failed'])
            continue
        print(f"   library load + first HIP call: {r['load_s']:.2f} s")
        for n, f in r['fits'].items():
            print(f"   n = {n}: first fit {f['first_ms']:.1f} ms, median {f['median_ms']:.2f} ms")
        print('   singular trained K (n = 256):', r['singular'])
    print('JSON ' + json.dumps(rows))


if __name__ == '__main__':
    main()
