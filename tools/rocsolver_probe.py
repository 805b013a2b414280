"""Times the rocSOLVER path of the library phase by phase and lists the ROCm images the process holds (round 3: after the
loader fix -- rocBLAS / rocSOLVER bound to the images already mapped or to the copy beside the rocBLAS in use, mapped BEFORE
the first HIP call; gapflow_amd/_lib.py: _preload_rocsolver, csrc/gp_kernels.hip: roclibs).  Run through tools/rocsolver_diag.sh,
which takes the stacks with rocgdb if a phase stalls.

    GPF_USE_ROCSOLVER=1 python tools/rocsolver_probe.py [late]      # late: GPF_ROCSOLVER_NO_PRELOAD, dlopen after HIP init"""
import ctypes as C
import os
import re
import sys
import time

import numpy as np

sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
t0 = time.time()


def stamp(what):
    print(f'[{time.time() - t0:8.2f} s] {what}', flush=True)


def images():
    seen = sorted({ln.split()[-1] for ln in open('/proc/self/maps')
                   if re.search(r'rocblas|rocsolver|amdhip64|rocroller|hipblaslt|hsa-runtime', ln) and ln.split()[-1].startswith('/')})
    for s in seen:
        print('      ', s, flush=True)


from gapflow_amd import _lib  # noqa: E402
if 'late' in sys.argv:
    _lib._preload_rocsolver = lambda: None          # round 2's order: rocSOLVER is first opened inside gpf_gp_fit, HIP already up
stamp('loading libgapflow_hip.so (pins the HIP runtime; maps rocBLAS + rocSOLVER first unless `late`)')
lib = _lib.load()
images()
stamp('first HIP call (gpf_device_count)')
assert lib.gpf_device_count() >= 1
rng = np.random.default_rng(4)
n, d, m = 300, 3, 2
X = rng.uniform(0.5, 1.0, (n, d))
Y = rng.standard_normal((n, m))
Xc, Yc, sc = _lib.f64c(X), _lib.f64c(Y), _lib.f64c(np.array([2.0, 0.8, 1.5]))
L, alpha, logdet = np.empty((n, n)), np.empty((n, m)), C.c_double(0)
stamp('gpf_gp_fit #1 (rocblas handle, rocsolver_dpotrf / dpotrs)')
_lib.check(lib.gpf_gp_fit(0, n, d, m, _lib.as_dp(Xc), _lib.as_dp(Yc), 1.2, _lib.as_dp(sc), 0.05, _lib.as_dp(L), _lib.as_dp(alpha), C.byref(logdet)))
stamp('gpf_gp_fit #1 done')
images()
_lib.check(lib.gpf_gp_fit(0, n, d, m, _lib.as_dp(Xc), _lib.as_dp(Yc), 1.2, _lib.as_dp(sc), 0.05, _lib.as_dp(L), _lib.as_dp(alpha), C.byref(logdet)))
stamp('gpf_gp_fit #2 done')
K = 1.2 * (lambda r: (1 + np.sqrt(3) * r) * np.exp(-np.sqrt(3) * r))(np.sqrt((((X[:, None] - X[None]) * np.array([2.0, 0.8, 1.5]))**2).sum(-1))) + 0.05**2 * np.eye(n)
print('      max |L L^T - K| / max K =', np.abs(np.tril(L) @ np.tril(L).T - K).max() / K.max(), flush=True)
