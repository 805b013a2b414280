"""Wall time per step of the headline workload with no validity checks (for timing experiments with deliberately
incomplete kernels, e.g. -DGPF_K2_NO_POSTPASS): python tools/time_raw.py N [steps].  GPF_LIB_PATH selects the build."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import WORKLOAD_YAML
from gapflow_amd import Problem

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
text = WORKLOAD_YAML.format(N=n)
EOS_PROPS = {       # the reference's default parameter sets (io.py)
    'PL': 'EOS: PL\n    rho0: 1.1853\n    P0: 101325.\n    alpha: 0.',
    'vdW': 'EOS: vdW\n    M: 39.948\n    T: 100.\n    a: 1.355\n    b: 0.03201\n    rho0: 1400.',
    'cubic': 'EOS: cubic\n    a: 15.2\n    b: -9.6\n    c: 3.35\n    d: -0.07\n    rho0: 0.8',
    'BWR': 'EOS: BWR\n    T: 2.\n    gamma: 3.0\n    rho0: 0.8',
    'Bayada': 'EOS: Bayada\n    rho_l: 850.\n    rho_v: 0.019\n    c_l: 1600.\n    c_v: 352.\n    rho0: 850.',
}
if os.environ.get('EOS') in EOS_PROPS:
    import re
    text = re.sub(r'EOS: DH\n    P0: 101325\.\n    rho0: 877\.7007\n    C1: 3\.5e10\n    C2: 1\.23', EOS_PROPS[os.environ['EOS']], text)
    assert 'EOS: DH' not in text, text
if os.environ.get('EOS') == 'MT':          # Murnaghan-Tait instead of Dowson-Higginson (a pow() per pressure)
    text = text.replace('EOS: DH', 'EOS: MT').replace('rho0: 877.7007', 'rho0: 700.\n    K: 0.557e9\n    n: 7.33').replace('P0: 101325.', 'P0: 0.101e6')
if os.environ.get('GAP') == '2d':          # asperity gap with a cross flow: topography planes
    text = text.replace("type: journal\n    CR: 1.e-2\n    eps: 0.7\n    U: 0.1\n    V: 0.", "type: asperity\n    hmin: 2.e-6\n    hmax: 1.e-5\n    num: 1\n    U: 0.1\n    V: 0.05")
p = Problem.from_string(text)
p._pre_run()
p._advance(10, honor_stop=False)
torch.cuda.synchronize()
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    p._advance(steps, honor_stop=False)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / steps)
print(f"N={n} EOS={os.environ.get('EOS', 'DH')} gap={os.environ.get('GAP', 'x-only')} {os.path.basename(os.environ.get('GPF_LIB_PATH', 'default'))}: {best * 1e6:.1f} us/step, state invalid={p._scalars().invalid}", flush=True)
