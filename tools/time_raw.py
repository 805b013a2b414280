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
p = Problem.from_string(WORKLOAD_YAML.format(N=n))
p._pre_run()
p._advance(10, honor_stop=False)
torch.cuda.synchronize()
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    p._advance(steps, honor_stop=False)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / steps)
print(f"N={n} {os.environ.get('GPF_LIB_PATH', 'default')}: {best * 1e6:.1f} us/step", flush=True)
