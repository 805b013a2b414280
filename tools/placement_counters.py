"""Eight identical handles, ten steps each, one handle after the other (GPF_PLAN_TUNE=0: no trial launches): run under
rocprofv3 --pmc ... --kernel-trace, the k_step2 dispatches group by handle and the counters can be held against the handles'
speeds (tools/placement_probe.py).  Prints the HIP-event kernel time per handle."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(GPF_PLAN_TUNE='0', GPF_CHUNKS='31', GPF_NT_STORES='0')
import bench
from gapflow_amd import Problem, _lib

lib = _lib.require_device()
text = bench.WORKLOAD_YAML.format(N=4096)
probs = []
for i in range(8):
    p = Problem.from_string(text)
    p._pre_run()
    probs.append(p)
kt, tt = C.c_double(0), C.c_double(0)
for rnd in range(2):
    for i, p in enumerate(probs):
        _lib.check(lib.gpf_step_timed(p._h, 10, C.byref(kt), C.byref(tt)))
        if rnd == 1:
            print(f'handle {i}: {kt.value / 10 * 1e3:.1f} us per launch', flush=True)
