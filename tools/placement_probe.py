"""Does the step kernel's speed depend on WHERE its fields lie in device memory?  Several identical handles of the benchmark problem
in one process (tools/ab_inprocess.py found the first two 9 % faster than the next four, at identical code and settings), timed in
alternating batches; then the first two are destroyed and two new ones created (do they inherit the fast memory?); optionally a
large dummy allocation is made first.

    python tools/placement_probe.py [--dummy-gb 20] [--handles 6]"""
import ctypes as C
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gapflow_amd import Problem, _lib

args = sys.argv[1:]
dummy_gb, nh = 0, 6
while args:
    if args[0] == '--dummy-gb':
        dummy_gb = float(args[1])
    elif args[0] == '--handles':
        nh = int(args[1])
    args = args[2:]
os.environ.update(GPF_CHUNKS='31', GPF_NT_STORES='0')
lib = _lib.require_device()
hip = C.CDLL('libamdhip64.so.7')
hip.hipMalloc.argtypes, hip.hipMalloc.restype = [C.POINTER(C.c_void_p), C.c_size_t], C.c_int
hip.hipFree.argtypes = [C.c_void_p]
dummy = C.c_void_p()
if dummy_gb:
    assert hip.hipMalloc(C.byref(dummy), int(dummy_gb * 2**30)) == 0
    print(f'dummy allocation of {dummy_gb} GiB at {dummy.value:#x}')
text = bench.WORKLOAD_YAML.format(N=4096)


def make():
    p = Problem.from_string(text)
    p._pre_run()
    p._advance(5, honor_stop=False)
    return p


def timeall(probs, batches=10, steps=40):
    kt, tt = C.c_double(0), C.c_double(0)
    res = {k: [] for k in probs}
    for b in range(batches):
        for k, p in (list(probs.items()) if b % 2 == 0 else list(probs.items())[::-1]):
            _lib.check(lib.gpf_step_timed(p._h, steps, C.byref(kt), C.byref(tt)))
            res[k].append(kt.value / steps * 1e3)
    return {k: statistics.median(v[2:]) for k, v in res.items()}


probs = {f'h{i + 1}': make() for i in range(nh)}
print('six handles created in order   :', {k: round(v, 1) for k, v in timeall(probs).items()})
for k in ('h1', 'h2'):
    del probs[k]
import gc
gc.collect()
probs.update({'n1': make(), 'n2': make()})
print('h1, h2 destroyed, n1, n2 created:', {k: round(v, 1) for k, v in timeall(probs).items()})
