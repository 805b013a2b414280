"""A/B of run-time choices of the step kernel inside ONE process: the variants are separate handles of the same problem, timed in
short alternating batches (HIP events around every launch, gpf_step_timed), so that clock and power drift -- +-10 % between the
rounds of tools/ab_step.py on this pool -- hits both alike.  A variant = environment variables read when its handle plans its
first step (GPF_NT_STORES, GPF_CHUNKS, GPF_TOPO_PLANES ...).

    python tools/ab_inprocess.py [--n 4096] [--batches 24] [--steps 40] [--gap journal|asperity] name:VAR=VAL[,VAR=VAL] ...
    e.g. python tools/ab_inprocess.py plain:GPF_NT_STORES=0 nt:GPF_NT_STORES=1

`lib=path/to/variant.so` among a variant's settings loads ANOTHER build of the library into the same process (compile-time choices:
python -m gapflow_amd.build --variant NAME -DGPF_ONLY_EOS_DH -D...); the handles of the variants are independent."""
import ctypes as C
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gapflow_amd import Problem, _lib


def main():
    args = sys.argv[1:]
    n, batches, steps, gap = 4096, 24, 40, 'journal'
    while args and args[0].startswith('--'):
        k, v = args[0], args[1]
        n = int(v) if k == '--n' else n
        batches = int(v) if k == '--batches' else batches
        steps = int(v) if k == '--steps' else steps
        gap = v if k == '--gap' else gap
        args = args[2:]
    text = bench.WORKLOAD_YAML.format(N=n)
    if gap == 'asperity':
        text = text.replace("type: journal\n    CR: 1.e-2\n    eps: 0.7\n    U: 0.1\n    V: 0.",
                            "type: asperity\n    hmin: 2.e-6\n    hmax: 1.e-5\n    num: 1\n    U: 0.1\n    V: 0.05")
    _lib.load()
    variants = []
    for a in args:
        name, rest = a.split(':', 1)
        env = dict(p.split('=', 1) for p in rest.split(',') if p)
        libpath = env.pop('lib', None)
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        default_lib = _lib._lib
        if libpath:                     # a second (third ...) build in this process: bound like the default one
            other = C.CDLL(os.path.abspath(libpath))
            for fname, (res, argt) in _lib.SIGNATURES.items():
                fn = getattr(other, fname)
                fn.restype, fn.argtypes = res, argt
            _lib._lib = other
        try:
            prob = Problem.from_string(text)
        finally:
            _lib._lib = default_lib
        prob._pre_run()
        prob._advance(5, honor_stop=False)          # plans with this variant's environment
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        print(f'{name}: {prob._lib.gpf_plan_note(prob._h).decode()}', flush=True)
        variants.append((name, prob, []))
    kt, tt = C.c_double(0), C.c_double(0)
    for b in range(batches):
        order = variants if b % 2 == 0 else variants[::-1]
        for name, prob, times in order:
            _lib.check(prob._lib.gpf_step_timed(prob._h, steps, C.byref(kt), C.byref(tt)))
            times.append((kt.value / steps * 1e3, tt.value / steps * 1e3))
    base = variants[0][2]
    print(f'{gap} gap, {n}^2, {batches} alternating batches of {steps} steps; kernel / whole step, us')
    for name, prob, times in variants:
        k = [t[0] for t in times[2:]]
        s = [t[1] for t in times[2:]]
        ratio = statistics.median(t[0] / b[0] for t, b in zip(times[2:], base[2:]))
        print(f'{name:>14s}: kernel median {statistics.median(k):7.1f} (min {min(k):7.1f}, max {max(k):7.1f})   step median {statistics.median(s):7.1f}'
              f'   paired kernel ratio to {variants[0][0]}: {ratio:.4f}')


if __name__ == '__main__':
    main()
