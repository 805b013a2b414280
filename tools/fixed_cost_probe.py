"""What a launch of k_step2 costs before it moves a byte: the fused step on grids of Nx x 4096 cells for a few Nx, kernel time (HIP
events around every launch) and whole step, so that slope and intercept can be read off; run it with builds that leave parts out
(GPF_LIB_PATH=gapflow_amd/lib/variants/<name>.so; python -m gapflow_amd.build --variant <name> -DGPF_ONLY_EOS_DH
-DGPF_K2_NO_POSTPASS | -DGPF_K2_NO_FINISH: timing experiments, wrong results).

    python tools/fixed_cost_probe.py [rows ...]"""
import contextlib
import ctypes as C
import io
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gapflow_amd import Problem, _lib

rows = [int(a) for a in sys.argv[1:]] or [64, 128, 256, 512, 1024, 2048]
print(f"{'rows':>6s} {'kernel us':>10s} {'step us':>9s}   plan", flush=True)
for nx in rows:
    text = bench.WORKLOAD_YAML.format(N=4096).replace('Nx: 4096', f'Nx: {nx}')
    with contextlib.redirect_stdout(io.StringIO()):
        prob = Problem.from_string(text)
    prob._pre_run()
    prob._advance(20, honor_stop=False)
    kt, tt = C.c_double(0), C.c_double(0)
    best = (1e9, 1e9)
    for _ in range(3):
        _lib.check(prob._lib.gpf_step_timed(prob._h, 300, C.byref(kt), C.byref(tt)))
        best = min(best, (kt.value / 300 * 1e3, tt.value / 300 * 1e3))
    print(f'{nx:6d} {best[0]:10.1f} {best[1]:9.1f}   {prob._lib.gpf_plan_note(prob._h).decode()[:60]}', flush=True)
    del prob
