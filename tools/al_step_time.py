"""Where an active-learning step spends its time: 2-D slider with the three surrogates, active learning on for the pressure
model, a few steps.  Usage: python tools/al_step_time.py [n] [ntrain] [rtol of the pressure model]"""
import contextlib
import io
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import GP_YAML
from gapflow_amd import Problem

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 256
rtol = sys.argv[3] if len(sys.argv) > 3 else '0.1'
text = GP_YAML.format(n=n, nt=nt).replace('atol: 1., rtol: 0.1, obs_stddev: 100., active_learning: False', f'atol: 1., rtol: {rtol}, obs_stddev: 1.e5, active_learning: True, max_steps: 2, pause_steps: 3')
text = text.replace('obs_stddev: 1., active_learning: False', 'obs_stddev: 500., active_learning: False')
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    prob = Problem.from_string(text)
    prob._pre_run()
    t0 = time.perf_counter()
    steps = 6
    for _ in range(steps):
        prob.update()
    prob._scalars()
    wall = time.perf_counter() - t0
print(f"{n}x{n}, {nt} initial points: {wall / steps * 1e3:.1f} ms per step over {steps} steps; database now {prob.database.size} points")
for name, m in prob._gp_models.items():
    print(f"  {name}: train {m.cumtime_train.total_seconds():.3f} s, infer/variance calls {m.cumtime_infer.total_seconds():.3f} s, AL {m.use_active_learning}")
