"""Per-step latency of the slab step on ONE rank: all-gather transport vs peer-to-peer transport, for the whole
4096^2 grid and for one eighth of it (what a rank owns at 8 GPUs), each in the arrangements the library offers:

    (default)     k_step2 + k_begin_slab per step, plan and placement as the handle times them (csrc/api_plan.inc: plan_step2)
    notune        GPF_PLAN_TUNE=0: rule-of-thumb plan, first placement
    chunks=N      GPF_CHUNKS=N: N row chunks per strip instead of what plan_step2 picks
(the `folded` / `every-step` pair of profiles/r03_slab/ was measured with the patch kept there)

A one-rank group exercises every kernel and the collective's fixed cost, not the xGMI hop.
Usage: python tools/p2p_one_rank.py            (P2P_CASES=512:p2p,... restricts the list)"""
import contextlib
import io
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29533')

import torch
import torch.distributed as dist

from bench import WORKLOAD_YAML
from gapflow_amd.slab import SlabProblem

torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
DEFAULT = '4096:allgather,4096:p2p,512:allgather,512:p2p,512:p2p:notune,512:p2p:chunks=31,512:p2p:chunks=62'
CASES = os.environ.get('P2P_CASES', DEFAULT).split(',')
NSTEPS = int(os.environ.get('P2P_STEPS', 400))
print(f"{'rows':>5s} {'transport':>9s} {'arrangement':>12s} {'us/step':>9s}   state after the run", flush=True)
for case in CASES:
    parts = case.split(':')
    nx, mode, opts = int(parts[0]), parts[1], parts[2:]
    env = {}
    for o in opts:
        if o == 'notune':
            env['GPF_PLAN_TUNE'] = '0'
        elif o.startswith('chunks='):
            env['GPF_CHUNKS'] = o.split('=')[1]
    for k in ('GPF_PLAN_TUNE', 'GPF_CHUNKS'):
        os.environ.pop(k, None)
    os.environ.update(env)
    text = WORKLOAD_YAML.format(N=4096).replace('Nx: 4096', f'Nx: {nx}')
    with contextlib.redirect_stdout(io.StringIO()):         # the set-up listing (printed whatever `silent` says, as the reference does)
        prob = SlabProblem.from_string(text, device=0)
    if mode == 'p2p':
        assert prob.connect_p2p()
    prob.pre_run()
    prob.advance(20)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        prob.advance(NSTEPS)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / NSTEPS)
    st = prob.state()
    print(f"{nx:5d} {mode:>9s} {(' '.join(opts) or 'default'):>12s} {best * 1e6:9.1f}   step={st.step} dt={st.dt:.6e} ekin={st.ekin:.12e}", flush=True)
    del prob
dist.destroy_process_group()
