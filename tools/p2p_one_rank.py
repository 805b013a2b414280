"""Per-step latency of the slab step on ONE rank: all-gather transport vs peer-to-peer transport, for the whole
4096^2 grid and for one eighth of it (what a rank owns at 8 GPUs).  A one-rank group exercises every kernel and the
collective's fixed cost, not the xGMI hop.  Usage: python tools/p2p_one_rank.py"""
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
CASES = os.environ.get('P2P_CASES', '4096:allgather,4096:p2p,512:allgather,512:p2p').split(',')
for nx in (4096, 512):
    text = WORKLOAD_YAML.format(N=4096).replace('Nx: 4096', f'Nx: {nx}')
    for mode in ('allgather', 'p2p'):
        if f'{nx}:{mode}' not in CASES:
            continue
        prob = SlabProblem.from_string(text, device=0)
        if mode == 'p2p':
            assert prob.connect_p2p()
        prob.pre_run()
        prob.advance(20)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        prob.advance(400)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 400
        st = prob.state()
        print(f"Nx={nx:5d} Ny=4096 {mode:9s}: {dt * 1e6:8.1f} us/step  step={st.step} dt={st.dt:.6e} ekin={st.ekin:.12e}", flush=True)
        del prob
dist.destroy_process_group()
