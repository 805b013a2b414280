"""Experiment: can two ranks share one GPU over RCCL?  (Used once to see whether the halo ring can be
exercised on a 1-GPU box; not part of the test suite.)"""
import os, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch, torch.distributed as dist
from gapflow_amd.slab import SlabProblem
from gapflow_amd import Problem
SIM = open(os.path.join(os.path.dirname(__file__), '..', 'tests', 'test_gpu_slab.py')).read().split('SIM = """')[1].split('"""')[0]
torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
rank, world = dist.get_rank(), dist.get_world_size()
slab = SlabProblem.from_string(SIM)
slab.pre_run(); slab.advance(20)
st = slab.state()
serial = Problem.from_string(SIM); serial._pre_run(); serial._advance(20, honor_stop=False)
L = slab.layout
ref = serial.q[:, L.rows()]
err = max(np.abs(slab.local_q()[c][1:-1] - ref[c][1:-1]).max() / np.abs(ref[c]).max() for c in range(3))
print(f"rank {rank}/{world}: step {st.step} dt {st.dt:.6e} (serial {serial.dt:.6e}) max rel err interior {err:.2e}", flush=True)
dist.barrier(); dist.destroy_process_group()
