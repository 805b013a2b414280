"""Command line entry: ``python -m gapflow_amd -i input.yaml`` runs one problem to completion
(same flag as the reference's ``python -m GaPFlow -i``, GaPFlow/__main__.py:28-48).

Several GPUs of one node: start one process per GPU,

    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m gapflow_amd -i input.yaml

and the problem is cut into x-slabs (gapflow_amd/slab.py); rank 0 writes the output.  GPF_SLAB_TRANSPORT=p2p selects the
peer-to-peer mailbox transport instead of one all-gather per step."""
import argparse
import os
import sys

from . import Problem


def main(argv=None):
    cli = argparse.ArgumentParser(prog='python -m gapflow_amd',
                                  description="Advance a GaPFlow YAML problem on an MI355X.")
    cli.add_argument('-i', '--input', dest='filename', required=True, metavar='YAML', help="problem definition")
    cli.add_argument('--device', type=int, default=0, help="HIP device ordinal of a single-process run (default 0)")
    opts = cli.parse_args(argv)
    if int(os.environ.get('WORLD_SIZE', '1')) > 1:
        import torch
        import torch.distributed as dist
        from .slab import SlabProblem
        local = int(os.environ.get('LOCAL_RANK', '0'))
        torch.cuda.set_device(local)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        try:
            SlabProblem.from_yaml(opts.filename, device=local).run()
        finally:
            dist.destroy_process_group()
        return 0
    Problem.from_yaml(opts.filename, device=opts.device).run()
    return 0


if __name__ == "__main__":
    sys.exit(main())
