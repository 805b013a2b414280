#!/bin/bash
# Rehearsal of bench.py's multi-rank code path with a one-rank RCCL group (all a 1-GPU box allows).
cd "$(dirname "$0")/.."
WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 python - "$@" <<'PY'
import sys, json
sys.argv = ['bench.py', '--gpus', '1', '--steps', '60', '--warmup', '6']
import bench, argparse
args = argparse.Namespace(gpus=1, steps=60, warmup=6, no_cpu=True)
print(json.dumps(bench.run_slabs(args, 0, 1)))
PY
