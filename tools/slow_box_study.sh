#!/bin/bash
# On the occasional slow box of the pool (same build: 175 instead of 150 us per step, every buffer of the placement pool alike) look
# for what still helps: piece sizes of the scattered fields, chunk counts, cache hints.  Exits at once on a fast box.
t=$(timeout -k 10 170 python bench.py --steps 20 --warmup 5 --no-cpu --no-variants 2>/dev/null | python3 -c "import json,sys; print(int(json.loads(sys.stdin.read())['ms_per_step']*1e3))")
echo "quick step time: $t us"
if [ "$t" -lt 165 ]; then echo "fast box: nothing to study"; exit 0; fi
rocm-smi --showmemorypartition --showcomputepartition --showclocks 2>&1 | grep -v "^=\|^$\|WARNING" | head -12
F="^-\|^  -\|^\*\|^    -\|amdgpu.ids"
echo "== piece sizes (search on)"
timeout -k 10 300 python tools/ab_inprocess.py --gap journal --batches 6 s16: s2:GPF_SCATTER_MB=2 s64:GPF_SCATTER_MB=64 s0:GPF_SCATTER_MB=0 s4:GPF_SCATTER_MB=4 2>&1 | grep -v "$F" | cut -c1-260
echo "== chunk counts (no search, scattered)"
B="GPF_NT=2,GPF_PLACEMENT_TRIES=0"
timeout -k 10 300 python tools/ab_inprocess.py --gap journal --batches 6 c62:$B,GPF_CHUNKS=62 c46:$B,GPF_CHUNKS=46 c31:$B,GPF_CHUNKS=31 c93:$B,GPF_CHUNKS=93 c124:$B,GPF_CHUNKS=124 2>&1 | grep "kernel median"
echo "== GP step (fp64-bound: the clocks)"
timeout -k 10 170 python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('gp ms', d['variants']['gp_2048x2048_512pts']['ms_per_step'], 'stream', d['roofline']['stream_ceiling_GBps'])"
