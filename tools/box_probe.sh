#!/bin/bash
# What kind of box is this?  Partition modes, clocks and the quick step time, to correlate the slow boxes of the pool with their set-up.
rocm-smi --showmemorypartition --showcomputepartition 2>&1 | grep -v "^=\|^$" | head -8
rocm-smi --showclocks 2>&1 | grep -i "sclk\|mclk\|fclk" | head -6
rocm-smi --showpower --showmaxpower 2>&1 | grep -i "power" | head -4
rocminfo 2>/dev/null | grep -i "gfx950\|Compute Unit\|Max Clock\|Memory Properties" | head -6
timeout -k 10 170 python bench.py --steps 20 --warmup 5 --no-cpu --no-gp 2>/dev/null | python3 -c "
import json,sys,re
d=json.loads(sys.stdin.read()); r=d['roofline']
print('step us', d['ms_per_step']*1e3, 'stream', r['stream_ceiling_GBps'], re.search(r'placement:[^;]*', r['plan']).group(0))"
