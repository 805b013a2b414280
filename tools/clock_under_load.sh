#!/bin/bash
# Core clock and package power WHILE the step kernel runs (a long bench in the background, rocm-smi sampled in its timed region), and
# the same for the fp64-bound GP step: is the occasional slow box of the pool a part that throttles under the step kernel's mix of
# memory traffic and fp64 arithmetic?
sample() { for i in 1 2 3; do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -i "clk clock level\|Package Power (W)\|junction\|Sensor memory" | sed 's/GPU\[0\]\s*: //; s/clock level: [0-9S]*: //; s/Current Socket Graphics Package Power (W)/W/; s/Temperature (Sensor \([a-z]*\)) (C)/T\1/' | tr '\n' ' '; echo; sleep 1.0; done; }
echo "== step kernel (4096^2, 20000 steps in the background)"
( timeout -k 10 170 python bench.py --steps 20000 --warmup 50 --no-cpu --no-variants 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('step us', round(d['ms_per_step']*1e3,1))" ) &
sleep 6; sample; wait
echo "== GP step (2048^2, 512 points)"
( timeout -k 10 170 python tools/bench_gp.py --steps 1200 2>/dev/null | tail -n 1 | cut -c1-160 ) &
sleep 6; sample; wait
