"""A/B timing of step-kernel builds on ONE box (box-to-box variation of the pool is +-7 %, so only alternating runs on
the same device compare):

    python tools/ab_step.py [--rounds R] [--steps K] name=path/to/lib.so[,ENV=VAL...] ...

Runs bench.py --no-cpu once per variant and round (alternating), with GPF_LIB_PATH set, and prints per variant the
median kernel / whole-step time of the x-only-gap workload (journal bearing) and of the 2-D-gap variant (asperity)."""
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = sys.argv[1:]
    rounds, steps = 3, 100
    while args and args[0].startswith('--'):
        if args[0] == '--rounds':
            rounds = int(args[1])
        elif args[0] == '--steps':
            steps = int(args[1])
        args = args[2:]
    variants = []
    for a in args:
        name, rest = a.split('=', 1)
        parts = rest.split(',')
        env = dict(p.split('=', 1) for p in parts[1:])
        variants.append((name, parts[0], env))
    res = {v[0]: [] for v in variants}
    for r in range(rounds):
        for name, lib, env in variants:
            e = dict(os.environ, **env)
            if lib:
                e['GPF_LIB_PATH'] = os.path.join(ROOT, lib)
            out = subprocess.run([sys.executable, 'bench.py', '--no-cpu', '--no-gp', '--steps', str(steps)], cwd=ROOT, env=e,
                                 capture_output=True, text=True, timeout=600)
            if out.returncode != 0:
                print(name, 'FAILED', out.stderr[-500:], flush=True)
                continue
            d = json.loads(out.stdout.strip().splitlines()[-1])
            v = d['variants']['asperity_gap_2d_V0.05']
            res[name].append((d['roofline']['kernel_ms'], d['ms_per_step'], v['kernel_ms'], v['ms_per_step']))
            ceil = d['roofline'].get('stream_ceiling_GBps', 0.0), v['roofline'].get('stream_ceiling_GBps', 0.0)
            print(f"round {r} {name:>12}: [box streams {ceil[0]:.0f} / {ceil[1]:.0f} GB/s] line kernel {d['roofline']['kernel_ms']*1e3:7.1f} step {d['ms_per_step']*1e3:7.1f} us | "
                  f"2-D gap kernel {v['kernel_ms']*1e3:7.1f} step {v['ms_per_step']*1e3:7.1f} us", flush=True)
    print('--- medians (us) ---')
    for name, rows in res.items():
        if rows:
            med = [statistics.median(x[i] for x in rows) * 1e3 for i in range(4)]
            print(f"{name:>12}: line kernel {med[0]:7.1f} step {med[1]:7.1f} | 2-D gap kernel {med[2]:7.1f} step {med[3]:7.1f}")


if __name__ == '__main__':
    main()
