"""Counters of the step kernel for one library build, on the GPU box:

    python tools/pmc_step.py OUTDIR [--planes] [--lib path/to/lib.so]

Separate rocprofv3 --pmc passes (never combined with other traces) of `bench.py --steps 6 --warmup 2 --no-cpu
--no-variants`; prints the per-launch means for the step kernel and the derived figures (HBM bytes with the gfx950
FETCH_SIZE correction of MI355X_MICROARCH.md, VALU-active share per wave, wait share)."""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = [['FETCH_SIZE'], ['WRITE_SIZE'], ['SQ_ACTIVE_INST_VALU', 'SQ_INSTS_VALU', 'SQ_WAVES', 'SQ_BUSY_CYCLES'],
          ['GRBM_GUI_ACTIVE', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY']]


def main():
    out = os.path.abspath(sys.argv[1])
    env = dict(os.environ, TMPDIR='/tmp')
    if '--planes' in sys.argv:
        env['GPF_TOPO_PLANES'] = '1'
    if '--lib' in sys.argv:
        env['GPF_LIB_PATH'] = os.path.join(ROOT, sys.argv[sys.argv.index('--lib') + 1])
    os.makedirs(out, exist_ok=True)
    sums, counts = {}, {}
    for i, group in enumerate(PASSES):
        d = os.path.join(out, f'pmc{i}')
        with open(os.path.join(out, f'pmc{i}.log'), 'w') as f:
            subprocess.run(['rocprofv3', '--pmc'] + group + ['--kernel-trace', '--output-format', 'csv', '-d', d, '-o', 'pmc', '--',
                            'python3', 'bench.py', '--steps', '6', '--warmup', '2', '--no-cpu', '--no-variants'],
                           stdout=f, stderr=subprocess.STDOUT, cwd=ROOT, env=env, timeout=600)
        for fn in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            for row in csv.DictReader(open(fn)):
                name = row['Kernel_Name'].split('(')[0].replace('void ', '')
                key = (name, row['Counter_Name'])
                sums[key] = sums.get(key, 0.0) + float(row['Counter_Value'])
                counts.setdefault(key, set()).add(row['Dispatch_Id'])
        subprocess.run(['rm', '-rf', d])
    means = {}
    for (name, ctr), v in sums.items():
        means.setdefault(name, {})[ctr] = v / max(1, len(counts[(name, ctr)]))
    json.dump(means, open(os.path.join(out, 'pmc_means_per_launch.json'), 'w'), indent=1, sort_keys=True)
    for k, m in means.items():
        if 'k_step' not in k:
            continue
        d = dict(m)
        if 'FETCH_SIZE' in m and 'WRITE_SIZE' in m:
            d['hbm_bytes'] = (2 * m['FETCH_SIZE'] + m['WRITE_SIZE']) * 1024
        if 'SQ_WAVE_CYCLES' in m:
            d['valu_active_share_per_wave'] = m.get('SQ_ACTIVE_INST_VALU', 0) / m['SQ_WAVE_CYCLES']
            d['wait_share'] = m.get('SQ_WAIT_ANY', 0) / m['SQ_WAVE_CYCLES']
        print(k, json.dumps(d, indent=1, sort_keys=True))


if __name__ == '__main__':
    main()
