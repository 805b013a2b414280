"""Profile set of the GP configuration (BASELINE.json configs[3]) on the GPU box:

    python tools/collect_gp_profile.py gpurun_out/profile_gp

  bench_gp.json              python tools/bench_gp.py
  kernel_stats.csv           rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/bench_gp.py --steps 3
  pmc_means_per_launch.json  per-kernel counter means (separate --pmc passes) with
                             valu_busy_frac = SQ_ACTIVE_INST_VALU * 4 / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs)   [wave64: 4 cycles per issue]
                             mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs ... / (GRBM_GUI_ACTIVE / 8)"""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = [['SQ_ACTIVE_INST_VALU', 'SQ_INSTS_VALU', 'SQ_BUSY_CYCLES', 'SQ_WAVES'], ['SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_INSTS_VALU_MFMA_MOPS_F64'],
          ['SQ_INSTS_MFMA', 'SQ_BUSY_CU_CYCLES'], ['GRBM_GUI_ACTIVE', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY']]


def main():
    out = os.path.abspath(sys.argv[1])
    os.makedirs(out, exist_ok=True)
    env = dict(os.environ, TMPDIR='/tmp')
    with open(os.path.join(out, 'bench_gp.json'), 'w') as f:
        subprocess.run([sys.executable, 'tools/bench_gp.py'], stdout=f, stderr=subprocess.DEVNULL, cwd=ROOT, env=env, timeout=600, check=True)
    kt = os.path.join(out, 'kt')
    subprocess.run(['rocprofv3', '--kernel-trace', '--stats', '--output-format', 'csv', '-d', kt, '-o', 'kt', '--',
                    'python3', 'tools/bench_gp.py', '--steps', '3'], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=ROOT, env=env, timeout=600)
    for fn in glob.glob(os.path.join(kt, '**', '*kernel_stats.csv'), recursive=True):
        os.replace(fn, os.path.join(out, 'kernel_stats.csv'))
    sums, counts = {}, {}
    for i, group in enumerate(PASSES):
        d = os.path.join(out, f'pmc{i}')
        subprocess.run(['rocprofv3', '--pmc'] + group + ['--kernel-trace', '--output-format', 'csv', '-d', d, '-o', 'pmc', '--',
                        'python3', 'tools/bench_gp.py', '--steps', '1'], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=ROOT, env=env, timeout=600)
        for fn in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            for row in csv.DictReader(open(fn)):
                name = row['Kernel_Name'].split('(')[0].replace('void ', '')
                if name.startswith('Cijk_'):
                    name = 'rocBLAS GEMM ' + name[:48]
                key = (name, row['Counter_Name'])
                sums[key] = sums.get(key, 0.0) + float(row['Counter_Value'])
                counts.setdefault(key, set()).add(row['Dispatch_Id'])
    means = {}
    for (name, ctr), v in sums.items():
        means.setdefault(name, {})[ctr] = v / max(1, len(counts[(name, ctr)]))
    for name, m in means.items():
        if m.get('GRBM_GUI_ACTIVE'):
            per_xcd = m['GRBM_GUI_ACTIVE'] / 8.0
            if 'SQ_ACTIVE_INST_VALU' in m:
                m['valu_busy_frac'] = m['SQ_ACTIVE_INST_VALU'] * 4.0 / 1024.0 / per_xcd
            if 'SQ_VALU_MFMA_BUSY_CYCLES' in m:
                m['mfma_busy_frac'] = m['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0 / per_xcd
    json.dump(means, open(os.path.join(out, 'pmc_means_per_launch.json'), 'w'), indent=1, sort_keys=True)
    for d in glob.glob(os.path.join(out, 'pmc[0-9]')) + [kt]:
        subprocess.run(['rm', '-rf', d])
    for k in sorted(means):
        if 'k_gp' in k or 'GEMM' in k:
            print(k[:60], {a: round(b, 3) for a, b in means[k].items() if a.endswith('_frac')})


if __name__ == '__main__':
    main()
