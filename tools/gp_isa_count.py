"""Static instruction mix of the posterior-mean kernel's inner loop (one Matern-3/2 evaluation per iteration and lane), from the
gfx950 assembly hipcc produces (CPU, cross-compile):

    python tools/gp_isa_count.py [out.json]

For each instantiation k_gp_mean<D, M, WITH_GRAD> the innermost loop (the one over the training points of a chunk in LDS) is
located and its instructions are classified: fp64 VALU (v_*_f64, 4 cycles per wave64 instruction on a SIMD), other VALU,
LDS, scalar.  bench.py turns `fp64_per_evaluation` into the issue-level fraction of the GP variants: evaluations/s x fp64
instructions per evaluation / (256 CUs x 4 SIMDs x 16 lanes x clock), beside SURVEY 8(d)'s flop-count fraction, which prices
exp and sqrt at one flop each."""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'gapflow_amd', 'csrc')
VARIANTS = ['2, 1, false', '2, 1, true', '3, 2, false']


def innermost_loops(lines):
    """(start, end) of loops that contain no other loop header."""
    heads = []
    for i, l in enumerate(lines):
        if 'Loop Header' in l:          # on the label's own line, or on the comment line that follows a label with a 'Parent Loop' note
            heads.append(i if re.match(r'^\.LBB\d+_\d+:', l) else i - 1)
    out = []
    for h in heads:
        m = re.match(r'^\.(LBB\d+_\d+):', lines[h])
        if not m:
            continue
        name = m.group(1)
        ends = [i for i, l in enumerate(lines) if re.search(r's_cbranch\w* \.' + name + r'\b', l)]
        if ends and not any(h < k <= ends[-1] for k in heads):
            out.append((h, ends[-1]))
    return out


def classify(body):
    c = {'fp64_valu': 0, 'other_valu': 0, 'lds': 0, 'scalar': 0, 'other': 0}
    for l in body:
        t = l.strip().split()
        if not t or t[0][0] in ';.' or t[0].endswith(':'):
            continue
        op = t[0]
        if op.startswith('v_') and ('_f64' in op or op.startswith(('v_ldexp_f64', 'v_rsq_f64', 'v_rcp_f64', 'v_cvt_i32_f64', 'v_rndne_f64'))):
            c['fp64_valu'] += 1
        elif op.startswith('v_'):
            c['other_valu'] += 1
        elif op.startswith('ds_'):
            c['lds'] += 1
        elif op.startswith('s_'):
            c['scalar'] += 1
        else:
            c['other'] += 1
    return c


def main():
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, 't.hip')
        with open(src, 'w') as f:
            f.write('#include <hip/hip_runtime.h>\n#include <string>\n#include "gp_kernels.hip"\nusing namespace gpf;\n')
            for v in VARIANTS:
                f.write(f'template __global__ void gpf::k_gp_mean<{v}>(const GpModelDev, const GpFieldArgs);\n')
        asm = os.path.join(tmp, 't.s')
        subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=fast', '-I', CSRC, '-S',
                        '--cuda-device-only', src, '-o', asm], check=True, capture_output=True)
        text = open(asm).read().split('\n')
    result = {}
    starts = [i for i, l in enumerate(text) if re.match(r'^_ZN3gpf9k_gp_meanI.*:', l)]
    for a in starts:
        b = next(i for i in range(a, len(text)) if text[i].startswith('.Lfunc_end'))
        fn = text[a:b]
        loops = innermost_loops(fn)
        # the evaluation loop is the innermost loop with the most fp64 instructions
        best = max(loops, key=lambda se: classify(fn[se[0]:se[1] + 1])['fp64_valu'])
        body = fn[best[0]:best[1] + 1]
        c = classify(body)
        # the compiler may unroll: evaluations per iteration = number of v_rsq_f64 (one square root per evaluation)
        per_iter = max(1, sum(1 for l in body if l.strip().startswith('v_rsq_f64')))
        name = re.sub(r'^_ZN3gpf9k_gp_meanI', 'k_gp_mean<', text[a].split(':')[0])
        key = {'Li2ELi1ELb0E': 'k_gp_mean<2, 1, false>', 'Li2ELi1ELb1E': 'k_gp_mean<2, 1, true>', 'Li3ELi2ELb0E': 'k_gp_mean<3, 2, false>'}
        label = next((v for k, v in key.items() if k in text[a]), name)
        result[label] = {'evaluations_per_loop_iteration': per_iter,
                         'fp64_per_evaluation': c['fp64_valu'] / per_iter, 'other_valu_per_evaluation': c['other_valu'] / per_iter,
                         'lds_per_evaluation': c['lds'] / per_iter, 'scalar_per_evaluation': c['scalar'] / per_iter}
        print(label, result[label])
    if len(sys.argv) > 1:
        json.dump(result, open(sys.argv[1], 'w'), indent=1, sort_keys=True)


if __name__ == '__main__':
    sys.exit(main())
