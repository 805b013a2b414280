"""The hand-pipelined row loads of k_step2 hide their register writes from hipcc (inline asm): the compiler could legally
copy, spill or reuse a destination register before the data has landed.  tools/audit_step_isa.py compiles a set of kernel
instantiations to gfx950 assembly (hipcc cross-compiles without a GPU) and checks that no compiler instruction touches a
destination register between its load and the hand-written wait that guards it.  Run with the CPU suite: it is the
build-time guard of that contract."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which('hipcc') is None and not os.path.exists('/opt/rocm/bin/hipcc'), reason='needs hipcc')
def test_no_compiler_instruction_touches_a_row_buffer_in_flight():
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'audit_step_isa.py')], capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith('_ZN3gpf7k_step2')]
    assert len(lines) >= 8 and all('registers safe' in l for l in lines), res.stdout[-3000:]
    # the benchmarked instantiations (constant viscosity, no slip-length field) keep the march free of spills
    for l in lines:
        if 'ELb0ELb0E' in l:
            assert 'compiler-visible memory accesses' not in l, l


def test_hot_kernels_of_the_shipped_build_do_not_spill():
    """`python -m gapflow_amd.build` keeps hipcc's resource report: the fused variance kernel (16 waves per workgroup: 128
    registers per wave is all there is) and the benchmarked step kernels must not use scratch memory -- the one time the
    variance kernel did, it ran 3.5x slower."""
    import re
    path = os.path.join(ROOT, 'gapflow_amd', 'lib', 'resource_usage.txt')
    if not os.path.exists(path):
        pytest.skip('no resource report (library not built here)')
    text = open(path).read()
    seen = 0
    for block in re.split(r'remark: Function Name: ', text)[1:]:
        name = block.split()[0]
        if 'k_gp_var_fused' in name or re.search(r'k_step2ILi0ELb0ELb0E', name):
            seen += 1
            assert int(re.search(r'ScratchSize \[bytes/lane\]: (\d+)', block).group(1)) == 0, name
    assert seen >= 12
