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
