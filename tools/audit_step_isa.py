"""ISA audit of k_step2's hand-pipelined row loads (csrc/step2_kernel.hip), on the CPU (hipcc cross-compiles):

    python tools/audit_step_isa.py [--all]

The row loads are inline-asm `global_load_dwordx4`; hipcc treats their destination registers as written when the asm
statement ends, so it could legally copy, spill or reuse them before the data has landed (CDNA4 guide 5.7, item 1).  For
a few instantiations of the kernel this script compiles the kernel to assembly and checks that
  * between each row load and the hand-written `s_waitcnt vmcnt(N)` that guards its row (two loop bodies later, and from
    the prologue's loads up to the loop) no compiler instruction mentions a destination register,
  * and reports scratch accesses / compiler-visible vector loads inside the march loop (hipcc's `vmcnt` waits for them
    drain the pipeline: a speed matter, not a correctness one; the piezo-viscosity variants spill).
Exit status 0 = no register hazard."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'gapflow_amd', 'csrc')
VARIANTS = ['0, false, false, 1, 0', '0, false, false, -1, 0', '0, false, false, 1, 1', '0, false, false, -1, 1', '0, false, false, 1, 2',
            '0, false, false, 1, 3', '0, false, false, -1, 3', '2, false, false, 1, 3', '5, false, false, -1, 3', '0, true, false, 1, 0',
            '0, false, true, 1, 0', '5, true, true, -1, 0',
            # the equations of state that go through pow / exp / log (one wave per SIMD, fewer rows ahead: Step2Weight) -- their
            # closures decide the register allocation, so each heavy law is audited, one of them on the planes kernel
            '1, false, false, 1, 3', '3, false, false, 1, 0', '3, false, false, -1, 3', '6, false, false, -1, 1']


def regs_of(tok):
    out = set()
    for m in re.finditer(r'v\[(\d+):(\d+)\]', tok):
        out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r'\bv(\d+)\b', tok):
        out.add(int(m.group(1)))
    return out


def loop_span(lines, start):
    name = re.match(r'^(\.LBB\d+_\d+):', lines[start]).group(1)[2:]
    end = max([i for i, l in enumerate(lines) if re.search(r'in Loop: Header=' + name + r'\b', l)] + [start])
    while not lines[end + 1].startswith(('.LBB', '; %bb.')) and 'Lfunc_end' not in lines[end + 1]:     # the next block's header
        end += 1
    return end


def asm_events(block):
    """(index, kind, registers) of a block: 'load' / 'wait' inside inline asm, 'ins' for compiler instructions."""
    events, in_asm = [], False
    for i, l in enumerate(block):
        t = l.strip()
        if t.startswith(';;#ASMSTART'):
            in_asm = True
        elif t.startswith(';;#ASMEND'):
            in_asm = False
        elif in_asm and t.startswith('global_load_dwordx4'):
            events.append((i, 'load', regs_of(t.split(',')[0])))
        elif in_asm and t.startswith('s_waitcnt'):
            events.append((i, 'wait', set()))
        elif not in_asm and t and t[0] not in ';.':
            events.append((i, 'ins', regs_of(t)))
    return events


def audit(lines):
    """Every march loop of the function (the kernel holds one per compile-time flavour of the march, entered from a common
    prologue): returns (row-load groups per loop, register hazards, compiler-visible memory accesses inside the loops)."""
    headers = [i for i, l in enumerate(lines) if 'Inner Loop Header' in l]
    loops = [(h, loop_span(lines, h)) for h in headers]
    loops = [(h, e) for h, e in loops if any(k == 'load' for _, k, _ in asm_events(lines[h:e + 1]))]
    problems, slow, ngroups = [], [], []
    first = loops[0][0]
    # the prologue's loads stay in flight until a loop consumes them: nothing before the loops may touch their registers
    inflight = set()
    for i, kind, regs in asm_events(lines[:first]):
        if kind == 'load':
            inflight |= regs
        elif kind == 'ins' and regs & inflight:
            problems.append(f'prologue line {i}: {lines[i].strip()}')
    prev_end = first - 1
    for h, e in loops:
        # straight-line code between the previous loop and this one (this loop's pre-header, the other loop's exit path)
        for i, kind, regs in asm_events(lines[prev_end + 1:h]):
            if kind == 'wait':
                break                       # the post-loop drain: nothing is in flight beyond it
            if kind == 'ins' and regs & inflight:
                problems.append(f'line {prev_end + 1 + i} (between loops): {lines[prev_end + 1 + i].strip()}')
        body = lines[h:e + 1]
        events = asm_events(body)
        for i, kind, regs in events:
            t = body[i].strip()
            if kind == 'ins' and (t.startswith('scratch_') or t.startswith('global_load') or t.startswith('buffer_load')):
                slow.append(f'loop line {i}: compiler-visible memory access {t}')
        groups, cur = [], None
        for idx, kind, regs in events:
            if kind == 'load':
                cur = cur or []
                cur.append((idx, regs))
            elif kind == 'wait' and cur:
                groups.append(cur)
                cur = None
        waits = [next(idx for idx, kind, _ in events if kind == 'wait' and idx > g[0][0]) for g in groups]
        n = len(groups)
        ngroups.append(n)
        bufregs = set()
        for k, loads in enumerate(groups):
            target = waits[(k + n - 1) % n]         # a row is consumed n - 1 = AHEAD loop bodies after its request
            for gidx, regs in loads:
                bufregs |= regs
                span = [(i, r) for i, kind, r in events if kind == 'ins' and (gidx < i < target if target > gidx else (i > gidx or i < target))]
                for i, r in span:
                    if r & regs:
                        problems.append(f'loop line {i}: touches in-flight v{sorted(r & regs)}: {body[i].strip()}')
        # after the loop: the last (dummy) requests are in flight until the hand-written drain
        for i, kind, regs in asm_events(lines[e + 1:e + 400]):
            if kind == 'wait':
                break
            if kind == 'ins' and regs & bufregs and not lines[e + 1 + i].strip().startswith(('s_', ';')):
                problems.append(f'line {e + 1 + i} (loop exit, before the drain): {lines[e + 1 + i].strip()}')
        inflight |= bufregs
        prev_end = e
    return ngroups, problems, slow


def main():
    everything = '--all' in sys.argv        # every instantiation of the shipped library (compiles api.hip: minutes)
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, 't.hip')
        with open(src, 'w') as f:
            if everything:
                f.write('#include "api.hip"\n')
            else:
                f.write('#include <hip/hip_runtime.h>\n#include "step_kernel.hip"\n#include "aux_kernels.hip"\n#include "step2_kernel.hip"\nusing namespace gpf;\n')
                for v in VARIANTS:
                    f.write(f'template __global__ void gpf::k_step2<{v}>(const Step2Args, const Phys);\n')
        asm = os.path.join(tmp, 't.s')
        subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=fast', '-I', CSRC, '-S',
                        '--cuda-device-only', src, '-o', asm] + os.environ.get('GPF_EXTRA_FLAGS', '').split(), check=True, capture_output=True)
        text = open(asm).read().split('\n')
    bad = 0
    starts = [i for i, l in enumerate(text) if re.match(r'^_ZN3gpf7k_step2I.*:', l)]
    for a in starts:
        b = next(i for i in range(a, len(text)) if text[i].startswith('.Lfunc_end'))
        n, problems, slow = audit(text[a:b])
        print(text[a].split(':')[0], f'{"+".join(map(str, n))} row-load groups in the march loop(s),', 'registers safe' if not problems else f'{len(problems)} REGISTER HAZARDS',
              '' if not slow else f'; {len(slow)} compiler-visible memory accesses in the loop (spills: slow, not wrong)')
        for p in problems[:10]:
            print('   ', p)
        bad += len(problems)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
