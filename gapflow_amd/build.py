"""Builds libgapflow_hip.so (gfx950 only) in-tree with hipcc.

    python -m gapflow_amd.build [--force] [--no-strict]

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels to the
GPU box with the working tree.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, 'libgapflow_hip.so')
SOURCES = ['api.hip']
DEPS = sorted(f for f in os.listdir(CSRC) if f.endswith(('.hip', '.hpp', '.inc'))) + [os.path.join('..', '..', 'include', 'gapflow_hip.h')]
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-Wno-unused-value',
         '-ffp-contract=fast']
LIBS = ['-ldl']     # rocSOLVER / rocBLAS are dlopen'ed by the GP entry points (csrc/gp_kernels.hip)


STRICT_LIB = os.path.join(LIBDIR, 'variants', 'strict.so')
STRICT_FLAGS = ['-DGPF_STRICT_ATOMICS', '-DGPF_ONLY_EOS_DH']


def is_stale(lib=None):
    lib = lib or LIB
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.exists(os.path.join(CSRC, d)) and os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build_strict_variant(force=False, verbose=False):
    """The memory-model cross-check build (aux_kernels.hip: release / acquire orders on the in-launch hand-offs instead of
    the write-through + drained-counter form), Dowson-Higginson kernels only: half a minute of compile time.
    tests/test_gpu_extras.py compares its fields with the default library's bit for bit."""
    if not force and not is_stale(STRICT_LIB):
        if verbose:
            print('up to date, reused', STRICT_LIB)
        return STRICT_LIB
    return build_library(out=STRICT_LIB, extra_flags=STRICT_FLAGS, verbose=verbose)


def build_library(force=False, verbose=False, out=None, extra_flags=()):
    """Compile csrc/*.hip into lib/libgapflow_hip.so; returns the library path.

    `out` / `extra_flags` build an A/B variant next to it (lib/variants/<name>.so, selected at run time with
    GPF_LIB_PATH); the default library is rebuilt only when a source is newer (or with force=True)."""
    lib = out or LIB
    if out is None and not force and not is_stale():
        if verbose:
            print('up to date, reused', LIB)
        return LIB
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    extra = os.environ.get('GPF_EXTRA_FLAGS', '').split() + list(extra_flags)       # diagnostics, e.g. -DGPF_STUB_CLOSURE
    cmd = [HIPCC] + FLAGS + extra + ['-Rpass-analysis=kernel-resource-usage', '-o', lib] + SOURCES + LIBS
    res = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
    with open(os.path.splitext(lib)[0] + '.resource_usage.txt' if out else os.path.join(LIBDIR, 'resource_usage.txt'), 'w') as f:
        f.write(res.stderr)
    if res.returncode != 0:
        errs = [ln for ln in res.stderr.splitlines() if 'remark:' not in ln]
        first = [i for i, ln in enumerate(errs) if ' error' in ln]        # the errors themselves (warnings follow them by the hundred)
        shown = [ln for i in first[:8] for ln in errs[i:i + 6]] + ['...'] + errs[-10:]
        raise RuntimeError('hipcc failed:\n' + '\n'.join(shown))
    if verbose:
        print('compiled', lib)
    return lib


if __name__ == '__main__':
    if '--variant' in sys.argv:         # python -m gapflow_amd.build --variant NAME [flags...]
        i = sys.argv.index('--variant')
        build_library(out=os.path.join(LIBDIR, 'variants', sys.argv[i + 1] + '.so'), extra_flags=sys.argv[i + 2:], verbose=True)
    else:
        build_library(force='--force' in sys.argv, verbose=True)
        if '--no-strict' not in sys.argv:       # keep the cross-check variant in step with the sources (its ABI must match)
            build_strict_variant(force='--force' in sys.argv, verbose=True)
