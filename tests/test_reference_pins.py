"""Gap profiles and the YAML input contract against outputs of the REFERENCE'S OWN functions
(tests/golden/make_reference_pins.py compiles topography.py:38-170 and io.py:38-57, 100-452 out of the reference's source files
with `ast` -- no stand-in for the libraries those files import at module level -- and freezes what they return):

  topo_profiles.npz   x, y, [h, dh/dx, dh/dy] of every profile type on a 1-D and a 2-D grid, with and without `flip`
  io_sanitized.json   the sanitised dict (or the exception type) for every examples/config/*.yaml, every YAML document the
                      reference's tests hold as a string, and 47 synthetic inputs that walk the sanitiser's branches

Both the oracle (oracle/topography.py, oracle/config.py) and the product (gapflow_amd/topography.py, gapflow_amd/io.py) must
reproduce them BITWISE / key for key.  The GPU test sends the product's planes through the device layout and back."""
import contextlib
import io
import json
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
TOPO = np.load(os.path.join(GOLDEN, 'topo_profiles.npz'))
TOPO_CASES = json.loads(str(TOPO['cases']))
with open(os.path.join(GOLDEN, 'io_sanitized.json')) as f:
    IO_CASES = json.load(f)
REST = "\nproperties: {EOS: DH, shear: 0.1, bulk: 0.}\nnumerics: {}\noptions: {silent: True}"


def quiet(fn, *a):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a)


def plain(o):
    return json.loads(json.dumps(o, default=lambda x: x.item() if hasattr(x, 'item') else str(x)))


def differences(a, b, path=''):
    """Key-for-key, type-for-type comparison of two sanitised dicts (a: ours, b: the reference's)."""
    out = []
    if isinstance(a, dict) and isinstance(b, dict):
        for k in sorted(set(a) | set(b)):
            if k not in a:
                out.append(f'{path}/{k}: missing (reference: {b[k]!r})')
            elif k not in b:
                out.append(f'{path}/{k}: not in the reference ({a[k]!r})')
            else:
                out += differences(a[k], b[k], f'{path}/{k}')
    elif isinstance(a, list) and isinstance(b, list) and len(a) == len(b):
        for i, (x, y) in enumerate(zip(a, b)):
            out += differences(x, y, f'{path}[{i}]')
    elif a != b or type(a) is not type(b):
        out.append(f'{path}: {a!r} ({type(a).__name__}) vs reference {b!r} ({type(b).__name__})')
    return out


def readers():
    from oracle.config import read_yaml_input as oracle_reader
    from gapflow_amd.io import read_yaml_input as product_reader
    return {'oracle': oracle_reader, 'product': product_reader}


@pytest.mark.parametrize('who', ['oracle', 'product'])
@pytest.mark.parametrize('case', TOPO_CASES, ids=[c['key'] for c in TOPO_CASES])
def test_profiles_are_bitwise_the_reference_functions_output(case, who):
    d = quiet(readers()[who], io.StringIO(case['yaml'] + REST))
    if case['flip']:
        d['geometry']['flip'] = True
    np.random.seed(case['seed'])                # asperity, num > 1: topography.py:141-146 draws from the global generator
    if who == 'oracle':
        from oracle.topography import build_topography
        with np.errstate(all='ignore'):
            topo, x, y = build_topography(d['grid'], d['geometry'])
    else:
        from gapflow_amd.topography import Topography
        with np.errstate(all='ignore'):
            t = Topography(d['grid'], d['geometry'], d['properties'])
        topo, x, y = t.full, t.x, t.y
    key = case['key']
    assert np.array_equal(x, TOPO[key + '_x']) and np.array_equal(y, TOPO[key + '_y']), 'cell-centre coordinates'
    for c, name in enumerate(('h', 'dh/dx', 'dh/dy')):
        assert np.array_equal(topo[c], TOPO[key + '_topo'][c], equal_nan=True), \
            f'{name}: max |difference| {np.nanmax(np.abs(topo[c] - TOPO[key + "_topo"][c])):.3e}'


@pytest.mark.parametrize('who', ['oracle', 'product'])
@pytest.mark.parametrize('name', sorted(IO_CASES))
def test_sanitised_input_equals_the_reference_functions_output(name, who):
    entry, reader = IO_CASES[name], readers()[who]
    if 'raises' in entry:
        with pytest.raises(Exception) as err:
            quiet(reader, io.StringIO(entry['yaml']))
        assert type(err.value).__name__ == entry['raises'], f'{type(err.value).__name__}: {err.value}'
        return
    diff = differences(plain(quiet(reader, io.StringIO(entry['yaml']))), entry['sanitized'])
    assert not diff, '\n'.join(diff)


@pytest.mark.gpu
@pytest.mark.parametrize('case', [c for c in TOPO_CASES if c['key'].startswith('g2d')], ids=lambda c: c['key'])
def test_device_topography_is_bitwise_the_reference_functions_output(hiplib, case):
    """What the kernels read: the product's planes after upload into the padded device layout and download again."""
    from gapflow_amd import Problem, _lib
    from gapflow_amd.io import read_yaml_input
    d = quiet(read_yaml_input, io.StringIO(case['yaml'] + REST))
    if case['flip']:
        d['geometry']['flip'] = True
    np.random.seed(case['seed'])
    with np.errstate(all='ignore'):
        prob = quiet(Problem, d['options'], d['grid'], d['numerics'], d['properties'], d['geometry'])
    prob._sync_to_device()
    dev = prob._download(_lib.FIELD_TOPO, 3)
    assert np.array_equal(dev, TOPO[case['key'] + '_topo'], equal_nan=True)
