"""Host-side logic of gapflow_amd and the C-ABI surface, without a GPU (no compute calls)."""
import ctypes
import io
import os
import re
import types

import numpy as np
import pytest

from helpers import STEP_CASES, load_case, input_dict
from gapflow_amd import io as gio
from gapflow_amd import topography as gtopo
from oracle import config as ocfg
from oracle.topography import build_topography

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _same(a, b, path=''):
    assert type(a) is type(b) or (isinstance(a, (int, float)) and isinstance(b, (int, float))), f'{path}: {a!r} vs {b!r}'
    if isinstance(a, dict):
        assert list(a.keys()) == list(b.keys()) or set(a) == set(b), f'{path}: keys {set(a) ^ set(b)}'
        for k in a:
            _same(a[k], b[k], f'{path}.{k}')
    else:
        assert a == b, f'{path}: {a!r} != {b!r}'


@pytest.mark.parametrize('name', STEP_CASES)
def test_sanitised_input_matches_oracle(name, capsys):
    """gapflow_amd.io (product) and oracle.config (checker) sanitise every fixture YAML identically."""
    _, yaml_text, meta = load_case(name)
    a = gio.read_yaml_input(io.StringIO(yaml_text))
    b = ocfg.read_yaml_input(io.StringIO(yaml_text))
    _same(a, b)


def test_input_defaults_and_quirks():
    d = gio.read_yaml_input(io.StringIO("""
options: {}
grid: {Nx: 10, Lx: 1., dy: 2.}
geometry: {type: journal, hmin: 1., hmax: 2.}
numerics: {}
properties: {shear: 1., EOS: DH, T0: 300.}
gp: {press: {}, shear: {active_dims: {x: [0, 1]}}}
db: {init_width: 3}
"""))
    assert d['options'] == {'output': 'example', 'write_freq': 1000, 'use_tstamp': True, 'silent': False}
    g = d['grid']
    assert (g['Nx'], g['dx'], g['Ny'], g['dy'], g['Ly'], g['dim']) == (10, 0.1, 1, 2., 2., 1)
    assert g['bc_xE_P'] == [True] * 3 and 'bc_xE_D_val' not in g
    assert d['numerics'] == {'tol': 1e-6, 'max_it': 1000, 'dt': 3e-10, 'adaptive': False, 'CFL': 0.5, 'MC_order': 1}
    p = d['properties']
    assert (p['rho0'], p['P0'], p['C1'], p['C2'], p['bulk']) == (877.7007, 101325., 3.5e10, 1.23, -1.)
    assert 'T0' not in p                                   # unknown keys are dropped silently
    assert d['gp']['press']['active_dims'] == [0, 3] and d['gp']['shear']['active_dims_x'] == [0, 1]
    assert d['gp']['shear']['active_dims_y'] == [0, 2, 3] and d['gp']['press']['active_learning'] is True
    assert d['db']['init_seed'] == 3                        # io.py:439 reads the init_width key
    assert d['md'] is None


def test_input_errors():
    with pytest.raises(IOError):
        gio.sanitize_grid({'Nx': 10, 'Ny': 1, 'dy': 1.})            # neither dx nor Lx
    with pytest.raises(IOError):
        gio.sanitize_grid({'Nx': 4, 'dx': 1., 'Ny': 4, 'dy': 1., 'yS': ['D', 'N', 'N'], 'yN': ['D', 'N', 'N']})   # yS_D mandatory
    with pytest.raises(AssertionError):
        gio.sanitize_grid({'Nx': 4, 'dx': 1., 'Ny': 1, 'dy': 1., 'xE': ['P', 'P', 'P'], 'xW': ['D', 'N', 'N']})
    with pytest.raises(IOError):
        gio.sanitize_geometry({'type': 'sphere'})
    with pytest.raises(IOError):
        gio.sanitize_properties({'EOS': 'DH'})                      # no shear viscosity
    with pytest.raises(IOError):
        gio.sanitize_properties({'shear': 1., 'EOS': 'ideal'})


@pytest.mark.parametrize('name', STEP_CASES)
def test_topography_matches_fixture(name):
    fx, yaml_text, meta = load_case(name)
    d = input_dict(yaml_text, meta, ocfg.read_yaml_input)
    topo = gtopo.Topography(d['grid'], d['geometry'], d['properties'])
    np.testing.assert_array_equal(topo.full, fx['topo'])
    ref, xx, yy = build_topography(d['grid'], d['geometry'])
    np.testing.assert_array_equal(topo.x, xx)
    np.testing.assert_array_equal(topo.y, yy)


def test_topography_profiles_and_gradient_stencil():
    grid = {'Nx': 32, 'Ny': 6, 'Lx': 2., 'Ly': 1., 'dx': 2. / 32, 'dy': 1. / 6}
    prop = {'elastic': {'enabled': False}}
    cd = gtopo.Topography(grid, {'type': 'cdc', 'hmin': 1., 'hmax': 2., 'b': 0.25, 'flip': False}, prop)
    assert cd.h.min() == 1. and cd.h.max() == 2.
    assert set(np.unique(np.sign(cd.dh_dx))) == {-1., 0., 1.}
    par = gtopo.Topography(grid, {'type': 'parabolic', 'hmin': 1., 'hmax': 2., 'flip': False}, prop)
    # h setter re-differentiates with np.gradient (topography.py:273-295): second order in the interior
    par.h = par.h.copy()
    xx = par.x
    exact = 2 * (4. / grid['Lx']**2) * (xx - grid['Lx'] / 2.)
    np.testing.assert_allclose(par.dh_dx[1:-1], exact[1:-1], atol=1e-12)
    el = gtopo.Topography(grid, {'type': 'journal', 'CR': 1e-2, 'eps': 0.5, 'flip': False}, {'elastic': {'enabled': True}})
    assert el.elastic and np.all(el.deformation == 0)       # the device takes over after the first elastic update


def test_elastic_green_functions_match_the_oracle_restatement():
    """gapflow_amd/elastic.py (product, host set-up) against oracle/elastic.py for every half-space variant, incl. the
    1-D line-contact rule (topography.py:366-380).  Both restate published forms; the oracle is pinned to analytic
    solutions in tests/test_oracle_elastic.py."""
    import warnings
    from gapflow_amd.elastic import ElasticDeformation
    from oracle.elastic import ElasticDeformation as Oracle

    def grid(Nx, Ny, perX, perY):
        g = {'Nx': Nx, 'Ny': Ny, 'Lx': 3e-4, 'Ly': 2e-4, 'dx': 3e-4 / Nx, 'dy': 2e-4 / Ny}
        for side, per in (('xE', perX), ('xW', perX), ('yS', perY), ('yN', perY)):
            g[f'bc_{side}_P'] = [per] * 3
        return g
    for Nx, Ny, perX, perY in ((12, 8, True, True), (12, 8, False, False), (12, 8, False, True), (12, 8, True, False), (20, 1, False, True)):
        g = grid(Nx, Ny, perX, perY)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            a, b = ElasticDeformation(50e9, 0.3, 0.1, g, 4), Oracle(50e9, 0.3, 0.1, g, 4)
        assert a.periodicity == b.periodicity and a.shape_fft == b.pad
        np.testing.assert_allclose(a.greens, b.greens, rtol=1e-13, atol=1e-30)
        assert a.area_per_pt == b.area_per_pt


def test_edge_rules_resolution():
    """problem.py:676-768 quirks: low-y ghost takes the yN value, high-y ghost the yS value."""
    from gapflow_amd.problem import Problem
    g = gio.sanitize_grid({'Nx': 4, 'dx': 1., 'Ny': 4, 'dy': 1., 'xE': ['D', 'N', 'N'], 'xW': ['D', 'N', 'N'],
                           'xE_D': 2., 'xW_D': 3., 'yS': ['D', 'N', 'N'], 'yN': ['D', 'N', 'N'], 'yS_D': 5., 'yN_D': 7.})
    rules, values = Problem._edge_rules(types.SimpleNamespace(grid=g))
    assert rules == [[1, 2, 2]] * 4
    assert values == [3., 2., 7., 5.]      # ix=0 <- xW, ix=Nx+1 <- xE, iy=0 <- yN (sic), iy=Ny+1 <- yS (sic)
    g = gio.sanitize_grid({'Nx': 4, 'dx': 1., 'Ny': 4, 'dy': 1.})
    rules, values = Problem._edge_rules(types.SimpleNamespace(grid=g))
    assert rules == [[0, 0, 0]] * 4
    g = gio.sanitize_grid({'Nx': 4, 'dx': 1., 'Ny': 1, 'dy': 1., 'xE': ['D', 'N', 'N'], 'xW': ['D', 'D', 'N']})
    with pytest.raises(NotImplementedError):
        Problem._edge_rules(types.SimpleNamespace(grid=g))


# ---------------------------------------------------------------------------------------------
# C ABI
# ---------------------------------------------------------------------------------------------

def _declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'gapflow_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(gpf_[a-z_0-9]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from gapflow_amd import _lib
    from gapflow_amd.build import build_library
    lib = ctypes.CDLL(build_library())
    names = _declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/gapflow_hip.h but not exported'
    assert sorted(_lib.SIGNATURES) == names, 'ctypes signatures out of sync with the header'


def test_struct_layouts_match_header():
    from gapflow_amd import _lib
    # gpf_config: 2 i32, 6 f64, i32(+pad), 8 f64, i32(+pad), 4 f64, 12 i32, 4 f64, 2 i32, i32(+pad), 3 f64, i64, 2 i32
    assert ctypes.sizeof(_lib.GpfConfig) == 8 + 48 + 8 + 64 + 8 + 32 + 48 + 32 + 8 + 8 + 24 + 8 + 8 + 8 + 32
    assert ctypes.sizeof(_lib.GpfScalars) == 8 + 8 * 8 + 8
    assert _lib.GpfConfig.bc_rule.offset == 8 + 48 + 8 + 64 + 8 + 32


@pytest.mark.skipif(os.path.exists('/dev/kfd'), reason='a GPU is present: the no-device error path cannot be seen')
def test_fails_loudly_without_a_device():
    """No CPU fallback: constructing a Problem or calling an operator without an MI355X raises."""
    from gapflow_amd import Problem, integrate, _lib
    with pytest.raises(_lib.GapflowHipError):
        Problem.from_string("""
options: {silent: True}
grid: {Nx: 8, dx: 1.e-5, Ny: 1, dy: 1.}
geometry: {type: journal, CR: 1.e-2, eps: 0.5, U: 0.1}
numerics: {}
properties: {shear: 0.1, bulk: 0., EOS: DH}
""")
    q = np.ones((3, 4, 4))
    with pytest.raises(_lib.GapflowHipError):
        integrate.predictor_corrector(q, q[0], q, 1)
    with pytest.raises(_lib.GapflowHipError):
        integrate.source(q, q, q, np.ones((6, 4, 4)), np.ones((6, 4, 4)))
    lib = _lib.load()
    a = (ctypes.c_double * 48)()
    assert lib.gpf_predictor_corrector(4, 4, a, a, a, 1, a, a) == -3       # GPF_ERR_NO_DEVICE
    assert b'no HIP device' in lib.gpf_last_error()


def test_reference_module_paths_exist():
    """A script written against the reference imports these paths (examples/slip_1d_lj_mock.py:5-8,
    GaPFlow/__init__.py:36-37, models/__init__.py:24-26); LAMMPS runners are refused by name."""
    import gapflow_amd
    from gapflow_amd.problem import Problem
    from gapflow_amd.io import read_yaml_input  # noqa: F401
    from gapflow_amd.db import Database
    from gapflow_amd.md import Mock  # noqa: F401
    from gapflow_amd.models import Pressure, WallStress, BulkStress  # noqa: F401
    from gapflow_amd.integrate import predictor_corrector, source  # noqa: F401
    assert gapflow_amd.Problem is Problem and gapflow_amd.Database is Database
    import gapflow_amd.md as md
    with pytest.raises(NotImplementedError):
        md.LennardJones


def test_argument_errors_are_reported_not_crashed():
    """Error convention of the C ABI (0 or a negative gpf_status, message in gpf_last_error): bad arguments are
    rejected before any device work, so this runs without a GPU."""
    from gapflow_amd import _lib
    lib = _lib.load()
    a = (ctypes.c_double * 64)()
    P = ctypes.c_void_p
    INVALID = -1
    assert lib.gpf_predictor_corrector(0, 4, a, a, a, 1, a, a) == INVALID
    assert lib.gpf_source(4, 4, a, a, a, a, None, a) == INVALID
    assert lib.gpf_viscous_stress(4, a, a, None, None, a, a, 0.1, 0., 0., 0, None, None, None) == INVALID     # no output
    assert b'no output' in lib.gpf_last_error()
    assert lib.gpf_viscous_stress(0, a, a, None, None, a, a, 0.1, 0., 0., 0, a, None, None) == INVALID
    assert lib.gpf_eos(99, a, 4, a, a, None) == INVALID and b'unknown equation of state' in lib.gpf_last_error()
    assert lib.gpf_eos(0, a, 4, a, None, None) == INVALID
    assert lib.gpf_viscosity(7, 0, a, 0.1, 4, a, None, None, 0., 0., a) == INVALID
    assert lib.gpf_viscosity(0, 99, a, 0.1, 4, a, None, None, 0., 0., a) == INVALID and b'unknown law' in lib.gpf_last_error()
    # handle-taking entry points refuse a null handle
    for name, args in (('gpf_step_p2p', (None, 1, 0)), ('gpf_p2p_export', (None, a, 64)), ('gpf_stage_message', (None,)),
                       ('gpf_close_step_local', (None,)), ('gpf_step_local', (None, 0))):
        assert getattr(lib, name)(*args) < 0, name
    assert lib.gpf_destroy(None) == 0


def test_grid_beyond_32_bit_offsets_is_refused():
    from gapflow_amd import _lib
    lib = _lib.load()
    cfg = _lib.GpfConfig()
    cfg.Nx, cfg.Ny, cfg.dx, cfg.dy = 50000, 50000, 1e-5, 1e-5
    h = ctypes.c_void_p()
    assert lib.gpf_create(ctypes.byref(cfg), ctypes.byref(h)) == -1 and b'2e9 cells' in lib.gpf_last_error()
