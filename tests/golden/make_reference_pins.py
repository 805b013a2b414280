#!/usr/bin/env python3
"""Pin the gap profiles and the YAML input contract with the reference's OWN functions.

Runs ONLY in the build container (it reads /root/reference); what it writes is data: inputs + expected outputs.

`GaPFlow/topography.py` and `GaPFlow/io.py` cannot be imported as modules here (module-level `import ContactMechanics`,
`from muGrid.Field import wrap_field`, `import polars`: not installed), but the functions this path needs are pure NumPy /
pure Python.  The files are parsed with `ast`, the module-level FunctionDef nodes named below are compiled unchanged into a
namespace that holds numpy / yaml / os / datetime -- no stand-in for any missing library -- and evaluated:

  topography.py:38-170   create_midpoint_grid, journal_bearing, inclined_slider, parabolic_slider, cdc, asperity
                         -> topo_profiles.npz   (every profile, two grids, with and without `flip`)
  io.py:38-57, 100-452   print_header, print_dict, read_yaml_input, sanitize_*
                         -> io_sanitized.json   (the sanitised dict of every examples/config/*.yaml and of every YAML
                                                 document held as a string by the reference's tests)

The flip of topography.py:229-236, 251-254 (six statements inside Topography.__init__, which needs a muGrid field
collection) is restated in `arrange`.  asperity() with num > 1 draws from the global NumPy generator (topography.py:141-146):
seeded here and in the tests with the value stored beside the arrays.

    python tests/golden/make_reference_pins.py
"""
import ast
import contextlib
import glob
import io
import json
import os
import sys

import numpy as np
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/'


def reference_functions(path, names, namespace):
    """Compile the named module-level functions of a reference source file, unchanged, into `namespace`."""
    with open(path) as f:
        tree = ast.parse(f.read(), filename=path)
    picked = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    missing = set(names) - {n.name for n in picked}
    assert not missing, f'{path}: no function named {sorted(missing)}'
    exec(compile(ast.Module(body=picked, type_ignores=[]), path, 'exec'), namespace)
    return namespace


TOPO = reference_functions(REF + 'GaPFlow/topography.py',
                           ['create_midpoint_grid', 'journal_bearing', 'inclined_slider', 'parabolic_slider', 'cdc', 'asperity'],
                           {'np': np})
import datetime as _dt  # noqa: E402
IO = reference_functions(REF + 'GaPFlow/io.py',
                         ['print_header', 'print_dict', 'read_yaml_input', 'sanitize_options', 'sanitize_grid', 'sanitize_geometry',
                          'sanitize_properties', 'sanitize_numerics', 'sanitize_gp', 'sanitize_db', 'sanitize_md'],
                         {'yaml': yaml, 'os': os, 'datetime': _dt.datetime, 'np': np})


def sanitized(text):
    with contextlib.redirect_stdout(io.StringIO()):
        return IO['read_yaml_input'](io.StringIO(text))


GRIDS = {
    'g1d': "grid: {Lx: 0.1, Ly: 1., Nx: 100, Ny: 1, xE: ['P', 'P', 'P'], xW: ['P', 'P', 'P'], yS: ['P', 'P', 'P'], yN: ['P', 'P', 'P']}",
    'g2d': "grid: {dx: 1.3e-5, dy: 0.7e-5, Nx: 24, Ny: 17}",
}
GEOS = {
    'journal': 'geometry: {type: journal, CR: 1.e-2, eps: 0.7, U: 0.1, V: 0.03FLIP}',
    'inclined': 'geometry: {type: inclined, hmax: 6.6e-5, hmin: 1.e-5, U: 50., V: 4.FLIP}',
    'parabolic': 'geometry: {type: parabolic, hmin: 1.2e-5, hmax: 6.e-5, U: 0.12, V: 0.FLIP}',
    'cdc': 'geometry: {type: cdc, hmin: 1.e-6, hmax: 1.e-5, b: 0.3e-4, U: 1., V: 0.FLIP}',
    'asperity1': 'geometry: {type: asperity, hmin: 2.e-6, hmax: 1.e-5, num: 1, U: 10., V: 5.FLIP}',
    'asperity3': 'geometry: {type: asperity, hmin: 2.e-6, hmax: 1.e-5, num: 3, U: 10., V: 5.FLIP}',
}
SEED = 20260313


def arrange(h, dh_dx, dh_dy, flip):
    """topography.py:229-236, 251-254: planes [h, dh/dx, dh/dy] as the solver holds them."""
    if flip:
        return np.stack([h.T, dh_dy.T, dh_dx.T])
    return np.stack([h, dh_dx, dh_dy])


def topo_fixture():
    out, cases = {}, []
    for gname, gtext in GRIDS.items():
        for pname, ptext in GEOS.items():
            for flip in (False, True):
                if flip and gname == 'g1d':
                    continue                        # a flipped 1-D grid would need Nx = 1 (tests/test_flip_axes.py uses square 2-D grids)
                if flip and gname == 'g2d':
                    gtext_used = "grid: {dx: 1.3e-5, dy: 0.7e-5, Nx: 20, Ny: 20}"     # h.T must have the field's shape
                else:
                    gtext_used = gtext
                text = gtext_used + '\n' + ptext.replace('FLIP', '')
                d = sanitized(text + "\nproperties: {EOS: DH, shear: 0.1, bulk: 0.}\nnumerics: {}\noptions: {}")
                grid, geo = d['grid'], d['geometry']
                xx, yy = TOPO['create_midpoint_grid'](grid)
                np.random.seed(SEED)
                if geo['type'] == 'asperity':
                    h, hx, hy = TOPO['asperity'](xx, yy, grid, geo)
                else:
                    fn = {'journal': 'journal_bearing', 'inclined': 'inclined_slider', 'parabolic': 'parabolic_slider', 'cdc': 'cdc'}[geo['type']]
                    h, hx, hy = TOPO[fn](xx, grid, geo)
                key = f'{gname}_{pname}' + ('_flip' if flip else '')
                out[key + '_x'], out[key + '_y'] = xx, yy
                out[key + '_topo'] = arrange(np.asarray(h, float), np.asarray(hx, float), np.asarray(hy, float), flip)
                cases.append({'key': key, 'yaml': text, 'flip': flip, 'seed': SEED})
    out['cases'] = np.array(json.dumps(cases))
    np.savez_compressed(os.path.join(HERE, 'topo_profiles.npz'), **out)
    print(f'topo_profiles.npz: {len(cases)} profiles')


def yaml_documents_of_tests():
    """YAML documents the reference's tests hold as plain string constants (f-string templates are not documents)."""
    docs = {}
    for path in sorted(glob.glob(REF + 'tests/*.py')):
        with open(path) as f:
            tree = ast.parse(f.read())
        k = 0
        for node in ast.walk(tree):
            if isinstance(node, ast.Constant) and isinstance(node.value, str) and 'grid:' in node.value and 'geometry:' in node.value:
                try:
                    raw = yaml.full_load(node.value)
                except yaml.YAMLError:
                    continue
                if isinstance(raw, dict) and 'grid' in raw:
                    docs[f'tests/{os.path.basename(path)}#{k}'] = node.value
                    k += 1
    return docs


BASE = "grid: {Nx: 8, dx: 1.e-5, Ny: 1, dy: 1.}\ngeometry: {type: inclined, hmax: 2.e-5, hmin: 1.e-5}\nnumerics: {}\noptions: {}\n"


def synthetic_documents():
    """Inputs written for this fixture (not the reference's) that walk the sanitiser's branches: defaults of every equation of
    state, piezo-viscosity and shear-thinning law, the elastic section, boundary-condition mixes, gp / db defaults, and the
    inputs the reference refuses (the exception type is the expected output)."""
    docs = {}
    for eos in ('DH', 'PL', 'vdW', 'MT', 'cubic', 'BWR', 'Bayada', 'MD'):
        docs[f'synthetic/eos_{eos}_defaults'] = BASE + f"properties: {{EOS: {eos}, shear: 0.1, bulk: 0.}}\n"
    docs['synthetic/eos_vdW_rho0'] = BASE + "properties: {EOS: vdW, shear: 0.1, bulk: 0., rho0: 30., T: 120.}\n"
    for law in ('Barus', 'Roelands', 'Dukler', 'McAdams', 'nonsense'):
        docs[f'synthetic/piezo_{law}'] = BASE + f"properties: {{EOS: DH, shear: 0.1, bulk: 0., piezo: {{name: {law}}}}}\n"
    docs['synthetic/piezo_Roelands_values'] = BASE + "properties: {EOS: DH, shear: 0.1, bulk: 0., piezo: {name: Roelands, z: 0.5, p_ref: 1.e8}}\n"
    for law in ('Carreau', 'Eyring', 'nonsense'):
        docs[f'synthetic/thinning_{law}'] = BASE + f"properties: {{EOS: DH, shear: 0.1, bulk: 0., thinning: {{name: {law}}}}}\n"
    docs['synthetic/elastic_defaults'] = BASE + "properties: {EOS: DH, shear: 0.1, bulk: 0., elastic: {}}\n".replace('elastic: {}', 'elastic: {E: 1.e9}')
    docs['synthetic/elastic_values'] = BASE + "properties: {EOS: DH, shear: 0.1, bulk: 0., elastic: {E: 5.e10, v: 0.25, alpha_underrelax: 0.01, n_images: 3}}\n"
    props = "properties: {EOS: DH, shear: 0.1, bulk: 0.}\n"
    geo = "geometry: {type: inclined, hmax: 2.e-5, hmin: 1.e-5}\nnumerics: {}\noptions: {}\n"
    docs['synthetic/grid_Lx_Ly'] = "grid: {Nx: 7, Lx: 0.3, Ny: 3, Ly: 0.2}\n" + geo + props
    docs['synthetic/grid_defaults_Nx'] = "grid: {dx: 0.1, dy: 0.2}\n" + geo + props
    docs['synthetic/grid_bc_mix'] = ("grid: {Nx: 9, dx: 1.e-5, Ny: 4, dy: 2.e-5, xE: ['D', 'N', 'N'], xW: ['N', 'D', 'N'], xE_D: 870., xW_D: 12.5, "
                                     "yS: ['D', 'N', 'N'], yN: ['N', 'N', 'D'], yS_D: 1., yN_D: 2.}\n" + geo + props)
    docs['synthetic/grid_bc_dirichlet_default_x'] = "grid: {Nx: 9, dx: 1.e-5, Ny: 1, dy: 1., xE: ['D', 'N', 'N'], xW: ['D', 'N', 'N']}\n" + geo + props
    docs['synthetic/grid_bc_dirichlet_missing_y'] = "grid: {Nx: 9, dx: 1.e-5, Ny: 4, dy: 1., yS: ['D', 'N', 'N'], yN: ['D', 'N', 'N']}\n" + geo + props
    docs['synthetic/grid_bc_periodic_mismatch'] = "grid: {Nx: 9, dx: 1.e-5, Ny: 1, dy: 1., xE: ['P', 'P', 'P'], xW: ['D', 'N', 'N']}\n" + geo + props
    docs['synthetic/grid_bc_bad_letter'] = "grid: {Nx: 9, dx: 1.e-5, Ny: 1, dy: 1., xE: ['X', 'P', 'P']}\n" + geo + props
    docs['synthetic/grid_no_size_x'] = "grid: {Nx: 9, Ny: 1, dy: 1.}\n" + geo + props
    docs['synthetic/grid_no_size_y'] = "grid: {Nx: 9, dx: 1., Ny: 1}\n" + geo + props
    g = "grid: {Nx: 8, dx: 1.e-5, Ny: 1, dy: 1.}\nnumerics: {}\noptions: {}\n" + props
    docs['synthetic/geo_journal_cr_eps'] = g + "geometry: {type: journal, CR: 1.e-2, eps: 0.7, U: 0.1}\n"
    docs['synthetic/geo_journal_hmin_hmax'] = g + "geometry: {type: journal, hmin: 1.e-6, hmax: 3.e-6, V: 2.}\n"
    docs['synthetic/geo_journal_incomplete'] = g + "geometry: {type: journal, CR: 1.e-2}\n"
    docs['synthetic/geo_cdc'] = g + "geometry: {type: cdc, hmin: 1.e-6, hmax: 3.e-6, b: 2.e-5, flip: True}\n"
    docs['synthetic/geo_asperity_default_num'] = g + "geometry: {type: asperity, hmin: 1.e-6, hmax: 3.e-6}\n"
    docs['synthetic/geo_unknown'] = g + "geometry: {type: wedge, hmin: 1.e-6, hmax: 3.e-6}\n"
    docs['synthetic/geo_inclined_missing_height'] = g + "geometry: {type: inclined, hmax: 3.e-6}\n"
    docs['synthetic/props_no_shear'] = BASE + "properties: {EOS: DH, bulk: 0.}\n"
    docs['synthetic/props_no_bulk'] = BASE + "properties: {EOS: DH, shear: 0.1}\n"
    docs['synthetic/props_unknown_eos'] = BASE + "properties: {EOS: ideal, shear: 0.1, bulk: 0.}\n"
    docs['synthetic/numerics_values'] = BASE.replace('numerics: {}', 'numerics: {tol: 1.e-9, max_it: 77, dt: 2.e-9, adaptive: 1, CFL: 0.25, MC_order: 0}') + props
    docs['synthetic/options_values'] = BASE.replace('options: {}', 'options: {output: data/x, write_freq: 13, use_tstamp: False, silent: 1}') + props
    docs['synthetic/gp_defaults'] = BASE + props + "gp: {press: {}, shear: {}}\ndb: {}\n".replace('press: {}', 'press: {atol: 2.}').replace('shear: {}', 'shear: {rtol: 0.3}').replace('db: {}', 'db: {init_size: 4}')
    docs['synthetic/gp_press_only'] = BASE + props + "gp: {press: {obs_stddev: 3., active_dims: [0, 1, 3], max_steps: 2, pause_steps: 7, active_learning: False}}\ndb: {init_method: sobol, init_width: 0.05, init_seed: 7}\n"
    docs['synthetic/gp_shear_dims'] = BASE + props + "gp: {shear: {active_dims: {x: [0, 1], y: [0, 2]}, fix_noise: False}}\ndb: {init_method: rand, dtool_path: data/train}\n"
    docs['synthetic/db_bad_method'] = BASE + props + "gp: {press: {atol: 1.}}\ndb: {init_method: grid}\n"
    docs['synthetic/md_passthrough'] = BASE + props + "md: {system: lj, ncpu: 4, nested: {a: 1}}\n"
    docs['synthetic/section_missing'] = "grid: {Nx: 8, dx: 1.e-5, Ny: 1, dy: 1.}\n" + props
    return docs


def plain(o):
    if isinstance(o, dict):
        return {str(k): plain(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [plain(v) for v in o]
    if isinstance(o, (np.floating, np.integer, np.bool_)):
        return o.item()
    return o


def io_fixture():
    docs = {}
    for path in sorted(glob.glob(REF + 'examples/config/*.yaml')):
        with open(path) as f:
            docs['examples/config/' + os.path.basename(path)] = f.read()
    docs.update(yaml_documents_of_tests())
    docs.update(synthetic_documents())
    out = {}
    for name, text in docs.items():
        try:
            out[name] = {'yaml': text, 'sanitized': plain(sanitized(text))}
        except Exception as e:      # noqa: BLE001  -- what the reference raises on this input is part of the contract
            out[name] = {'yaml': text, 'raises': type(e).__name__}
    with open(os.path.join(HERE, 'io_sanitized.json'), 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(f'io_sanitized.json: {len(out)} documents ({sum("raises" in v for v in out.values())} raise)')


if __name__ == '__main__':
    topo_fixture()
    io_fixture()
