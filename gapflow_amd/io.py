"""YAML input layer: same sections, defaults and quirks as the reference's GaPFlow/io.py.

Accepts the reference's input files unchanged and produces the same sanitised dictionaries
(io.py:100-452), which are the input contract of ``Problem.from_yaml``.  Output helpers
(history.csv, config.yml, output directory) follow io.py:60-97.
"""
import csv
import os
from datetime import datetime

import yaml

_BC_TYPES = ('P', 'N', 'D')


def print_header(s, n=60, f0='*', f1=' '):
    # io.py:38-47
    if len(s) > n:
        n = len(s) + 4
    w = n + len(s) % 2
    b = (w - len(s)) // 2 - 1
    print(w * f0)
    print(f0 + b * f1 + s + b * f1 + f0)
    print(w * f0)


def print_dict(d):
    # io.py:50-57
    for k, v in d.items():
        if isinstance(v, dict):
            print(f'  - {k}:')
            for kk, vv in v.items():
                print(f'    - {kk:<23s}: {vv}')
        else:
            print(f'  - {k:<25s}: {v}')


def _get_output_path(name, use_tstamp=True):
    stamp = datetime.now().replace(microsecond=0).strftime("%Y-%m-%d_%H%M%S") + '_' if use_tstamp else ''
    return os.path.join(os.path.dirname(name), stamp + os.path.basename(name))


def create_output_directory(name, use_tstamp=True):
    # io.py:74-86
    outdir = _get_output_path(name, use_tstamp)
    if not os.path.exists(outdir):
        os.makedirs(outdir)
    elif len(os.listdir(outdir)) > 0:
        raise RuntimeError('Output path exists and is not empty.')
    print_header(f"Writing output into: {outdir}", f0=' ', f1=' ')
    return outdir


def write_yaml(output_dict, fname):
    with open(fname, 'w') as f:
        yaml.dump(output_dict, f)


def history_to_csv(fname, out):
    """Column-wise dict -> csv with a header row (the reference writes it through polars, io.py:95-97)."""
    keys = list(out.keys())
    with open(fname, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(keys)
        for row in zip(*[out[k] for k in keys]):
            w.writerow([float(v) if not isinstance(v, (int, str)) else v for v in row])


def read_yaml_input(file):
    """io.py:100-125: every known section is sanitised, absent sections become None."""
    print_header("PROBLEM SETUP")
    sanitizers = (('options', sanitize_options), ('grid', sanitize_grid), ('geometry', sanitize_geometry),
                  ('numerics', sanitize_numerics), ('properties', sanitize_properties), ('gp', sanitize_gp),
                  ('db', sanitize_db), ('md', sanitize_md))
    raw = yaml.full_load(file)
    out = {}
    for key, func in sanitizers:
        print(f'- {key}:')
        val = raw.get(key)
        out[key] = func(val) if val is not None else None
    print_header("PROBLEM SETUP COMPLETED")
    return out


def sanitize_options(d):
    out = {'output': str(d.get('output', 'example')),
           'write_freq': int(d.get('write_freq', 1000)),
           'use_tstamp': bool(d.get('use_tstamp', True)),
           'silent': bool(d.get('silent', False))}
    print_dict(out)
    return out


def _axis(out, d, ax, n_default):
    N, L, h = 'N' + ax, 'L' + ax, 'd' + ax
    out[N] = int(d.get(N, n_default))
    if L in d.keys():
        out[L] = float(d.get(L, 1.))
        out[h] = out[L] / out[N]
    elif h in d.keys():
        out[h] = float(d.get(h, 0.1))
        out[L] = out[h] * out[N]
    else:
        raise IOError(f"Must specify grid size ({N}) with either {h} or {L}.")


def _edge(out, d, side, default_value):
    bc = list(d.get(side, ['P', 'P', 'P']))
    assert all(b in _BC_TYPES for b in bc)
    out[f'bc_{side}_P'] = [b == 'P' for b in bc]
    out[f'bc_{side}_D'] = [b == 'D' for b in bc]
    out[f'bc_{side}_N'] = [b == 'N' for b in bc]


def sanitize_grid(d):
    # io.py:140-223
    out = {}
    _axis(out, d, 'x', 100)
    _axis(out, d, 'y', 1)
    out['dim'] = int(out['Nx'] > 1) + int(out['Ny'] > 1)
    for pair, default in ((('xE', 'xW'), 1.), (('yS', 'yN'), None)):
        for side in pair:
            _edge(out, d, side, default)
        for side in pair:
            if any(out[f'bc_{side}_D']):
                out[f'bc_{side}_D_val'] = d.get(f'{side}_D', default)
                if out[f'bc_{side}_D_val'] is None:
                    raise IOError("Need to specify Dirichlet BC value")
        a, b = pair
        assert all(p == q for p, q in zip(out[f'bc_{a}_P'], out[f'bc_{b}_P']))
    print_dict(out)
    return out


def sanitize_geometry(d):
    # io.py:226-265
    out = {'U': float(d.get('U', 1.)), 'V': float(d.get('V', 0.)),
           'type': str(d.get('type', 'none')), 'flip': bool(d.get('flip', False))}
    if out['type'] not in ('journal', 'inclined', 'parabolic', 'cdc', 'asperity'):
        raise IOError("Specify a valid geometry type")
    if out['type'] == 'journal':
        # the reference's test `"CR" and 'eps' in d` only looks at 'eps' (io.py:240)
        if 'eps' in d.keys():
            out['CR'] = float(d.get('CR'))
            out['eps'] = float(d.get('eps'))
        elif 'hmax' in d.keys():
            out['hmin'] = float(d.get('hmin'))
            out['hmax'] = float(d.get('hmax'))
        else:
            raise IOError("Need to specify either clearance ratio and eccentrity or min/max gap height")
    elif out['type'] == 'inclined':
        out['hmax'] = float(d.get('hmax'))
        out['hmin'] = float(d.get('hmin'))
    else:
        out['hmin'] = float(d.get('hmin'))
        out['hmax'] = float(d.get('hmax'))
        if out['type'] == 'cdc':
            out['b'] = float(d.get('b'))
        elif out['type'] == 'asperity':
            out['num'] = int(d.get('num', 1))
    print_dict(out)
    return out


_EOS_TABLE = {
    'DH': (('rho0', 'P0', 'C1', 'C2'), (877.7007, 101325, 3.5e10, 1.23)),
    'PL': (('rho0', 'P0', 'alpha'), (1.1853, 101325, 0.)),
    'vdW': (('M', 'T', 'a', 'b'), (39.948, 100., 1.355, 0.03201)),
    'MT': (('rho0', 'P0', 'K', 'n'), (700., 0.101e6, .557e9, 7.33)),
    'cubic': (('a', 'b', 'c', 'd'), (15.2, -9.6, 3.35, -0.07)),
    'BWR': (('T', 'gamma'), (2., 3.0)),
    'Bayada': (('rho_l', 'rho_v', 'c_l', 'c_v'), (850., 0.019, 1600., 352.)),
    'MD': (('rho0',), (1.,)),
}
_PIEZO_TABLE = {
    'Roelands': (('mu_inf', 'p_ref', 'z'), (1.e-3, 1.96e8, 0.68)),
    'Barus': (('aB',), (20e-9,)),
    'Dukler': (('eta_v', 'rho_l', 'rho_v'), (3.9e-5, 850., 0.019)),
    'McAdams': (('eta_v', 'rho_l', 'rho_v'), (3.9e-5, 850., 0.019)),
}
_THINNING_TABLE = {
    'Carreau': (('mu_inf', 'lam', 'a', 'N'), (1.e-9, 1e-6, 2., 0.6)),
    'Eyring': (('tauE',), (5.e5,)),
}


def sanitize_properties(d):
    # io.py:268-378
    out = {}
    out['shear'] = float(d.get('shear', -1.))
    if out['shear'] < 0.:
        raise IOError("Specify a a (non-negative) shear viscosity")
    out['bulk'] = float(d.get('bulk', -1.))        # not validated by the reference either (io.py:276-278)
    out['EOS'] = str(d.get('EOS', 'none'))
    if out['EOS'] not in _EOS_TABLE:
        raise IOError("Specify a valid equation of state")
    keys, defaults = _EOS_TABLE[out['EOS']]
    for k, de in zip(keys, defaults):
        out[k] = float(d.get(k, de))
    if 'rho0' not in out.keys():
        out['rho0'] = float(d.get('rho0', 1.))
    for section, table in (('piezo', _PIEZO_TABLE), ('thinning', _THINNING_TABLE)):
        if section in d.keys():
            name = str(d[section].get('name', 'none'))
            out[section] = {'name': name}
            if name in table:
                for k, de in zip(*table[name]):
                    out[section][k] = float(d[section].get(k, de))
    if 'elastic' in d.keys():
        e = d['elastic']
        out['elastic'] = {'enabled': True, 'E': float(e.get('E', 210e09)), 'v': float(e.get('v', 0.3)),
                          'alpha_underrelax': float(e.get('alpha_underrelax', 1e-03)),
                          'n_images': int(e.get('n_images', 10))}
    else:
        out['elastic'] = {'enabled': False}
    print_dict(out)
    return out


def sanitize_numerics(d):
    out = {'tol': float(d.get('tol', 1e-6)), 'max_it': int(d.get('max_it', 1000)),
           'dt': float(d.get('dt', 3e-10)), 'adaptive': bool(d.get('adaptive', False)),
           'CFL': float(d.get('CFL', 0.5)), 'MC_order': int(d.get('MC_order', 1))}
    print_dict(out)
    return out


def sanitize_gp(d):
    # io.py:397-428
    out = {'press_gp': bool('press' in d.keys()), 'shear_gp': bool('shear' in d.keys())}
    for sk in ('press', 'shear'):
        if sk not in d.keys():
            continue
        ds = d[sk]
        o = {'atol': float(ds.get('atol', 1.)), 'rtol': float(ds.get('rtol', 0.5)),
             'obs_stddev': float(ds.get('obs_stddev', 0.)), 'fix_noise': bool(ds.get('fix_noise', True)),
             'max_steps': int(ds.get('max_steps', 5)), 'pause_steps': int(ds.get('pause_steps', 100)),
             'active_learning': bool(ds.get('active_learning', True))}
        if sk == 'press':
            o['active_dims'] = list(ds.get('active_dims', [0, 3]))
        else:
            ad = ds.get('active_dims', {})
            o['active_dims_x'] = ad.get('x', [0, 1, 3])
            o['active_dims_y'] = ad.get('y', [0, 2, 3])
        out[sk] = o
    print_dict(out)
    return out


def sanitize_db(d):
    # io.py:431-445; 'init_seed' is read from the 'init_width' key there (io.py:439) -- kept
    out = {'dtool_path': d.get('dtool_path', None), 'init_size': int(d.get('init_size', 5)),
           'init_method': str(d.get('init_method', 'lhc')), 'init_width': float(d.get('init_width', 1e-2)),
           'init_seed': int(d.get('init_width', 123))}
    assert out['init_method'] in ('rand', 'lhc', 'sobol')
    print_dict(out)
    return out


def sanitize_md(d):
    print_dict(d)
    return d
