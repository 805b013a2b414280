"""Input sanitisation, restated from GaPFlow/io.py:128-445 (plain dict in, plain dict out).

Test infrastructure only (see oracle/__init__.py).
"""
import yaml


def _bc_masks(out, d, side, default_val, mandatory):
    # io.py:170-219 -- per-component 'P'/'D'/'N' lists become boolean masks
    bc = list(d.get(side, ['P', 'P', 'P']))
    assert all(b in ('P', 'N', 'D') for b in bc)
    for t in 'PDN':
        out[f'bc_{side}_{t}'] = [b == t for b in bc]
    if any(out[f'bc_{side}_D']):
        val = d.get(f'{side}_D', default_val)
        if val is None:
            raise IOError("Need to specify Dirichlet BC value")
        out[f'bc_{side}_D_val'] = val


def sanitize_options(d):
    # io.py:128-137
    return {'output': str(d.get('output', 'example')),
            'write_freq': int(d.get('write_freq', 1000)),
            'use_tstamp': bool(d.get('use_tstamp', True)),
            'silent': bool(d.get('silent', False))}


def sanitize_grid(d):
    # io.py:140-223
    out = {}
    for ax in 'xy':
        N, L, dd = f'N{ax}', f'L{ax}', f'd{ax}'
        out[N] = int(d.get(N, 100 if ax == 'x' else 1))
        if L in d:
            out[L] = float(d[L])
            out[dd] = out[L] / out[N]
        elif dd in d:
            out[dd] = float(d[dd])
            out[L] = out[dd] * out[N]
        else:
            raise IOError(f"Must specify grid size ({N}) with either {dd} or {L}.")
    out['dim'] = int(out['Nx'] > 1) + int(out['Ny'] > 1)
    _bc_masks(out, d, 'xE', 1., False)
    _bc_masks(out, d, 'xW', 1., False)
    assert out['bc_xE_P'] == out['bc_xW_P']
    _bc_masks(out, d, 'yS', None, True)
    _bc_masks(out, d, 'yN', None, True)
    assert out['bc_yS_P'] == out['bc_yN_P']
    return out


def sanitize_geometry(d):
    # io.py:226-265
    out = {'U': float(d.get('U', 1.)), 'V': float(d.get('V', 0.)),
           'type': str(d.get('type', 'none')), 'flip': bool(d.get('flip', False))}
    t = out['type']
    if t not in ('journal', 'inclined', 'parabolic', 'cdc', 'asperity'):
        raise IOError("Specify a valid geometry type")
    if t == 'journal':
        if 'eps' in d:      # ("CR" and 'eps' in d) == ('eps' in d), io.py:240
            out['CR'] = float(d.get('CR'))
            out['eps'] = float(d.get('eps'))
        elif 'hmax' in d:
            out['hmin'] = float(d.get('hmin'))
            out['hmax'] = float(d.get('hmax'))
        else:
            raise IOError("Need to specify either clearance ratio and eccentrity or min/max gap height")
    else:
        out['hmin'] = float(d.get('hmin'))
        out['hmax'] = float(d.get('hmax'))
        if t == 'cdc':
            out['b'] = float(d.get('b'))
        if t == 'asperity':
            out['num'] = int(d.get('num', 1))
    return out


_EOS_DEFAULTS = {
    'DH': (['rho0', 'P0', 'C1', 'C2'], [877.7007, 101325, 3.5e10, 1.23]),
    'PL': (['rho0', 'P0', 'alpha'], [1.1853, 101325, 0.]),
    'vdW': (['M', 'T', 'a', 'b'], [39.948, 100., 1.355, 0.03201]),
    'MT': (['rho0', 'P0', 'K', 'n'], [700., 0.101e6, .557e9, 7.33]),
    'cubic': (['a', 'b', 'c', 'd'], [15.2, -9.6, 3.35, -0.07]),
    'BWR': (['T', 'gamma'], [2., 3.0]),
    'Bayada': (['rho_l', 'rho_v', 'c_l', 'c_v'], [850., 0.019, 1600., 352.]),
    'MD': (['rho0'], [1.]),
}


def sanitize_properties(d):
    # io.py:268-378
    out = {'shear': float(d.get('shear', -1.))}
    if out['shear'] < 0.:
        raise IOError("Specify a a (non-negative) shear viscosity")
    out['bulk'] = float(d.get('bulk', -1.))     # never validated in the reference (io.py:276-278)
    out['EOS'] = str(d.get('EOS', 'none'))
    if out['EOS'] not in _EOS_DEFAULTS:
        raise IOError("Specify a valid equation of state")
    keys, defaults = _EOS_DEFAULTS[out['EOS']]
    for k, de in zip(keys, defaults):
        out[k] = float(d.get(k, de))
    if 'rho0' not in out:
        out['rho0'] = float(d.get('rho0', 1.))
    if 'piezo' in d:
        name = str(d['piezo'].get('name', 'none'))
        out['piezo'] = {'name': name}
        table = {'Roelands': (['mu_inf', 'p_ref', 'z'], [1.e-3, 1.96e8, 0.68]),
                 'Barus': (['aB'], [20e-9]),
                 'Dukler': (['eta_v', 'rho_l', 'rho_v'], [3.9e-5, 850., 0.019]),
                 'McAdams': (['eta_v', 'rho_l', 'rho_v'], [3.9e-5, 850., 0.019])}
        if name in table:
            for k, de in zip(*table[name]):
                out['piezo'][k] = float(d['piezo'].get(k, de))
    if 'thinning' in d:
        name = str(d['thinning'].get('name', 'none'))
        out['thinning'] = {'name': name}
        table = {'Carreau': (['mu_inf', 'lam', 'a', 'N'], [1.e-9, 1e-6, 2., 0.6]),
                 'Eyring': (['tauE'], [5.e5])}
        if name in table:
            for k, de in zip(*table[name]):
                out['thinning'][k] = float(d['thinning'].get(k, de))
    if 'elastic' in d:
        e = d['elastic']
        out['elastic'] = {'enabled': True, 'E': float(e.get('E', 210e09)), 'v': float(e.get('v', 0.3)),
                          'alpha_underrelax': float(e.get('alpha_underrelax', 1e-03)),
                          'n_images': int(e.get('n_images', 10))}
    else:
        out['elastic'] = {'enabled': False}
    return out


def sanitize_numerics(d):
    # io.py:381-394
    return {'tol': float(d.get('tol', 1e-6)), 'max_it': int(d.get('max_it', 1000)),
            'dt': float(d.get('dt', 3e-10)), 'adaptive': bool(d.get('adaptive', False)),
            'CFL': float(d.get('CFL', 0.5)), 'MC_order': int(d.get('MC_order', 1))}


def sanitize_gp(d):
    # io.py:397-428
    out = {'press_gp': 'press' in d, 'shear_gp': 'shear' in d}
    for sk in ('press', 'shear'):
        if sk in d:
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
    return out


def sanitize_db(d):
    # io.py:431-445 ('init_seed' reads the 'init_width' key -- reference quirk, io.py:439)
    out = {'dtool_path': d.get('dtool_path', None), 'init_size': int(d.get('init_size', 5)),
           'init_method': str(d.get('init_method', 'lhc')), 'init_width': float(d.get('init_width', 1e-2)),
           'init_seed': int(d.get('init_width', 123))}
    assert out['init_method'] in ('rand', 'lhc', 'sobol')
    return out


def read_yaml_input(stream):
    # io.py:100-125 (sections absent from the file map to None)
    raw = yaml.full_load(stream)
    funcs = {'options': sanitize_options, 'grid': sanitize_grid, 'geometry': sanitize_geometry,
             'numerics': sanitize_numerics, 'properties': sanitize_properties, 'gp': sanitize_gp,
             'db': sanitize_db, 'md': lambda d: d}
    return {k: (f(raw[k]) if raw.get(k) is not None else None) for k, f in funcs.items()}
