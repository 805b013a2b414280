"""Shared helpers of the parity tests: golden fixture loading and problem construction."""
import ast
import io
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

STEP_CASES = sorted(f[len('step_'):-len('.npz')] for f in os.listdir(GOLDEN) if f.startswith('step_'))


def load_case(name):
    d = np.load(os.path.join(GOLDEN, f'step_{name}.npz'))
    meta = ast.literal_eval(str(d['meta']))
    return d, str(d['yaml']), meta


def input_dict(yaml_text, meta, reader):
    """Sanitised input dict of a fixture, including the flip of tests/test_flip_axes.py:73-76."""
    with io.StringIO(yaml_text) as f:
        d = reader(f)
    if meta.get('flip'):
        d['geometry']['V'] = d['geometry']['U']
        d['geometry']['U'] = 0.
        d['geometry']['flip'] = True
    return d


def comp_err(a, b):
    """Per-component max|a-b| / max|b_c| (scale 1 for an identically zero component) -> array (ncomp,)."""
    a, b = np.asarray(a, float), np.asarray(b, float)
    if a.ndim == 2:
        a, b = a[None], b[None]
    return np.array([np.abs(x - y).max() / (np.abs(y).max() or 1.) for x, y in zip(a, b)])


def rel_err(a, b):
    return comp_err(a, b).max()


FIELD_RTOL = 1e-9      # BASELINE.json north_star: fp64 fields within 1e-9 relative


def field_tol(fx, s):
    """Tolerance per component of q at snapshot s: 1e-9, unless the problem itself is more sensitive.

    `sens_<s>` in the fixture is the oracle's own response to a one-ulp perturbation of the initial
    field (tests/golden/make_golden.py); with dp/drho ~ 1e8 that noise reaches 1e-9..1e-8 of the flux
    scale within a step for the stiff DH cases, so no two correct implementations agree better."""
    return np.maximum(FIELD_RTOL, 10. * fx[f'sens_{s}'])


# golden history columns: step, time, dt(next), ekin, residual, vsound, vmax, mass
HISTORY_RTOL = np.array([0., 1e-12, 1e-10, 1e-11, 1e-6, 1e-11, 1e-10, 1e-12])


def history_tol(fx):
    return np.maximum(HISTORY_RTOL, 10. * fx['sens_history'])
