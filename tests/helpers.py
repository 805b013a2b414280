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


def prepare(problem, fixture, meta):
    """Apply the fixture's initial state (slip-length field, seeded wave) to a constructed problem.

    Works for oracle.problem.OracleProblem and gapflow_amd.Problem alike."""
    return problem


def rel_err(a, b):
    """max |a-b| / max|b| per leading component (fields are compared relative to their own scale)."""
    a, b = np.asarray(a, float), np.asarray(b, float)
    if a.ndim == 2:
        a, b = a[None], b[None]
    worst = 0.
    for x, y in zip(a, b):
        scale = np.max(np.abs(y))
        worst = max(worst, np.max(np.abs(x - y)) / (scale if scale > 0 else 1.))
    return worst
