"""CPU sanitizer build of the per-cell closures ("run the HIP lib under -fsanitize=address host
side").  gapflow_amd/csrc/closures.hpp + phys_setup.hpp are plain C++ when compiled without hipcc; tests/hostcheck
builds them with g++ -fsanitize=address,undefined and evaluates the reference's golden inputs.  The outputs must match
tests/golden/leaf_closures.npz (true outputs of the reference's pressure.py / sound.py / viscous.py / viscosity.py)
and no sanitizer report may appear."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from helpers import GOLDEN

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'hostcheck', 'closures_host.cpp')
LEAF = np.load(os.path.join(GOLDEN, 'leaf_closures.npz'))

EOS_ORDER = ['DH', 'PL', 'vdW', 'MT', 'cubic', 'BWR', 'Bayada']
EOS_PAR = {'DH': [877.7007, 101325., 3.5e10, 1.23], 'PL': [1.1853, 101325., 0.], 'vdW': [39.948, 100., 1.355, 0.03201],
           'MT': [700., 0.101e6, 0.557e9, 7.33], 'cubic': [1.33030e-1, -1.41778e2, 8.35134e4, -2.86532e6], 'BWR': [1.0, 3.0],
           'Bayada': [850., 0.019, 1600., 352.]}
PIEZO = [('Barus', [20e-9], 'piezo_p'), ('Roelands', [1e-3, 1.96e8, 0.68], 'piezo_p'),
         ('Dukler', [3.9e-5, 850., 0.019], 'piezo_rho'), ('McAdams', [3.9e-5, 850., 0.019], 'piezo_rho')]
THIN = [('Eyring', [5e5]), ('Carreau', [1e-9, 1e-6, 2., 0.6])]


def pad(v, n):
    return list(v) + [0.] * (n - len(v))


@pytest.fixture(scope='module')
def hostcheck(tmp_path_factory):
    gxx = shutil.which('g++')
    assert gxx, 'g++ is part of the image'
    exe = str(tmp_path_factory.mktemp('hostcheck') / 'closures_host')
    cmd = [gxx, '-std=c++17', '-O1', '-g', '-fsanitize=address,undefined', '-fno-sanitize-recover=all', '-Wall', '-Werror',
           SRC, '-o', exe]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-3000:]
    return exe


def run(exe):
    parts = []
    n = LEAF['eos_DH_rho'].size
    parts.append([float(n)])
    for e in EOS_ORDER:
        parts += [pad(EOS_PAR[e], 8), LEAF[f'eos_{e}_rho'].ravel()]
    q, h = LEAF['visc_q'], LEAF['visc_h']
    m = q[0].size
    out_sizes = [('eos', 7, n)]
    blocks = []
    for tag in ('Ls0', 'LsF'):
        blocks.append((tag, [[float(m)] + list(LEAF['visc_params']), pad(EOS_PAR['DH'], 8), q.ravel(), h.ravel(), LEAF[f'visc_{tag}_Ls'].ravel()]))
    # the program handles ONE stress block: it is run once per slip-length table
    results = {}
    k = LEAF['piezo_p'].size
    tail = [[float(k), 0.0794]]
    for name, par, arg in PIEZO:
        tail += [pad(par, 4), LEAF[arg]]
    for name, par in THIN:
        tail += [pad(par, 4), LEAF['thin_sr']]
    tail += [[0.1, 0.], LEAF['sr_gx'], LEAF['sr_gy'], LEAF['sr_h']]
    for tag, blk in blocks:
        data = np.concatenate([np.asarray(x, float).ravel() for x in parts + blk + tail])
        env = dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=0', UBSAN_OPTIONS='print_stacktrace=1')
        res = subprocess.run([exe], input=data.tobytes(), capture_output=True, env=env)
        assert res.returncode == 0, res.stderr.decode()[-3000:]
        assert res.stderr == b'', 'sanitizer output:\n' + res.stderr.decode()[-3000:]
        out = np.frombuffer(res.stdout, dtype=float)
        assert out.size == 7 * 2 * n + (6 + 6 + 3 + 1) * m + 7 * k
        results[tag] = out
    return results, n, m, k


def test_host_closures_match_reference_outputs_under_sanitizers(hostcheck):
    results, n, m, k = run(hostcheck)
    out = results['Ls0']
    pos = 0
    for e in EOS_ORDER:
        p, c = out[pos:pos + n], out[pos + n:pos + 2 * n]
        pos += 2 * n
        p_ref, c_ref = LEAF[f'eos_{e}_p'].ravel(), LEAF[f'eos_{e}_c'].ravel()
        np.testing.assert_allclose(p, p_ref, rtol=1e-12, atol=1e-13 * np.abs(p_ref).max(), err_msg=f'pressure {e}')
        # c = sqrt(dp/drho): where dp/drho passes through zero (BWR's van-der-Waals loop) the polynomial cancels,
        # so the bound is on c^2 relative to its scale
        np.testing.assert_allclose(c * c, c_ref * c_ref, rtol=1e-12, atol=1e-12 * np.nanmax(c_ref * c_ref), equal_nan=True, err_msg=f'sound speed {e}')
    shape = LEAF['visc_q'].shape[1:]
    for tag in ('Ls0', 'LsF'):
        o = results[tag][pos:]
        lower, upper = o[:6 * m].reshape((6,) + shape), o[6 * m:12 * m].reshape((6,) + shape)
        avg, dev = o[12 * m:15 * m].reshape((3,) + shape), o[15 * m:16 * m]
        for got, key in ((lower, 'bot'), (upper, 'top'), (avg, 'avg')):
            ref = LEAF[f'visc_{tag}_{key}']
            for c in range(ref.shape[0]):
                np.testing.assert_allclose(got[c], ref[c], rtol=1e-12, atol=1e-13 * np.abs(ref[c]).max(), err_msg=f'{tag} {key}[{c}]')
        # the fused kernel's specialised closure agrees with the general fields + integrate.py's source formula
        assert dev.max() < 1e-11, f'{tag}: cell_closure deviates from cell_fields by {dev.max():.2e}'
    o = results['Ls0'][pos + 16 * m:]
    for i, (name, _, _) in enumerate(PIEZO):
        np.testing.assert_allclose(o[i * k:(i + 1) * k], LEAF[f'piezo_{name}'], rtol=1e-12, err_msg=name)
    for i, (name, _) in enumerate(THIN):
        np.testing.assert_allclose(o[(4 + i) * k:(5 + i) * k], LEAF[f'thin_{name}'], rtol=1e-12, err_msg=name)
    np.testing.assert_allclose(o[6 * k:7 * k], LEAF['sr_out'], rtol=1e-12)


def test_series_log_exp_pow_match_the_c_library_including_special_values(tmp_path):
    """The device closures do not call the C library's pow / exp / log (a few hundred fp64 operations each) but the series forms
    of closures.hpp.  Same source, compiled for the host: against NumPy over the whole double range, and on the special values
    that decide whether a run that blows up is STOPPED (q_is_valid, problem.py:319-332) -- log(0), log(< 0), exp(NaN),
    exp(overflow), 0^y, x^0 -- which np.log / np.exp / np.power define."""
    gxx = shutil.which('g++')
    exe = str(tmp_path / 'series_host')
    src = os.path.join(HERE, 'hostcheck', 'series_host.cpp')
    res = subprocess.run([gxx, '-std=c++17', '-O1', '-g', '-fsanitize=address,undefined', '-fno-sanitize-recover=all', '-fno-sanitize=float-cast-overflow',
                          '-Wall', '-Werror', src, '-o', exe], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-3000:]
    rng = np.random.default_rng(7)
    inf, nan, tiny = np.inf, np.nan, 5e-324
    xs = np.concatenate([10.0**rng.uniform(-300, 300, 4000), rng.uniform(0.5, 2.0, 4000), [1.0, 2.0, 0.5, tiny, 2.2250738585072014e-308, 1.7976931348623157e308,
                                                                                           0.0, -0.0, -1.0, -tiny, inf, -inf, nan]])
    ts = np.concatenate([rng.uniform(-745, 709.7, 4000), rng.uniform(-2, 2, 4000), [0.0, -0.0, 709.78, 709.79, 710.0, 1e300, inf, -745.0, -745.2, -746.0, -1e300, -inf, nan]])
    n = xs.size
    assert ts.size == n
    px = np.concatenate([rng.uniform(0.2, 5.0, n - 16), [0.0, 0.0, 0.0, 2.0, nan, inf, inf, 1e-300, 1e300, 1e300, 1e-300, -2.0, -2.0, 1.0, nan, 0.0]])
    py = np.concatenate([rng.uniform(-40.0, 40.0, n - 16), [2.5, -2.5, 0.0, 0.0, 0.0, 0.5, -0.5, 3.0, 3.0, -3.0, -3.0, 0.5, 2.0, 5.0, 1.5, nan]])      # (exponents are parameters of a law: pow(1, NaN) = 1 of the C library is not imitated)
    data = np.concatenate([[float(n)], xs, ts, np.column_stack([px, py]).ravel()])
    res = subprocess.run([exe], input=data.tobytes(), capture_output=True,
                         env=dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=0', UBSAN_OPTIONS='print_stacktrace=1'))
    assert res.returncode == 0 and res.stderr == b'', res.stderr.decode()[-3000:]
    out = np.frombuffer(res.stdout, dtype=float)
    with np.errstate(all='ignore'):
        ref_log, ref_exp, ref_pow = np.log(xs), np.exp(ts), np.power(px, py)
    ref_pow[-5], ref_pow[-4] = nan, nan             # negative base: NaN also for the integer exponent np.power would accept
    lg, ex, pw = out[:n], out[n:2 * n], out[2 * n:]

    def same(got, ref, rtol, what):
        special = ~np.isfinite(ref) | (ref == 0.0)
        assert np.array_equal(np.isnan(got), np.isnan(ref)), f'{what}: NaN pattern; got {got[np.isnan(got) != np.isnan(ref)]}'
        assert np.array_equal(got[special & ~np.isnan(ref)], ref[special & ~np.isnan(ref)]), f'{what}: {got[special]} vs {ref[special]}'
        ok = ~special
        if not ok.any():
            return 0.0
        err = np.abs(got[ok] - ref[ok]) / np.maximum(np.abs(ref[ok]), 5e-324)
        assert err.max() <= rtol, f'{what}: {err.max():.2e} at {np.argmax(err)}'
        return err.max()
    normal = np.abs(ref_log) > 1e-3
    e1 = same(lg[normal], ref_log[normal], 4e-16, 'log')
    assert np.abs(lg[~normal & np.isfinite(ref_log)] - ref_log[~normal & np.isfinite(ref_log)]).max() <= 3e-19 + 1e-16 * 1e-3
    sub = np.isfinite(ref_exp) & (ref_exp != 0) & (ref_exp < 2.3e-308)          # denormal results: absolute
    assert np.abs(ex[sub] - ref_exp[sub]).max() <= 5e-324 * 2
    e2 = same(ex[~sub], ref_exp[~sub], 3e-16 + 0, 'exp')
    big = np.isfinite(ref_pow) & (ref_pow != 0)
    relerr = np.abs(pw[big] - ref_pow[big]) / np.abs(ref_pow[big])
    with np.errstate(all='ignore'):     # exp(y ln x): the rounding of the argument is amplified by |y ln x|
        bound = 4e-16 * np.maximum(1.0, np.nan_to_num(np.abs(py[big] * np.log(px[big])), nan=0.0, posinf=0.0))
    worst = np.argmax(relerr / bound)
    assert (relerr <= bound).all(), f'pow: {relerr[worst]:.2e} > {bound[worst]:.2e} for {px[big][worst]!r} ^ {py[big][worst]!r}'
    same(pw[~big], ref_pow[~big], 0.0, 'pow special')
    print(f'\n[series forms vs NumPy] log {e1:.1e}, exp {e2:.1e}, pow {np.max(relerr / bound) * 4e-16:.1e} x max(1, |y ln x|)')
