"""Fixed-form constitutive closures, restated in NumPy.  Test infrastructure only.

Follows (reference file:line, relative to /root/reference):
  GaPFlow/models/pressure.py:35-325   eos_pressure and the seven EOS
  GaPFlow/models/sound.py:35-329      eos_sound_velocity
  GaPFlow/models/viscous.py:37-786    stress_bottom / stress_top / stress_avg
  GaPFlow/models/viscosity.py:34-318  piezo-viscosity, shear thinning
  GaPFlow/models/stress.py:289-362, 427-459, 600-622  (which closure feeds which field)

The solver never passes gradients of q to the stress functions
(stress.py:328-344 leaves dqx=dqy=None -> 0), so the polynomials are written
here without the terms that multiply dqx/dqy; they are otherwise kept in the
reference's term order.  Only ``slip="top"`` -- the single branch the solver reaches (stress.py:328-344
pass no ``slip``) -- is restated.
"""
import numpy as np

R_GAS = 8.31446261815324  # = scipy.constants.gas_constant (exact k_B*N_A), pressure.py:32

# The 32 parameters x1..x32 of the modified Benedict-Webb-Rubin EOS of the
# Lennard-Jones fluid (Johnson, Zollweg & Gubbins, Mol. Phys. 78 (1993) 591) --
# the data table the reference reads from GaPFlow/models/bwr_coeffs.txt at
# pressure.py:255-256 and sound.py:250-251.
BWR_COEFFS = np.array([
    0.8623085097507421,
    2.976218765822098,
    -8.402230115796038,
    0.1054136629203555,
    -0.8564583828174598,
    1.582759470107601,
    0.7639421948305453,
    1.753173414312048,
    2.798291772190376e+03,
    -4.8394220260857657e-02,
    0.9963265197721935,
    -3.698000291272493e+01,
    2.084012299434647e+01,
    8.305402124717285e+01,
    -9.574799715203068e+02,
    -1.477746229234994e+02,
    6.398607852471505e+01,
    1.603993673294834e+01,
    6.805916615864377e+01,
    -2.791293578795945e+03,
    -6.245128304568454,
    -8.116836104958410e+03,
    1.488735559561229e+01,
    -1.059346754655084e+04,
    -1.131607632802822e+02,
    -8.867771540418822e+03,
    -3.986982844450543e+01,
    -4.689270299917261e+03,
    2.593535277438717e+02,
    -2.694523589434903e+03,
    -7.218487631550215e+02,
    1.721802063863269e+02,
])


def _bwr_x():
    return BWR_COEFFS


# ----------------------------------------------------------------------------
# Equations of state (pressure.py) and their sound speeds (sound.py)
# ----------------------------------------------------------------------------

def _bayada_consts(rho_l, rho_v, c_l, c_v):
    # pressure.py:303-304
    N = rho_v * c_v**2 * rho_l * c_l**2 * (rho_v - rho_l) / (rho_v**2 * c_v**2 - rho_l**2 * c_l**2)
    Pcav = rho_v * c_v**2 - N * np.log(rho_v**2 * c_v**2 / (rho_l**2 * c_l**2))
    return N, Pcav


def _bwr_poly(rho, T, gamma):
    # pressure.py:258-272
    x = _bwr_x()
    return (rho * T +
            rho**2 * (x[0] * T + x[1] * np.sqrt(T) + x[2] + x[3] / T + x[4] / T**2) +
            rho**3 * (x[5] * T + x[6] + x[7] / T + x[8] / T**2) +
            rho**4 * (x[9] * T + x[10] + x[11] / T) +
            rho**5 * x[12] +
            rho**6 * (x[13] / T + x[14] / T**2) +
            rho**7 * (x[15] / T) +
            rho**8 * (x[16] / T + x[17] / T**2) +
            rho**9 * (x[18] / T**2) +
            np.exp(-gamma * rho**2) * (rho**3 * (x[19] / T**2 + x[20] / T**3) +
                                       rho**5 * (x[21] / T**2 + x[22] / T**4) +
                                       rho**7 * (x[23] / T**2 + x[24] / T**3) +
                                       rho**9 * (x[25] / T**2 + x[26] / T**4) +
                                       rho**11 * (x[27] / T**2 + x[28] / T**3) +
                                       rho**13 * (x[29] / T**2 + x[30] / T**3 + x[31] / T**4)))


def eos_pressure(rho, prop):
    """p(rho) for prop['EOS'] in DH, PL, vdW, MT, cubic, BWR, Bayada (pressure.py:35-76)."""
    rho = np.asarray(rho, dtype=float)
    eos = prop['EOS']
    if eos == 'DH':         # pressure.py:79-109 (density clamped at 0.99*C2*rho0)
        r = np.minimum(rho, 0.99 * prop['C2'] * prop['rho0'])
        return prop['P0'] + (prop['C1'] * (r / prop['rho0'] - 1.)) / (prop['C2'] - r / prop['rho0'])
    if eos == 'PL':         # pressure.py:112-137
        return prop['P0'] * (rho / prop['rho0'])**(1. / (1. - 0.5 * prop['alpha']))
    if eos == 'vdW':        # pressure.py:140-173
        md = rho / prop['M'] * 1000.
        a = prop['a'] / 10.
        b = prop['b'] / 1000.
        return R_GAS * prop['T'] * md / (1. - b * md) - a * md**2
    if eos == 'MT':         # pressure.py:176-205
        return prop['K'] / prop['n'] * ((rho / prop['rho0'])**prop['n'] - 1) + prop['P0']
    if eos == 'cubic':      # pressure.py:208-230
        return prop['a'] * rho**3 + prop['b'] * rho**2 + prop['c'] * rho + prop['d']
    if eos == 'BWR':        # pressure.py:233-274
        return _bwr_poly(rho, prop['T'], prop['gamma'])
    if eos == 'Bayada':     # pressure.py:277-325 (three branches in alpha)
        rho_l, rho_v, c_l, c_v = prop['rho_l'], prop['rho_v'], prop['c_l'], prop['c_v']
        N, Pcav = _bayada_consts(rho_l, rho_v, c_l, c_v)
        alpha = (rho - rho_l) / (rho_v - rho_l)
        with np.errstate(all='ignore'):
            den = rho_l * (rho_v * c_v**2 * (1 - alpha) + rho_l * c_l**2 * alpha)
            mix = Pcav + N * np.log(rho_v * c_v**2 * rho / den)
        return np.where(alpha < 0, Pcav + (rho - rho_l) * c_l**2,
                        np.where(alpha <= 1, mix, c_v**2 * rho))
    raise ValueError(eos)


def eos_sound_speed(rho, prop):
    """c(rho) = sqrt(dp/drho) (sound.py:35-329).  NB: DH uses the *unclamped* density (sound.py:109)."""
    rho = np.asarray(rho, dtype=float)
    eos = prop['EOS']
    if eos == 'DH':
        c2 = prop['C1'] * prop['rho0'] * (prop['C2'] - 1.0) * (1 / rho)**2 / ((prop['C2'] * prop['rho0'] / rho - 1.0)**2)
    elif eos == 'PL':
        al = prop['alpha']
        c2 = -2.0 * prop['P0'] * (rho / prop['rho0'])**(-2.0 / (al - 2.0)) / ((al - 2) * rho)
    elif eos == 'vdW':
        md = rho / prop['M'] * 1000.
        a = prop['a'] / 10.
        b = prop['b'] / 1000.
        c2 = R_GAS * prop['T'] / (1. - b * md)**2 - 2. * a * md
    elif eos == 'MT':
        c2 = prop['K'] / prop['rho0']**prop['n'] * rho**(prop['n'] - 1)
    elif eos == 'cubic':
        c2 = 3 * prop['a'] * rho**2 + 2 * prop['b'] * rho + prop['c']
    elif eos == 'BWR':      # sound.py:232-285
        x = _bwr_x()
        T, g = prop['T'], prop['gamma']
        e = [x[19] / T**2 + x[20] / T**3, x[21] / T**2 + x[22] / T**4, x[23] / T**2 + x[24] / T**3,
             x[25] / T**2 + x[26] / T**4, x[27] / T**2 + x[28] / T**3, x[29] / T**2 + x[30] / T**3 + x[31] / T**4]
        pre = sum(rho**(3 + 2 * k) * e[k] for k in range(6))
        dpre = sum((3. + 2 * k) * rho**(2 + 2 * k) * e[k] for k in range(6))
        c2 = (T
              + 2.0 * rho * (x[0] * T + x[1] * np.sqrt(T) + x[2] + x[3] / T + x[4] / T**2)
              + 3.0 * rho**2 * (x[5] * T + x[6] + x[7] / T + x[8] / T**2)
              + 4.0 * rho**3 * (x[9] * T + x[10] + x[11] / T)
              + 5.0 * rho**4 * x[12]
              + 6.0 * rho**5 * (x[13] / T + x[14] / T**2)
              + 7.0 * rho**6 * (x[15] / T)
              + 8.0 * rho**7 * (x[16] / T + x[17] / T**2)
              + 9.0 * rho**8 * (x[18] / T**2)
              + np.exp(-g * rho**2) * dpre
              - 2.0 * rho * g * np.exp(-g * rho**2) * pre)
    elif eos == 'Bayada':   # sound.py:288-329
        rho_l, rho_v, c_l, c_v = prop['rho_l'], prop['rho_v'], prop['c_l'], prop['c_v']
        alpha = (rho - rho_l) / (rho_v - rho_l)
        with np.errstate(all='ignore'):
            mix = rho_v * rho_l * (c_v * c_l)**2 / (alpha * rho_l * c_l**2 + (1 - alpha) * rho_v * c_v**2) / rho
        c2 = np.where(alpha < 0, c_l**2, np.where(alpha <= 1, mix, c_v**2))
    else:
        raise ValueError(eos)
    return np.sqrt(c2)


# ----------------------------------------------------------------------------
# Viscosity models (viscosity.py)
# ----------------------------------------------------------------------------

def piezoviscosity(p, mu0, pz):
    """viscosity.py:34-66, 150-262 (p is the density field for the Bayada mixture laws)."""
    name = pz['name']
    if name == 'Barus':
        return mu0 * np.exp(pz['aB'] * p)
    if name == 'Roelands':
        return mu0 * np.exp(np.log(mu0 / pz['mu_inf']) * (-1 + (1 + p / pz['p_ref'])**pz['z']))
    if name in ('Dukler', 'McAdams'):
        alpha = (p - pz['rho_l']) / (pz['rho_v'] - pz['rho_l'])
        if name == 'Dukler':
            return alpha * pz['eta_v'] + (1 - alpha) * mu0
        M = alpha * pz['rho_v'] / p
        return pz['eta_v'] * mu0 / (mu0 * M + pz['eta_v'] * (1 - M))
    return np.ones_like(p) * mu0


def shear_rate_avg(dp_dx, dp_dy, h, u1, u2, mu):
    """viscosity.py:99-141 (mean of |wall shear rates| of the Newtonian profile)."""
    gp = np.hypot(dp_dx, dp_dy)
    du_p = h * gp / (2 * mu)
    du_c = (u2 - u1) / h
    return (np.abs(du_p + du_c) + np.abs(-du_p + du_c)) / 2.


def shear_thinning_factor(shear_rate, mu0, th):
    """viscosity.py:69-96, 265-318."""
    name = th['name']
    if name == 'Eyring':
        tau0 = mu0 * shear_rate
        return th['tauE'] / tau0 * np.arcsinh(tau0 / th['tauE'])
    if name == 'Carreau':
        mu = th['mu_inf'] + (mu0 - th['mu_inf']) * (1 + (th['lam'] * shear_rate)**th['a'])**((th['N'] - 1) / th['a'])
        return mu / mu0
    return np.ones_like(shear_rate)


# ----------------------------------------------------------------------------
# Viscous stresses (viscous.py), grad q == 0
# ----------------------------------------------------------------------------

def _slip_parabola(h, W, m, lo, hi):
    """u(z) = a z^2 + b z + c with u(0) = W + lo u'(0), u(h) = -hi u'(h), mean(u) = m; coefficients and their
    partial derivatives with respect to h and m.  This is the velocity model behind viscous.py / profiles.py
    (profiles.py:60-135 lists the resulting u(z) per slip keyword); derived here, not transcribed."""
    D = h * h + 4 * h * (lo + hi) + 12 * lo * hi
    Dh = 2 * h + 4 * (lo + hi)
    Na = 3 * (h + 2 * hi) * W - 6 * (h + lo + hi) * m
    Nb = -4 * (h + 3 * hi) * W + 6 * (h + 2 * hi) * m
    Nc = h * (h + 4 * hi) * W + 6 * lo * (h + 2 * hi) * m
    a, b, c = Na / (h * D), Nb / D, Nc / D
    ah = ((3 * W - 6 * m) * (h * D) - Na * (D + h * Dh)) / (h * D)**2
    bh = ((-4 * W + 6 * m) * D - Nb * Dh) / D**2
    ch = (((2 * h + 4 * hi) * W + 6 * lo * m) * D - Nc * Dh) / D**2
    am, bm, cm = -6 * (h + lo + hi) / (h * D), 6 * (h + 2 * hi) / D, 6 * lo * (h + 2 * hi) / D
    # without slip at the lower wall u(0) = W exactly: keep c's derivatives exactly zero instead of a rounding residue
    noslip = np.asarray(lo) == 0
    c, ch = np.where(noslip, W, c), np.where(noslip, 0., ch)
    return (a, b, c), (ah, bh, ch), (am, bm, cm)


def viscous_general(where, q, h, U, V, eta, zeta, Ls, dqx=None, dqy=None, slip="top"):
    """Newtonian stress of the parabolic profile at the lower wall (where=0), the upper wall (1) or averaged over the
    gap (2), all six Voigt components, with gradient terms (viscous.py:37-786 in full generality).  slip="top": only
    the upper wall slips; any other keyword: both walls (viscous.py:105, 427, 716 -- its second branch)."""
    rho, jx, jy = q[0], q[1], q[2]
    h0, hx, hy = h[0], h[1], h[2]
    zero = np.zeros(np.broadcast(rho, h0).shape)
    dqx = [zero, zero, zero] if dqx is None else dqx
    dqy = [zero, zero, zero] if dqy is None else dqy
    lo = 0. * Ls if slip == "top" else Ls
    hi = Ls
    mu, mv = jx / rho, jy / rho
    (ua, ub, uc), uh, um = _slip_parabola(h0, U, mu, lo, hi)
    (va, vb, vc), vh, vm = _slip_parabola(h0, V, mv, lo, hi)
    w2 = {0: 0., 1: h0 * h0, 2: h0 * h0 / 3}[where]
    w1 = {0: 0., 1: h0, 2: h0 / 2}[where]
    at = lambda k: k[0] * w2 + k[1] * w1 + k[2]
    mux, muy = (dqx[1] - mu * dqx[0]) / rho, (dqy[1] - mu * dqy[0]) / rho
    mvx, mvy = (dqx[2] - mv * dqx[0]) / rho, (dqy[2] - mv * dqy[0]) / rho
    ux, uy = at(uh) * hx + at(um) * mux, at(uh) * hy + at(um) * muy
    vx, vy = at(vh) * hx + at(vm) * mvx, at(vh) * hy + at(vm) * mvy
    zf = {0: 0., 1: 2 * h0, 2: h0}[where]
    uz, vz = ua * zf + ub, va * zf + vb
    v1 = zeta + 4 / 3 * eta
    v2 = zeta - 2 / 3 * eta
    return np.array([v1 * ux + v2 * vy, v2 * ux + v1 * vy, v2 * (ux + vy), eta * vz, eta * uz, eta * (uy + vx)])


def stress_bottom(q, h, U, V, eta, zeta, Ls, dqx=None, dqy=None, slip="top"):
    """Lower-wall viscous stress, Voigt order xx,yy,zz,yz,xz,xy (viscous.py:37-278)."""
    if slip != "top" or dqx is not None or dqy is not None:
        return viscous_general(0, q, h, U, V, eta, zeta, Ls, dqx, dqy, slip)
    rho, jx, jy = q[0], q[1], q[2]
    h0, hx, hy = h[0], h[1], h[2]
    v1 = zeta + 4 / 3 * eta
    v2 = zeta - 2 / 3 * eta
    tau = np.zeros((6,) + np.broadcast(rho, h0, Ls).shape)
    if slip == "top":       # viscous.py:88-105
        tau[3] = 2 * eta * (-6 * Ls * V * rho + 6 * Ls * jy - 2 * V * h0 * rho + 3 * h0 * jy) / (h0 * rho * (4 * Ls + h0))
        tau[4] = 2 * eta * (-6 * Ls * U * rho + 6 * Ls * jx - 2 * U * h0 * rho + 3 * h0 * jx) / (h0 * rho * (4 * Ls + h0))
        return tau
    raise AssertionError('unreachable')


def stress_top(q, h, U, V, eta, zeta, Ls, dqx=None, dqy=None, slip="top"):
    """Upper-wall viscous stress, Voigt order (viscous.py:281-609)."""
    if slip != "top" or dqx is not None or dqy is not None:
        return viscous_general(1, q, h, U, V, eta, zeta, Ls, dqx, dqy, slip)
    rho, jx, jy = q[0], q[1], q[2]
    h0, hx, hy = h[0], h[1], h[2]
    v1 = zeta + 4 / 3 * eta
    v2 = zeta - 2 / 3 * eta
    tau = np.zeros((6,) + np.broadcast(rho, h0, Ls).shape)
    if slip == "top":       # viscous.py:333-426
        den = rho**2 * (16 * Ls**2 + 8 * Ls * h0 + h0**2)

        def norm(va, vb):
            return (-3 * Ls * U * hx * rho**2 * va - 3 * Ls * V * hy * rho**2 * vb
                    + 9 * Ls * hx * rho * jx * va + 9 * Ls * hy * rho * jy * vb
                    - U * h0 * hx * rho**2 * va - V * h0 * hy * rho**2 * vb
                    + 3 * h0 * hx * rho * jx * va + 3 * h0 * hy * rho * jy * vb)
        tau[0] = 2 * norm(v1, v2) / den
        tau[1] = 2 * norm(v2, v1) / den
        tau[2] = 2 * v2 * norm(1., 1.) / den
        tau[3] = 2 * eta * (V * rho - 3 * jy) / (rho * (4 * Ls + h0))
        tau[4] = 2 * eta * (U * rho - 3 * jx) / (rho * (4 * Ls + h0))
        tau[5] = 2 * eta * (-3 * Ls * U * hy * rho**2 - 3 * Ls * V * hx * rho**2
                            + 9 * Ls * hx * rho * jy + 9 * Ls * hy * rho * jx
                            - U * h0 * hy * rho**2 - V * h0 * hx * rho**2
                            + 3 * h0 * hx * rho * jy + 3 * h0 * hy * rho * jx) / den
        return tau
    raise AssertionError('unreachable')


def stress_avg(q, h, U, V, eta, zeta, Ls, dqx=None, dqy=None, slip="top"):
    """Gap-averaged viscous stress xx, yy, xy (viscous.py:612-786)."""
    if slip not in ("top", "both"):         # viscous.py:663, 717: no third branch -- the array of zeros is returned
        return np.zeros((3,) + np.broadcast(q[0], h[0], Ls).shape)
    if slip != "top" or dqx is not None or dqy is not None:
        return viscous_general(2, q, h, U, V, eta, zeta, Ls, dqx, dqy, slip)[[0, 1, 5]]
    rho, jx, jy = q[0], q[1], q[2]
    h0, hx, hy = h[0], h[1], h[2]
    v1 = zeta + 4 / 3 * eta
    v2 = zeta - 2 / 3 * eta
    tau = np.zeros((3,) + np.broadcast(rho, h0, Ls).shape)
    den = h0 * rho**2 * (4 * Ls + h0)       # viscous.py:663-715

    def norm(va, vb):
        return (2 * Ls * U * hx * rho**2 * va + 2 * Ls * V * hy * rho**2 * vb
                - 2 * Ls * hx * rho * jx * va - 2 * Ls * hy * rho * jy * vb
                + h0 * hx * rho * jx * va + h0 * hy * rho * jy * vb)
    tau[0] = norm(v1, v2) / den
    tau[1] = norm(v2, v1) / den
    tau[2] = eta * (2 * Ls * U * hy * rho**2 + 2 * Ls * V * hx * rho**2
                    - 2 * Ls * hx * rho * jy - 2 * Ls * hy * rho * jx
                    + h0 * hx * rho * jy + h0 * hy * rho * jx) / den
    return tau
