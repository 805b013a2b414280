// Per-cell constitutive closures of the gap-averaged balance equations (fp64).
//
// Everything here is a pure function of one cell's state, usable from HIP device
// code and from host code (tests/hostcheck builds the same header with g++).
//
// Reference arithmetic (paths relative to the reference root):
//   GaPFlow/models/pressure.py:79-325   equations of state
//   GaPFlow/models/sound.py:84-329      dp/drho
//   GaPFlow/models/viscous.py:88-105    lower-wall stress, slip = "top"
//   GaPFlow/models/viscous.py:333-426   upper-wall stress, slip = "top"
//   GaPFlow/models/viscous.py:663-715   gap-averaged stress, slip = "top"
//   GaPFlow/models/viscosity.py:150-262 piezo-viscosity
//   GaPFlow/integrate.py:80-198         flux vectors and source term
//
// The solver never passes grad(q) to the stress functions (stress.py:328-344), so the
// terms multiplying dqx/dqy vanish identically and the polynomials factor:
//   D  = 4 Ls + h
//   Bx = 2 Ls U rho + (h - 2 Ls) jx          (avg)       By likewise with V, jy
//   Ax = (3 Ls + h)(3 jx - U rho)            (top wall)  Ay likewise
//   tau_xx     = (v1 hx Bx + v2 hy By) / (h rho D)
//   tau_xx^top = 2 (v1 hx Ax + v2 hy Ay) / (rho D^2)     ...
// which is the same rational function as the reference's expanded form (checked against
// the reference's own outputs in tests/golden/leaf_closures.npz to ~1e-15).
#pragma once

#include <math.h>

#if defined(__HIPCC__)
#define GPF_HD __host__ __device__ __forceinline__
#else
#define GPF_HD inline
#endif

namespace gpf {

enum { EOS_DH = 0, EOS_PL = 1, EOS_VDW = 2, EOS_MT = 3, EOS_CUBIC = 4, EOS_BWR = 5, EOS_BAYADA = 6 };
enum { PIEZO_NONE = 0, PIEZO_BARUS = 1, PIEZO_ROELANDS = 2, PIEZO_DUKLER = 3, PIEZO_MCADAMS = 4 };
enum { THIN_NONE = 0, THIN_EYRING = 1, THIN_CARREAU = 2 };

// Material + kinematic constants, preprocessed on the host (see make_phys in api.hip).
struct Phys {
    double U, V, eta, zeta;
    double v1, v2;      // zeta + 4/3 eta, zeta - 2/3 eta for the constant-viscosity case
    double inv_dx, inv_dy;
    int eos, piezo;
    double e[12];       // EOS constants, meaning per EOS documented in make_phys
    double x[32];       // BWR: temperature-folded polynomial coefficients
    double pz[4];       // piezo-viscosity constants
    int thinning;
    double th[4];       // shear-thinning constants: Eyring tauE | Carreau mu_inf, lam, a, N
};

// Reciprocal.  On gfx950 an IEEE f64 division expands to ~15 VALU instructions (div_scale, rcp,
// four fma, div_fmas, div_fixup); v_rcp_f64 (24 good bits) + two Newton steps is 5 and was
// measured at 0 ulp from 1.0/x over 2^20 random operands spanning 2^-60..2^60 on MI355X.  All
// operands here (h, rho, 4Ls+h, C2-rho/rho0 ...) are far from the denormal range.
GPF_HD double rcp(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(x);
    double e = fma(-x, r, 1.0);
    r = fma(r, e, r);
    e = fma(-x, r, 1.0);
    return fma(r, e, r);
#else
    return 1.0 / x;
#endif
}

// ---- viscous stresses of the general slip model (stateless operator gpf_viscous_stress) ----------------------
// The in-plane velocity across the gap is the parabola u(z) = a z^2 + b z + c fixed by
//     u(0) = W + lo u'(0)      lower wall moving with W, Navier slip length lo
//     u(h) = -hi u'(h)         upper wall at rest, slip length hi
//     1/h int_0^h u dz = m     m = j / rho, the gap-averaged velocity
// (the model behind GaPFlow/models/viscous.py and profiles.py: slip="top" is lo = 0, hi = Ls; every other keyword
// takes viscous.py's second branch, lo = hi = Ls).  With D = h^2 + 4 h (lo + hi) + 12 lo hi:
//     a = [3 (h + 2 hi) W - 6 (h + lo + hi) m] / (h D),  b = [-4 (h + 3 hi) W + 6 (h + 2 hi) m] / D,
//     c = [h (h + 4 hi) W + 6 lo (h + 2 hi) m] / D
// The Newtonian stress needs du/dz = 2 a z + b and the in-plane derivatives at fixed z,
//     du/dx = (a_h z^2 + b_h z + c_h) dh/dx + (a_m z^2 + b_m z + c_m) dm/dx,   dm/dx = (dj/dx - m drho/dx) / rho
// evaluated at z = 0 (lower wall), z = h (upper wall) and averaged over the gap.
struct Parabola { double a, b, c, ah, bh, ch, am, bm, cm; };

GPF_HD Parabola slip_parabola(double h, double W, double m, double lo, double hi) {
    const double D = h * h + 4.0 * h * (lo + hi) + 12.0 * lo * hi, Dh = 2.0 * h + 4.0 * (lo + hi);
    const double iD = 1.0 / D, ihD = 1.0 / (h * D);
    const double Na = 3.0 * (h + 2.0 * hi) * W - 6.0 * (h + lo + hi) * m;
    const double Nb = -4.0 * (h + 3.0 * hi) * W + 6.0 * (h + 2.0 * hi) * m;
    const double Nc = h * (h + 4.0 * hi) * W + 6.0 * lo * (h + 2.0 * hi) * m;
    Parabola p;
    p.a = Na * ihD; p.b = Nb * iD; p.c = Nc * iD;
    p.ah = ((3.0 * W - 6.0 * m) * (h * D) - Na * (D + h * Dh)) * ihD * ihD;
    p.bh = ((-4.0 * W + 6.0 * m) * D - Nb * Dh) * iD * iD;
    p.ch = (((2.0 * h + 4.0 * hi) * W + 6.0 * lo * m) * D - Nc * Dh) * iD * iD;
    p.am = -6.0 * (h + lo + hi) * ihD; p.bm = 6.0 * (h + 2.0 * hi) * iD; p.cm = 6.0 * lo * (h + 2.0 * hi) * iD;
    if (lo == 0.0) { p.c = W; p.ch = 0.0; }     // u(0) = W exactly: no rounding residue in the lower wall's normal stresses
    return p;
}

// where: 0 lower wall, 1 upper wall, 2 gap average.  out: xx, yy, zz, yz, xz, xy (Voigt order of viscous.py:85).
GPF_HD void viscous_general(int where, const double q[3], const double hh[3], const double dqx[3], const double dqy[3],
                            double U, double V, double eta, double zeta, double lo, double hi, double out[6]) {
    const double h = hh[0], irho = 1.0 / q[0];
    const double mu = q[1] * irho, mv = q[2] * irho;
    const Parabola pu = slip_parabola(h, U, mu, lo, hi), pv = slip_parabola(h, V, mv, lo, hi);
    // weights of (z^2, z, 1) at the requested place
    const double w2 = where == 0 ? 0.0 : (where == 1 ? h * h : h * h / 3.0);
    const double w1 = where == 0 ? 0.0 : (where == 1 ? h : 0.5 * h);
    auto at = [&](double c2, double c1, double c0) { return c2 * w2 + c1 * w1 + c0; };
    const double mux = (dqx[1] - mu * dqx[0]) * irho, muy = (dqy[1] - mu * dqy[0]) * irho;
    const double mvx = (dqx[2] - mv * dqx[0]) * irho, mvy = (dqy[2] - mv * dqy[0]) * irho;
    const double ux = at(pu.ah, pu.bh, pu.ch) * hh[1] + at(pu.am, pu.bm, pu.cm) * mux;
    const double uy = at(pu.ah, pu.bh, pu.ch) * hh[2] + at(pu.am, pu.bm, pu.cm) * muy;
    const double vx = at(pv.ah, pv.bh, pv.ch) * hh[1] + at(pv.am, pv.bm, pv.cm) * mvx;
    const double vy = at(pv.ah, pv.bh, pv.ch) * hh[2] + at(pv.am, pv.bm, pv.cm) * mvy;
    // du/dz = 2 a z + b: b at the lower wall, 2 a h + b at the upper one, a h + b on average
    const double zf = where == 0 ? 0.0 : (where == 1 ? 2.0 * h : h);
    const double uz = pu.a * zf + pu.b, vz = pv.a * zf + pv.b;
    const double v1 = zeta + (4.0 / 3.0) * eta, v2 = zeta - (2.0 / 3.0) * eta;
    out[0] = v1 * ux + v2 * vy;
    out[1] = v2 * ux + v1 * vy;
    out[2] = v2 * (ux + vy);
    out[3] = eta * vz;
    out[4] = eta * uz;
    out[5] = eta * (uy + vx);
}

// ---- pow for the equations of state ---------------------------------------------------------------------------
// The library pow costs a few hundred fp64 operations (extended-precision log, special cases): with three of them per
// cell-update the Murnaghan-Tait and power-law steps were compute-bound at three times the Dowson-Higginson step time.
// For the operands an equation of state sees (x > 0 a density ratio, |y ln x| far below 700) a plain exp(y ln x) is
// accurate to a few 1e-16 |y ln x|: ln through the mantissa in [1/sqrt2, sqrt2) and the atanh series (11 terms), exp by
// reduction with ln 2 and a degree-13 polynomial -- about 45 operations.  The *_series forms are plain C++ (compiled for the
// host too: tests/hostcheck compares them with std::log / std::exp / std::pow, special values included); the device code
// calls them, the host statement of the closures keeps the C library.
// Special values follow np.log / np.exp / np.power, because a state that blows up must END the run (q_is_valid,
// problem.py:319-332) instead of carrying on with finite garbage: log(0) = -inf, log(x < 0) = NaN, exp(NaN) = NaN,
// exp(t > 709.78) = inf, x^0 = 1, 0^y = 0 / inf for y > 0 / y < 0.  (A negative base gives NaN: np.power does the same for
// the fractional exponents an equation of state has.)
GPF_HD double fast_log_series(double x) {
    int e;
    double m = __builtin_frexp(x, &e);                      // m in [0.5, 1)
    const bool low = m < 0.70710678118654752;
    m = low ? m + m : m;                                    // [1/sqrt2, sqrt2)
    e = low ? e - 1 : e;
    const double z = (m - 1.0) * rcp(m + 1.0);              // |z| <= 0.1716
    const double w = z * z;
    double p = 1.0 / 23.0;
    p = fma(p, w, 1.0 / 21.0); p = fma(p, w, 1.0 / 19.0); p = fma(p, w, 1.0 / 17.0); p = fma(p, w, 1.0 / 15.0);
    p = fma(p, w, 1.0 / 13.0); p = fma(p, w, 1.0 / 11.0); p = fma(p, w, 1.0 / 9.0); p = fma(p, w, 1.0 / 7.0);
    p = fma(p, w, 1.0 / 5.0); p = fma(p, w, 1.0 / 3.0);
    const double lm = fma(z + z, p * w, z + z);             // ln m = 2 z (1 + w p)
    const double de = (double)e;
    double r = fma(de, 0.69314718036912382, fma(de, 1.9082149292705877e-10, lm));      // e ln2 (hi + lo) + ln m
    r = x == __builtin_inf() ? x : r;
    r = x == 0.0 ? -__builtin_inf() : r;
    return x < 0.0 ? __builtin_nan("") : r;                 // (a NaN argument comes out of the series as NaN)
}

GPF_HD double fast_exp_series(double t) {
    const double tc = fmin(fmax(t, -745.2), 709.8);         // (fmin / fmax drop a NaN: restored below)
    const double n = __builtin_rint(tc * 1.4426950408889634);
    double f = fma(n, -0.69314718036912382, tc);
    f = fma(n, -1.9082149292705877e-10, f);
    double p = 1.6059043836821613e-10;
    p = fma(p, f, 2.08767569878681e-09); p = fma(p, f, 2.505210838544172e-08); p = fma(p, f, 2.755731922398589e-07);
    p = fma(p, f, 2.7557319223985893e-06); p = fma(p, f, 2.48015873015873e-05); p = fma(p, f, 1.984126984126984e-04);
    p = fma(p, f, 1.388888888888889e-03); p = fma(p, f, 8.333333333333333e-03); p = fma(p, f, 4.1666666666666664e-02);
    p = fma(p, f, 1.6666666666666666e-01); p = fma(p, f, 0.5); p = fma(p, f, 1.0); p = fma(p, f, 1.0);
    // 2^n in two factors: n reaches -1075 and 1024, outside the exponents a single scale factor can hold
    const int ni = (int)n, h1 = ni / 2;
    double r = ldexp(ldexp(p, h1), ni - h1);
    r = t > 709.782712893384 ? __builtin_inf() : r;
    r = t < -745.13321910194122 ? 0.0 : r;
    return t != t ? t : r;
}

GPF_HD double pow_pos_series(double x, double y) {
    const double r = fast_exp_series(y * fast_log_series(x));       // x = 0: exp(-+inf) = 0 / inf
    return y == 0.0 ? 1.0 : r;                                      // (0 * inf would be NaN)
}

GPF_HD double fast_log(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return fast_log_series(x);
#else
    return log(x);
#endif
}

GPF_HD double fast_exp(double t) {
#if defined(__HIP_DEVICE_COMPILE__)
    return fast_exp_series(t);
#else
    return exp(t);
#endif
}

// x^y for x >= 0 (a negative base: NaN, like np.power with a fractional exponent)
GPF_HD double pow_pos(double x, double y) {
#if defined(__HIP_DEVICE_COMPILE__)
    return pow_pos_series(x, y);
#else
    return x < 0.0 ? __builtin_nan("") : pow(x, y);
#endif
}

// ---- equations of state -------------------------------------------------------------------

template <int EOS>
GPF_HD double eos_pressure(double rho, const Phys& P) {
    if (EOS == EOS_DH) {
        // e0=rho0 e1=P0 e2=C1 e3=C2 e4=0.99*C2*rho0 (clamp) e5=1/rho0
        double r = fmin(rho, P.e[4]);
        double s = r * P.e[5];
        return P.e[1] + (P.e[2] * (s - 1.0)) * rcp(P.e[3] - s);
    } else if (EOS == EOS_PL) {
        // e0=rho0 e1=P0 e2=alpha e3=1/(1-alpha/2) e5=1/rho0 e6=-2/(alpha-2) e7=-2 P0/(alpha-2)
        return P.e[1] * pow_pos(rho * P.e[5], P.e[3]);
    } else if (EOS == EOS_VDW) {
        // e0=1000/M e1=R*T e2=a/10 e3=b/1000
        double md = rho * P.e[0];
        return P.e[1] * md * rcp(1.0 - P.e[3] * md) - P.e[2] * md * md;
    } else if (EOS == EOS_MT) {
        // e0=rho0 e1=P0 e2=K e3=n e5=1/rho0 e6=K/n e7=K/rho0
        return P.e[6] * (pow_pos(rho * P.e[5], P.e[3]) - 1.0) + P.e[1];
    } else if (EOS == EOS_CUBIC) {
        return ((P.e[0] * rho + P.e[1]) * rho + P.e[2]) * rho + P.e[3];
    } else if (EOS == EOS_BWR) {
        // x[0..8]: coefficients of rho^1..rho^9 ; x[9..14]: of rho^3,5,..,13 inside exp(-gamma rho^2); e0=gamma
        double r2 = rho * rho;
        double poly = P.x[8];
        for (int k = 7; k >= 0; --k) poly = poly * rho + P.x[k];
        poly *= rho;
        double ex = P.x[14];
        for (int k = 13; k >= 9; --k) ex = ex * r2 + P.x[k];
        ex *= r2 * rho;
        return poly + fast_exp(-P.e[0] * r2) * ex;
    } else {
        // Bayada-Chupin: e0=rho_l e1=rho_v e2=c_l^2 e3=c_v^2 e4=N e5=Pcav e6=1/(rho_v-rho_l)
        double alpha = (rho - P.e[0]) * P.e[6];
        if (alpha < 0.0) return P.e[5] + (rho - P.e[0]) * P.e[2];
        if (alpha <= 1.0) {
            double den = P.e[0] * (P.e[1] * P.e[3] * (1.0 - alpha) + P.e[0] * P.e[2] * alpha);
            return P.e[5] + P.e[4] * fast_log(P.e[1] * P.e[3] * rho * rcp(den));
        }
        return P.e[3] * rho;
    }
}

// dp/drho (the square of the sound speed); NaN/negative values propagate like np.sqrt would.
template <int EOS>
GPF_HD double eos_c2(double rho, const Phys& P) {
    if (EOS == EOS_DH) {
        // C1 rho0 (C2-1) / rho^2 / (C2 rho0/rho - 1)^2 = C1 rho0 (C2-1) / (C2 rho0 - rho)^2, *unclamped* rho (sound.py:109)
        double it = rcp(P.e[7] - rho);                // e7 = C2*rho0
        return P.e[6] * (it * it);                    // e6 = C1*rho0*(C2-1)
    } else if (EOS == EOS_PL) {
        // -2 P0 (rho/rho0)^(-2/(alpha-2)) / ((alpha-2) rho)
        return P.e[7] * pow_pos(rho * P.e[5], P.e[6]) * rcp(rho);
    } else if (EOS == EOS_VDW) {
        double md = rho * P.e[0];
        double t = 1.0 - P.e[3] * md;
        return P.e[1] * rcp(t * t) - 2.0 * P.e[2] * md;
    } else if (EOS == EOS_MT) {
        // K / rho0^n rho^(n-1) = (K / rho0) (rho / rho0)^(n-1): one pow, of a ratio near one
        return P.e[7] * pow_pos(rho * P.e[5], P.e[3] - 1.0);
    } else if (EOS == EOS_CUBIC) {
        return (3.0 * P.e[0] * rho + 2.0 * P.e[1]) * rho + P.e[2];
    } else if (EOS == EOS_BWR) {
        double r2 = rho * rho;
        // d/drho of sum_{k=1..9} x[k-1] rho^k
        double dpoly = 9.0 * P.x[8];
        for (int k = 7; k >= 0; --k) dpoly = dpoly * rho + (double)(k + 1) * P.x[k];
        // exp part: E(rho) = sum_{m=0..5} x[9+m] rho^(3+2m)
        double ex = P.x[14], dex = 13.0 * P.x[14];
        for (int m = 4; m >= 0; --m) {
            ex = ex * r2 + P.x[9 + m];
            dex = dex * r2 + (double)(3 + 2 * m) * P.x[9 + m];
        }
        ex *= r2 * rho;     // E
        dex *= r2;          // E'
        double g = fast_exp(-P.e[0] * r2);
        return dpoly + g * dex - 2.0 * rho * P.e[0] * g * ex;
    } else {
        double alpha = (rho - P.e[0]) * P.e[6];
        if (alpha < 0.0) return P.e[2];
        if (alpha <= 1.0)
            return P.e[1] * P.e[0] * (P.e[3] * P.e[2]) * rcp((alpha * P.e[0] * P.e[2] + (1.0 - alpha) * P.e[1] * P.e[3]) * rho);
        return P.e[3];
    }
}

// ---- viscosity ----------------------------------------------------------------------------

// `arg` is the pressure, or the density when the EOS is Bayada-Chupin (stress.py:307-310)
GPF_HD double piezo_eta(double eta0, double arg, const Phys& P) {
    switch (P.piezo) {
    case PIEZO_BARUS:    return eta0 * fast_exp(P.pz[0] * arg);
    case PIEZO_ROELANDS: return eta0 * fast_exp(P.pz[3] * (-1.0 + pow_pos(1.0 + arg / P.pz[1], P.pz[2])));   // pz3 = ln(mu0/mu_inf)
    case PIEZO_DUKLER: {
        double a = (arg - P.pz[1]) / (P.pz[2] - P.pz[1]);
        return a * P.pz[0] + (1.0 - a) * eta0;
    }
    case PIEZO_MCADAMS: {
        double a = (arg - P.pz[1]) / (P.pz[2] - P.pz[1]);
        double M = a * P.pz[2] / arg;
        return P.pz[0] * eta0 / (eta0 * M + P.pz[0] * (1.0 - M));
    }
    default: return eta0;
    }
}

// Shear-thinning factor eta/mu0 at the mean wall shear rate of the Newtonian profile
// (viscosity.py:69-141, 265-318; stress.py:314-324 passes U and V as the two wall velocities).
// mean of the |wall shear rates| of the Newtonian profile between walls moving with u1, u2 (viscosity.py:110-141)
GPF_HD double shear_rate_avg(double dp_dx, double dp_dy, double h, double u1, double u2, double mu) {
    const double gp = hypot(dp_dx, dp_dy);
    const double du_p = h * gp / (2.0 * mu);
    const double du_c = (u2 - u1) / h;
    return (fabs(du_p + du_c) + fabs(-du_p + du_c)) / 2.0;
}

// mu(shear rate) / mu0 (viscosity.py:69-96, 262-318)
GPF_HD double thinning_factor(double rate, double mu0, const Phys& P) {
    if (P.thinning == THIN_EYRING) {
        const double tau0 = mu0 * rate;
        return P.th[0] / tau0 * asinh(tau0 / P.th[0]);
    }
    if (P.thinning == THIN_CARREAU) {
        const double mu = P.th[0] + (mu0 - P.th[0]) * pow(1.0 + pow(P.th[1] * rate, P.th[2]), (P.th[3] - 1.0) / P.th[2]);
        return mu / mu0;
    }
    return 1.0;
}

GPF_HD double thinning_eta(double mu0, double dp_dx, double dp_dy, double h, const Phys& P) {
    if (P.thinning != THIN_EYRING && P.thinning != THIN_CARREAU) return mu0;
    return mu0 * thinning_factor(shear_rate_avg(dp_dx, dp_dy, h, P.U, P.V, mu0), mu0, P);
}

// ---- one cell: fluxes and source ------------------------------------------------------------

struct CellIn {
    double rho, jx, jy;     // q
    double h, hx, hy;       // gap height and slopes
    double Ls;              // slip length ('extra' field)
};

// F_x = (jx, p + tau_xx, tau_xy), F_y = (jy, tau_xy, p + tau_yy)  (integrate.py:133-198)
struct CellFlux {
    double fx1, fx2;        // p + tau_xx, tau_xy   (fx0 = jx)
    double fy2;             // p + tau_yy           (fy0 = jy, fy1 = tau_xy = fx2)
    double s0, s1, s2;      // source term (integrate.py:117-130)
    double p;
};

// Gap-averaged stress, both wall stresses, pressure: everything a stage needs from one cell.
// Topography-only reciprocals of a cell; both stages of a step evaluate their closure on the same
// topography, so the stencil kernel computes these once per cell and step.
struct TopoRcp {
    double ih, iD;          // 1/h, 1/(4 Ls + h)
};

template <bool HAS_LS>
GPF_HD TopoRcp topo_rcp(const CellIn& c) {
    TopoRcp t;
    t.ih = rcp(c.h);
    t.iD = HAS_LS ? rcp(4.0 * c.Ls + c.h) : t.ih;
    return t;
}

// HAS_LS = false is the Ls == 0 specialisation (D = h, B = h j, ...), the common case; PIEZO = false
// takes the constant viscosities from Phys (no per-cell exp/pow).
template <int EOS, bool WITH_SOURCE, bool HAS_LS = true, bool PIEZO = true>
GPF_HD void cell_closure(const CellIn& c, const TopoRcp& t, const Phys& P, CellFlux& o) {
#ifdef GPF_STUB_CLOSURE     // diagnostic build: memory-access pattern of the step without its arithmetic
    o.p = c.rho; o.fx1 = c.rho + c.h; o.fx2 = c.jx + c.hx; o.fy2 = c.jy + c.hy;
    o.s0 = c.h; o.s1 = c.hx; o.s2 = c.hy;
    return;
#endif
    const double U = P.U, V = P.V;
    const double p = eos_pressure<EOS>(c.rho, P);
    const double eta = (!PIEZO || P.piezo == PIEZO_NONE) ? P.eta : piezo_eta(P.eta, (EOS == EOS_BAYADA) ? c.rho : p, P);
    const double v1 = PIEZO ? P.zeta + (4.0 / 3.0) * eta : P.v1;
    const double v2 = PIEZO ? P.zeta - (2.0 / 3.0) * eta : P.v2;

    const double ih = t.ih, iD = t.iD, ir = rcp(c.rho);
    const double irD = ir * iD;         // 1/(rho D)
    const double ihrD = ih * irD;       // 1/(h rho D)

    const double Urho = U * c.rho, Vrho = V * c.rho;
    const double Bx = HAS_LS ? 2.0 * c.Ls * Urho + (c.h - 2.0 * c.Ls) * c.jx : c.h * c.jx;
    const double By = HAS_LS ? 2.0 * c.Ls * Vrho + (c.h - 2.0 * c.Ls) * c.jy : c.h * c.jy;
    const double hxBx = c.hx * Bx, hyBy = c.hy * By;
    const double txx = (v1 * hxBx + v2 * hyBy) * ihrD;
    const double tyy = (v2 * hxBx + v1 * hyBy) * ihrD;
    const double txy = eta * (c.hy * Bx + c.hx * By) * ihrD;

    o.p = p;
    o.fx1 = p + txx;
    o.fx2 = txy;
    o.fy2 = p + tyy;

    if (WITH_SOURCE) {
        const double w = HAS_LS ? 3.0 * c.Ls + c.h : c.h;
        const double gx = 3.0 * c.jx - Urho, gy = 3.0 * c.jy - Vrho;
        const double Ax = w * gx, Ay = w * gy;
        const double t2 = 2.0 * irD * iD;                         // 2/(rho D^2)
        const double hxAx = c.hx * Ax, hyAy = c.hy * Ay;
        const double txx_t = (v1 * hxAx + v2 * hyAy) * t2;
        const double tyy_t = (v2 * hxAx + v1 * hyAy) * t2;
        const double txy_t = eta * (c.hy * Ax + c.hx * Ay) * t2;
        const double e2 = 2.0 * eta * irD;                        // 2 eta/(rho D)
        const double txz_t = -e2 * gx;                            // 2 eta (U rho - 3 jx)/(rho D)
        const double tyz_t = -e2 * gy;
        // bottom: 2 eta [(6Ls+3h) j - (6Ls+2h) W rho] / (h rho D)
        const double a3 = HAS_LS ? 6.0 * c.Ls + 3.0 * c.h : 3.0 * c.h;
        const double a2 = HAS_LS ? 6.0 * c.Ls + 2.0 * c.h : 2.0 * c.h;
        const double e2h = e2 * ih;
        const double txz_b = e2h * (a3 * c.jx - a2 * Urho);
        const double tyz_b = e2h * (a3 * c.jy - a2 * Vrho);

        o.s0 = -(c.jx * c.hx + c.jy * c.hy) * ih;
        o.s1 = ((txx - txx_t) * c.hx + (txy - txy_t) * c.hy + txz_t - txz_b) * ih;
        o.s2 = ((txy - txy_t) * c.hx + (tyy - tyy_t) * c.hy + tyz_t - tyz_b) * ih;
    }
}

template <int EOS, bool WITH_SOURCE, bool HAS_LS = true, bool PIEZO = true>
GPF_HD void cell_closure(const CellIn& c, const Phys& P, CellFlux& o) {
    cell_closure<EOS, WITH_SOURCE, HAS_LS, PIEZO>(c, topo_rcp<HAS_LS>(c), P, o);
}

// ---- no slip-length field, constant viscosity (the common case): D = h ----------------------------------------------
// With Ls = 0 the polynomials of viscous.py's slip-top branch lose their h-dependence up to the slopes per gap height
// a = hx/h, b = hy/h, and everything is linear in the velocities m = j / rho:
//   tau_xx = v1 a m_x + v2 b m_y        tau_yy = v2 a m_x + v1 b m_y        tau_xy = eta (b m_x + a m_y)
//   tau - tau^top:  d_xx = v1 a P_x + v2 b P_y,  d_yy = v2 a P_x + v1 b P_y,  d_xy = eta (b P_x + a P_y),  P = 2 W - 5 m
//   tau_xz^top - tau_xz^bot = -2 eta (6 m_x - 3 U)/h                         (yz alike)
//   s0 = -(a jx + b jy),   s1 = d_xx a + d_xy b - E (6 m_x - 3 U),   s2 = d_xy a + d_yy b - E (6 m_y - 3 V),   E = 2 eta/h^2
// -- the same rational functions as cell_closure<.., HAS_LS = false, PIEZO = false> (tests/hostcheck compares the two)
// in two thirds of the arithmetic; a, b and 1/h are formed once per cell and step and serve both stages.
struct GapCoef { double ih, a, b; };

GPF_HD GapCoef gap_coefficients(double h, double hx, double hy) {
    GapCoef g;
    g.ih = rcp(h);
    g.a = hx * g.ih;
    g.b = hy * g.ih;
    return g;
}

// p(rho) and 1/rho together.  Dowson-Higginson divides by (C2 - rho_c/rho0), the velocities by rho: ONE reciprocal of the product
// serves both (a reciprocal with its two Newton steps costs as much as eight multiplications on the fp64 pipe, and the two
// divisions were a quarter of a closure's arithmetic); the other laws keep their own forms.
template <int EOS>
GPF_HD double pressure_and_inverse_density(double rho, const Phys& P, double& inv_rho) {
    if (EOS == EOS_DH) {
        const double r = fmin(rho, P.e[4]);             // as eos_pressure<EOS_DH>
        const double s = r * P.e[5];
        const double t = P.e[3] - s;
        const double inv = rcp(rho * t);
        inv_rho = inv * t;
        return P.e[1] + (P.e[2] * (s - 1.0)) * (inv * rho);
    }
    inv_rho = rcp(rho);
    return eos_pressure<EOS>(rho, P);
}

template <int EOS>
GPF_HD void cell_closure_ls0(double rho, double jx, double jy, const GapCoef& g, const Phys& P, CellFlux& o) {
#ifdef GPF_STUB_CLOSURE     // diagnostic build: memory-access pattern of the step without its arithmetic
    o.p = rho; o.fx1 = rho + g.ih; o.fx2 = jx + g.a; o.fy2 = jy + g.b; o.s0 = g.ih; o.s1 = g.a; o.s2 = g.b;
    return;
#endif
    double ir;
    const double p = pressure_and_inverse_density<EOS>(rho, P, ir);
    const double mx = jx * ir, my = jy * ir;
    const double ax = g.a * mx, by = g.b * my;
    const double txx = fma(P.v1, ax, P.v2 * by);
    const double tyy = fma(P.v2, ax, P.v1 * by);
    const double txy = P.eta * fma(g.b, mx, g.a * my);
    o.p = p;
    o.fx1 = p + txx;
    o.fx2 = txy;
    o.fy2 = p + tyy;
    const double px = fma(-5.0, mx, 2.0 * P.U), py = fma(-5.0, my, 2.0 * P.V);
    const double apx = g.a * px, bpy = g.b * py;
    const double dxx = fma(P.v1, apx, P.v2 * bpy);
    const double dyy = fma(P.v2, apx, P.v1 * bpy);
    const double dxy = P.eta * fma(g.b, px, g.a * py);
    const double E = (2.0 * P.eta) * (g.ih * g.ih);
    const double wx = fma(6.0, mx, -3.0 * P.U), wy = fma(6.0, my, -3.0 * P.V);
    o.s0 = -fma(g.a, jx, g.b * jy);
    o.s1 = fma(dxx, g.a, fma(dxy, g.b, -(E * wx)));
    o.s2 = fma(dxy, g.a, fma(dyy, g.b, -(E * wy)));
}

// ---- gaps that vary along x only: h = h(x), dh/dy = 0, no slip-length field, constant viscosity -----------------
// journal, inclined, parabolic and cdc profiles (topography.py:57-130).  With D = h and hy = 0 every stress of
// viscous.py's slip-top branch is LINEAR in the velocities m = j / rho, with coefficients that depend on the row only:
//   tau_xx = A m_x,  tau_yy = B m_x,  tau_xy = C m_y          A = v1 hx/h, B = v2 hx/h, C = eta hx/h
//   tau_xx^top = 2A (3 m_x - U),  tau_xy^top = 2C (3 m_y - V)
//   tau_xz^top = -2 eta (3 m_x - U)/h,  tau_xz^bot = 2 eta (3 m_x - 2U)/h          (yz alike with m_y, V)
//   s0 = -(hx/h) jx
//   s1 = [(tau_xx - tau_xx^top) hx + tau_xz^top - tau_xz^bot]/h = -(5 A hx/h + 12 eta/h^2) m_x + U (2 A hx/h + 6 eta/h^2)
//   s2 = [(tau_xy - tau_xy^top) hx + tau_yz^top - tau_yz^bot]/h = -(5 C hx/h + 12 eta/h^2) m_y + V (2 C hx/h + 6 eta/h^2)
// -- the same rational functions as cell_closure with hy = Ls = 0 (tests/hostcheck compares the two), a third of the
// arithmetic.  The contractions are spelled out so that every kernel rounds alike.
struct RowCoef { double A, B, C, S0, S1a, S1b, S2a, S2b; };

GPF_HD RowCoef row_coefficients(double h, double hx, const Phys& P) {
    const double ih = rcp(h);
    const double g = hx * ih;
    const double e = P.eta * (ih * ih);
    RowCoef r;
    r.A = P.v1 * g; r.B = P.v2 * g; r.C = P.eta * g; r.S0 = -g;
    const double Ag = r.A * g, Cg = r.C * g;
    r.S1a = fma(-5.0, Ag, -12.0 * e); r.S1b = P.U * fma(2.0, Ag, 6.0 * e);
    r.S2a = fma(-5.0, Cg, -12.0 * e); r.S2b = P.V * fma(2.0, Cg, 6.0 * e);
    return r;
}

template <int EOS>
GPF_HD void cell_closure_xonly(double rho, double jx, double jy, const RowCoef& r, const Phys& P, CellFlux& o) {
#ifdef GPF_STUB_CLOSURE
    o.p = rho; o.fx1 = rho + r.A; o.fx2 = jx + r.C; o.fy2 = jy + r.B; o.s0 = r.S0; o.s1 = r.S1a; o.s2 = r.S2a;
    return;
#endif
    double ir;
    const double p = pressure_and_inverse_density<EOS>(rho, P, ir);
    const double mx = jx * ir, my = jy * ir;
    o.p = p;
    o.fx1 = fma(r.A, mx, p);
    o.fx2 = r.C * my;
    o.fy2 = fma(r.B, mx, p);
    o.s0 = r.S0 * jx;
    o.s1 = fma(r.S1a, mx, r.S1b);
    o.s2 = fma(r.S2a, my, r.S2b);
}

// The full set of derived fields the reference keeps per cell (for gpf_update_closures).
struct CellFields {
    double p;
    double tau[3];          // xx, yy, xy
    double lower[6];        // Voigt xx,yy,zz,yz,xz,xy
    double upper[6];
};

// eta_override >= 0: shear viscosity already evaluated by the caller (shear thinning needs grad p)
template <int EOS>
GPF_HD void cell_fields(const CellIn& c, const Phys& P, CellFields& o, double eta_override = -1.0) {
    const double U = P.U, V = P.V;
    const double p = eos_pressure<EOS>(c.rho, P);
    const double eta = eta_override >= 0.0 ? eta_override
                     : (P.piezo == PIEZO_NONE) ? P.eta : piezo_eta(P.eta, (EOS == EOS_BAYADA) ? c.rho : p, P);
    const double v1 = P.zeta + (4.0 / 3.0) * eta;
    const double v2 = P.zeta - (2.0 / 3.0) * eta;
    const double D = 4.0 * c.Ls + c.h;
    const double den = c.h * c.rho * D;
    const double Urho = U * c.rho, Vrho = V * c.rho;
    const double hm = c.h - 2.0 * c.Ls;
    const double Bx = 2.0 * c.Ls * Urho + hm * c.jx;
    const double By = 2.0 * c.Ls * Vrho + hm * c.jy;
    o.p = p;
    o.tau[0] = (v1 * c.hx * Bx + v2 * c.hy * By) / den;
    o.tau[1] = (v2 * c.hx * Bx + v1 * c.hy * By) / den;
    o.tau[2] = eta * (c.hy * Bx + c.hx * By) / den;
    const double w = 3.0 * c.Ls + c.h;
    const double gx = 3.0 * c.jx - Urho, gy = 3.0 * c.jy - Vrho;
    const double Ax = w * gx, Ay = w * gy;
    const double dT = c.rho * D * D;
    o.upper[0] = 2.0 * (v1 * c.hx * Ax + v2 * c.hy * Ay) / dT;
    o.upper[1] = 2.0 * (v2 * c.hx * Ax + v1 * c.hy * Ay) / dT;
    o.upper[2] = 2.0 * v2 * (c.hx * Ax + c.hy * Ay) / dT;
    o.upper[3] = -2.0 * eta * gy / (c.rho * D);
    o.upper[4] = -2.0 * eta * gx / (c.rho * D);
    o.upper[5] = 2.0 * eta * (c.hy * Ax + c.hx * Ay) / dT;
    const double a3 = 6.0 * c.Ls + 3.0 * c.h, a2 = 6.0 * c.Ls + 2.0 * c.h;
    o.lower[0] = o.lower[1] = o.lower[2] = o.lower[5] = 0.0;
    o.lower[3] = 2.0 * eta * (a3 * c.jy - a2 * Vrho) / den;
    o.lower[4] = 2.0 * eta * (a3 * c.jx - a2 * Urho) / den;
}

}  // namespace gpf
