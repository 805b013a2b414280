// Host build of the per-cell closures (gapflow_amd/csrc/closures.hpp + phys_setup.hpp) for the CPU sanitizer test:
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all closures_host.cpp -o closures_host
// Reads little-endian doubles from stdin in the order written by tests/test_hostcheck.py, evaluates
//   * eos_pressure / sqrt(eos_c2) for the seven equations of state,
//   * cell_fields (gap-averaged stress, both wall stresses) for the slip-top branch,
//   * the fused kernel's specialised cell_closure against cell_fields + integrate.py's source formula,
//   * piezo-viscosity, shear-thinning factor and mean wall shear rate,
// and writes the results as doubles to stdout.  No GPU, no HIP: the same header the kernels compile.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../gapflow_amd/csrc/phys_setup.hpp"

using namespace gpf;

static std::vector<double> rd(size_t n) {
    std::vector<double> v(n);
    if (n && std::fread(v.data(), sizeof(double), n, stdin) != n) { std::fprintf(stderr, "short read\n"); std::exit(2); }
    return v;
}
static void wr(const std::vector<double>& v) { std::fwrite(v.data(), sizeof(double), v.size(), stdout); }

template <int EOS>
static void eos_table(const Phys& P, const std::vector<double>& rho) {
    std::vector<double> p(rho.size()), c(rho.size());
    for (size_t i = 0; i < rho.size(); ++i) { p[i] = eos_pressure<EOS>(rho[i], P); c[i] = std::sqrt(eos_c2<EOS>(rho[i], P)); }
    wr(p); wr(c);
}

int main() {
    const double zero4[4] = {0, 0, 0, 0};
    // ---- 1. equations of state: [n] then per EOS: 8 parameters + n densities ----
    const size_t n = (size_t)rd(1)[0];
    for (int eos = 0; eos < 7; ++eos) {
        const std::vector<double> par = rd(8), rho = rd(n);
        Phys P;
        setup_phys(P, 0.1, 0.0, 0.1, 0.0, 1e-5, 1e-5, eos, par.data(), PIEZO_NONE, zero4, THIN_NONE, zero4);
        switch (eos) {
        case EOS_DH: eos_table<EOS_DH>(P, rho); break;
        case EOS_PL: eos_table<EOS_PL>(P, rho); break;
        case EOS_VDW: eos_table<EOS_VDW>(P, rho); break;
        case EOS_MT: eos_table<EOS_MT>(P, rho); break;
        case EOS_CUBIC: eos_table<EOS_CUBIC>(P, rho); break;
        case EOS_BWR: eos_table<EOS_BWR>(P, rho); break;
        default: eos_table<EOS_BAYADA>(P, rho); break;
        }
    }
    // ---- 2. stresses: [m, U, V, eta, zeta] + DH parameters(8) + q(3m) h(3m) Ls(m) ----
    {
        const std::vector<double> hd = rd(5), par = rd(8);
        const size_t m = (size_t)hd[0];
        const std::vector<double> q = rd(3 * m), h = rd(3 * m), Ls = rd(m);
        Phys P;
        setup_phys(P, hd[1], hd[2], hd[3], hd[4], 1e-5, 1e-5, EOS_DH, par.data(), PIEZO_NONE, zero4, THIN_NONE, zero4);
        std::vector<double> lower(6 * m), upper(6 * m), avg(3 * m), dev(m);
        for (size_t i = 0; i < m; ++i) {
            CellIn c;
            c.rho = q[i]; c.jx = q[m + i]; c.jy = q[2 * m + i]; c.h = h[i]; c.hx = h[m + i]; c.hy = h[2 * m + i]; c.Ls = Ls[i];
            CellFields f;
            cell_fields<EOS_DH>(c, P, f);
            for (int k = 0; k < 6; ++k) { lower[k * m + i] = f.lower[k]; upper[k * m + i] = f.upper[k]; }
            for (int k = 0; k < 3; ++k) avg[k * m + i] = f.tau[k];
            // the fused kernel's closure (fluxes + source) against the general fields + integrate.py:117-130
            CellFlux g;
            cell_closure<EOS_DH, true, true, true>(c, P, g);
            const double s0 = -(c.jx * c.hx + c.jy * c.hy) / c.h;
            const double s1 = ((f.tau[0] - f.upper[0]) * c.hx + (f.tau[2] - f.upper[5]) * c.hy + f.upper[4] - f.lower[4]) / c.h;
            const double s2 = ((f.tau[2] - f.upper[5]) * c.hx + (f.tau[1] - f.upper[1]) * c.hy + f.upper[3] - f.lower[3]) / c.h;
            auto rel = [](double a, double b, double scale) { return std::fabs(a - b) / (std::fabs(scale) + 1e-300); };
            const double sc = std::fabs(s1) + std::fabs(s2) + 1e-300;
            double d = rel(g.fx1, f.p + f.tau[0], f.p);
            d = std::fmax(d, rel(g.fx2, f.tau[2], std::fabs(f.tau[0]) + std::fabs(f.tau[1]) + std::fabs(f.tau[2])));
            d = std::fmax(d, rel(g.fy2, f.p + f.tau[1], f.p));
            d = std::fmax(d, rel(g.s0, s0, s0));
            d = std::fmax(d, rel(g.s1, s1, sc));
            d = std::fmax(d, rel(g.s2, s2, sc));
            // the x-only-gap closure against the general one on the same cell with hy = Ls = 0
            {
                CellIn cx = c;
                cx.hy = 0.0; cx.Ls = 0.0;
                CellFlux a, b;
                cell_closure<EOS_DH, true, false, false>(cx, P, a);
                cell_closure_xonly<EOS_DH>(cx.rho, cx.jx, cx.jy, row_coefficients(cx.h, cx.hx, P), P, b);
                const double ss = std::fabs(a.s1) + std::fabs(a.s2) + 1e-300;
                d = std::fmax(d, rel(b.fx1, a.fx1, a.fx1));
                d = std::fmax(d, rel(b.fx2, a.fx2, std::fabs(a.fx2) + std::fabs(a.fx1 - a.p) + 1e-300));
                d = std::fmax(d, rel(b.fy2, a.fy2, a.fy2));
                d = std::fmax(d, rel(b.s0, a.s0, a.s0));
                d = std::fmax(d, rel(b.s1, a.s1, ss));
                d = std::fmax(d, rel(b.s2, a.s2, ss));
            }
            // the Ls = 0 closure against the general one on the same cell with Ls = 0 (hx, hy as they are)
            {
                CellIn c0 = c;
                c0.Ls = 0.0;
                CellFlux a, b;
                cell_closure<EOS_DH, true, false, false>(c0, P, a);
                cell_closure_ls0<EOS_DH>(c0.rho, c0.jx, c0.jy, gap_coefficients(c0.h, c0.hx, c0.hy), P, b);
                const double ss = std::fabs(a.s1) + std::fabs(a.s2) + 1e-300;
                const double ts = std::fabs(a.fx2) + std::fabs(a.fx1 - a.p) + std::fabs(a.fy2 - a.p) + 1e-300;
                d = std::fmax(d, rel(b.fx1, a.fx1, a.fx1));
                d = std::fmax(d, rel(b.fx2, a.fx2, ts));
                d = std::fmax(d, rel(b.fy2, a.fy2, a.fy2));
                d = std::fmax(d, rel(b.fx1 - b.p, a.fx1 - a.p, ts));
                d = std::fmax(d, rel(b.fy2 - b.p, a.fy2 - a.p, ts));
                d = std::fmax(d, rel(b.s0, a.s0, std::fabs(c0.jx * c0.hx / c0.h) + std::fabs(c0.jy * c0.hy / c0.h)));
                d = std::fmax(d, rel(b.s1, a.s1, ss));
                d = std::fmax(d, rel(b.s2, a.s2, ss));
            }
            dev[i] = d;
        }
        wr(lower); wr(upper); wr(avg); wr(dev);
    }
    // ---- 3. viscosity laws: [k, mu0] ; 4 piezo laws: 4 parameters + k arguments ; 2 thinning laws: 4 parameters + k rates ;
    //         shear_rate_avg: [u1, u2] + gx(k) gy(k) h(k) ----
    {
        const std::vector<double> hd = rd(2);
        const size_t k = (size_t)hd[0];
        const double mu0 = hd[1];
        const double dh[8] = {877.7007, 101325., 3.5e10, 1.23, 0, 0, 0, 0};
        for (int law = PIEZO_BARUS; law <= PIEZO_MCADAMS; ++law) {
            const std::vector<double> par = rd(4), arg = rd(k);
            Phys P;
            setup_phys(P, 0.1, 0.0, mu0, 0.0, 1e-5, 1e-5, EOS_DH, dh, law, par.data(), THIN_NONE, zero4);
            std::vector<double> out(k);
            for (size_t i = 0; i < k; ++i) out[i] = piezo_eta(mu0, arg[i], P);
            wr(out);
        }
        for (int law = THIN_EYRING; law <= THIN_CARREAU; ++law) {
            const std::vector<double> par = rd(4), rate = rd(k);
            Phys P;
            setup_phys(P, 0.1, 0.0, mu0, 0.0, 1e-5, 1e-5, EOS_DH, dh, PIEZO_NONE, zero4, law, par.data());
            std::vector<double> out(k);
            for (size_t i = 0; i < k; ++i) out[i] = thinning_factor(rate[i], mu0, P);
            wr(out);
        }
        const std::vector<double> uu = rd(2), gx = rd(k), gy = rd(k), hh = rd(k);
        std::vector<double> out(k);
        for (size_t i = 0; i < k; ++i) out[i] = shear_rate_avg(gx[i], gy[i], hh[i], uu[0], uu[1], mu0);
        wr(out);
    }
    return 0;
}
