// Host-side preprocessing of the material constants into gpf::Phys (plain C++: shared by api.hip and the
// sanitizer build of tests/hostcheck).  Parameter meaning per law: include/gapflow_hip.h (gpf_config).
#pragma once

#include <cmath>
#include <cstring>
#include "closures.hpp"

namespace gpf {

inline void setup_phys(Phys& P, double U, double V, double eta, double zeta, double dx, double dy, int eos, const double* e,
                       int piezo, const double* z, int thinning, const double* th) {
    std::memset(&P, 0, sizeof(P));
    P.U = U; P.V = V; P.eta = eta; P.zeta = zeta;
    P.v1 = zeta + (4.0 / 3.0) * eta; P.v2 = zeta - (2.0 / 3.0) * eta;      // viscous.py:82-83
    P.inv_dx = 1.0 / dx; P.inv_dy = 1.0 / dy;
    P.eos = eos; P.piezo = piezo;
    switch (eos) {
    case EOS_DH:     // rho0, P0, C1, C2
        P.e[0] = e[0]; P.e[1] = e[1]; P.e[2] = e[2]; P.e[3] = e[3];
        P.e[4] = 0.99 * e[3] * e[0]; P.e[5] = 1.0 / e[0]; P.e[6] = e[2] * e[0] * (e[3] - 1.0); P.e[7] = e[3] * e[0];
        break;
    case EOS_PL:     // rho0, P0, alpha
        P.e[0] = e[0]; P.e[1] = e[1]; P.e[2] = e[2]; P.e[3] = 1.0 / (1.0 - 0.5 * e[2]);
        P.e[5] = 1.0 / e[0]; P.e[6] = -2.0 / (e[2] - 2.0); P.e[7] = -2.0 * e[1] / (e[2] - 2.0);       // 1/rho0, sound-speed exponent and prefactor
        break;
    case EOS_VDW:    // M, T, a, b   (pressure.py:168-173)
        P.e[0] = 1000.0 / e[0]; P.e[1] = 8.31446261815324 * e[1]; P.e[2] = e[2] / 10.0; P.e[3] = e[3] / 1000.0;
        break;
    case EOS_MT:     // rho0, P0, K, n
        for (int i = 0; i < 4; ++i) P.e[i] = e[i];
        P.e[5] = 1.0 / e[0]; P.e[6] = e[2] / e[3]; P.e[7] = e[2] / e[0];      // 1/rho0, K/n, K/rho0
        break;
    case EOS_CUBIC:  // a, b, c, d
        for (int i = 0; i < 4; ++i) P.e[i] = e[i];
        break;
    case EOS_BWR: {  // T, gamma; x1..x32 of Johnson, Zollweg & Gubbins (1993), pressure.py:255-272
        static const double x[32] = {
            0.8623085097507421, 2.976218765822098, -8.402230115796038, 0.1054136629203555, -0.8564583828174598,
            1.582759470107601, 0.7639421948305453, 1.753173414312048, 2.798291772190376e+03, -4.8394220260857657e-02,
            0.9963265197721935, -3.698000291272493e+01, 2.084012299434647e+01, 8.305402124717285e+01,
            -9.574799715203068e+02, -1.477746229234994e+02, 6.398607852471505e+01, 1.603993673294834e+01,
            6.805916615864377e+01, -2.791293578795945e+03, -6.245128304568454, -8.116836104958410e+03,
            1.488735559561229e+01, -1.059346754655084e+04, -1.131607632802822e+02, -8.867771540418822e+03,
            -3.986982844450543e+01, -4.689270299917261e+03, 2.593535277438717e+02, -2.694523589434903e+03,
            -7.218487631550215e+02, 1.721802063863269e+02};
        const double T = e[0], T2 = T * T, T3 = T2 * T, T4 = T2 * T2;
        P.e[0] = e[1];
        P.x[0] = T;
        P.x[1] = x[0] * T + x[1] * std::sqrt(T) + x[2] + x[3] / T + x[4] / T2;
        P.x[2] = x[5] * T + x[6] + x[7] / T + x[8] / T2;
        P.x[3] = x[9] * T + x[10] + x[11] / T;
        P.x[4] = x[12];
        P.x[5] = x[13] / T + x[14] / T2;
        P.x[6] = x[15] / T;
        P.x[7] = x[16] / T + x[17] / T2;
        P.x[8] = x[18] / T2;
        P.x[9] = x[19] / T2 + x[20] / T3;
        P.x[10] = x[21] / T2 + x[22] / T4;
        P.x[11] = x[23] / T2 + x[24] / T3;
        P.x[12] = x[25] / T2 + x[26] / T4;
        P.x[13] = x[27] / T2 + x[28] / T3;
        P.x[14] = x[29] / T2 + x[30] / T3 + x[31] / T4;
        break;
    }
    case EOS_BAYADA: {   // rho_l, rho_v, c_l, c_v (pressure.py:303-304)
        const double rl = e[0], rv = e[1], cl2 = e[2] * e[2], cv2 = e[3] * e[3];
        const double N = rv * cv2 * rl * cl2 * (rv - rl) / (rv * rv * cv2 - rl * rl * cl2);
        const double Pcav = rv * cv2 - N * std::log(rv * rv * cv2 / (rl * rl * cl2));
        P.e[0] = rl; P.e[1] = rv; P.e[2] = cl2; P.e[3] = cv2; P.e[4] = N; P.e[5] = Pcav; P.e[6] = 1.0 / (rv - rl);
        break;
    }
    }
    switch (piezo) {
    case PIEZO_BARUS: P.pz[0] = z[0]; break;
    case PIEZO_ROELANDS: P.pz[0] = z[0]; P.pz[1] = z[1]; P.pz[2] = z[2]; P.pz[3] = std::log(eta / z[0]); break;
    case PIEZO_DUKLER:
    case PIEZO_MCADAMS: P.pz[0] = z[0]; P.pz[1] = z[1]; P.pz[2] = z[2]; break;
    }
    P.thinning = thinning;
    for (int i = 0; i < 4; ++i) P.th[i] = th[i];
}

}  // namespace gpf
