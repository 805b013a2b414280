// Device-side data layout and step state shared by the kernels and the C ABI.
#pragma once

#include <stdint.h>
#include "closures.hpp"

namespace gpf {

// Fields live in HBM as structure-of-arrays planes, iy contiguous:
//   element (c, ix, iy) of a field at  base[c*plane + ix*pitch + off + iy]
// `off` = 15 places the first interior column (iy = 1) on a 128-byte boundary and
// `pitch` is a multiple of 16 doubles, so every interior row segment starts 128-B aligned.
struct Layout {
    int Nx, Ny;         // interior cells
    int pitch, off;
    long long plane;    // (Nx+2)*pitch
    __host__ __device__ __forceinline__ long long at(int ix, int iy) const {
        return (long long)ix * pitch + off + iy;
    }
};

enum { BC_P = 0, BC_D = 1, BC_N = 2 };

// Ghost-edge rules, resolved by the host from problem.py:676-768.
struct Edges {
    int rule[4][3];     // [edge][component]; edge 0: ix=0, 1: ix=Nx+1, 2: iy=0, 3: iy=Ny+1
    double value[4];    // Dirichlet target per edge
    int halo[2];        // kind of row ix=0 / ix=Nx+1: 0 physical ghost (local rule), 1 slab halo (a neighbour's
                        // interior row), 2 periodic seam (the domain's ghost row, filled by the ring exchange)
};

// Everything that changes from step to step and must not round-trip through the host.
struct StepState {
    double dt;              // step size of the next step
    double simtime;
    double ekin, ekin_old;
    double residual;
    double vmax2, c2max;    // max (jx^2+jy^2)/rho, max dp/drho over all cells
    double rbuf[5];         // last <=5 residuals (deque(maxlen=5), problem.py:435)
    int rcount, rhead;
    long long step;
    long long max_it;
    double tol, CFL, hmin;  // hmin = min(dx, dy)
    int adaptive;
    int mc_order;
    int parity;             // which of the two q buffers holds the current state
    int invalid;            // 0 ok, 1 NaN, 2 negative density
    int converged;
    double dt_last;         // step size of the last committed step (gpf_update_closures re-runs its predictor stage)
};

struct LogEntry {           // == gpf_scalars_t in include/gapflow_hip.h
    long long step;
    double simtime, dt, ekin, ekin_old, residual, v_max, v_sound, mass;
    int invalid, converged;
};

// Partial reductions: one record per wave of the step kernel (and per block of the others).
struct Partial {
    double ekin;        // sum
    double vmax2;       // max, NaN-propagating
    double c2max;       // max, NaN-propagating (negative dp/drho counts as NaN, like np.sqrt)
    double flags;       // 1: NaN seen, 2: rho<0 seen (bit-ored as small ints)
};

// NaN-propagating maximum (np.max semantics)
__host__ __device__ __forceinline__ double nanmax(double m, double x) {
    return (x > m || x != x) ? x : m;
}

}  // namespace gpf
