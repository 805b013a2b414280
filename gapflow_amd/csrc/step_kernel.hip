// Edge work of the fused MacCormack step for slabs, and what the step kernels share.
//
// The step itself lives in step2_kernel.hip (k_step2: both stages, the time average, the per-step reductions and -- for
// single-handle problems -- ghost cells and the commit in ONE launch).  This file holds
//   * the ghost rule, the halt test and the predictor-direction table used by every step kernel,
//   * predictor_value: the one FMA sequence every stage-1 value is formed with,
//   * k_ghost_stage1: the stage-1 field on the downwind ghost row / column of a SLAB (whose outer rows are filled by a
//     neighbour exchange, so the step kernel cannot derive them itself), shared with k_begin_slab (aux_kernels.hip).
//
// Why one pass over HBM is possible: the predictor differences upwind (F[n]-F[n-1]) and the corrector downwind
// (F[n+1]-F[n]); the composed stencil of a full step is
//   (n-1,m) (n,m-1) (n,m) (n+1,m) (n,m+1) (n+1,m-1) (n-1,m+1)
// i.e. radius 1.  So a step needs q and the topography once (read) and q once (write): 72 B per cell (80 B with a
// slip-length field) -- the algorithmic minimum of BASELINE.md.
//
// "Logical" indices: n (rows) and m (columns) always increase DOWNWIND of the predictor:
//   ix = n for D=+1, Nx+1-n for D=-1  (likewise iy from m).
// In these coordinates both MC orders run the same code; only the sign D of the one-sided differences and the address
// mapping change (problem.py:521-522).
//
// Ghost cells.  Stage-1 input is whatever the array holds (including ghost cells the user may have left stale,
// tests/test_wave_decay.py:101).  The stage-1 RESULT at a physical downwind ghost (row Nx+1 / column Ny+1 in logical
// terms) is not computed by the stencil but follows the ghost rule applied to the stage-1 field (problem.py:560).
#include <hip/hip_runtime.h>
#include "device_types.hpp"

namespace gpf {

// value a ghost cell of edge e takes from its source cell's value v (problem.py:758-766)
__device__ __forceinline__ double ghost_rule(const Edges& E, int e, int c, double v) {
    return E.rule[e][c] == BC_D ? 2.0 * E.value[e] - v : v;
}

__device__ __forceinline__ bool halted(const StepState* st, int honor_stop) {
    return st->invalid != 0 || (honor_stop && (st->converged || st->step >= st->max_it));
}

// q - dt (cx dFx + cy dFy - S): the predictor's update (problem.py:558) with its contractions spelled out, so that every
// kernel that forms a stage-1 value -- the march of k_step2, its ghost-row prologue, k_ghost_stage1 / k_begin_slab for
// slabs -- rounds it identically: an N-slab run is then bit-identical to the 1-slab run of the same problem.
__device__ __forceinline__ double predictor_value(double q, double dt, double cx, double dfx, double cy, double dfy, double s) {
    return fma(-dt, fma(cx, dfx, fma(cy, dfy, -s)), q);
}

// direction of the predictor of step number `step` (problem.py:521-522)
__device__ __forceinline__ int direction_of_step(const StepState* st, long long step) {
    if (st->mc_order == 0) return (step % 2 == 0) ? 1 : -1;
    return ((st->mc_order + 1) / 2) ? 1 : -1;      // [[-1,1],[1,-1]][(switch+1)//2]
}
__device__ __forceinline__ int predictor_direction(const StepState* st) { return direction_of_step(st, st->step); }

// ---------------------------------------------------------------------------------------------
// Stage-1 values on the physical downwind ghost row / column (problem.py:560 after stage 1):
//   periodic : q1(ghost) = q1(partner), partner = first interior cell on the other side
//   Neumann  : q1(ghost) = q1(adjacent)                (problem.py:766)
//   Dirichlet: q1(ghost) = 2*target - q1(adjacent)     (problem.py:758-764)
// q1 at the source cell is the ordinary predictor result there: q1 = q - dt R.  k_step reads finished ghost values
// g1 = rule(q - dt R), written at the start of every step by k_ghost_stage1 (or, for slabs, by the previous step's
// k_begin_slab once the neighbours' rows are in).  Handing k_step the dt-independent pair (rule-folded q, R) instead --
// prepared one launch earlier, inside the ghost fill -- was measured and rejected: the three extra loads and FMAs in
// k_step's row loop cost 6 % of its time at 4096^2 (11 % with a 2-D gap), more than the launch they save.
// ---------------------------------------------------------------------------------------------
struct GhostArgs {
    const double* qa; const double* qb;
    const double* topo; const double* Ls;
    const double* seam[2];      // per x edge: [2 rows][4: h,hx,hy,Ls][pitch] = topography of (source row, its
                                // upwind row) on the far side of a periodic slab seam, or nullptr
    double* g1x; double* g1y;   // finished ghost values [3][pitch], [3][Nx+2]
    const StepState* st;
    Layout L; Edges E;
    int honor_stop;
};

// The field as it is stored, ghost cells included (they may be stale after a user edit: that is what the
// reference's first stage would read, tests/test_wave_decay.py:101).
struct StoredField {
    const double* q; Layout L;
    __device__ __forceinline__ double get(int ix, int iy, int c) const { return q[c * L.plane + L.at(ix, iy)]; }
};
// A slab's field whose two outer rows are still in this rank's mailbox (peer-to-peer transport): what StoredField
// will show once the rows have been scattered.
struct MailField {
    const double* q; Layout L;
    const double* row_lo; const double* row_hi;     // [3][pitch] each, or nullptr: read the stored row
    __device__ __forceinline__ double get(int ix, int iy, int c) const {
        if (ix == 0 && row_lo) return row_lo[c * L.pitch + L.off + iy];
        if (ix == L.Nx + 1 && row_hi) return row_hi[c * L.pitch + L.off + iy];
        return q[c * L.plane + L.at(ix, iy)];
    }
};

// Work item t of a step's stage-1 ghost data: t in [0, Ny) is column t+1 of the downwind ghost ROW, t in [Ny, Ny+Nx)
// is row t-Ny+1 of the downwind ghost COLUMN.  Its value needs the predictor rate R at a source cell, i.e. closures at
// three cells: the source (with the source term), its x-upwind and its y-upwind neighbour.
struct Stage1Item {
    bool active, row;
    int edge;                       // whose rule applies: 0/1 for a row, 2/3 for a column
    int ix_src, iy_src, ix_up;      // source cell; x-upwind row (redirected across a slab seam)
    const double *ts, *tu, *ls, *lu;    // optional topography rows of (source, upwind) beyond a periodic slab seam
    int out;                        // index into g1x / g1y
};

__device__ __forceinline__ Stage1Item stage1_item(const GhostArgs& a, int D, int t) {
    const Layout& L = a.L;
    Stage1Item it;
    it.active = t < L.Ny + L.Nx;
    it.row = t < L.Ny;
    it.ts = it.tu = it.ls = it.lu = nullptr;
    if (it.row) {
        const int iy = t + 1;
        it.edge = D > 0 ? 1 : 0;
        if (a.E.halo[it.edge] == 1) it.active = false;         // a neighbour's real cell: the stencil computes it
        const bool periodic = a.E.rule[it.edge][0] == BC_P;
        if (!periodic) {
            it.ix_src = D > 0 ? L.Nx : 1;
            it.ix_up = it.ix_src - D;
        } else if (a.E.halo[it.edge] != 2) {
            it.ix_src = D > 0 ? 1 : L.Nx;                       // partner cell on the other side of the domain
            it.ix_up = it.ix_src - D;
        } else {
            // periodic seam between slabs: the partner row's q sits in this slab's halo/ghost row, its upwind
            // neighbour is this slab's last interior row; their topography is static data
            it.ix_src = D > 0 ? L.Nx + 1 : 0;
            it.ix_up = D > 0 ? L.Nx : 1;
            it.ts = a.seam[it.edge]; it.tu = a.seam[it.edge] + 4 * L.pitch;
            it.ls = it.ts + 3 * L.pitch; it.lu = it.tu + 3 * L.pitch;
        }
        it.iy_src = iy;
        it.out = L.off + iy;
    } else {
        const int ix = t - L.Ny + 1;
        it.edge = D > 0 ? 3 : 2;
        const bool periodic = a.E.rule[it.edge][0] == BC_P;
        it.iy_src = periodic ? (D > 0 ? 1 : L.Ny) : (D > 0 ? L.Ny : 1);
        it.ix_src = ix;
        it.ix_up = ix - D;
        it.out = ix;
    }
    return it;
}

// One tile of 64 work items by a 256-thread block.  The three closures of an item are evaluated by three different
// WAVES (wave 0: source cell, wave 1: x-upwind, wave 2: y-upwind; wave 3 idles), which cuts the dependent arithmetic
// chain of this latency-bound job to a third; the upwind fluxes travel through LDS.
// PIEZO and the closure variant (with source term) are the march's own, so that a value formed here is rounded exactly
// like the same value formed by the stencil on the other side of a slab boundary.
// XONLY: the step kernel runs its x-only-gap closure (TOPO = 3); so does this one.
template <int EOS, bool HAS_LS, bool PIEZO, bool XONLY, class Field>
__device__ __forceinline__ void ghost_stage1_tile(const Field& fld, const GhostArgs& a, const Phys& P, int D, int t0, double dt,
                                                  double (*sm)[3][64]) {
    const Layout& L = a.L;
    const int role = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const Stage1Item it = stage1_item(a, D, t0 + lane);
    auto cell = [&](int cx_, int cy_, const double* trow, const double* lrow, CellIn& c) {
        c.rho = fld.get(cx_, cy_, 0); c.jx = fld.get(cx_, cy_, 1); c.jy = fld.get(cx_, cy_, 2);
        if (trow) {
            c.h = trow[L.off + cy_]; c.hx = trow[L.pitch + L.off + cy_]; c.hy = trow[2 * L.pitch + L.off + cy_];
            c.Ls = (HAS_LS && lrow) ? lrow[L.off + cy_] : 0.0;
        } else {
            const long long o = L.at(cx_, cy_);
            c.h = a.topo[o]; c.hx = a.topo[o + L.plane]; c.hy = a.topo[o + 2 * L.plane];
            c.Ls = HAS_LS ? a.Ls[o] : 0.0;
        }
    };
    CellIn c;
    CellFlux f;
    if (it.active) {
        if (role == 0) {
            cell(it.ix_src, it.iy_src, it.ts, it.ls, c);
            if (XONLY) cell_closure_xonly<EOS>(c.rho, c.jx, c.jy, row_coefficients(c.h, c.hx, P), P, f);
            else if (!HAS_LS && !PIEZO) cell_closure_ls0<EOS>(c.rho, c.jx, c.jy, gap_coefficients(c.h, c.hx, c.hy), P, f);
            else cell_closure<EOS, true, HAS_LS, PIEZO>(c, P, f);
        } else if (role == 1) {
            cell(it.ix_up, it.iy_src, it.tu, it.lu, c);
            if (XONLY) cell_closure_xonly<EOS>(c.rho, c.jx, c.jy, row_coefficients(c.h, c.hx, P), P, f);
            else if (!HAS_LS && !PIEZO) cell_closure_ls0<EOS>(c.rho, c.jx, c.jy, gap_coefficients(c.h, c.hx, c.hy), P, f);
            else cell_closure<EOS, true, HAS_LS, PIEZO>(c, P, f);
            sm[0][0][lane] = c.jx; sm[0][1][lane] = f.fx1; sm[0][2][lane] = f.fx2;
        } else if (role == 2) {
            cell(it.ix_src, it.iy_src - D, it.ts, it.ls, c);
            if (XONLY) cell_closure_xonly<EOS>(c.rho, c.jx, c.jy, row_coefficients(c.h, c.hx, P), P, f);
            else if (!HAS_LS && !PIEZO) cell_closure_ls0<EOS>(c.rho, c.jx, c.jy, gap_coefficients(c.h, c.hx, c.hy), P, f);
            else cell_closure<EOS, true, HAS_LS, PIEZO>(c, P, f);
            sm[1][0][lane] = c.jy; sm[1][1][lane] = f.fx2; sm[1][2][lane] = f.fy2;
        }
    }
    __syncthreads();
    if (it.active && role == 0) {
        const double cx = (double)D * P.inv_dx, cy = (double)D * P.inv_dy;
        double v[3];
        v[0] = predictor_value(c.rho, dt, cx, c.jx - sm[0][0][lane], cy, c.jy - sm[1][0][lane], f.s0);
        v[1] = predictor_value(c.jx, dt, cx, f.fx1 - sm[0][1][lane], cy, f.fx2 - sm[1][1][lane], f.s1);
        v[2] = predictor_value(c.jy, dt, cx, f.fx2 - sm[0][2][lane], cy, f.fy2 - sm[1][2][lane], f.s2);
        double* g = it.row ? a.g1x : a.g1y;
        const int stride = it.row ? L.pitch : L.Nx + 2;
        for (int k = 0; k < 3; ++k) g[k * stride + it.out] = ghost_rule(a.E, it.edge, k, v[k]);
    }
    __syncthreads();                // the LDS tile is free again
}

// stage-1 ghost data of the step about to run, from the stored field (ghost cells and halo rows included)
template <int EOS, bool HAS_LS, bool PIEZO, bool XONLY>
__global__ __launch_bounds__(256) void k_ghost_stage1(const GhostArgs a, const Phys P) {
    __shared__ double sm[2][3][64];
    if (halted(a.st, a.honor_stop)) return;
    const Layout& L = a.L;
    StoredField fld;
    fld.q = a.st->parity ? a.qb : a.qa; fld.L = L;
    const int D = predictor_direction(a.st);
    const double dt = a.st->dt;
    const int ntiles = (L.Ny + L.Nx + 63) / 64;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) ghost_stage1_tile<EOS, HAS_LS, PIEZO, XONLY>(fld, a, P, D, tile * 64, dt, sm);
}

}  // namespace gpf
