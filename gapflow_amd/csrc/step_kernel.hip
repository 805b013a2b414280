// Fused MacCormack step for gfx950: both stages, the time average and the per-step
// reductions in ONE pass over HBM.
//
// Replaces, per time step (paths relative to the reference root):
//   GaPFlow/problem.py:528-563          q0 copy, two stages, (q + q0)/2
//   GaPFlow/models/stress.py:289-362, 427-459, 600-622   closures, twice per step
//   GaPFlow/integrate.py:38-130         flux differences + source, twice per step
//   GaPFlow/problem.py:319-357          NaN / rho<0 checks, Ekin, v_max, v_sound
//
// Why one pass is possible: the predictor differences upwind (F[n]-F[n-1]) and the corrector
// downwind (F[n+1]-F[n]); the composed stencil of a full step is
//   (n-1,m) (n,m-1) (n,m) (n+1,m) (n,m+1) (n+1,m-1) (n-1,m+1)
// i.e. radius 1.  So a step needs q and the topography once (read) and q once (write):
// 72 B per cell (80 B with a slip-length field) -- the algorithmic minimum of BASELINE.md.
//
// Mapping to the hardware
//   * one 64-lane wavefront owns a strip of 62 output columns (iy, the contiguous axis) and
//     marches along ix over a chunk of rows.  Lane l holds column m0-1+l, so lanes 0 and 63
//     are the strip's halo columns and every global access of the wave is one contiguous
//     512-B segment per plane.
//   * y-neighbours come from the adjacent lane through wavefront shuffles (`__shfl_up/down`,
//     ds_bpermute: no LDS allocation, no barrier); x-neighbours are the wave's own registers
//     from the previous row.  Waves never synchronise, so a 256-thread block is just four
//     adjacent strips that share their halo lines in the CU's L1.
//   * each cell's closure is evaluated exactly twice per step (once per stage) plus the halo
//     lanes/rows (2/64 columns, 2/rows_per_chunk rows).
//   * rows n+1 are requested before row n is computed (the loads below sit one iteration
//     ahead), so ~7 planes x 512 B per wave are always in flight.
//
// "Logical" indices: n (rows) and m (columns) always increase DOWNWIND of the predictor:
//   ix = n for D=+1, Nx+1-n for D=-1  (likewise iy from m).
// In these coordinates both MC orders run the same code; only the sign D of the one-sided
// differences and the address mapping change (problem.py:521-522).
//
// Ghost cells.  Stage-1 input is whatever the array holds (including ghost cells the user
// may have left stale, tests/test_wave_decay.py:101).  The stage-1 RESULT at a physical
// downwind ghost (row Nx+1 / column Ny+1 in logical terms) is not computed by the stencil
// but follows the ghost rule applied to the stage-1 field (problem.py:560); those values
// are prepared by k_ghost_stage1 into g1x / g1y and picked up here.
#include <hip/hip_runtime.h>
#include "device_types.hpp"

namespace gpf {

constexpr int STRIP = 62;       // output columns per wavefront

struct StepArgs {
    const double* qa;           // q buffer 0 (3 planes)
    const double* qb;           // q buffer 1
    const double* topo;         // h, hx, hy planes
    const double* topo_line;    // TOPO = 1: [3][Nx+2] profile over ix; TOPO = 2: [3][Ny+2] profile over iy
    const double* Ls;           // slip-length plane or nullptr
    const double* g1x;          // [3][pitch]  stage-1 field on the downwind physical ghost row
    const double* g1y;          // [3][Nx+2]   ... on the downwind physical ghost column
    const StepState* st;
    Partial* partials;          // one per (chunk, strip)
    Layout L;
    Edges E;
    int rows_per_chunk;
    int nstrips;
    int honor_stop;
};

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

template <int EOS, bool HAS_LS, bool PIEZO, int D, int TOPO>
__device__ __forceinline__ void step_strip(const StepArgs& a, const Phys& P, const double* __restrict__ qin,
                                           double* __restrict__ qout) {
    const Layout L = a.L;
    const int lane = threadIdx.x & 63;
    const int strip = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (strip >= a.nstrips) return;                         // wave-uniform
    const int chunk = blockIdx.y;

    // columns
    const int m_raw = strip * STRIP + lane;                 // logical column, 0 = upwind ghost
    const int m = m_raw <= L.Ny + 1 ? m_raw : L.Ny + 1;
    const int iy = D > 0 ? m : L.Ny + 1 - m;
    const bool col_out = lane >= 1 && lane <= STRIP && m_raw >= 1 && m_raw <= L.Ny;
    const bool col_dw_ghost = (m_raw == L.Ny + 1);          // physical downwind ghost column
    // rows of this chunk: outputs n_first..n_last, marching n_first-1 .. n_last+1
    const int n_first = chunk * a.rows_per_chunk + 1;
    const int n_last = min(n_first + a.rows_per_chunk - 1, L.Nx);
    const bool dw_row_is_ghost = (n_last == L.Nx) && a.E.halo[D > 0 ? 1 : 0] != 1;

    const double dt = a.st->dt;
    const double cx = (double)D * P.inv_dx, cy = (double)D * P.inv_dy;

    const double* __restrict__ q0p = qin;
    const double* __restrict__ q1p = qin + L.plane;
    const double* __restrict__ q2p = qin + 2 * L.plane;
    const double* __restrict__ hp = a.topo;
    const double* __restrict__ hxp = a.topo + L.plane;
    const double* __restrict__ hyp = a.topo + 2 * L.plane;
    double* __restrict__ qo0 = qout;
    double* __restrict__ qo1 = qout + L.plane;
    double* __restrict__ qo2 = qout + 2 * L.plane;

    // TOPO: most gap profiles vary along one axis only (journal, inclined, parabolic, cdc: h = h(x)).  Then a third
    // of the step's HBM reads is redundant: TOPO = 1 reads one (h, hx, hy) triple per ROW through the scalar cache
    // (the row index is wave-uniform), TOPO = 2 keeps the lane's column triple in registers for the whole march.
    // The planes stay resident for every other kernel; the values are bitwise the same.
    double lh = 0.0, lhx = 0.0, lhy = 0.0;
    if (TOPO == 2) {
        lh = a.topo_line[iy]; lhx = a.topo_line[(L.Ny + 2) + iy]; lhy = a.topo_line[2 * (L.Ny + 2) + iy];
    }
    // per-lane element offsets fit 32 bits (a plane is < 2^28 doubles); the plane bases stay in SGPRs
    auto load = [&](int n, CellIn& c) {
        const int ix = D > 0 ? n : L.Nx + 1 - n;
        const int o = ix * L.pitch + L.off + iy;
        c.rho = q0p[o]; c.jx = q1p[o]; c.jy = q2p[o];
        if (TOPO == 0) {
            c.h = hp[o]; c.hx = hxp[o]; c.hy = hyp[o];
        } else if (TOPO == 1) {
            c.h = a.topo_line[ix]; c.hx = a.topo_line[(L.Nx + 2) + ix]; c.hy = a.topo_line[2 * (L.Nx + 2) + ix];
        } else {
            c.h = lh; c.hx = lhx; c.hy = lhy;
        }
        c.Ls = HAS_LS ? a.Ls[o] : 0.0;
    };

    // rows n+1 and n+2 are in flight while row n is computed (with a line topography a row is only three
    // loads per lane, and one row ahead leaves too few bytes in flight to cover the HBM latency)
    CellIn cur, nxt, nxt2;
    load(n_first - 1, cur);
    load(n_first, nxt);

    // carried from the previous row
    double fx1p0 = 0, fx1p1 = 0, fx1p2 = 0;     // stage-1 x-flux of row n-1
    double part0 = 0, part1 = 0, part2 = 0;     // row n-1: q(t0) + q1 - dt*(-cx*Fx2 + cy*dFy2 - S2)
    // reductions over this wave's output cells
    double r_ekin = 0.0, r_v2 = 0.0, r_c2 = (EOS == EOS_DH) ? __builtin_inf() : 0.0;
    int r_flags = 0;

    for (int n = n_first - 1; n <= n_last + 1; ++n) {
        if (n < n_last) load(n + 2, nxt2);
        const bool first = (n == n_first - 1);
        const bool last = (n == n_last + 1);
        const int ix = D > 0 ? n : L.Nx + 1 - n;

        const TopoRcp tr = topo_rcp<HAS_LS>(cur);           // shared by both stages of this cell

        // ---- stage 1 at (n, m) ----
        double q10, q11, q12;
        if (last && dw_row_is_ghost) {
            q10 = a.g1x[0 * L.pitch + L.off + iy];
            q11 = a.g1x[1 * L.pitch + L.off + iy];
            q12 = a.g1x[2 * L.pitch + L.off + iy];
        } else {
            CellFlux f;
            cell_closure<EOS, true, HAS_LS, PIEZO>(cur, tr, P, f);
            const double fy0 = cur.jy, fy1 = f.fx2, fy2 = f.fy2;
            const double u0 = __shfl_up(fy0, 1), u1 = __shfl_up(fy1, 1), u2 = __shfl_up(fy2, 1);
            q10 = cur.rho - dt * (cx * (cur.jx - fx1p0) + cy * (fy0 - u0) - f.s0);
            q11 = cur.jx - dt * (cx * (f.fx1 - fx1p1) + cy * (fy1 - u1) - f.s1);
            q12 = cur.jy - dt * (cx * (f.fx2 - fx1p2) + cy * (fy2 - u2) - f.s2);
            fx1p0 = cur.jx; fx1p1 = f.fx1; fx1p2 = f.fx2;
            if (col_dw_ghost) {
                q10 = a.g1y[0 * (L.Nx + 2) + ix];
                q11 = a.g1y[1 * (L.Nx + 2) + ix];
                q12 = a.g1y[2 * (L.Nx + 2) + ix];
            }
        }

        if (!first) {
            // ---- stage 2 closure at (n, m) on the stage-1 field ----
            CellIn c1 = cur;
            c1.rho = q10; c1.jx = q11; c1.jy = q12;
            CellFlux g;
            cell_closure<EOS, true, HAS_LS, PIEZO>(c1, tr, P, g);
            const double gy0 = q12, gy1 = g.fx2, gy2 = g.fy2;
            const double d0 = __shfl_down(gy0, 1), d1 = __shfl_down(gy1, 1), d2 = __shfl_down(gy2, 1);

            // ---- finish row n-1: corrector + time average (problem.py:558, 563) ----
            if (n > n_first) {
                const double o0 = 0.5 * (part0 - dt * cx * q11);
                const double o1 = 0.5 * (part1 - dt * cx * g.fx1);
                const double o2 = 0.5 * (part2 - dt * cx * g.fx2);
                if (col_out) {
                    const int ixo = D > 0 ? n - 1 : L.Nx + 2 - n;
                    const int o = ixo * L.pitch + L.off + iy;
                    qo0[o] = o0;
                    qo1[o] = o1;
                    qo2[o] = o2;
                    const double v2 = (o1 * o1 + o2 * o2) * rcp(o0);
                    // a row next to a periodic slab seam also stands in for the far slab's ghost row
                    const double w = 1.0 + ((ixo == 1 && a.E.halo[0] == 2) ? 1.0 : 0.0) +
                                     ((ixo == L.Nx && a.E.halo[1] == 2) ? 1.0 : 0.0);
                    r_ekin += w * (v2 * 0.5);
                    // Flags instead of NaN-propagating maxima in the hot loop: a NaN in any component makes v2 NaN
                    // (flag 1: the state is invalid and the step is undone, problem.py:319-332, so the maxima are
                    // then irrelevant); an imaginary sound speed raises flag 4 and commit_step turns c2max into NaN,
                    // which is what np.sqrt(...).max() yields in the reference (stress.py:539).
                    r_v2 = fmax(r_v2, v2);
                    if (v2 != v2) r_flags |= 1;
                    if (o0 < 0.0) r_flags |= 2;
                    if (EOS == EOS_DH) {
                        // dp/drho = K / (C2 rho0 - rho)^2 grows monotonically towards the pole: its maximum sits at
                        // the cell closest to it, so one reciprocal per WAVE (below) replaces one per cell
                        r_c2 = fmin(r_c2, fabs(P.e[7] - o0));
                    } else {
                        const double c2 = eos_c2<EOS>(o0, P);
                        if (!(c2 >= 0.0)) r_flags |= 4;
                        r_c2 = fmax(r_c2, c2);
                    }
                }
            }
            // ---- open row n (an output row unless this is the downwind extra row) ----
            part0 = (cur.rho + q10) - dt * (-cx * q11 + cy * (d0 - gy0) - g.s0);
            part1 = (cur.jx + q11) - dt * (-cx * g.fx1 + cy * (d1 - gy1) - g.s1);
            part2 = (cur.jy + q12) - dt * (-cx * g.fx2 + cy * (d2 - gy2) - g.s2);
        }
        cur = nxt;
        nxt = nxt2;
    }

    // ---- wave reduction, one record per wave ----
    for (int s = 32; s >= 1; s >>= 1) {
        r_ekin += __shfl_down(r_ekin, s);
        r_v2 = fmax(r_v2, __shfl_down(r_v2, s));
        const double oc = __shfl_down(r_c2, s);
        r_c2 = (EOS == EOS_DH) ? fmin(r_c2, oc) : fmax(r_c2, oc);
        r_flags |= __shfl_down(r_flags, s);
    }
    if (lane == 0) {
        if (EOS == EOS_DH) {                    // same expression as eos_c2<EOS_DH> at the cell nearest the pole
            const double it = rcp(r_c2);
            r_c2 = (r_c2 == __builtin_inf()) ? 0.0 : P.e[6] * (it * it);
        }
        Partial p;
        p.ekin = r_ekin; p.vmax2 = r_v2; p.c2max = r_c2; p.flags = (double)r_flags;
        a.partials[(long long)chunk * a.nstrips + strip] = p;
    }
}

// D = direction of the predictor, chosen by the host from the step index (problem.py:521-522);
// the device-side step counter is checked against it so a disagreement can never go unnoticed.
template <int EOS, bool HAS_LS, bool PIEZO, int D, int TOPO>
__global__ __launch_bounds__(256) void k_step(const StepArgs a, const Phys P) {
    if (halted(a.st, a.honor_stop)) return;
    if (predictor_direction(a.st) != D) __builtin_trap();
    const int par = a.st->parity;
    const double* qin = par ? a.qb : a.qa;
    double* qout = const_cast<double*>(par ? a.qa : a.qb);
    step_strip<EOS, HAS_LS, PIEZO, D, TOPO>(a, P, qin, qout);
}

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
template <int EOS, bool HAS_LS, bool PIEZO, class Field>
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
            cell_closure<EOS, true, HAS_LS, PIEZO>(c, P, f);
        } else if (role == 1) {
            cell(it.ix_up, it.iy_src, it.tu, it.lu, c);
            cell_closure<EOS, true, HAS_LS, PIEZO>(c, P, f);
            sm[0][0][lane] = c.jx; sm[0][1][lane] = f.fx1; sm[0][2][lane] = f.fx2;
        } else if (role == 2) {
            cell(it.ix_src, it.iy_src - D, it.ts, it.ls, c);
            cell_closure<EOS, true, HAS_LS, PIEZO>(c, P, f);
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
template <int EOS, bool HAS_LS, bool PIEZO>
__global__ __launch_bounds__(256) void k_ghost_stage1(const GhostArgs a, const Phys P) {
    __shared__ double sm[2][3][64];
    if (halted(a.st, a.honor_stop)) return;
    const Layout& L = a.L;
    StoredField fld;
    fld.q = a.st->parity ? a.qb : a.qa; fld.L = L;
    const int D = predictor_direction(a.st);
    const double dt = a.st->dt;
    const int ntiles = (L.Ny + L.Nx + 63) / 64;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) ghost_stage1_tile<EOS, HAS_LS, PIEZO>(fld, a, P, D, tile * 64, dt, sm);
}

}  // namespace gpf
