// Small problems (a few hundred cells: the reference's 1-D examples have Nx = 100-200) are launch-bound: three launches of
// ~5-10 us each per time step, for microseconds of arithmetic.  k_small_steps advances such a problem by MANY time steps in
// ONE launch of ONE workgroup: the field, the topography and the fluxes live in LDS (<= 160 KB), phases are separated by
// workgroup barriers, the scalars are reduced in the block and committed by thread 0 -- no launches, no device-scope
// traffic between steps.  Arithmetic and order are those of the stage-wise pipeline (problem.py:509-586 line by line:
// closures -> flux differences + source -> update -> ghost cells, twice; average; ghost cells; scalars; commit).
#pragma once


namespace gpf {

constexpr int SMALL_DOUBLES_PER_CELL = 16;     // q0 3, q 3, fluxes 6 (fx1 fx2 fy2 s0 s1 s2), topography 3, slip length 1

struct SmallArgs {
    double* qa; double* qb;
    const double* topo; const double* Ls;
    StepState* st;
    LogEntry* log; long long log_base, log_cap;
    Layout L; Edges E;
    int nsteps, honor_stop;
};

template <int EOS, bool HAS_LS>
__global__ __launch_bounds__(512) void k_small_steps(const SmallArgs a, const Phys P) {
    extern __shared__ double lds[];
    __shared__ Acc sm[16];
    __shared__ int sh_flags[2];
    __shared__ StepState sst;                       // the run state stays on-chip for the whole batch
    const Layout& G = a.L;                          // layout of the global planes
    const int w = G.Ny + 2, nc = (G.Nx + 2) * w;
    Layout D;                                       // dense layout of the LDS planes
    D.Nx = G.Nx; D.Ny = G.Ny; D.pitch = w; D.off = 0; D.plane = nc;
    double* q0 = lds;
    double* q = q0 + 3 * nc;
    double* F = q + 3 * nc;
    double* T = F + 6 * nc;                         // h, hx, hy, Ls
    const int tid = threadIdx.x, nt = blockDim.x;
    if (tid == 0) sst = *a.st;
    __syncthreads();
    StepState* st = &sst;

    {   // current field and topography -> LDS
        const double* src = st->parity ? a.qb : a.qa;
        for (int t = tid; t < nc; t += nt) {
            const long long o = G.at(t / w, t % w);
            for (int c = 0; c < 3; ++c) { q0[c * nc + t] = src[c * G.plane + o]; T[c * nc + t] = a.topo[c * G.plane + o]; }
            T[3 * nc + t] = HAS_LS ? a.Ls[o] : 0.0;
        }
    }
    __syncthreads();

    // problem.py:676-707 in ONE phase: the reference fills x edges over all columns and then y edges over all rows, so a
    // corner is rule_y(rule_x(.)); every ghost cell is derived here from interior values directly (cf. ghost_fill_block)
    auto ghosts = [&](double* f) {
        const int nrow = 2 * w, ncol = 2 * D.Nx;
        for (int t = tid; t < nrow + ncol; t += nt) {
            double v[3];
            int cell;
            if (t < nrow) {
                const int e = t / w, iy = t % w;
                cell = (int)D.at(e == 0 ? 0 : D.Nx + 1, iy);
                if (iy >= 1 && iy <= D.Ny) {
                    for (int c = 0; c < 3; ++c) v[c] = ghost_x(f, D, a.E, e, c, iy);
                } else {
                    const int ey = iy == 0 ? 2 : 3;
                    const int src = (a.E.rule[ey][0] == BC_P) ? (ey == 2 ? D.Ny : 1) : (ey == 2 ? 1 : D.Ny);
                    for (int c = 0; c < 3; ++c) {
                        const double gx = ghost_x(f, D, a.E, e, c, src);
                        v[c] = a.E.rule[ey][c] == BC_D ? 2.0 * a.E.value[ey] - gx : gx;
                    }
                }
            } else {
                const int u = t - nrow, e = 2 + u / D.Nx, ix = 1 + u % D.Nx;
                cell = (int)D.at(ix, e == 2 ? 0 : D.Ny + 1);
                for (int c = 0; c < 3; ++c) v[c] = ghost_y(f, D, a.E, e, c, ix);
            }
            for (int c = 0; c < 3; ++c) f[c * nc + cell] = v[c];
        }
        __syncthreads();
    };

    int committed = 0;                              // steps this launch has committed (block-uniform)
    for (int step = 0; step < a.nsteps; ++step) {
        if (st->invalid != 0 || (a.honor_stop && (st->converged || st->step >= st->max_it))) break;     // block-uniform
        const double dt = st->dt;
        const int dir0 = direction_of_step(st, st->step);
        for (int stage = 0; stage < 2; ++stage) {
            const double* qin = stage == 0 ? q0 : q;        // stage 1 reads the current field, writes the working one
            const int dir = stage == 0 ? dir0 : -dir0;
            // Pressure / WallStress / BulkStress.update on the whole array, ghost cells included (problem.py:536-540)
            for (int t = tid; t < nc; t += nt) {
                CellIn c;
                c.rho = qin[t]; c.jx = qin[nc + t]; c.jy = qin[2 * nc + t];
                c.h = T[t]; c.hx = T[nc + t]; c.hy = T[2 * nc + t]; c.Ls = T[3 * nc + t];
                CellFlux f;
                cell_closure<EOS, true, HAS_LS, true>(c, P, f);
                F[t] = f.fx1; F[nc + t] = f.fx2; F[2 * nc + t] = f.fy2; F[3 * nc + t] = f.s0; F[4 * nc + t] = f.s1; F[5 * nc + t] = f.s2;
            }
            __syncthreads();
            // predictor_corrector + source + update (integrate.py:38-130, problem.py:558) on interior cells; the ghost
            // cells the reference also updates are overwritten right after
            const double cx = (double)dir * P.inv_dx, cy = (double)dir * P.inv_dy;
            double nv[3][3];                        // <= 1200 cells on 512 threads: at most three per thread
            int nk = 0, cells[3];
            for (int t = tid; t < nc; t += nt) {
                const int ix = t / w, iy = t % w;
                if (ix < 1 || ix > D.Nx || iy < 1 || iy > D.Ny) continue;
                const int xu = t - dir * w, yu = t - dir;          // upwind neighbours in x and y
                nv[nk][0] = qin[t] - dt * (cx * (qin[nc + t] - qin[nc + xu]) + cy * (qin[2 * nc + t] - qin[2 * nc + yu]) - F[3 * nc + t]);
                nv[nk][1] = qin[nc + t] - dt * (cx * (F[t] - F[xu]) + cy * (F[nc + t] - F[nc + yu]) - F[4 * nc + t]);
                nv[nk][2] = qin[2 * nc + t] - dt * (cx * (F[nc + t] - F[nc + xu]) + cy * (F[2 * nc + t] - F[2 * nc + yu]) - F[5 * nc + t]);
                cells[nk++] = t;
            }
            __syncthreads();
            for (int k = 0; k < nk; ++k)
                for (int c = 0; c < 3; ++c) q[c * nc + cells[k]] = nv[k][c];
            __syncthreads();
            ghosts(q);
        }
        // second-order temporal averaging over the whole array (problem.py:563); validity before the ghost update (:565)
        int pre = 0;
        for (int t = tid; t < nc; t += nt) {
            for (int c = 0; c < 3; ++c) q[c * nc + t] = 0.5 * (q[c * nc + t] + q0[c * nc + t]);
            const double r = q[t], jx = q[nc + t], jy = q[2 * nc + t];
            if (r != r || jx != jx || jy != jy) pre |= 1;
            if (r < 0.0) pre |= 2;
        }
        if (tid == 0) sh_flags[0] = 0;
        __syncthreads();
        if (pre) atomicOr(&sh_flags[0], pre);
        __syncthreads();
        ghosts(q);
        Acc acc; acc.zero();
        for (int t = tid; t < nc; t += nt) acc.cell<EOS>(q[t], q[nc + t], q[2 * nc + t], T[t], P);
        acc = block_reduce(acc, sm);
        if (tid == 0) {
            const int flags = (sh_flags[0] & 3) | (acc.flags & 4);
            commit_step(st, acc.ekin, acc.v2, acc.c2, flags, a.log, a.log_base, a.log_cap);
            sh_flags[1] = st->invalid;
        }
        __syncthreads();
        if (sh_flags[1]) break;                     // rolled back: q0 still holds the last valid state
        double* tmp = q0; q0 = q; q = tmp;          // the averaged field is the current one now
        committed += 1;
    }
    // the current state -> the buffer the (final) parity designates; the run state back to global memory
    __syncthreads();
    if (tid == 0) *a.st = sst;
    double* dst = st->parity ? a.qb : a.qa;
    double* prev = st->parity ? a.qa : a.qb;        // ... and the state before the last committed step -> the other buffer, where the
                                                    // launch-per-step kernels leave it too (gpf_update_closures redoes its predictor stage)
    for (int t = tid; t < nc; t += nt) {
        const long long o = G.at(t / w, t % w);
        for (int c = 0; c < 3; ++c) dst[c * G.plane + o] = q0[c * nc + t];
        if (committed > 0 && st->invalid == 0)
            for (int c = 0; c < 3; ++c) prev[c * G.plane + o] = q[c * nc + t];
    }
}

}  // namespace gpf
