// Everything around the fused step: ghost-cell rules, per-step scalar reductions and the
// device-side dt / residual update, the reference-ordered unfused stage pipeline, the
// stateless integrate.py operators, and host<->device layout conversion.
#include <hip/hip_runtime.h>
#include "device_types.hpp"

namespace gpf {

// ---------------------------------------------------------------------------------------------
// layout conversion: reference layout [c][ix][iy] (pitch Ny+2)  <->  padded planes
// ---------------------------------------------------------------------------------------------
__global__ void k_pack(const double* __restrict__ src, double* __restrict__ dst, Layout L, int ncomp) {
    const long long ncell = (long long)(L.Nx + 2) * (L.Ny + 2);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < ncell * ncomp;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i / ncell);
        const long long r = i - c * ncell;
        const int ix = (int)(r / (L.Ny + 2)), iy = (int)(r - (long long)ix * (L.Ny + 2));
        dst[c * L.plane + L.at(ix, iy)] = src[i];
    }
}

__global__ void k_unpack(const double* __restrict__ src, double* __restrict__ dst, Layout L, int ncomp) {
    const long long ncell = (long long)(L.Nx + 2) * (L.Ny + 2);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < ncell * ncomp;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i / ncell);
        const long long r = i - c * ncell;
        const int ix = (int)(r / (L.Ny + 2)), iy = (int)(r - (long long)ix * (L.Ny + 2));
        dst[i] = src[c * L.plane + L.at(ix, iy)];
    }
}

// ---------------------------------------------------------------------------------------------
// ghost-cell rules (problem.py:676-768), one value
// ---------------------------------------------------------------------------------------------
// x edge e in {0: ix=0, 1: ix=Nx+1}: returns the ghost value of component c in column iy
__device__ __forceinline__ double ghost_x(const double* q, const Layout& L, const Edges& E, int e, int c, int iy) {
    const int r = E.rule[e][c];
    const int adj = e == 0 ? 1 : L.Nx;
    const int src = (r == BC_P) ? (e == 0 ? L.Nx : 1) : adj;
    const double v = q[c * L.plane + L.at(src, iy)];
    return r == BC_D ? 2.0 * E.value[e] - v : v;
}
// y edge e in {2: iy=0, 3: iy=Ny+1}
__device__ __forceinline__ double ghost_y(const double* q, const Layout& L, const Edges& E, int e, int c, int ix) {
    const int r = E.rule[e][c];
    const int adj = e == 2 ? 1 : L.Ny;
    const int src = (r == BC_P) ? (e == 2 ? L.Ny : 1) : adj;
    const double v = q[c * L.plane + L.at(ix, src)];
    return r == BC_D ? 2.0 * E.value[e] - v : v;
}

// x edges for all columns (ghost columns included, as the reference does), then -- in a second
// launch or after a barrier -- y edges for all rows: corners end up as rule_y(rule_x(.)).
__global__ void k_bc_x(double* q, Layout L, Edges E) {
    const int iy = blockIdx.x * blockDim.x + threadIdx.x;
    if (iy > L.Ny + 1) return;
    for (int e = 0; e < 2; ++e) {
        if (E.halo[e]) continue;
        const int ix = e == 0 ? 0 : L.Nx + 1;
        double v[3];
        for (int c = 0; c < 3; ++c) v[c] = ghost_x(q, L, E, e, c, iy);
        for (int c = 0; c < 3; ++c) q[c * L.plane + L.at(ix, iy)] = v[c];
    }
}
__global__ void k_bc_y(double* q, Layout L, Edges E) {
    const int ix = blockIdx.x * blockDim.x + threadIdx.x;
    if (ix > L.Nx + 1) return;
    for (int e = 2; e < 4; ++e) {
        const int iy = e == 2 ? 0 : L.Ny + 1;
        double v[3];
        for (int c = 0; c < 3; ++c) v[c] = ghost_y(q, L, E, e, c, ix);
        for (int c = 0; c < 3; ++c) q[c * L.plane + L.at(ix, iy)] = v[c];
    }
}

// ---------------------------------------------------------------------------------------------
// scalars
// ---------------------------------------------------------------------------------------------
struct Acc {
    double ekin, v2, c2, mass;
    int flags;
    __device__ __forceinline__ void zero() { ekin = 0; v2 = 0; c2 = 0; mass = 0; flags = 0; }
    template <int EOS>
    __device__ __forceinline__ void cell(double r, double jx, double jy, double h, const Phys& P, double w = 1.0) {
        const double v = (jx * jx + jy * jy) / r;
        ekin += w * (v * 0.5);
        v2 = nanmax(v2, v);
        const double c = eos_c2<EOS>(r, P);
        if (!(c >= 0.0)) flags |= 4;
        c2 = fmax(c2, c);
        mass += w * (r * h);
        if (r != r || jx != jx || jy != jy) flags |= 1;
        if (r < 0.0) flags |= 2;
    }
    __device__ __forceinline__ void merge(const Acc& o) {
        ekin += o.ekin; v2 = nanmax(v2, o.v2); c2 = nanmax(c2, o.c2); mass += o.mass; flags |= o.flags;
    }
};

__device__ __forceinline__ Acc wave_reduce(Acc a) {
    for (int s = 32; s >= 1; s >>= 1) {
        Acc o;
        o.ekin = __shfl_down(a.ekin, s); o.v2 = __shfl_down(a.v2, s); o.c2 = __shfl_down(a.c2, s);
        o.mass = __shfl_down(a.mass, s); o.flags = __shfl_down(a.flags, s);
        a.merge(o);
    }
    return a;
}

// block reduction; result valid in thread 0.  smem: at least blockDim.x/64 entries.
__device__ __forceinline__ Acc block_reduce(Acc a, Acc* smem) {
    a = wave_reduce(a);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) smem[w] = a;
    __syncthreads();
    Acc r; r.zero();
    if (threadIdx.x == 0) {
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r.merge(smem[i]);
    }
    __syncthreads();
    return r;
}

struct ScalarPartial { double ekin, v2, c2, mass, flags; };

// Full-array scalars of an arbitrary state (Problem.mass / kinetic_energy / v_max / v_sound,
// problem.py:334-352).  Rows [row0, row1] are summed: a slab leaves out halo rows it does not own.
template <int EOS>
__global__ __launch_bounds__(256) void k_scalars(const double* q, const double* topo, Layout L, Phys P, int row0,
                                                 int row1, ScalarPartial* out) {
    __shared__ Acc sm[4];
    Acc a; a.zero();
    const long long w = L.Ny + 2;
    const long long n = (long long)(row1 - row0 + 1) * w;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int ix = row0 + (int)(i / w), iy = (int)(i % w);
        const long long o = L.at(ix, iy);
        a.cell<EOS>(q[o], q[o + L.plane], q[o + 2 * L.plane], topo[o], P);
    }
    a = block_reduce(a, sm);
    if (threadIdx.x == 0) {
        ScalarPartial p = {a.ekin, a.v2, a.c2, a.mass, (double)a.flags};
        out[blockIdx.x] = p;
    }
}

// ---------------------------------------------------------------------------------------------
// commit: dt / residual / convergence bookkeeping (problem.py:565-586), one thread
// ---------------------------------------------------------------------------------------------
// dt_crit = min(dx, dy) / (v_max + v_sound)  (problem.py:354-356)
__device__ __forceinline__ double critical_dt(const StepState* st, double v2, double c2) {
    return st->hmin / (sqrt(v2) + sqrt(c2));
}

__device__ inline void commit_step(StepState* st, double ekin, double v2, double c2, int flags, LogEntry* log,
                                   long long log_base, long long log_cap) {
    if (flags & 4) c2 = __builtin_nan("");      // an imaginary sound speed somewhere: np.sqrt -> NaN -> max -> NaN
    flags &= 3;
    if (flags) {
        // invalid state: keep the pre-step field (parity not flipped) and stop (problem.py:588-610)
        st->invalid = (flags & 1) ? 1 : 2;
    } else {
        const double dt_crit = critical_dt(st, v2, c2);
        const double cfl = st->dt / dt_crit;
        const double res = fabs(ekin - st->ekin_old) / st->ekin_old / cfl;
        st->residual = res;
        if (st->rcount < 5) { st->rbuf[st->rcount++] = res; }
        else { st->rbuf[st->rhead] = res; st->rhead = (st->rhead + 1) % 5; }
        bool conv = true;
        for (int i = 0; i < st->rcount; ++i) conv = conv && (st->rbuf[i] < st->tol);
        st->converged = conv ? 1 : 0;
        st->ekin_old = ekin;
        st->ekin = ekin; st->vmax2 = v2; st->c2max = c2;
        st->step += 1;
        st->simtime += st->dt;
        st->dt_last = st->dt;
        if (st->adaptive) st->dt = st->CFL * dt_crit;
        st->parity ^= 1;
    }
    if (log) {
        const long long k = (flags ? st->step : st->step - 1) - log_base;
        if (k >= 0 && k < log_cap) {
            LogEntry e;
            e.step = st->step; e.simtime = st->simtime; e.dt = st->dt; e.ekin = st->ekin; e.ekin_old = st->ekin_old;
            e.residual = st->residual; e.v_max = sqrt(st->vmax2); e.v_sound = sqrt(st->c2max); e.mass = 0.0;
            e.invalid = st->invalid; e.converged = st->converged;
            log[k] = e;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// finish of the fused step: reduce the per-wave records of k_step and the per-block records of the edge kernel and
// commit.  Runs in the LAST block of the edge kernel to arrive (finish_if_last), so a step needs no separate launch.
// ---------------------------------------------------------------------------------------------
// Peer-to-peer slab transport (one process per GPU, mailboxes mapped into every peer through HIP IPC).
// A rank's mailbox lives in ITS OWN memory (fine-grained) and is written by its peers over xGMI:
//     flag[slot][r]     sequence number of rank r's latest message in this slot (release-stored last)
//     rec[slot][r][8]   rank r's record of the step (Ekin share, max v^2, max c^2, validity flags)
//     rows[slot][0|1]   the row that fills my row 0 (lower neighbour's last row) / my row Nx+1
// Two slots alternate with the sequence number: a rank can only send message n+2 after it has seen every
// rank's message n+1, which those ranks sent after consuming message n -- so slot n%2 is free again.
// ---------------------------------------------------------------------------------------------
constexpr long long P2P_TIMEOUT_TICKS = 3000000000ll;       // 30 s of the 100 MHz constant clock
constexpr int P2P_MAX_RANKS = 16;
constexpr size_t P2P_ROWS_OFFSET = 4096;      // bytes; header below is 2*16*8 + 2*16*64 = 2304
struct MailHeader {
    unsigned long long flag[2][P2P_MAX_RANKS];
    double rec[2][P2P_MAX_RANKS][8];
};
__host__ __device__ inline size_t p2p_mailbox_bytes(int pitch) { return P2P_ROWS_OFFSET + (size_t)2 * 2 * 3 * pitch * sizeof(double); }
__device__ inline double* p2p_rows(char* box, int slot, int side, int pitch) {
    return (double*)(box + P2P_ROWS_OFFSET) + (size_t)(slot * 2 + side) * 3 * pitch;
}
struct P2PArgs {
    int on, nranks, rank, rank_lo, rank_hi;   // rank_lo: whose LAST row fills my row 0; rank_hi: whose FIRST row fills my row Nx+1
    char* box[P2P_MAX_RANKS];                 // every rank's mailbox as mapped here (box[rank] = my own)
    unsigned long long* seq;                  // messages sent so far by this handle (== received from every peer)
    const double* qa; const double* qb;
};

struct FinishArgs {
    const Partial* partials;    // k_step's per-wave records (nstep_partials of them)
    StepState* st;
    LogEntry* log; long long log_base, log_cap;
    double* out;            // if non-null: slab mode, write the 8-double local record here instead of committing
    int honor_stop;
    Layout L; Edges E;
    int nstep_partials;
    Partial* block_partials;// one record per block of the edge kernel (its slice of k_step's records)
    unsigned int* arrive;   // blocks of the edge kernel that are done (reset by the last one)
    double* msg;            // slab, all-gather transport: [first row | last row] of the local message
    P2PArgs p2p;
};

// The hand-over of the per-block records to the block that finishes the step uses no fences.  A device-scope release
// writes back the XCD's whole L2 and costs 2-5 us PER BLOCK, serialised per XCD (measured: the same kernel with 160
// blocks fencing took 32 us, with 34 blocks 16 us; tools/fence_probe.hip), so instead every hand-off word is
//   - stored write-through by ONE lane (relaxed agent-scope atomic store = global_store ... sc1),
//   - drained (s_waitcnt vmcnt(0)) by that lane before it -- the same lane -- adds to the arrival counter,
//   - loaded by the last arriver with relaxed agent-scope atomic loads (sc1: served past the CU's L1), after the
//     workgroup barrier that follows the add whose return value told it that it is last
// (the 'write-through payload + counter + sc1 loads' form of the CDNA4 guide, Guideline 16 / MI355X_MICROARCH.md
// "Valid forms").  Everything else these blocks store -- ghost cells, stage-1 pairs, the slab message -- is read by
// LATER launches only.
// Memory-model note.  The fast path orders these hand-offs by construction (write-through stores, drained with
// s_waitcnt vmcnt(0) before the same lane's arrival add, sc1 loads after the add that identified the last arriver) rather
// than through the HIP/LLVM memory model: relaxed agent-scope atomics carry no ordering of their own.  What is assumed of
// gfx950: (1) a relaxed agent-scope atomic store is a write-through `global_store ... sc1` whose completion `vmcnt` counts;
// (2) atomic adds of one lane to the same L2 are performed in issue order once earlier stores have drained; (3) a relaxed
// agent-scope atomic load (`sc1`) is served from L2, past the CU's L1.  -DGPF_STRICT_ATOMICS builds the same kernels with
// release / acquire orders on the arrival add and on the record loads (gapflow_amd/build.py: build_strict_variant, built by
// __graft_entry__.build(); tests/test_gpu_extras.py runs goldens on both builds): results must be identical.
#ifdef GPF_STRICT_ATOMICS
#define GPF_ORDER_ARRIVE __ATOMIC_ACQ_REL
#define GPF_ORDER_LOAD __ATOMIC_ACQUIRE
#else
#define GPF_ORDER_ARRIVE __ATOMIC_RELAXED
#define GPF_ORDER_LOAD __ATOMIC_RELAXED
#endif

__device__ __forceinline__ void publish_partial(Partial* slot, const Acc& acc) {      // one lane
    unsigned long long* w = reinterpret_cast<unsigned long long*>(slot);
    __hip_atomic_store(w + 0, (unsigned long long)__double_as_longlong(acc.ekin), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(w + 1, (unsigned long long)__double_as_longlong(acc.v2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(w + 2, (unsigned long long)__double_as_longlong(acc.c2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(w + 3, (unsigned long long)__double_as_longlong((double)acc.flags), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void merge_published(Acc& acc, const Partial* slot) {
    const unsigned long long* w = reinterpret_cast<const unsigned long long*>(slot);
    const double ekin = __longlong_as_double((long long)__hip_atomic_load(w + 0, GPF_ORDER_LOAD, __HIP_MEMORY_SCOPE_AGENT));
    const double v2 = __longlong_as_double((long long)__hip_atomic_load(w + 1, GPF_ORDER_LOAD, __HIP_MEMORY_SCOPE_AGENT));
    const double c2 = __longlong_as_double((long long)__hip_atomic_load(w + 2, GPF_ORDER_LOAD, __HIP_MEMORY_SCOPE_AGENT));
    const double fl = __longlong_as_double((long long)__hip_atomic_load(w + 3, GPF_ORDER_LOAD, __HIP_MEMORY_SCOPE_AGENT));
    acc.ekin += ekin; acc.v2 = nanmax(acc.v2, v2); acc.c2 = nanmax(acc.c2, c2); acc.flags |= (int)fl;
}

// the last block: combine the blocks' records and commit, or -- slab -- publish this rank's record
// (pieces instead of a FinishArgs: k_step2 passes references into its own kernel arguments -- a by-value copy of the
// mailbox table, indexed by thread, would live in scratch memory)
__device__ inline void finish_tail(StepState* st, LogEntry* log, long long log_base, long long log_cap, double* out,
                                   const Partial* block_partials, const P2PArgs& p2p, Acc* sm) {
    __shared__ double rec_sm[8];
    Acc acc; acc.zero();
    for (int i = threadIdx.x; i < (int)gridDim.x; i += blockDim.x) merge_published(acc, block_partials + i);
    acc = block_reduce(acc, sm);
    if (threadIdx.x == 0) {
        if (out) {
            // slab mode: NaN maxima travel as +inf so that any reduction order keeps them
            const double inf = __builtin_inf();
            out[0] = acc.ekin;
            out[1] = acc.v2 != acc.v2 ? inf : acc.v2;
            out[2] = acc.c2 != acc.c2 ? inf : acc.c2;
            out[3] = (double)acc.flags;
            out[4] = out[5] = out[6] = out[7] = 0.0;
            for (int k = 0; k < 8; ++k) rec_sm[k] = out[k];
        } else {
            commit_step(st, acc.ekin, acc.v2, acc.c2, acc.flags, log, log_base, log_cap);
        }
    }
    if (p2p.on) {
        // message n = seq + 1: the rows are in the neighbours' mailboxes (send_rows_block: write-through stores,
        // drained before the blocks arrived); now the record to everybody, then the flags
        __syncthreads();
        const P2PArgs& c = p2p;
        const unsigned long long n = *c.seq + 1;
        const int slot = (int)(n & 1);
        if (threadIdx.x < c.nranks) {
            MailHeader* hd = (MailHeader*)c.box[threadIdx.x];
            for (int k = 0; k < 8; ++k) hd->rec[slot][c.rank][k] = rec_sm[k];
            // same lane, same peer: the system-scope release store orders the record before the flag
            __hip_atomic_store(&hd->flag[slot][c.rank], n, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

__device__ inline void finish_tail(const FinishArgs& a, Acc* sm) {
    finish_tail(a.st, a.log, a.log_base, a.log_cap, a.out, a.block_partials, a.p2p, sm);
}

// Called by every block of the edge kernel when its own work is done, with the reductions over the cells it wrote
// (`own`; zero for blocks that wrote none).  Folds in a slice of k_step's per-wave records, publishes one record per
// block, and lets the last block to arrive finish the step.
__device__ inline void finish_step(const FinishArgs& f, Acc own, Acc* sm) {
    __shared__ int last;
    const int per = (f.nstep_partials + (int)gridDim.x - 1) / (int)gridDim.x;
    const int i0 = blockIdx.x * per, i1 = min(i0 + per, f.nstep_partials);
    for (int i = i0 + threadIdx.x; i < i1; i += blockDim.x) {       // written by k_step, an earlier launch: plain loads
        const Partial p = f.partials[i];
        own.ekin += p.ekin; own.v2 = nanmax(own.v2, p.vmax2); own.c2 = nanmax(own.c2, p.c2max); own.flags |= (int)p.flags;
    }
    own = block_reduce(own, sm);        // its barriers also put every wave's drained row stores before the arrival
    if (threadIdx.x == 0) {
        publish_partial(f.block_partials + blockIdx.x, own);
        last = __hip_atomic_fetch_add(f.arrive, 1u, GPF_ORDER_ARRIVE, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1 : 0;
        if (last) __hip_atomic_store(f.arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!last) return;
    finish_tail(f, sm);
}

// A slab's first and last owned rows of the field the step has produced, ghost columns included (derived by the
// y rules, so this does not wait for the fill blocks): into the peers' mailboxes, or into the local all-gather message.
__device__ inline void send_rows_block(const double* q, const FinishArgs& f, int sblock, int nsblocks) {
    const Layout& L = f.L;
    double *dst_first, *dst_last;       // where my FIRST row (-> lower neighbour's row Nx+1) / LAST row goes
    if (f.p2p.on) {
        const P2PArgs& c = f.p2p;
        const int slot = (int)((*c.seq + 1) & 1);
        dst_first = (c.rank_lo >= 0 && f.E.halo[0]) ? p2p_rows(c.box[c.rank_lo], slot, 1, L.pitch) : nullptr;
        dst_last = (c.rank_hi >= 0 && f.E.halo[1]) ? p2p_rows(c.box[c.rank_hi], slot, 0, L.pitch) : nullptr;
    } else {
        dst_first = f.msg; dst_last = f.msg + 3 * L.pitch;
    }
    const int total = 6 * L.pitch;
    for (int t = sblock * blockDim.x + threadIdx.x; t < total; t += nsblocks * blockDim.x) {
        const int side = t / (3 * L.pitch), k = (t / L.pitch) % 3, i = t % L.pitch;
        double* dst = side ? dst_last : dst_first;
        if (!dst) continue;
        const int ix = side ? L.Nx : 1, iy = i - L.off;
        double v = 0.0;
        if (iy >= 1 && iy <= L.Ny) v = q[k * L.plane + (long long)ix * L.pitch + i];
        else if (iy == 0 || iy == L.Ny + 1) v = ghost_y(q, L, f.E, iy == 0 ? 2 : 3, k, ix);
        if (f.p2p.on)       // a peer's memory: write-through at system scope, drained below, before this block arrives
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst + k * L.pitch + i), (unsigned long long)__double_as_longlong(v),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        else dst[k * L.pitch + i] = v;
    }
    if (f.p2p.on) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---------------------------------------------------------------------------------------------
// ghost cells of the field k_step has just written (problem.py:576 -> 676-768), in parallel:
// one thread per ghost cell of the two ghost rows / two ghost columns; the four corners are
// rule_y(rule_x(.)) exactly as the reference's x-then-y order produces them.  Each block also
// leaves one reduction record covering the ghost cells it wrote (the reference's Ekin / v_max /
// v_sound run over ghost cells too, problem.py:342-347).
// ---------------------------------------------------------------------------------------------
struct GhostFillArgs {
    double* qa; double* qb;
    const StepState* st;
    Layout L; Edges E;
    int honor_stop;
};

template <int EOS>
__device__ __forceinline__ Acc ghost_fill_block(const GhostFillArgs& a, const Phys& P, int block, int nblocks) {
    const Layout& L = a.L;
    double* q = a.st->parity ? a.qa : a.qb;      // the buffer k_step has just written
    Acc acc; acc.zero();
    auto put = [&](int ix, int iy, const double v[3], double w) {
        const long long o = L.at(ix, iy);
        q[o] = v[0]; q[o + L.plane] = v[1]; q[o + 2 * L.plane] = v[2];
        acc.cell<EOS>(v[0], v[1], v[2], 0.0, P, w);
    };
    // work items: [0, 2*(Ny+2)) ghost rows incl. corners, then [.., + 2*Nx) ghost columns of interior rows
    const int nrow = 2 * (L.Ny + 2), ncol = 2 * L.Nx;
    for (int t = block * blockDim.x + threadIdx.x; t < nrow + ncol; t += nblocks * blockDim.x) {
        double v[3];
        if (t < nrow) {
            const int e = t / (L.Ny + 2), iy = t % (L.Ny + 2);
            if (a.E.halo[e]) continue;                        // filled by the neighbour exchange
            const int ix = e == 0 ? 0 : L.Nx + 1;
            if (iy >= 1 && iy <= L.Ny) {
                for (int c = 0; c < 3; ++c) v[c] = ghost_x(q, L, a.E, e, c, iy);
            } else {
                // corner: the y rule applied to the ghost-row value of the source column
                const int ey = iy == 0 ? 2 : 3;
                const int r0 = a.E.rule[ey][0];
                const int src = (r0 == BC_P) ? (ey == 2 ? L.Ny : 1) : (ey == 2 ? 1 : L.Ny);
                for (int c = 0; c < 3; ++c) {
                    const double gx = ghost_x(q, L, a.E, e, c, src);
                    v[c] = a.E.rule[ey][c] == BC_D ? 2.0 * a.E.value[ey] - gx : gx;
                }
            }
            put(ix, iy, v, 1.0);
        } else {
            const int u = t - nrow;
            const int e = 2 + u / L.Nx, ix = 1 + u % L.Nx;
            for (int c = 0; c < 3; ++c) v[c] = ghost_y(q, L, a.E, e, c, ix);
            // a row next to a periodic slab seam also stands in for the far slab's ghost row
            const double w = 1.0 + ((ix == 1 && a.E.halo[0] == 2) ? 1.0 : 0.0) + ((ix == L.Nx && a.E.halo[1] == 2) ? 1.0 : 0.0);
            put(ix, e == 2 ? 0 : L.Ny + 1, v, w);
        }
    }
    return acc;         // per thread; reduced in finish_step
}

// Edge work after k_step in ONE launch: blocks [0, nfill) write the ghost cells of the new field, the others (slabs
// only) ship the two boundary rows; the last block done finishes the step.
template <int EOS>
__global__ __launch_bounds__(256) void k_ghost_fill(const GhostFillArgs a, const FinishArgs f, int nfill, const Phys P) {
    __shared__ Acc sm[4];
    const StepState* st = a.st;
    if (st->invalid != 0 || (a.honor_stop && (st->converged || st->step >= st->max_it))) return;
    Acc own; own.zero();
    if ((int)blockIdx.x < nfill) own = ghost_fill_block<EOS>(a, P, blockIdx.x, nfill);
    else send_rows_block(st->parity ? a.qa : a.qb, f, blockIdx.x - nfill, gridDim.x - nfill);
    finish_step(f, own, sm);
}


// slab mode.  Every rank contributes ONE message per step to a single all-gather:
//     [ first interior row (3 x pitch) | last interior row (3 x pitch) | 8-double record ]
// taken from the field the step has just produced (buffer !parity); after the all-gather each rank scatters
// its neighbours' rows into its outer rows and reduces all records in rank order.
struct HaloArgs {
    double* qa; double* qb;
    double* msg;                // this rank's message, 6*pitch + 8 doubles
    const double* gathered;     // nranks messages (commit side)
    int rank_lo, rank_hi;       // rank whose LAST row fills my row 0 / whose FIRST row fills my row Nx+1; -1: none
    const StepState* st;
    Layout L; Edges E;
    int honor_stop;
    int work_parity;            // -1: the buffer the fused step wrote (!parity); 0/1: that buffer (stage-wise step)
    // shear thinning: the viscosity of a halo row needs grad p there, i.e. the density one row further into the
    // neighbour.  The message then carries two more rows (density of the second and of the second-to-last owned row)
    // and `beyond` receives the neighbours': [0] beyond row 0, [1] beyond row Nx+1 (pitch doubles each); else nullptr.
    double* beyond;
    int msg_rows;               // 6, or 8 with shear thinning (the two extra rows travel whether or not this slab uses them)
};
__global__ void k_halo_pack(const HaloArgs a) {
    if (a.st->invalid != 0 || (a.honor_stop && (a.st->converged || a.st->step >= a.st->max_it))) return;
    const double* q = a.work_parity >= 0 ? (a.work_parity ? a.qb : a.qa) : (a.st->parity ? a.qa : a.qb);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.L.pitch) return;
    for (int c = 0; c < 3; ++c) {
        a.msg[c * a.L.pitch + i] = q[c * a.L.plane + (long long)1 * a.L.pitch + i];
        a.msg[(3 + c) * a.L.pitch + i] = q[c * a.L.plane + (long long)a.L.Nx * a.L.pitch + i];
    }
    if (a.msg_rows == 8) {  // densities of the second and the second-to-last owned row (a one-row slab sends its only row)
        a.msg[6 * a.L.pitch + i] = q[(long long)min(2, a.L.Nx) * a.L.pitch + i];
        a.msg[7 * a.L.pitch + i] = q[(long long)max(a.L.Nx - 1, 1) * a.L.pitch + i];
    }
}
__global__ void k_halo_unpack(const HaloArgs a) {
    if (a.st->invalid != 0 || (a.honor_stop && (a.st->converged || a.st->step >= a.st->max_it))) return;
    double* q = a.work_parity >= 0 ? (a.work_parity ? a.qb : a.qa) : (a.st->parity ? a.qa : a.qb);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.L.pitch) return;
    const long long len = (long long)a.msg_rows * a.L.pitch + 8;
    for (int c = 0; c < 3; ++c) {
        if (a.E.halo[0] && a.rank_lo >= 0) q[c * a.L.plane + i] = a.gathered[a.rank_lo * len + (3 + c) * a.L.pitch + i];
        if (a.E.halo[1] && a.rank_hi >= 0)
            q[c * a.L.plane + (long long)(a.L.Nx + 1) * a.L.pitch + i] = a.gathered[a.rank_hi * len + c * a.L.pitch + i];
    }
    if (a.beyond) {     // beyond my row 0: the lower neighbour's second-to-last row; beyond my row Nx+1: the upper one's second
        if (a.E.halo[0] && a.rank_lo >= 0) a.beyond[i] = a.gathered[a.rank_lo * len + 7 * a.L.pitch + i];
        if (a.E.halo[1] && a.rank_hi >= 0) a.beyond[a.L.pitch + i] = a.gathered[a.rank_hi * len + 6 * a.L.pitch + i];
    }
}

// beyond rows of the new state = average of the working field's and the old state's, as the field itself (problem.py:563)
__global__ void k_beyond_average(double* state, const double* work, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) state[i] = (work[i] + state[i]) / 2.0;
}

// Receiving side of a slab step (either transport), fused with the start of the NEXT step.  Every block waits (bounded)
// for message seq+1 of every rank in this rank's own mailbox, works out what the commit will decide -- the same
// rank-ordered reduction of the records, the same dt -- scatters its share of the two neighbour rows into the outer
// rows and computes its share of the next step's stage-1 ghost values (reading the outer rows from the mailbox, since
// other blocks are still scattering them).  The last block to arrive writes the committed state.  Nobody waits for
// another block; no host, no collective library.
constexpr int INVALID_PEER_TIMEOUT = 3;
struct WaitArgs {
    double* qa; double* qb;
    StepState* st;
    LogEntry* log; long long log_base, log_cap;
    Layout L; Edges E;
    int honor_stop;
    unsigned int* arrive;       // [0] blocks done, [1] blocks whose wait for a peer timed out
    long long timeout_ticks;    // bound on the wait for the peers' flags (100 MHz constant clock)
    P2PArgs p2p;                // MAILBOX source: rank ids, mailboxes, message counter
    const double* gathered;     // all-gather source: nranks messages [first row | last row | record] of msg_len doubles
    long long msg_len;
    int nranks, rank_lo, rank_hi;
    int stage1;                 // also prepare the next step's stage-1 ghost data (the split step, GPF_STEP_UNFUSED_EDGES)
};
// MAILBOX: rows and records are waited for in this rank's mailbox (peer-to-peer transport); otherwise they are read from
// the buffer an all-gather has filled before this launch (same layout as gpf_slab_message, rank order).
template <int EOS, bool HAS_LS, bool PIEZO, bool XONLY, bool MAILBOX>
__global__ __launch_bounds__(256) void k_begin_slab(const GhostArgs g, const WaitArgs a, const Phys P) {
    __shared__ int missing, last;
    __shared__ double tiles[2][3][64];
    StepState* st = a.st;
    if (st->invalid != 0 || (a.honor_stop && (st->converged || st->step >= st->max_it))) return;
    const P2PArgs& c = a.p2p;
    const unsigned long long n = MAILBOX ? *c.seq + 1 : 0ull;
    const int slot = (int)(n & 1);
    MailHeader* hd = MAILBOX ? (MailHeader*)c.box[c.rank] : nullptr;
    const int nranks = MAILBOX ? c.nranks : a.nranks, rank_lo = MAILBOX ? c.rank_lo : a.rank_lo, rank_hi = MAILBOX ? c.rank_hi : a.rank_hi;
    if (threadIdx.x == 0) missing = 0;
    __syncthreads();
    if (MAILBOX && threadIdx.x < c.nranks) {
        const long long t0 = wall_clock64();
        bool ok = false;
        int probes = 0;
        for (;;) {
            // relaxed while spinning (an acquire would invalidate the caches on every probe); fenced once below
            ok = __hip_atomic_load(&hd->flag[slot][threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= n;
            if (ok) break;
            __builtin_amdgcn_s_sleep(2);
            if ((++probes & 255) == 0 && wall_clock64() - t0 > a.timeout_ticks) break;
        }
        if (!ok) atomicAdd(&missing, 1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    }
    __syncthreads();
    // A peer that never delivers must stop this handle, and the decision must be ONE decision: blocks time out on their
    // own clocks, so a flag that lands near the deadline is seen by some blocks and missed by others.  Every block --
    // timed out or not -- therefore goes through the arrival counter below; a block that missed a flag skips its share of
    // the work and says so in a second counter, and the last block to arrive commits only if nobody missed anything.
    const bool timed_out = missing != 0;
    // what the commit will decide (identical in every block and on every rank)
    double ekin = 0.0, v2 = 0.0, c2 = 0.0;
    int flags = 0;
    const Layout& L = a.L;
    for (int r = 0; r < nranks; ++r) {
        const double* p = MAILBOX ? hd->rec[slot][r] : a.gathered + r * a.msg_len + 6ll * L.pitch;
        ekin += p[0];
        v2 = fmax(v2, p[1]); c2 = fmax(c2, p[2]);
        flags |= (int)p[3];
    }
    const double inf = __builtin_inf();
    if (v2 == inf) v2 = __builtin_nan("");
    if (c2 == inf) c2 = __builtin_nan("");
    const int nblocks = gridDim.x * gridDim.y, block = blockIdx.y * gridDim.x + blockIdx.x;
    if (!timed_out && (flags & 3) == 0) {
        double* q = st->parity ? a.qa : a.qb;           // the field the step has produced (current after the commit)
        // my row 0 is the lower neighbour's LAST row, my row Nx+1 the upper neighbour's FIRST row
        const double* row_lo = !(rank_lo >= 0 && a.E.halo[0]) ? nullptr
                             : MAILBOX ? p2p_rows(c.box[c.rank], slot, 0, L.pitch) : a.gathered + rank_lo * a.msg_len + 3ll * L.pitch;
        const double* row_hi = !(rank_hi >= 0 && a.E.halo[1]) ? nullptr
                             : MAILBOX ? p2p_rows(c.box[c.rank], slot, 1, L.pitch) : a.gathered + rank_hi * a.msg_len;
        for (int t = block * blockDim.x + threadIdx.x; t < 6 * L.pitch; t += nblocks * blockDim.x) {
            const int side = t / (3 * L.pitch), k = (t / L.pitch) % 3, i = t % L.pitch;
            const double* src = side ? row_hi : row_lo;
            if (src) q[k * L.plane + (long long)(side ? L.Nx + 1 : 0) * L.pitch + i] = src[k * L.pitch + i];
        }
        // the split step: stage-1 ghost values of step+1 with the dt the commit is about to fix (the fused step kernel forms
        // them itself)
        if (a.stage1) {
        const double c2_eff = (flags & 4) ? __builtin_nan("") : c2;
        const double dt = st->adaptive ? st->CFL * critical_dt(st, v2, c2_eff) : st->dt;
        const int D = direction_of_step(st, st->step + 1);
        MailField fld;
        fld.q = q; fld.L = L; fld.row_lo = row_lo; fld.row_hi = row_hi;
        const int ntiles = (L.Ny + L.Nx + 63) / 64;
        for (int tile = block; tile < ntiles; tile += nblocks) ghost_stage1_tile<EOS, HAS_LS, PIEZO, XONLY>(fld, g, P, D, tile * 64, dt, tiles);
        }
    }
    // the last block to get here writes the committed state (the others have read everything they need from it)
    __syncthreads();
    if (threadIdx.x == 0) {
        if (timed_out) {
            __hip_atomic_fetch_add(a.arrive + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the miss is counted before this block arrives
        }
        last = __hip_atomic_fetch_add(a.arrive, 1u, GPF_ORDER_ARRIVE, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)nblocks - 1 ? 1 : 0;
    }
    __syncthreads();
    if (last && threadIdx.x == 0) {
        const unsigned int misses = __hip_atomic_load(a.arrive + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.arrive + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (misses) {
            st->invalid = INVALID_PEER_TIMEOUT;         // no commit, no sequence advance: the handle stops here
        } else {
            if (MAILBOX) *c.seq = n;
            commit_step(st, ekin, v2, c2, flags, a.log, a.log_base, a.log_cap);
        }
    }
}

// stage-wise slab step: validity from the pre-ghost-update scalars, values from the post-update ones (problem.py:565-578)
__global__ void k_record_unfused(const ScalarPartial* pre, const ScalarPartial* post, double* rec) {
    const double inf = __builtin_inf();
    rec[0] = post->ekin;
    rec[1] = post->v2 != post->v2 ? inf : post->v2;
    rec[2] = post->c2 != post->c2 ? inf : post->c2;
    rec[3] = (double)(((int)pre->flags & 3) | ((int)post->flags & 4));
    rec[4] = rec[5] = rec[6] = rec[7] = 0.0;
}

// slab mode: reduce the gathered per-slab records in rank order (identical on every rank), then commit
__global__ void k_commit_gathered(StepState* st, const double* gathered, long long len, long long rec_off, int nranks,
                                  LogEntry* log, long long log_base, long long log_cap, int honor_stop) {
    if (st->invalid != 0 || (honor_stop && (st->converged || st->step >= st->max_it))) return;
    double ekin = 0.0, v2 = 0.0, c2 = 0.0;
    int flags = 0;
    for (int r = 0; r < nranks; ++r) {
        const double* p = gathered + r * len + rec_off;
        ekin += p[0];
        v2 = fmax(v2, p[1]); c2 = fmax(c2, p[2]);
        flags |= (int)p[3];
    }
    const double inf = __builtin_inf();
    if (v2 == inf) v2 = __builtin_nan("");
    if (c2 == inf) c2 = __builtin_nan("");
    commit_step(st, ekin, v2, c2, flags, log, log_base, log_cap);
}

// reduce the per-block records of k_scalars; optionally (re)initialise the step state from them
__global__ __launch_bounds__(256) void k_scalars_final(const ScalarPartial* in, int n, ScalarPartial* out) {
    __shared__ Acc sm[4];
    Acc a; a.zero();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        Acc o; o.ekin = in[i].ekin; o.v2 = in[i].v2; o.c2 = in[i].c2; o.mass = in[i].mass; o.flags = (int)in[i].flags;
        a.merge(o);
    }
    a = block_reduce(a, sm);
    if (threadIdx.x == 0) {
        ScalarPartial p = {a.ekin, a.v2, a.c2, a.mass, (double)a.flags};
        out[0] = p;
    }
}

// ---------------------------------------------------------------------------------------------
// reference-ordered unfused pipeline, one kernel per reference function
// ---------------------------------------------------------------------------------------------
struct FieldPtrs {          // derived fields, padded planes
    double* p;              // 1
    double* tau;            // 3
    double* lower;          // 6
    double* upper;          // 6
};

// Pressure.update + WallStress.update (x and y objects summed) + BulkStress.update
template <int EOS, bool HAS_LS>
__global__ __launch_bounds__(256) void k_fields(const double* q, const double* topo, const double* Ls, FieldPtrs F,
                                                Layout L, Phys P) {
    const long long w = L.Ny + 2, n = (long long)(L.Nx + 2) * w;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long o = L.at((int)(i / w), (int)(i % w));
        CellIn c;
        c.rho = q[o]; c.jx = q[o + L.plane]; c.jy = q[o + 2 * L.plane];
        c.h = topo[o]; c.hx = topo[o + L.plane]; c.hy = topo[o + 2 * L.plane];
        c.Ls = HAS_LS ? Ls[o] : 0.0;
        CellFields f;
        cell_fields<EOS>(c, P, f);
        F.p[o] = f.p;
        for (int k = 0; k < 3; ++k) F.tau[o + k * L.plane] = f.tau[k];
        for (int k = 0; k < 6; ++k) { F.lower[o + k * L.plane] = f.lower[k]; F.upper[o + k * L.plane] = f.upper[k]; }
    }
}

// Shear thinning (stress.py:170-192, 314-326): two passes, because the viscosity of a cell needs grad p.
// Pass 1 stores p; pass 2 forms np.gradient(p) (central differences, one-sided at the ends of the array
// INCLUDING ghost cells), the thinned viscosity, and with it the stresses.
template <int EOS>
__global__ __launch_bounds__(256) void k_pressure(const double* q, double* p, Layout L, Phys P) {
    const long long w = L.Ny + 2, n = (long long)(L.Nx + 2) * w;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long o = L.at((int)(i / w), (int)(i % w));
        p[o] = eos_pressure<EOS>(q[o], P);
    }
}

// `beyond` (slabs only, else nullptr): density one row beyond row 0 ([0]) and beyond row Nx+1 ([1]) where that outer row
// is a neighbour's interior row (halo kind 1): np.gradient's central difference of the undivided array reaches there.
template <int EOS, bool HAS_LS>
__global__ __launch_bounds__(256) void k_fields_thinning(const double* q, const double* topo, const double* Ls, FieldPtrs F,
                                                         Layout L, Phys P, double dx, double dy, int halo_lo, int halo_hi,
                                                         const double* beyond) {
    const long long w = L.Ny + 2, n = (long long)(L.Nx + 2) * w;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int ix = (int)(i / w), iy = (int)(i % w);
        const long long o = L.at(ix, iy);
        CellIn c;
        c.rho = q[o]; c.jx = q[o + L.plane]; c.jy = q[o + 2 * L.plane];
        c.h = topo[o]; c.hx = topo[o + L.plane]; c.hy = topo[o + 2 * L.plane];
        c.Ls = HAS_LS ? Ls[o] : 0.0;
        const double pc = F.p[o];
        // np.gradient, edge_order 1
        const int ixm = ix > 0 ? ix - 1 : ix, ixp = ix < L.Nx + 1 ? ix + 1 : ix;
        const int iym = iy > 0 ? iy - 1 : iy, iyp = iy < L.Ny + 1 ? iy + 1 : iy;
        double dpx = (F.p[L.at(ixp, iy)] - F.p[L.at(ixm, iy)]) / ((ixp - ixm) * dx);
        if (beyond && ix == 0 && halo_lo == 1)
            dpx = (F.p[L.at(1, iy)] - eos_pressure<EOS>(beyond[L.off + iy], P)) / (2 * dx);
        if (beyond && ix == L.Nx + 1 && halo_hi == 1)
            dpx = (eos_pressure<EOS>(beyond[L.pitch + L.off + iy], P) - F.p[L.at(L.Nx, iy)]) / (2 * dx);
        const double dpy = (F.p[L.at(ix, iyp)] - F.p[L.at(ix, iym)]) / ((iyp - iym) * dy);
        const double mu0 = (P.piezo == PIEZO_NONE) ? P.eta : piezo_eta(P.eta, (EOS == EOS_BAYADA) ? c.rho : pc, P);
        const double eta = thinning_eta(mu0, dpx, dpy, c.h, P);
        CellFields f;
        cell_fields<EOS>(c, P, f, eta);
        for (int k = 0; k < 3; ++k) F.tau[o + k * L.plane] = f.tau[k];
        for (int k = 0; k < 6; ++k) { F.lower[o + k * L.plane] = f.lower[k]; F.upper[o + k * L.plane] = f.upper[k]; }
    }
}

// integrate.predictor_corrector: np.roll wraps over the whole (Nx+2)x(Ny+2) array (integrate.py:74-75)
__global__ __launch_bounds__(256) void k_fluxdiff(const double* q, const double* p, const double* tau, int d, double* fx,
                                                  double* fy, Layout L) {
    const long long w = L.Ny + 2, n = (long long)(L.Nx + 2) * w;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int ix = (int)(i / w), iy = (int)(i % w);
        const int ixn = (ix - d + (L.Nx + 2)) % (L.Nx + 2);      // roll(F, d)[ix] = F[ix-d]
        const int iyn = (iy - d + (L.Ny + 2)) % (L.Ny + 2);
        const long long o = L.at(ix, iy), ox = L.at(ixn, iy), oy = L.at(ix, iyn);
        const double Fx0 = q[o + L.plane], Fx1 = p[o] + tau[o], Fx2 = tau[o + 2 * L.plane];
        const double Fy0 = q[o + 2 * L.plane], Fy1 = Fx2, Fy2 = p[o] + tau[o + L.plane];
        const double Gx0 = q[ox + L.plane], Gx1 = p[ox] + tau[ox], Gx2 = tau[ox + 2 * L.plane];
        const double Gy0 = q[oy + 2 * L.plane], Gy1 = tau[oy + 2 * L.plane], Gy2 = p[oy] + tau[oy + L.plane];
        const double s = -(double)d;
        fx[o] = s * (Gx0 - Fx0); fx[o + L.plane] = s * (Gx1 - Fx1); fx[o + 2 * L.plane] = s * (Gx2 - Fx2);
        fy[o] = s * (Gy0 - Fy0); fy[o + L.plane] = s * (Gy1 - Fy1); fy[o + 2 * L.plane] = s * (Gy2 - Fy2);
    }
}

// integrate.source (integrate.py:117-130)
__global__ __launch_bounds__(256) void k_source(const double* q, const double* h, const double* tau, const double* lower,
                                                const double* upper, double* out, Layout L) {
    const long long w = L.Ny + 2, n = (long long)(L.Nx + 2) * w, S = L.plane;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long o = L.at((int)(i / w), (int)(i % w));
        const double h0 = h[o], hx = h[o + S], hy = h[o + 2 * S];
        out[o] = (-q[o + S] * hx - q[o + 2 * S] * hy) / h0;
        out[o + S] = ((tau[o] - upper[o]) * hx + (tau[o + 2 * S] - upper[o + 5 * S]) * hy + upper[o + 4 * S] - lower[o + 4 * S]) / h0;
        out[o + 2 * S] = ((tau[o + 2 * S] - upper[o + 5 * S]) * hx + (tau[o + S] - upper[o + S]) * hy + upper[o + 3 * S] - lower[o + 3 * S]) / h0;
    }
}

// q <- q - dt (fX/dx + fY/dy - src) over the whole array (problem.py:558)
// (dtp: the step size, read on the device: &st->dt for a step in progress, &st->dt_last when a finished step's predictor is redone)
__global__ __launch_bounds__(256) void k_axpy(double* q, const double* fx, const double* fy, const double* src,
                                              const double* dtp, double dx, double dy, Layout L) {
    const double dt = *dtp;
    const long long w = L.Ny + 2, n = (long long)(L.Nx + 2) * w;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long o = L.at((int)(i / w), (int)(i % w));
        for (int c = 0; c < 3; ++c) {
            const long long oc = o + c * L.plane;
            q[oc] = q[oc] - dt * (fx[oc] / dx + fy[oc] / dy - src[oc]);
        }
    }
}

// q <- (q + q0)/2 (problem.py:563)
__global__ __launch_bounds__(256) void k_average(double* q, const double* q0, Layout L) {
    const long long w = L.Ny + 2, n = (long long)(L.Nx + 2) * w;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long o = L.at((int)(i / w), (int)(i % w));
        for (int c = 0; c < 3; ++c) q[o + c * L.plane] = (q[o + c * L.plane] + q0[o + c * L.plane]) / 2.0;
    }
}

__global__ void k_copy3(const double* src, double* dst, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        dst[i] = src[i];
}

// commit for the unfused path: validity is judged BEFORE the final ghost update (problem.py:565),
// scalars after it (problem.py:576-578); both from k_scalars records.
__global__ void k_commit_unfused(StepState* st, const ScalarPartial* pre_bc, const ScalarPartial* post_bc) {
    const int flags = ((int)pre_bc->flags & 3) | ((int)post_bc->flags & 4);
    commit_step(st, post_bc->ekin, post_bc->v2, post_bc->c2, flags, nullptr, 0, 0);
}

// ---------------------------------------------------------------------------------------------
// stateless closure operators (GaPFlow/models/viscous.py, pressure.py, sound.py as functions)
// ---------------------------------------------------------------------------------------------
struct ViscousArgs {
    const double* q; const double* h; const double* dqx; const double* dqy;     // [3][n] each; dq* may be null (= 0)
    const double* eta; const double* Ls;                                        // [n]
    double U, V, zeta;
    int slip_both;              // 0: only the upper wall slips (viscous.py's first branch); 1: both walls
    double* out[3];             // lower wall [6][n], upper wall [6][n], gap average [3][n]; any may be null
    long long n;
};
__global__ __launch_bounds__(256) void k_viscous(const ViscousArgs a) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < a.n; i += (long long)gridDim.x * blockDim.x) {
        double q[3], h[3], dx[3] = {0.0, 0.0, 0.0}, dy[3] = {0.0, 0.0, 0.0};
        for (int c = 0; c < 3; ++c) {
            q[c] = a.q[c * a.n + i]; h[c] = a.h[c * a.n + i];
            if (a.dqx) dx[c] = a.dqx[c * a.n + i];
            if (a.dqy) dy[c] = a.dqy[c * a.n + i];
        }
        const double Ls = a.Ls[i], lo = a.slip_both ? Ls : 0.0;
        for (int w = 0; w < 3; ++w) {
            if (!a.out[w]) continue;
            double t[6];
            viscous_general(w, q, h, dx, dy, a.U, a.V, a.eta[i], a.zeta, lo, Ls, t);
            if (w < 2) {
                for (int c = 0; c < 6; ++c) a.out[w][c * a.n + i] = t[c];
            } else {
                a.out[2][i] = t[0]; a.out[2][a.n + i] = t[1]; a.out[2][2 * a.n + i] = t[5];     // xx, yy, xy (viscous.py:612)
            }
        }
    }
}

template <int EOS>
__global__ __launch_bounds__(256) void k_eos(const double* rho, long long n, Phys P, double* p, double* c) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        if (p) p[i] = eos_pressure<EOS>(rho[i], P);
        if (c) c[i] = sqrt(eos_c2<EOS>(rho[i], P));
    }
}

// models/viscosity.py: kind 0 piezoviscosity(arg, mu0), 1 shear_thinning_factor(rate, mu0), 2 shear_rate_avg(a0, a1, a2; u1, u2, mu0)
__global__ __launch_bounds__(256) void k_viscosity(int kind, const double* a0, const double* a1, const double* a2, long long n,
                                                   double mu0, double u1, double u2, Phys P, double* out) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        if (kind == 0) out[i] = piezo_eta(mu0, a0[i], P);
        else if (kind == 1) out[i] = thinning_factor(a0[i], mu0, P);
        else out[i] = shear_rate_avg(a0[i], a1[i], a2[i], u1, u2, mu0);
    }
}

}  // namespace gpf
