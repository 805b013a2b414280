// Fused MacCormack step, second generation: two columns per lane, 16-byte accesses, ONE launch per step.
//
// Same arithmetic and the same composed radius-1 stencil as step_kernel.hip (see the header there for the
// reference lines it replaces: problem.py:528-563, stress.py:289-622, integrate.py:38-130, problem.py:319-357,
// 571-586, 676-768).  What changed is the mapping to the machine:
//
//   * a wavefront owns a window of 128 consecutive columns -- lane l holds the pair (2l, 2l+1) of the window --
//     and produces 126 of them; every global access is ONE 16-byte load/store per lane (1 KB per wave
//     instruction) on a 16-byte boundary.  Half of the y-neighbours are the lane's own second column, the
//     other half one DPP/permute away: 6 cross-lane moves per row for 2 cells instead of 6 per cell.
//   * the strip halo drops from 2/64 to 2/128 columns, and the blocks are numbered so that neighbouring strips
//     run on the same XCD (blocks are dealt round-robin over the 8 XCDs, each with its own L2): the shared
//     halo lines are L2 hits instead of second HBM fetches.
//   * single-handle problems need no other launch: the wave that finishes the row next to an edge also writes
//     the ghost cells derived from it (problem.py:676-768, corners as rule_y(rule_x(.))), the waves of the last
//     row chunk / the last strip form the stage-1 field on the downwind ghost row / column themselves
//     (problem.py:560), and the last workgroup to finish reduces the per-block records and commits dt,
//     residual and step count on the device (problem.py:571-586).
//   * slabs (outer rows filled by a neighbour exchange) run the same launch: the waves that finish a slab's first /
//     last row also copy it -- ghost columns included -- into the all-gather message or straight into the
//     neighbour's mailbox, the stage-1 field across a periodic seam is formed from the halo row and the seam's
//     topography, and the last workgroup leaves the rank's RECORD (and, peer-to-peer, the sequence flags) instead of
//     committing; k_begin_slab (aux_kernels.hip) scatters the neighbours' rows and commits after the exchange.
//   * `fused = 0` (GPF_STEP_UNFUSED_EDGES=1) is the older split form, kept as a cross-check: ghost data from
//     g1x / g1y, edge kernels of step_kernel.hip / aux_kernels.hip before and after.
//
// Logical indices as in step_kernel.hip: n (rows) and m (columns) increase DOWNWIND of the predictor,
//   ix = n for D = +1, Nx+1-n for D = -1, likewise iy from m.
//
// Included by api.hip after step_kernel.hip (ghost_rule, halted, predictor_direction) and aux_kernels.hip (Acc,
// block_reduce, publish_partial, merge_published, commit_step).
#include <hip/hip_runtime.h>
#include "device_types.hpp"

namespace gpf {

constexpr int STRIP2 = 126;     // output columns per wavefront
#ifndef GPF_K2_AHEAD
#define GPF_K2_AHEAD 2          // rows requested ahead of the one being computed (1 .. 6)
#endif
#ifndef GPF_K2_AHEAD_LINE
#define GPF_K2_AHEAD_LINE 4     // ... for the x-only-gap kernels: three loads per row instead of six, so two rows ahead keep only
                                // 6 KB per wave in flight -- with one wave per SIMD (long marches, plan_step2) less than the
                                // bandwidth-latency product; 4 rows measured 1.7 % faster there, equal elsewhere.  (3 is unusable:
                                // with four row buffers hipcc renames them across the back edge and copies registers whose
                                // loads are in flight -- tools/audit_step_isa.py reports it; so is 5.  6 is clean -- 238 registers, two
                                // waves per SIMD still -- and buys nothing: 154.9 / 156.7 against 153.9 / 152.8 us, paired handles.)
#endif
#ifndef GPF_K2_MINWAVES
#define GPF_K2_MINWAVES 2       // waves per SIMD the register allocation is held to
#endif
#ifndef GPF_K2_MINWAVES_LINE
#define GPF_K2_MINWAVES_LINE GPF_K2_MINWAVES    // ... for the x-only-gap kernels (half the row buffers)
#endif
// Instantiations whose closures need more than the 256 registers of two waves per SIMD -- the equations of state that go through
// pow / exp / log (power law, Murnaghan-Tait, BWR, Bayada-Chupin) and the slip-length field together with piezo-viscosity --
// are compiled for ONE wave per SIMD (512 registers): held to 256 they spill into the march loop, which both costs the
// scratch traffic and makes hipcc drain the pipelined row loads (4096^2, Murnaghan-Tait: 535 us per step).  They also request
// fewer rows ahead.  plan_step2 reads the occupancy back and sizes the row chunks for it.
template <int EOS, bool HAS_LS, bool PIEZO>
struct Step2Weight {
    static constexpr bool heavy = EOS == EOS_PL || EOS == EOS_MT || EOS == EOS_BWR || EOS == EOS_BAYADA || (HAS_LS && PIEZO);
};

// Where the windows sit, per predictor direction (host: strip2_geom in api.hip).
//   logical column of window position p of strip s:  m = w0 + 126 s + p,  p = 2 lane + slot
// w0 is chosen so that (a) every lane's pair starts on a 16-byte boundary in memory for this direction and
// (b) the strip that produces column Ny has two idle lanes behind the ghost column Ny+1 ("wrap lanes").
struct Strip2Geom {
    int w0;
    int nstrips;
    int s_ghost, lane_ghost, slot_ghost;    // strip / lane / slot of logical column Ny+1 (the downwind ghost column)
    int lane_wrap;                          // first of the two wrap lanes of strip s_ghost
    int wrap_ma;                            // logical column of slot 0 of lane_wrap (0 or -1: the pair holding column 0)
    int wrap_src_lane, wrap_src_slot;       // where logical column 1 sits among the wrap lanes
};

struct Step2Args {
    const double* qa; const double* qb;
    const double* topo; const double* topo_line; const double* Ls;
    const double* g1x; const double* g1y;   // fused = 0: stage-1 ghost data prepared by k_ghost_stage1 / k_begin_slab
    const double* seam[2];                  // topography rows across a periodic slab seam (gpf_set_seam_topo), per x edge
    double* out;                            // slab: the rank's 8-double record goes here and nothing is committed
    double* msg;                            // slab, all-gather transport: [first row | last row] of the local message
    P2PArgs p2p;                            // slab, peer-to-peer transport: the neighbours' mailboxes
    StepState* st;
    Partial* partials;                      // fused = 0: one record per wave (chunk-major), folded by k_ghost_fill
    Partial* block_partials;                // fused = 1: one record per block, folded by the last block
    unsigned int* arrive;
    LogEntry* log; long long log_base, log_cap;
    Layout L; Edges E; Strip2Geom G;
    int nchunks;
    int fused;                              // bit 0: edge work inside this kernel; bit 1: non-temporal stores, bit 2: non-temporal loads (plan_step2)
    int honor_stop;
};

// cross-lane moves by one lane in the whole wavefront (GFX9 DPP wave shifts: a VALU move, no LDS round trip)
__device__ __forceinline__ double lane_from_below(double v) {          // lane l receives lane l-1's value
#ifdef GPF_SHUFFLE_BPERMUTE
    return __shfl_up(v, 1);
#else
    const long long b = __double_as_longlong(v);
    int lo = (int)b, hi = (int)(b >> 32);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);   // wave_shr:1
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
#endif
}
__device__ __forceinline__ double lane_from_above(double v) {          // lane l receives lane l+1's value
#ifdef GPF_SHUFFLE_BPERMUTE
    return __shfl_down(v, 1);
#else
    const long long b = __double_as_longlong(v);
    int lo = (int)b, hi = (int)(b >> 32);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);   // wave_shl:1
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
#endif
}

// The row loads of the march are issued through inline asm so that THIS file, not the compiler, decides when a
// wave waits for them.  hipcc's own s_waitcnt placement is exact in straight-line code but gives up at the loop back
// edge and at the exec-masked store branches: it drained every outstanding load (vmcnt(0)) once per row, which put
// the full HBM latency of the row just requested in front of every row's arithmetic.  vmcnt counts loads and stores
// together in issue order, so "at most N younger operations outstanding" needs only a LOWER bound N on the number of
// vector-memory instructions issued after the loads being waited for: the loads of the rows requested since (the
// compiler's stores in between only make the wait stricter, never unsafe).  Form (ii) of the CDNA4 guide, section 5.7:
// "=v" loads, then one wait statement naming every destination "+v" before the first consumer.
typedef double dpair __attribute__((ext_vector_type(2)));
// Cache policy of the march's traffic.  Every byte of q is written once per step and not read again before the next launch, so
// nothing is gained by keeping the stores in L2 / the Infinity Cache, and on this chip streaming with the non-temporal hint is
// measurably faster: an elementwise 3-in / 3-out kernel of the benchmark's size moves 5.7 TB/s with plain accesses and 6.5 TB/s
// with `nt` ones (tools/stream_probe.hip, profiles/r03_stream/).  In the march (one box, alternating runs, 4096^2, x-only gap /
// 2-D gap kernel): plain 189.6 / 279.0 us, nt stores 166.6 / 254.3, nt loads 195.8 / 265.4, both 178.4 / 261.8 -- the loads
// share their halo lines (two columns, two rows per window) with the neighbouring waves through L2, which the hint shortens.
// GPF_K2_NT: bit 0 loads, bit 1 stores; default: stores only.
#ifndef GPF_K2_NT
#define GPF_K2_NT 2
#endif
__device__ __forceinline__ void asm_load16(dpair& dst, const double* lane_ptr) {
#if GPF_K2_NT & 1
    asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(dst) : "v"(lane_ptr) : "memory");
#else
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(lane_ptr) : "memory");
#endif
}
template <int N>
__device__ __forceinline__ void asm_wait3(dpair& a, dpair& b, dpair& c) {
    asm volatile("s_waitcnt vmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void asm_wait1(dpair& a) {
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a) : "n"(N) : "memory");
}

// A wave-uniform load of data this kernel never writes (the topography profile, the slabs' stage-1 ghost column):
// through the constant address space, i.e. the scalar cache (s_load, counted by lgkmcnt).  As a plain load hipcc
// issues it as a vector load, and its vmcnt bookkeeping for that load -- blind to the asm row loads -- drains them.
__device__ __forceinline__ double uniform_load(const double* p) {
    typedef const __attribute__((address_space(4))) double* cptr;
    return *(cptr)(unsigned long long)p;
}

// one row as loaded: .x is the lower physical column of the lane's pair
template <int TOPO, bool HAS_LS>
struct RawRow {
    dpair q[3];
    dpair t[TOPO == 0 ? 3 : 1];     // h, hx, hy planes (TOPO = 0 only)
    dpair ls[HAS_LS ? 1 : 1];
    double th, thx, thy;            // TOPO = 1, 3: the row's (h, hx, hy), wave-uniform
};

// two adjacent cells of one row, as one lane holds them
struct Row2 { double rho[2], jx[2], jy[2], h[2], hx[2], hy[2], Ls[2]; };

// running reductions of one lane over the cells it wrote (ghost cells included, problem.py:342-347)
template <int EOS>
struct Red {
    double ekin, v2, c2;
    int flags;
    __device__ __forceinline__ void init() { ekin = 0.0; v2 = 0.0; c2 = (EOS == EOS_DH) ? __builtin_inf() : 0.0; flags = 0; }
    // Flags instead of NaN-propagating maxima (see step_kernel.hip): a NaN in any component makes v2 NaN (flag 1);
    // an imaginary sound speed raises flag 4 and commit_step turns c2max into NaN like np.sqrt(...).max().
    __device__ __forceinline__ void cell(double r, double jx, double jy, double w, const Phys& P) {
        const double v = (jx * jx + jy * jy) * rcp(r);
        ekin += w * (v * 0.5);
        v2 = fmax(v2, v);
        if (v != v) flags |= 1;
        if (r < 0.0) flags |= 2;
        if (EOS == EOS_DH) {
            // dp/drho = K / (C2 rho0 - rho)^2 peaks at the cell nearest the pole: one reciprocal per wave at the end
            c2 = fmin(c2, fabs(P.e[7] - r));
        } else {
            const double c = eos_c2<EOS>(r, P);
            if (!(c >= 0.0)) flags |= 4;
            c2 = fmax(c2, c);
        }
    }
};

template <int EOS, bool HAS_LS, bool PIEZO, int D, int TOPO>
__device__ __forceinline__ void step_strip2(const Step2Args& a, const Phys& P, const double* __restrict__ qin,
                                            double* __restrict__ qout, int strip, int chunk, int lane,
                                            double (*stash)[128], Acc& result) {
    const Layout L = a.L;
    const Strip2Geom G = a.G;
    const bool fused = (a.fused & 1) != 0;
    // the cache policy rides in the same word and is tested bit by bit where it is used: the march has no scalar register to spare
    // (one more live value tips some instantiations into a dead 16-byte spill slot, and a launch with a private segment costs ~4 us)
    const int policy = __builtin_amdgcn_readfirstlane(a.fused);

    // ---- columns of this lane ----
    const bool y_periodic = a.E.rule[2][0] == BC_P;
    const bool ghost_strip = strip == G.s_ghost;                                // wave-uniform
    const bool wrap_strip = fused && y_periodic && ghost_strip;
    const bool wrap_lane = wrap_strip && (lane == G.lane_wrap || lane == G.lane_wrap + 1);
    const int ma = wrap_lane ? G.wrap_ma + 2 * (lane - G.lane_wrap) : G.w0 + strip * STRIP2 + 2 * lane;
    const int iy_lo_raw = D > 0 ? ma : L.Ny - ma;                               // lower physical column of the pair (odd)
    const bool lane_ok = iy_lo_raw >= -1 && iy_lo_raw <= L.Ny + 1;
    const int iy_lo = lane_ok ? iy_lo_raw : -1;                                 // idle lanes read the pair (-1, 0)
    const int iy0 = D > 0 ? iy_lo : iy_lo + 1, iy1 = D > 0 ? iy_lo + 1 : iy_lo; // physical column of slot 0 / 1
    const bool out0 = lane_ok && !wrap_lane && lane >= 1 && ma >= 1 && ma <= L.Ny;
    const bool out1 = lane_ok && !wrap_lane && lane <= 62 && ma + 1 >= 1 && ma + 1 <= L.Ny;
    const bool is_ghost_lane = ghost_strip && lane == G.lane_ghost;
    // source of the ghost column's stage-1 value (fused): periodic -> logical column 1 among the wrap lanes,
    // otherwise logical column Ny, the slot just upwind of the ghost slot
    const int gsrc_lane = y_periodic ? G.wrap_src_lane : (G.slot_ghost ? G.lane_ghost : G.lane_ghost - 1);
    const int gsrc_slot = y_periodic ? G.wrap_src_slot : (G.slot_ghost ? 0 : 1);

    // ---- rows of this chunk: outputs n_first..n_last, marching n_first-1 .. n_last+1 ----
    const int n_first = 1 + (int)(((long long)chunk * L.Nx) / a.nchunks);
    const int n_last = (int)(((long long)(chunk + 1) * L.Nx) / a.nchunks);
    const int e_dw_x = D > 0 ? 1 : 0, e_dw_y = D > 0 ? 3 : 2;                    // downwind edges
    const bool dw_row_is_ghost = (n_last == L.Nx) && a.E.halo[e_dw_x] != 1;
    const bool x_periodic = a.E.rule[0][0] == BC_P;

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

    // TOPO: most gap profiles vary along one axis only (journal, inclined, parabolic, cdc: h = h(x)); a third of the
    // step's HBM reads is then redundant.  1 = one (h, hx, hy) triple per ROW through the scalar cache, 2 = the lane's
    // two column triples stay in registers for the whole march (flipped geometries); the values are bitwise those of
    // the planes.  3 = as 1 with dh/dy = 0 throughout: the x-only-gap closure (closures.hpp, cell_closure_xonly).
    double lh[2] = {0.0, 0.0}, lhx[2] = {0.0, 0.0}, lhy[2] = {0.0, 0.0};
    if (TOPO == 2) {
        const int c0 = min(max(iy0, 0), L.Ny + 1), c1 = min(max(iy1, 0), L.Ny + 1);
        lh[0] = a.topo_line[c0]; lhx[0] = a.topo_line[(L.Ny + 2) + c0]; lhy[0] = a.topo_line[2 * (L.Ny + 2) + c0];
        lh[1] = a.topo_line[c1]; lhx[1] = a.topo_line[(L.Ny + 2) + c1]; lhy[1] = a.topo_line[2 * (L.Ny + 2) + c1];
    }
    // Addressing: the row base is wave-uniform (scalar registers), the lane adds a constant 32-bit byte offset --
    // the `global_load_dwordx4 v, v_off, s[base]` form, no 64-bit vector address arithmetic in the loop.
    const unsigned int lane_bytes = (unsigned int)(L.off + iy_lo) * 8u;     // multiple of 16 (off + iy_lo is even)
    auto pair = [&](const double* __restrict__ rowp, double& s0, double& s1) {
        const double2 v = *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(rowp) + lane_bytes);
        s0 = D > 0 ? v.x : v.y; s1 = D > 0 ? v.y : v.x;
    };
    auto load = [&](int n, Row2& r) {
        const int ix = D > 0 ? n : L.Nx + 1 - n;
        const long long rb = (long long)ix * L.pitch;       // wave-uniform
        pair(q0p + rb, r.rho[0], r.rho[1]); pair(q1p + rb, r.jx[0], r.jx[1]); pair(q2p + rb, r.jy[0], r.jy[1]);
        if (TOPO == 0) {
            pair(hp + rb, r.h[0], r.h[1]); pair(hxp + rb, r.hx[0], r.hx[1]); pair(hyp + rb, r.hy[0], r.hy[1]);
        } else if (TOPO == 1 || TOPO == 3) {
            const double th = a.topo_line[ix], thx = a.topo_line[(L.Nx + 2) + ix], thy = TOPO == 1 ? a.topo_line[2 * (L.Nx + 2) + ix] : 0.0;
            r.h[0] = r.h[1] = th; r.hx[0] = r.hx[1] = thx; r.hy[0] = r.hy[1] = thy;
        } else {
            r.h[0] = lh[0]; r.h[1] = lh[1]; r.hx[0] = lhx[0]; r.hx[1] = lhx[1]; r.hy[0] = lhy[0]; r.hy[1] = lhy[1];
        }
        if (HAS_LS) pair(a.Ls + rb, r.Ls[0], r.Ls[1]);
        else r.Ls[0] = r.Ls[1] = 0.0;
    };
    // ---- the pipelined row loads of the march ----
    constexpr int NL = 3 + (TOPO == 0 ? 3 : 0) + (HAS_LS ? 1 : 0);       // vector loads per row (TOPO 1, 3: the gap travels as scalars)
    typedef RawRow<TOPO, HAS_LS> Raw;
    const double* const lane_q = reinterpret_cast<const double*>(reinterpret_cast<const char*>(qin) + lane_bytes);
    const double* const lane_t = reinterpret_cast<const double*>(reinterpret_cast<const char*>(a.topo) + lane_bytes);
    const double* const lane_ls = HAS_LS ? reinterpret_cast<const double*>(reinterpret_cast<const char*>(a.Ls) + lane_bytes) : nullptr;
    auto issue = [&](int n, Raw& r) {
        // Beyond the chunk's last row nothing is needed, but the number of loads in flight behind a row must not change
        // (the waits count them): the request then goes, for every plane, to the density row just requested -- one
        // cached kilobyte instead of a row of every plane.
        const bool dummy = n > n_last + 1;
        const int ix = D > 0 ? min(n, n_last + 1) : L.Nx + 1 - min(n, n_last + 1);
        const long long rb = (long long)ix * L.pitch;       // wave-uniform
        const long long p1 = dummy ? 0 : L.plane, p2 = dummy ? 0 : 2 * L.plane;
        // ONE asm statement requests the whole row, with or without the non-temporal hint (plan_step2 decides per handle: the hint
        // is worth 2-3 % where a step streams through HBM): the choice is a scalar branch inside the statement -- an if / else
        // around two sets of loads would make hipcc join their destinations with copies of registers whose loads are in flight.
        // "=&v": a destination must not share registers with the address of a later load of the same statement.
        const double* const a0 = lane_q + rb; const double* const a1 = lane_q + p1 + rb; const double* const a2 = lane_q + p2 + rb;
#define GPF_LD(d, a) "global_load_dwordx4 %" #d ", %" #a ", off\n\t"
#define GPF_LDNT(d, a) "global_load_dwordx4 %" #d ", %" #a ", off nt\n\t"
#define GPF_ROW_HEAD "s_cmp_lg_u32 %[nt], 0\n\ts_cbranch_scc1 .Lgpf_ldnt_%=\n\t"
#define GPF_ROW_MID "s_branch .Lgpf_lddone_%=\n.Lgpf_ldnt_%=:\n\t"
#define GPF_ROW_TAIL "\n.Lgpf_lddone_%=:"
        if constexpr (TOPO == 0 && HAS_LS) {
            const double* t0 = dummy ? lane_q : lane_t;
            const double* const a3 = t0 + rb; const double* const a4 = t0 + p1 + rb; const double* const a5 = t0 + p2 + rb;
            const double* const a6 = (dummy ? lane_q : lane_ls) + rb;
            asm volatile(GPF_ROW_HEAD GPF_LD(0, 7) GPF_LD(1, 8) GPF_LD(2, 9) GPF_LD(3, 10) GPF_LD(4, 11) GPF_LD(5, 12) GPF_LD(6, 13) GPF_ROW_MID
                         GPF_LDNT(0, 7) GPF_LDNT(1, 8) GPF_LDNT(2, 9) GPF_LDNT(3, 10) GPF_LDNT(4, 11) GPF_LDNT(5, 12) GPF_LDNT(6, 13) GPF_ROW_TAIL
                         : "=&v"(r.q[0]), "=&v"(r.q[1]), "=&v"(r.q[2]), "=&v"(r.t[0]), "=&v"(r.t[1]), "=&v"(r.t[2]), "=&v"(r.ls[0])
                         : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), [nt] "s"(policy & 4) : "memory", "scc");
        } else if constexpr (TOPO == 0) {
            const double* t0 = dummy ? lane_q : lane_t;
            const double* const a3 = t0 + rb; const double* const a4 = t0 + p1 + rb; const double* const a5 = t0 + p2 + rb;
            asm volatile(GPF_ROW_HEAD GPF_LD(0, 6) GPF_LD(1, 7) GPF_LD(2, 8) GPF_LD(3, 9) GPF_LD(4, 10) GPF_LD(5, 11) GPF_ROW_MID
                         GPF_LDNT(0, 6) GPF_LDNT(1, 7) GPF_LDNT(2, 8) GPF_LDNT(3, 9) GPF_LDNT(4, 10) GPF_LDNT(5, 11) GPF_ROW_TAIL
                         : "=&v"(r.q[0]), "=&v"(r.q[1]), "=&v"(r.q[2]), "=&v"(r.t[0]), "=&v"(r.t[1]), "=&v"(r.t[2])
                         : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), [nt] "s"(policy & 4) : "memory", "scc");
        } else if constexpr (HAS_LS) {
            const double* const a3 = (dummy ? lane_q : lane_ls) + rb;
            asm volatile(GPF_ROW_HEAD GPF_LD(0, 4) GPF_LD(1, 5) GPF_LD(2, 6) GPF_LD(3, 7) GPF_ROW_MID
                         GPF_LDNT(0, 4) GPF_LDNT(1, 5) GPF_LDNT(2, 6) GPF_LDNT(3, 7) GPF_ROW_TAIL
                         : "=&v"(r.q[0]), "=&v"(r.q[1]), "=&v"(r.q[2]), "=&v"(r.ls[0])
                         : "v"(a0), "v"(a1), "v"(a2), "v"(a3), [nt] "s"(policy & 4) : "memory", "scc");
        } else {
            asm volatile(GPF_ROW_HEAD GPF_LD(0, 3) GPF_LD(1, 4) GPF_LD(2, 5) GPF_ROW_MID GPF_LDNT(0, 3) GPF_LDNT(1, 4) GPF_LDNT(2, 5) GPF_ROW_TAIL
                         : "=&v"(r.q[0]), "=&v"(r.q[1]), "=&v"(r.q[2])
                         : "v"(a0), "v"(a1), "v"(a2), [nt] "s"(policy & 4) : "memory", "scc");
        }
#undef GPF_LD
#undef GPF_LDNT
#undef GPF_ROW_HEAD
#undef GPF_ROW_MID
#undef GPF_ROW_TAIL
        if (TOPO == 1 || TOPO == 3) {
            r.th = uniform_load(a.topo_line + ix); r.thx = uniform_load(a.topo_line + (L.Nx + 2) + ix);
            r.thy = TOPO == 1 ? uniform_load(a.topo_line + 2 * (L.Nx + 2) + ix) : 0.0;
        }
    };
    // wait until row r has landed: exactly AHEAD row requests (NL loads each) have been issued after it
    constexpr bool HEAVY = Step2Weight<EOS, HAS_LS, PIEZO>::heavy;
    // (TOPO 1 -- a row profile with all three of h, hx, hy, reached only through GPF_TOPO_GENERIC -- keeps six scalar registers per
    // row buffer: with five buffers hipcc runs out of them)
    constexpr int AHEAD_ROWS = PIEZO ? 1 : (TOPO == 3 ? (HEAVY ? 2 : GPF_K2_AHEAD_LINE) : GPF_K2_AHEAD);
    // The stores of the rows finished in between are vector-memory operations too and sit in the same counter, in issue order:
    // once the march is AHEAD rows past its first output row, every loop body since row r's request has issued three of them
    // (a wave with at least one output lane takes at least one of the three store branches below), so the exact number of younger
    // operations is AHEAD * (NL + 3).  Waiting for AHEAD * NL there would ask for the rows r+1, r+2 .. as well -- half the
    // requests in flight for nothing.  (gfx9 keeps loads and stores in ONE in-order counter; hipcc's own waits rely on that too.)
#ifndef GPF_K2_COUNT_STORES
#define GPF_K2_COUNT_STORES 1
#endif
    const bool stores_every_row = GPF_K2_COUNT_STORES && __builtin_amdgcn_ballot_w64(out0 || out1) != 0ull;
    auto arrive = [&](Raw& r, const int n) {
        constexpr int LOOSE = AHEAD_ROWS * NL, EXACT = AHEAD_ROWS * (NL + 3);
        static_assert(EXACT <= 63, "vmcnt is a 6-bit counter");
        // ONE asm statement names every destination of the row (two statements in an if / else make hipcc join their "+v" results
        // with register copies -- of registers whose loads are in flight: tools/audit_step_isa.py); the choice is a branch inside it
        // (readfirstlane: an "s" operand is handed over as it is, and hipcc keeps this wave-uniform value in a VGPR in some variants)
        const int exact = __builtin_amdgcn_readfirstlane((stores_every_row && n > n_first + AHEAD_ROWS) ? 1 : 0);
#define GPF_WAIT_ROW "s_cmp_lg_u32 %[ex], 0\n\ts_cbranch_scc1 .Lgpf_exact_%=\n\ts_waitcnt vmcnt(%[loose])\n.Lgpf_exact_%=:\n\ts_waitcnt vmcnt(%[exact])"
        if constexpr (TOPO == 0 && HAS_LS)
            asm volatile(GPF_WAIT_ROW : "+v"(r.q[0]), "+v"(r.q[1]), "+v"(r.q[2]), "+v"(r.t[0]), "+v"(r.t[1]), "+v"(r.t[2]), "+v"(r.ls[0])
                         : [ex] "s"(exact), [loose] "n"(LOOSE), [exact] "n"(EXACT) : "memory", "scc");
        else if constexpr (TOPO == 0)
            asm volatile(GPF_WAIT_ROW : "+v"(r.q[0]), "+v"(r.q[1]), "+v"(r.q[2]), "+v"(r.t[0]), "+v"(r.t[1]), "+v"(r.t[2])
                         : [ex] "s"(exact), [loose] "n"(LOOSE), [exact] "n"(EXACT) : "memory", "scc");
        else if constexpr (HAS_LS)
            asm volatile(GPF_WAIT_ROW : "+v"(r.q[0]), "+v"(r.q[1]), "+v"(r.q[2]), "+v"(r.ls[0])
                         : [ex] "s"(exact), [loose] "n"(LOOSE), [exact] "n"(EXACT) : "memory", "scc");
        else
            asm volatile(GPF_WAIT_ROW : "+v"(r.q[0]), "+v"(r.q[1]), "+v"(r.q[2])
                         : [ex] "s"(exact), [loose] "n"(LOOSE), [exact] "n"(EXACT) : "memory", "scc");
#undef GPF_WAIT_ROW
    };
    auto unpack = [&](const Raw& r, Row2& o) {
        o.rho[0] = D > 0 ? r.q[0].x : r.q[0].y; o.rho[1] = D > 0 ? r.q[0].y : r.q[0].x;
        o.jx[0] = D > 0 ? r.q[1].x : r.q[1].y; o.jx[1] = D > 0 ? r.q[1].y : r.q[1].x;
        o.jy[0] = D > 0 ? r.q[2].x : r.q[2].y; o.jy[1] = D > 0 ? r.q[2].y : r.q[2].x;
        if (TOPO == 0) {
            o.h[0] = D > 0 ? r.t[0].x : r.t[0].y; o.h[1] = D > 0 ? r.t[0].y : r.t[0].x;
            o.hx[0] = D > 0 ? r.t[1].x : r.t[1].y; o.hx[1] = D > 0 ? r.t[1].y : r.t[1].x;
            o.hy[0] = D > 0 ? r.t[2].x : r.t[2].y; o.hy[1] = D > 0 ? r.t[2].y : r.t[2].x;
        } else if (TOPO == 1 || TOPO == 3) {
            o.h[0] = o.h[1] = r.th; o.hx[0] = o.hx[1] = r.thx; o.hy[0] = o.hy[1] = r.thy;
        } else {
            o.h[0] = lh[0]; o.h[1] = lh[1]; o.hx[0] = lhx[0]; o.hx[1] = lhx[1]; o.hy[0] = lhy[0]; o.hy[1] = lhy[1];
        }
        if (HAS_LS) { o.Ls[0] = D > 0 ? r.ls[0].x : r.ls[0].y; o.Ls[1] = D > 0 ? r.ls[0].y : r.ls[0].x; }
        else o.Ls[0] = o.Ls[1] = 0.0;
    };
    // the closure of slot k of row r (its density and fluxes possibly replaced by a stage-1 state q)
    // Three forms of one closure: the x-only-gap form (TOPO = 3), the Ls = 0 / constant-viscosity form, the general one.
    constexpr bool LS0 = !HAS_LS && !PIEZO;
    auto closure = [&](const Row2& r, int k, const double* q, const TopoRcp& t, const GapCoef& gc, const RowCoef& rc, CellFlux& f) {
        if (TOPO == 3) {
            cell_closure_xonly<EOS>(q ? q[0] : r.rho[k], q ? q[1] : r.jx[k], q ? q[2] : r.jy[k], rc, P, f);
        } else if (LS0) {
            cell_closure_ls0<EOS>(q ? q[0] : r.rho[k], q ? q[1] : r.jx[k], q ? q[2] : r.jy[k], gc, P, f);
        } else {
            CellIn c;
            c.rho = q ? q[0] : r.rho[k]; c.jx = q ? q[1] : r.jx[k]; c.jy = q ? q[2] : r.jy[k];
            c.h = r.h[k]; c.hx = r.hx[k]; c.hy = r.hy[k]; c.Ls = r.Ls[k];
            cell_closure<EOS, true, HAS_LS, PIEZO>(c, t, P, f);
        }
    };
    auto cell_of = [&](const Row2& r, int k) {
        CellIn c;
        c.rho = r.rho[k]; c.jx = r.jx[k]; c.jy = r.jy[k]; c.h = r.h[k]; c.hx = r.hx[k]; c.hy = r.hy[k]; c.Ls = r.Ls[k];
        return c;
    };

    Red<EOS> red;
    red.init();
    const int seam_row_lo = a.E.halo[0] == 2 ? 1 : -1, seam_row_hi = a.E.halo[1] == 2 ? L.Nx : -1;

    // ---- fused: stage-1 field on a periodic downwind ghost row = the predictor's result on the partner row, logical
    //      row 1, from the stored rows 0 and 1 (problem.py:560, 682-695) ----
    if (fused && dw_row_is_ghost && x_periodic) {
        Row2 r0, r1;
        if (a.E.halo[e_dw_x] == 2) {
            // periodic seam between slabs: the partner row's state sits in this slab's outer row (logical row Nx+1), its
            // upwind neighbour is this slab's last row; their topography belongs to the far slab's rows 1 and 0
            load(L.Nx, r0);
            load(L.Nx + 1, r1);
            auto seam_topo = [&](const double* __restrict__ t, Row2& r) {
                if (TOPO == 1 || TOPO == 3) {
                    r.h[0] = r.h[1] = t[L.off + 1]; r.hx[0] = r.hx[1] = t[L.pitch + L.off + 1];
                    r.hy[0] = r.hy[1] = TOPO == 1 ? t[2 * L.pitch + L.off + 1] : 0.0;
                } else {
                    pair(t, r.h[0], r.h[1]); pair(t + L.pitch, r.hx[0], r.hx[1]); pair(t + 2 * L.pitch, r.hy[0], r.hy[1]);
                }
                if (HAS_LS) pair(t + 3 * L.pitch, r.Ls[0], r.Ls[1]);
            };
            seam_topo(a.seam[e_dw_x] + 4 * L.pitch, r0);
            seam_topo(a.seam[e_dw_x], r1);
        } else {
            load(0, r0);
            load(1, r1);
        }
        double fy[2][3], q1[2][3];
        CellFlux f0[2], f1[2];
        const RowCoef rc0 = TOPO == 3 ? row_coefficients(r0.h[0], r0.hx[0], P) : RowCoef();
        const RowCoef rc1 = TOPO == 3 ? row_coefficients(r1.h[0], r1.hx[0], P) : RowCoef();
        for (int k = 0; k < 2; ++k) {
            closure(r0, k, nullptr, topo_rcp<HAS_LS>(cell_of(r0, k)), gap_coefficients(r0.h[k], r0.hx[k], r0.hy[k]), rc0, f0[k]);
            closure(r1, k, nullptr, topo_rcp<HAS_LS>(cell_of(r1, k)), gap_coefficients(r1.h[k], r1.hx[k], r1.hy[k]), rc1, f1[k]);
            fy[k][0] = r1.jy[k]; fy[k][1] = f1[k].fx2; fy[k][2] = f1[k].fy2;
        }
        const double u0 = lane_from_below(fy[1][0]), u1 = lane_from_below(fy[1][1]), u2 = lane_from_below(fy[1][2]);
        const double up[2][3] = {{u0, u1, u2}, {fy[0][0], fy[0][1], fy[0][2]}};
        for (int k = 0; k < 2; ++k) {
            q1[k][0] = predictor_value(r1.rho[k], dt, cx, r1.jx[k] - r0.jx[k], cy, fy[k][0] - up[k][0], f1[k].s0);
            q1[k][1] = predictor_value(r1.jx[k], dt, cx, f1[k].fx1 - f0[k].fx1, cy, fy[k][1] - up[k][1], f1[k].s1);
            q1[k][2] = predictor_value(r1.jy[k], dt, cx, f1[k].fx2 - f0[k].fx2, cy, fy[k][2] - up[k][2], f1[k].s2);
            for (int c = 0; c < 3; ++c) stash[c][2 * lane + k] = q1[k][c];
        }
    }

    // slabs: the stage-1 field on the downwind ghost row was prepared by k_ghost_stage1 / k_begin_slab.  It is staged
    // here, outside the march: a compiler-visible vector load inside the loop would make hipcc drain the pipelined
    // row loads (its wait counts cannot see them).
    if (!fused && dw_row_is_ghost) {
        const int c0 = min(max(iy0, 0), L.Ny + 1), c1 = min(max(iy1, 0), L.Ny + 1);
        for (int c = 0; c < 3; ++c) {
            stash[c][2 * lane] = a.g1x[c * L.pitch + L.off + c0];
            stash[c][2 * lane + 1] = a.g1x[c * L.pitch + L.off + c1];
        }
    }

    // Rows n+1 .. n+AHEAD are in flight while row n is computed: at the top of row n the request for row n+AHEAD goes
    // out, then the wave waits for row n alone (requested AHEAD rows ago; a row of arithmetic lasts longer than an
    // HBM round trip).  AHEAD + 1 row buffers rotate by NAME: the loop body is a lambda instantiated once per
    // buffer, so no register copies are needed to advance the window.
    // (the piezo-viscosity closures -- exp / pow per cell -- leave no registers for the third buffer: one row ahead there)
    constexpr int AHEAD = AHEAD_ROWS;
    static_assert(AHEAD >= 1 && AHEAD <= 6, "1 to 6 rows ahead");
    constexpr int NB = AHEAD + 1;
    Raw rowbuf[NB];
    issue(n_first - 1, rowbuf[0]);
    if (AHEAD >= 2) issue(n_first, rowbuf[1 % NB]);
    if (AHEAD >= 3) issue(n_first + 1, rowbuf[2 % NB]);
    if (AHEAD >= 4) issue(n_first + 2, rowbuf[3 % NB]);
    if (AHEAD >= 5) issue(n_first + 3, rowbuf[4 % NB]);
    if (AHEAD >= 6) issue(n_first + 4, rowbuf[5 % NB]);

    // carried from the previous row, per slot
    double fx1p[2][3] = {{0, 0, 0}, {0, 0, 0}};     // stage-1 x-flux of row n-1
    double part[2][3] = {{0, 0, 0}, {0, 0, 0}};     // row n-1: q(t0) + q1 - dt*(-cx*Fx2 + cy*dFy2 - S2)

    // one row of the march: `raw` holds (or is about to hold) row n, `spare` is the buffer row n-1 has vacated
    auto march = [&](const int n, Raw& raw, Raw& spare) {
        // (beyond the chunk's last row `issue` sends a dummy request: the number of loads in flight behind `raw` is the
        // same in every iteration, and the wait needs no case distinction)
        issue(n + AHEAD, spare);
        arrive(raw, n);
        Row2 cur;
        unpack(raw, cur);
        const bool first = (n == n_first - 1);
        const bool last = (n == n_last + 1);
        const int ix = D > 0 ? n : L.Nx + 1 - n;

        TopoRcp tr[2] = {TopoRcp(), TopoRcp()};
        GapCoef gc[2] = {GapCoef(), GapCoef()};
        RowCoef rc = RowCoef();
        if (TOPO == 3) {
            rc = row_coefficients(cur.h[0], cur.hx[0], P);      // one set per row: the gap is the same in every column
        } else if (LS0) {
            gc[0] = gap_coefficients(cur.h[0], cur.hx[0], cur.hy[0]);   // once per cell and step: both stages use them
            gc[1] = gap_coefficients(cur.h[1], cur.hx[1], cur.hy[1]);
        } else {
            tr[0] = topo_rcp<HAS_LS>(cell_of(cur, 0));
            tr[1] = topo_rcp<HAS_LS>(cell_of(cur, 1));
        }

        // ---- stage 1 at (n, m) ----
        double q1[2][3];
        if (last && dw_row_is_ghost) {
            for (int k = 0; k < 2; ++k)
                for (int c = 0; c < 3; ++c) q1[k][c] = stash[c][2 * lane + k];
        } else {
            CellFlux f[2];
            double fy[2][3];
            for (int k = 0; k < 2; ++k) {
                closure(cur, k, nullptr, tr[k], gc[k], rc, f[k]);
                fy[k][0] = cur.jy[k]; fy[k][1] = f[k].fx2; fy[k][2] = f[k].fy2;
            }
            const double u0 = lane_from_below(fy[1][0]), u1 = lane_from_below(fy[1][1]), u2 = lane_from_below(fy[1][2]);
            const double up[2][3] = {{u0, u1, u2}, {fy[0][0], fy[0][1], fy[0][2]}};
            for (int k = 0; k < 2; ++k) {
                q1[k][0] = predictor_value(cur.rho[k], dt, cx, cur.jx[k] - fx1p[k][0], cy, fy[k][0] - up[k][0], f[k].s0);
                q1[k][1] = predictor_value(cur.jx[k], dt, cx, f[k].fx1 - fx1p[k][1], cy, fy[k][1] - up[k][1], f[k].s1);
                q1[k][2] = predictor_value(cur.jy[k], dt, cx, f[k].fx2 - fx1p[k][2], cy, fy[k][2] - up[k][2], f[k].s2);
                fx1p[k][0] = cur.jx[k]; fx1p[k][1] = f[k].fx1; fx1p[k][2] = f[k].fx2;
            }
            if (ghost_strip) {
                // stage-1 field on the downwind ghost COLUMN (logical column Ny+1): the ghost rule applied to the
                // predictor's result on its source column (problem.py:560) -- logical column 1 (periodic: computed by
                // the wrap lanes of this wave) or logical column Ny (the slot just upwind of the ghost slot)
                double gv[3];
                if (!fused) {
                    for (int c = 0; c < 3; ++c) gv[c] = uniform_load(a.g1y + c * (L.Nx + 2) + ix);
                } else {
                    for (int c = 0; c < 3; ++c)
                        gv[c] = ghost_rule(a.E, e_dw_y, c, __shfl(gsrc_slot ? q1[1][c] : q1[0][c], gsrc_lane));
                }
                if (is_ghost_lane) {
                    for (int c = 0; c < 3; ++c) {
                        if (G.slot_ghost) q1[1][c] = gv[c];
                        else q1[0][c] = gv[c];
                    }
                }
            }
            // fused, Dirichlet / Neumann in x: the ghost rule applied to the last interior row's predictor result
            if (fused && n == n_last && dw_row_is_ghost && !x_periodic) {
                for (int k = 0; k < 2; ++k)
                    for (int c = 0; c < 3; ++c) stash[c][2 * lane + k] = ghost_rule(a.E, e_dw_x, c, q1[k][c]);
            }
        }

        if (!first) {
            // ---- stage 2 closure at (n, m) on the stage-1 field ----
            CellFlux g[2];
            double gy[2][3];
            for (int k = 0; k < 2; ++k) {
                closure(cur, k, q1[k], tr[k], gc[k], rc, g[k]);
                gy[k][0] = q1[k][2]; gy[k][1] = g[k].fx2; gy[k][2] = g[k].fy2;
            }
            const double d0 = lane_from_above(gy[0][0]), d1 = lane_from_above(gy[0][1]), d2 = lane_from_above(gy[0][2]);
            const double dn[2][3] = {{gy[1][0], gy[1][1], gy[1][2]}, {d0, d1, d2}};

            // ---- finish row n-1: corrector + time average (problem.py:558, 563) ----
            if (n > n_first) {
                double o[2][3];
                for (int k = 0; k < 2; ++k) {
                    o[k][0] = 0.5 * (part[k][0] - dt * cx * q1[k][1]);
                    o[k][1] = 0.5 * (part[k][1] - dt * cx * g[k].fx1);
                    o[k][2] = 0.5 * (part[k][2] - dt * cx * g[k].fx2);
                }
                const int ixo = D > 0 ? n - 1 : L.Nx + 2 - n;
                const long long rbo = (long long)ixo * L.pitch;     // wave-uniform
                auto st16 = [&](double* __restrict__ rowp, double va, double vb) {
                    dpair v;
                    v.x = D > 0 ? va : vb; v.y = D > 0 ? vb : va;
                    dpair* p = reinterpret_cast<dpair*>(reinterpret_cast<char*>(rowp) + lane_bytes);
#if GPF_K2_NT & 2
                    // (only where the step's working set exceeds the 256-MiB Infinity Cache: a small grid -- one rank's slab of the
                    // 8-GPU run, 2 x 50 MB -- finds its previous output still on the die.)  The hinted store is written in asm:
                    // given `if (nt) __builtin_nontemporal_store(v, p); else *p = v;` hipcc sinks the two stores into one and
                    // drops the hint -- the library of the first half of round 3 held no `nt` store at all (llvm-objdump).
                    // s_nop: a store of more than 8 bytes must not be followed at once by a write of its data registers.
                    if (policy & 2) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 0" :: "v"(p), "v"(v) : "memory");
                    else *p = v;
#else
                    *p = v;
#endif
                };
                auto st8 = [&](double* __restrict__ rowp, unsigned int bytes, double v) {
                    *reinterpret_cast<double*>(reinterpret_cast<char*>(rowp) + bytes) = v;
                };
                if (out0 && out1) {
                    st16(qo0 + rbo, o[0][0], o[1][0]); st16(qo1 + rbo, o[0][1], o[1][1]); st16(qo2 + rbo, o[0][2], o[1][2]);
                } else if (out0) {
                    const unsigned int b = (unsigned int)(L.off + iy0) * 8u;
                    st8(qo0 + rbo, b, o[0][0]); st8(qo1 + rbo, b, o[0][1]); st8(qo2 + rbo, b, o[0][2]);
                } else if (out1) {
                    const unsigned int b = (unsigned int)(L.off + iy1) * 8u;
                    st8(qo0 + rbo, b, o[1][0]); st8(qo1 + rbo, b, o[1][1]); st8(qo2 + rbo, b, o[1][2]);
                }
                // a row next to a periodic slab seam also stands in for the far slab's ghost row
                const double w = 1.0 + (ixo == seam_row_lo ? 1.0 : 0.0) + (ixo == seam_row_hi ? 1.0 : 0.0);
                if (out0) red.cell(o[0][0], o[0][1], o[0][2], w, P);
                if (out1) red.cell(o[1][0], o[1][1], o[1][2], w, P);
            }
            // ---- open row n (an output row unless this is the downwind extra row) ----
            for (int k = 0; k < 2; ++k) {
                part[k][0] = (cur.rho[k] + q1[k][0]) - dt * (-cx * q1[k][1] + cy * (dn[k][0] - gy[k][0]) - g[k].s0);
                part[k][1] = (cur.jx[k] + q1[k][1]) - dt * (-cx * g[k].fx1 + cy * (dn[k][1] - gy[k][1]) - g[k].s1);
                part[k][2] = (cur.jy[k] + q1[k][2]) - dt * (-cx * g[k].fx2 + cy * (dn[k][2] - gy[k][2]) - g[k].s2);
            }
        }
    };
    for (int n = n_first - 1;;) {
        march(n, rowbuf[0], rowbuf[AHEAD]); if (++n > n_last + 1) break;
        march(n, rowbuf[1], rowbuf[0]); if (++n > n_last + 1) break;
        if (AHEAD >= 2) { march(n, rowbuf[2 % NB], rowbuf[1]); if (++n > n_last + 1) break; }
        if (AHEAD >= 3) { march(n, rowbuf[3 % NB], rowbuf[2 % NB]); if (++n > n_last + 1) break; }
        if (AHEAD >= 4) { march(n, rowbuf[4 % NB], rowbuf[3 % NB]); if (++n > n_last + 1) break; }
        if (AHEAD >= 5) { march(n, rowbuf[5 % NB], rowbuf[4 % NB]); if (++n > n_last + 1) break; }
        if (AHEAD >= 6) { march(n, rowbuf[6 % NB], rowbuf[5 % NB]); if (++n > n_last + 1) break; }
    }
    // The last requests (repeats of the final row) are still in flight: drain them while the buffers are still
    // live, or a late return would land in registers the code below has reused.
    for (int b = 0; b <= AHEAD; ++b) {
        asm_wait3<0>(rowbuf[b].q[0], rowbuf[b].q[1], rowbuf[b].q[2]);
        if (TOPO == 0) asm_wait3<0>(rowbuf[b].t[0], rowbuf[b].t[1], rowbuf[b].t[2]);
        if (HAS_LS) asm_wait1<0>(rowbuf[b].ls[0]);
    }

    // ---- fused: ghost cells of the field this step has produced (problem.py:576 -> 676-768) ----
    // Outside the march: the wave re-reads its OWN finished cells next to an edge (its stores are drained first; a CU's
    // L1 is coherent with its own stores), applies the ghost rule and writes the ghost cells; they enter the
    // reductions like every cell (problem.py:342-347).  x edge e: ghost row ix = e ? Nx+1 : 0; its source row is the
    // partner row (periodic) or the adjacent one; y edges alike; corners are rule_y(rule_x(.)) as the reference's
    // x-then-y order produces them.
#ifndef GPF_K2_NO_POSTPASS      // timing experiments only: results are wrong without it
    if (fused) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int src_row[2] = {x_periodic ? L.Nx : 1, x_periodic ? 1 : L.Nx};
        const int src_col[2] = {y_periodic ? L.Ny : 1, y_periodic ? 1 : L.Ny};      // y edge 2 (iy = 0), 3 (iy = Ny+1)
        const int ix_lo = D > 0 ? n_first : L.Nx + 1 - n_last, ix_hi = D > 0 ? n_last : L.Nx + 1 - n_first;
        auto get = [&](int ix, int iy, double v[3]) {
            const long long o = (long long)ix * L.pitch + L.off + iy;
            v[0] = qo0[o]; v[1] = qo1[o]; v[2] = qo2[o];
        };
        auto put = [&](int ix, int iy, const double v[3], double w) {
            const long long o = (long long)ix * L.pitch + L.off + iy;
            qo0[o] = v[0]; qo1[o] = v[1]; qo2[o] = v[2];
            red.cell(v[0], v[1], v[2], w, P);
        };
        // slabs: where this slab's first (e = 0) / last (e = 1) row travels -- the local all-gather message, or the
        // neighbour's mailbox (its slot of message seq + 1: row `1 - e` there, i.e. its outer row on my side)
        double* send_to[2] = {nullptr, nullptr};
        if (a.p2p.on) {
            const int slot = (int)((*a.p2p.seq + 1) & 1);
            if (a.E.halo[0] && a.p2p.rank_lo >= 0) send_to[0] = p2p_rows(a.p2p.box[a.p2p.rank_lo], slot, 1, L.pitch);
            if (a.E.halo[1] && a.p2p.rank_hi >= 0) send_to[1] = p2p_rows(a.p2p.box[a.p2p.rank_hi], slot, 0, L.pitch);
        } else if (a.msg) {
            if (a.E.halo[0]) send_to[0] = a.msg;
            if (a.E.halo[1]) send_to[1] = a.msg + 3 * L.pitch;
        }
        auto send = [&](int e, int iy, const double v[3]) {
            double* d = send_to[e] + L.off + iy;
            for (int c = 0; c < 3; ++c) {
                if (a.p2p.on)       // a peer's memory: write-through at system scope, drained before this block arrives
                    __hip_atomic_store(reinterpret_cast<unsigned long long*>(d + c * L.pitch), (unsigned long long)__double_as_longlong(v[c]),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                else d[c * L.pitch] = v[c];
            }
        };
        // ghost columns: one lane per row of the chunk
        const int m_out_lo = max(G.w0 + strip * STRIP2 + 1, 1), m_out_hi = min(G.w0 + strip * STRIP2 + STRIP2, L.Ny);
        for (int ey = 0; ey < 2; ++ey) {
            const int msrc = D > 0 ? src_col[ey] : L.Ny + 1 - src_col[ey];
            if (msrc < m_out_lo || msrc > m_out_hi) continue;                       // wave-uniform: not this strip's column
            for (int ix = ix_lo + lane; ix <= ix_hi; ix += 64) {
                double v[3], g[3];
                get(ix, src_col[ey], v);
                for (int c = 0; c < 3; ++c) g[c] = ghost_rule(a.E, 2 + ey, c, v[c]);
                // a row next to a periodic slab seam also stands in for the far slab's ghost row
                put(ix, ey ? L.Ny + 1 : 0, g, 1.0 + (ix == seam_row_lo ? 1.0 : 0.0) + (ix == seam_row_hi ? 1.0 : 0.0));
                if (ix == 1 && send_to[0]) send(0, ey ? L.Ny + 1 : 0, g);
                if (ix == L.Nx && send_to[1]) send(1, ey ? L.Ny + 1 : 0, g);
            }
        }
        // ghost rows (and corners): the lanes re-read the source row at their own output columns
        for (int e = 0; e < 2; ++e) {
            if (a.E.halo[e]) continue;                                              // a neighbour's row: the exchange fills it
            if (src_row[e] < ix_lo || src_row[e] > ix_hi) continue;                 // wave-uniform: not this chunk's row
            const int ixg = e ? L.Nx + 1 : 0;
            for (int k = 0; k < 2; ++k) {
                if (!(k ? out1 : out0)) continue;
                const int iy = k ? iy1 : iy0;
                double v[3], g[3], gc[3];
                get(src_row[e], iy, v);
                for (int c = 0; c < 3; ++c) g[c] = ghost_rule(a.E, e, c, v[c]);
                put(ixg, iy, g, 1.0);
                for (int ey = 0; ey < 2; ++ey) {
                    if (iy != src_col[ey]) continue;
                    for (int c = 0; c < 3; ++c) gc[c] = ghost_rule(a.E, 2 + ey, c, g[c]);
                    put(ixg, ey ? L.Ny + 1 : 0, gc, 1.0);
                }
            }
        }
        // slabs: the first / last row, at the lanes' own output columns
        for (int e = 0; e < 2; ++e) {
            const int ixb = e ? L.Nx : 1;
            if (!send_to[e] || ixb < ix_lo || ixb > ix_hi) continue;                // wave-uniform
            for (int k = 0; k < 2; ++k) {
                if (!(k ? out1 : out0)) continue;
                double v[3];
                get(ixb, k ? iy1 : iy0, v);
                send(e, k ? iy1 : iy0, v);
            }
        }
        if (a.p2p.on) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#endif

    // ---- wave reduction: the record is valid in lane 0 ----
    for (int s = 32; s >= 1; s >>= 1) {
        red.ekin += __shfl_down(red.ekin, s);
        red.v2 = fmax(red.v2, __shfl_down(red.v2, s));
        const double oc = __shfl_down(red.c2, s);
        red.c2 = (EOS == EOS_DH) ? fmin(red.c2, oc) : fmax(red.c2, oc);
        red.flags |= __shfl_down(red.flags, s);
    }
    if (EOS == EOS_DH) {                    // same expression as eos_c2<EOS_DH> at the cell nearest the pole
        const double it = rcp(red.c2);
        red.c2 = (red.c2 == __builtin_inf()) ? 0.0 : P.e[6] * (it * it);
    }
#ifdef GPF_K2_NO_POSTPASS
    red.flags = 0;      // (timing experiment: keep stepping whatever the stale ghost cells do to the field)
#endif
    result.ekin = red.ekin; result.v2 = red.v2; result.c2 = red.c2; result.mass = 0.0; result.flags = red.flags;
}

// D = direction of the predictor, chosen by the host from the step index (problem.py:521-522) and checked against the
// device-side step counter.  Grid: gridDim.x = a multiple of 8 blocks of 4 waves; wave w of the XCD-ordered numbering
// works on strip w % nstrips of chunk w / nstrips.
template <int EOS, bool HAS_LS, bool PIEZO, int D, int TOPO>
__global__ __launch_bounds__(256, (Step2Weight<EOS, HAS_LS, PIEZO>::heavy ? 1 : ((TOPO == 1 || TOPO == 3) ? GPF_K2_MINWAVES_LINE : GPF_K2_MINWAVES)))
void k_step2(const Step2Args a, const Phys P) {
    __shared__ double stash[4][3][128];         // per wave: stage-1 field on the downwind ghost row (fused)
    __shared__ Acc red_sm[4];
    __shared__ int s_last;
    if (halted(a.st, a.honor_stop)) return;
    if (predictor_direction(a.st) != D) __builtin_trap();
    const int par = a.st->parity;
    const double* qin = par ? a.qb : a.qa;
    double* qout = const_cast<double*>(par ? a.qa : a.qb);

    // blocks b and b + 8 share an XCD (round-robin dispatch): give each XCD a contiguous range of waves, so that
    // neighbouring strips -- which share their halo columns' cache lines -- meet in the same L2
    const int nb = gridDim.x;
    const int lb = (int)(blockIdx.x & 7) * (nb >> 3) + (int)(blockIdx.x >> 3);
    // the wave index is wave-uniform, but only readfirstlane tells the compiler: strip, chunk, the row loop, its
    // branches and the row base addresses then live in scalar registers
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int w = lb * 4 + wv;
    const int strip = w % a.G.nstrips, chunk = w / a.G.nstrips;
    const bool active = chunk < a.nchunks;      // wave-uniform

    Acc own;
    own.zero();
    if (active) step_strip2<EOS, HAS_LS, PIEZO, D, TOPO>(a, P, qin, qout, strip, chunk, lane, stash[wv], own);

    if (!(a.fused & 1)) {
        if (active && lane == 0) {
            Partial p;
            p.ekin = own.ekin; p.vmax2 = own.v2; p.c2max = own.c2; p.flags = (double)own.flags;
            a.partials[w] = p;
        }
        return;
    }
#ifdef GPF_K2_NO_FINISH         // timing experiments only: block 0 commits its own record, nobody waits for anybody
    if (blockIdx.x == 0 && threadIdx.x == 0) commit_step(a.st, own.ekin + 1.0, own.v2, own.c2, own.flags, a.log, a.log_base, a.log_cap);
    return;
#endif
    // ---- fused: one record per block; the last block to arrive folds them and commits the step ----
    // (the barrier below also puts every wave's drained stores -- message rows included -- before the arrival)
    if (lane == 0) red_sm[wv] = own;
    __syncthreads();
    if (threadIdx.x == 0) {
        Acc tot = red_sm[0];
        for (int i = 1; i < 4; ++i) { tot.ekin += red_sm[i].ekin; tot.v2 = fmax(tot.v2, red_sm[i].v2); tot.c2 = fmax(tot.c2, red_sm[i].c2); tot.flags |= red_sm[i].flags; }
        publish_partial(a.block_partials + blockIdx.x, tot);       // write-through + drained: see aux_kernels.hip
        const unsigned int t = __hip_atomic_fetch_add(a.arrive, 1u, GPF_ORDER_ARRIVE, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == (unsigned int)nb - 1) ? 1 : 0;
        if (s_last) __hip_atomic_store(a.arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!s_last) return;
    // (a slab leaves its record in the message / the peers' mailboxes instead: the commit follows the exchange)
    finish_tail(a.st, a.log, a.log_base, a.log_cap, a.out, a.block_partials, a.p2p, red_sm);
}

}  // namespace gpf
