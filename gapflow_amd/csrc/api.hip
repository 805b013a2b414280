// C ABI of libgapflow_hip.so (see include/gapflow_hip.h for the contract of every entry point).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/gapflow_hip.h"
#include "phys_setup.hpp"
#include "step_kernel.hip"
#include "aux_kernels.hip"
#include "step2_kernel.hip"
#include "gp_kernels.hip"
#include "elastic_kernels.hip"
#include "small_kernel.hip"

using namespace gpf;

static thread_local std::string g_err;
static const bool g_debug = std::getenv("GPF_DEBUG") != nullptr;
#define DBG(...) do { if (g_debug) { std::fprintf(stderr, "[gpf] " __VA_ARGS__); std::fprintf(stderr, "\n"); std::fflush(stderr); } } while (0)

static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

// Common prologue of the C-ABI calls on a handle: select its device; unless the call only reads the state, forget what is
// cached about it (gpf_handle::gp_state_mean).
struct gpf_handle;
static int enter(gpf_handle* h, bool reads_only = false);

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(GPF_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));               \
    } while (0)

#define GPF_TRY(expr)                                                                                   \
    do {                                                                                                \
        int r_ = (expr);                                                                                \
        if (r_ != GPF_OK) return r_;                                                                    \
    } while (0)

static_assert(sizeof(LogEntry) == sizeof(gpf_scalars_t), "LogEntry must mirror gpf_scalars_t");
// k_step2's 16-byte pair loads may reach one pair past the last column of the last row of a buffer
static constexpr size_t PLANE_PAD_BYTES = 256;

struct gpf_handle {
    gpf_config cfg;
    Layout L;
    Edges E;
    Phys P;
    hipStream_t stream = nullptr;
    // device state
    size_t field_bytes = 0;                 // size of each of the three field allocations q[0], q[1], topo (plan_placement may re-home them)
    double* q[2] = {nullptr, nullptr};      // ping-pong, 3 planes each
    double* topo = nullptr;                 // 3 planes
    double* topo_line = nullptr;            // [3][max(Nx,Ny)+2]: the profile when the topography varies along one axis only
    int topo_mode = 0;                      // 0: 2-D planes, 1: function of ix only, 2: function of iy only
    bool topo_hy0 = false;                  // topo_mode 1 and dh/dy == 0 in every row
    double* Ls = nullptr;                   // 1 plane (allocated on first non-zero upload)
    double* g1 = nullptr;                   // g1x [3][pitch], g1y [3][Nx+2]: stage-1 ghost values of the step about to run
    double* seam = nullptr;                 // [2 edges][2 rows][4: h,hx,hy,Ls][pitch]
    bool has_seam[2] = {false, false};
    // peer-to-peer slab transport (gpf_p2p_*): my mailbox, every rank's mailbox as mapped here, message counter
    struct { bool on = false; int nranks = 0, rank = 0, rank_lo = -1, rank_hi = -1; char* mine = nullptr;
             char* box[P2P_MAX_RANKS] = {}; unsigned long long* seq = nullptr; } p2p;
    double* halo = nullptr;                 // this slab's all-gather message: first row, last row (3 x pitch each), [thinning: density of
                                            // the second and second-to-last row,] 8-double record
    double* beyond = nullptr;               // thinning slabs: density one row beyond each outer row, [state | working field][2][pitch]
    StepState* st = nullptr;
    Partial* partials = nullptr;
    unsigned int* arrive = nullptr;         // [0] blocks done: k_step2 (fused) / k_ghost_fill (finish_step); [1], [2]: k_begin_slab's arrivals and time-outs
    long long p2p_timeout_ticks = P2P_TIMEOUT_TICKS;
    bool split_edges = false;               // GPF_STEP_UNFUSED_EDGES at gpf_create: edge work in separate launches
    bool g1_ready = false;                  // g1 already holds the next step's stage-1 ghost values (k_begin_slab wrote them)
    Partial* block_partials = nullptr;      // one record per edge-kernel block
    int nghost_blocks = 0;
    ScalarPartial* spart = nullptr;         // k_scalars block records (+ 4 totals at the end)
    int nspart = 0;
    LogEntry* log = nullptr;
    long long log_cap = 0;
    double* stage = nullptr;                // contiguous staging for upload/download
    size_t stage_doubles = 0;
    // unfused pipeline scratch (lazy)
    double* fields = nullptr;               // p(1) tau(3) lower(6) upper(6)
    double* work = nullptr;                 // fx(3) fy(3) src(3)
    // GP surrogate models: 0 pressure, 1 wall shear xz, 2 wall shear yz
    struct GpHost {
        bool set = false;
        GpModelDev dev;
        double *Z = nullptr, *alpha = nullptr, *L = nullptr, *Linv = nullptr, *W = nullptr;      // W: work matrix (L^-T)
        int cap = 0;                            // training points the buffers can hold (grown in steps of 128)
        bool has_linv = false;
        double xscale0 = 1.0;
        double inv_scale[GP_MAX_D] = {};        // kernel length scales of the fit (fscale = inv_scale / x_scale)
        double yscale_fit = 1.0;                // output scale at the fit (dev.yscale may be updated: gpf_gp_set_scales)
    } gp[3];
    // The sound-speed pass that closes a stage-wise step (d mean / d rho of the pressure surrogate on the new state) also
    // leaves the posterior MEAN of that state here; the first closure evaluation of the next step -- same state, same model --
    // copies it instead of running the pass again (one of seven posterior-mean passes per step).  Every C-ABI call except the
    // ones that only read the state drops it (enter()).
    double* gp_state_mean = nullptr;        // one plane
    bool gp_state_mean_valid = false;
    long long gp_state_mean_step = -1;      // the state's step count it belongs to
    bool gp_state_mean_fresh = false;       // the pass has just written it (gpf_close_step decides whether it stands)
    bool open_first_closures = false;       // gpf_stage_closures has not run yet in the open step
    bool gp_reuse_state_mean = true;        // GPF_GP_NO_STATE_MEAN at gpf_create turns it off
    long long gp_passes[3] = {0, 0, 0}, gp_reused = 0;
    double* gpvar = nullptr;                // 3 variance planes
    double* gpscratch = nullptr;            // block maxima + 4 result slots
    int gpscratch_n = 0;
    // elastic half-space (gpf_elastic_*): transform grid, Green's function in Fourier space, work buffers, state planes
    struct { bool on = false; int px = 0, py = 0, relative = 0; double alpha = 1.0, scale = 1.0; void* plan_f = nullptr; void* plan_b = nullptr;
             double2* greens = nullptr; double2* spec = nullptr; double* dense = nullptr;
             double* u_prev = nullptr; double* h0 = nullptr; double* deformation = nullptr; } el;
    double* gptile = nullptr;               // Ks tile for the variance solve
    size_t gptile_doubles = 0;
    void* blas = nullptr;
    bool step_open = false;                 // an unfused step is in progress on buffer parity^1
    int open_parity = 0;
    bool has_q = false, has_topo = false, pre_run_done = false;
    bool prev_state_valid = false;          // the OTHER q buffer holds the state before the last committed fused step (gpf_update_closures)
    long long host_step = 0;                // step count at the last sync
    long long next_step = 0;                // index of the next step to be enqueued (== device step unless halted)
    // two-columns-per-lane step kernel (step2_kernel.hip): window geometry per predictor direction [0: D=+1, 1: D=-1],
    // row chunks, grid; `fused`: ghost cells, stage-1 ghost data and the commit happen inside the kernel (no slab halo)
    Strip2Geom geom2[2];
    int nchunks2 = 0, nblocks2 = 0, npartials2_cap = 0, nblock_partials_cap = 0;
    bool plan2_valid = false;
    int nt_policy2 = 0;                     // k_step2's traffic with the non-temporal hint: 0 none, 1 the stores, 2 stores and loads (plan_step2)
    StepState* st_trial = nullptr;          // plan_step2's timing launches commit into this copy of the run state
    double* plan_master = nullptr;          // copy of the current state while plan_step2 runs its trials
    char plan2_note[400] = "";              // how the plan was arrived at (gpf_plan_note)
};

static int enter(gpf_handle* h, bool reads_only) {
    HIP_TRY(hipSetDevice(h->cfg.device));
    if (!reads_only) h->gp_state_mean_valid = false;
    return GPF_OK;
}

// ---------------------------------------------------------------------------------------------
static void make_phys(const gpf_config& c, Phys& P) {
    setup_phys(P, c.U, c.V, c.eta, c.zeta, c.dx, c.dy, c.eos, c.eos_par, c.piezo, c.piezo_par, c.thinning, c.thinning_par);
}

static int blocks_for(long long n, int bs = 256, int cap = 4096) {
    long long b = (n + bs - 1) / bs;
    return (int)std::max<long long>(1, std::min<long long>(b, cap));
}

#ifdef GPF_ONLY_EOS_DH   // experiment builds (tools/ab_step.py variants): one equation of state, a seventh of the compile time
#define EOS_DISPATCH(eos, ...)                                                                          \
    do {                                                                                                \
        if ((eos) == GPF_EOS_DH) { constexpr int EOS_ = EOS_DH; __VA_ARGS__; }      /* gpf_create refuses the others */ \
    } while (0)
#else
#define EOS_DISPATCH(eos, ...)                                                                          \
    switch (eos) {                                                                                      \
    case GPF_EOS_DH: { constexpr int EOS_ = EOS_DH; __VA_ARGS__; } break;                              \
    case GPF_EOS_PL: { constexpr int EOS_ = EOS_PL; __VA_ARGS__; } break;                              \
    case GPF_EOS_VDW: { constexpr int EOS_ = EOS_VDW; __VA_ARGS__; } break;                            \
    case GPF_EOS_MT: { constexpr int EOS_ = EOS_MT; __VA_ARGS__; } break;                              \
    case GPF_EOS_CUBIC: { constexpr int EOS_ = EOS_CUBIC; __VA_ARGS__; } break;                        \
    case GPF_EOS_BWR: { constexpr int EOS_ = EOS_BWR; __VA_ARGS__; } break;                            \
    default: { constexpr int EOS_ = EOS_BAYADA; __VA_ARGS__; } break;                                  \
    }
#endif

// slip-length field x piezo-viscosity: the edge kernels are specialised like the step kernel they serve
#define LS_PIEZO_DISPATCH(ls, pz, ...)                                                                  \
    do {                                                                                                \
        if (ls) {                                                                                       \
            constexpr bool LS_ = true;                                                                  \
            if (pz) { constexpr bool PZ_ = true; __VA_ARGS__; } else { constexpr bool PZ_ = false; __VA_ARGS__; }     \
        } else {                                                                                        \
            constexpr bool LS_ = false;                                                                 \
            if (pz) { constexpr bool PZ_ = true; __VA_ARGS__; } else { constexpr bool PZ_ = false; __VA_ARGS__; }     \
        }                                                                                               \
    } while (0)

// ---------------------------------------------------------------------------------------------
extern "C" const char* gpf_last_error(void) { return g_err.c_str(); }

extern "C" int gpf_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static int ensure_stage(gpf_handle* h, size_t doubles) {
    if (h->stage_doubles >= doubles) return GPF_OK;
    if (h->stage) HIP_TRY(hipFree(h->stage));
    h->stage = nullptr; h->stage_doubles = 0;
    HIP_TRY(hipMalloc(&h->stage, doubles * sizeof(double)));
    h->stage_doubles = doubles;
    return GPF_OK;
}

#include "api_fields.inc"

extern "C" int gpf_create(const gpf_config* cfg, gpf_handle** out) {
    if (!cfg || !out) return fail(GPF_ERR_INVALID, "gpf_create: null argument");
    if (cfg->Nx < 1 || cfg->Ny < 1) return fail(GPF_ERR_INVALID, "gpf_create: Nx, Ny must be >= 1");
    if (!(cfg->dx > 0) || !(cfg->dy > 0)) return fail(GPF_ERR_INVALID, "gpf_create: dx, dy must be > 0");
    // the kernels index cells of one plane with 32-bit offsets (a 46000^2 grid would still fit the card's memory)
    if (((long long)cfg->Nx + 2) * ((long long)cfg->Ny + 2 + 64) > 2000000000ll)
        return fail(GPF_ERR_INVALID, "gpf_create: more than 2e9 cells per plane; cut the domain into x-slabs (gapflow_amd/slab.py)");
    if (cfg->eos < GPF_EOS_DH || cfg->eos > GPF_EOS_BAYADA) return fail(GPF_ERR_INVALID, "gpf_create: unknown EOS id");
    if (cfg->thinning < 0 || cfg->thinning > GPF_THINNING_CARREAU) return fail(GPF_ERR_INVALID, "gpf_create: unknown shear-thinning id");
    for (int e = 0; e < 4; ++e) {
        int np = 0;
        for (int c = 0; c < 3; ++c) {
            int r = cfg->bc_rule[e][c];
            if (r < 0 || r > 2) return fail(GPF_ERR_INVALID, "gpf_create: bad ghost rule");
            np += (r == GPF_BC_PERIODIC);
        }
        if (np != 0 && np != 3)
            return fail(GPF_ERR_INVALID, "gpf_create: an edge must be periodic for all components or for none "
                                         "(problem.py:682-707 applies the periodic copy only when all three are 'P')");
    }
#ifdef GPF_ONLY_EOS_DH
    if (cfg->eos != GPF_EOS_DH) return fail(GPF_ERR_INVALID, "gpf_create: this build (-DGPF_ONLY_EOS_DH) holds the Dowson-Higginson kernels only");
#endif
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(GPF_ERR_NO_DEVICE, "gpf_create: no HIP device visible (this library has no CPU path)");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(GPF_ERR_INVALID, "gpf_create: device ordinal out of range");
    HIP_TRY(hipSetDevice(cfg->device));

    gpf_handle* h = new gpf_handle();
    h->cfg = *cfg;
    Layout& L = h->L;
    L.Nx = cfg->Nx; L.Ny = cfg->Ny; L.off = 15;
    L.pitch = ((cfg->Ny + 2 + L.off + 15) / 16) * 16;
    if (const char* s = std::getenv("GPF_PITCH_PAD")) L.pitch += (std::atoi(s) / 16) * 16;     // experiments: doubles appended to every row
    L.plane = (long long)(cfg->Nx + 2) * L.pitch;
    if (const char* s = std::getenv("GPF_PLANE_PAD")) L.plane += (std::atoll(s) & ~1ll);     // experiments: doubles between planes
    for (int e = 0; e < 4; ++e) {
        for (int c = 0; c < 3; ++c) h->E.rule[e][c] = cfg->bc_rule[e][c];
        h->E.value[e] = cfg->bc_value[e];
    }
    h->E.halo[0] = cfg->halo_lo; h->E.halo[1] = cfg->halo_hi;
    h->split_edges = std::getenv("GPF_STEP_UNFUSED_EDGES") != nullptr;
    h->gp_reuse_state_mean = std::getenv("GPF_GP_NO_STATE_MEAN") == nullptr;
    make_phys(*cfg, h->P);

    const size_t plane_b = (size_t)L.plane * sizeof(double);
    auto cleanup = [&](int code) { gpf_destroy(h); return code; };
#define HIP_TRY_C(expr)                                                                                 \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return cleanup(fail(GPF_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)));      \
    } while (0)
    h->field_bytes = (3 * plane_b + PLANE_PAD_BYTES + 255) & ~(size_t)255;
    for (int b = 0; b < 2; ++b) {
        HIP_TRY_C(field_malloc((void**)&h->q[b], h->field_bytes));
        if (field_zero(0, h->q[b], h->field_bytes) != GPF_OK) { gpf_destroy(h); return GPF_ERR_HIP; }
    }
    HIP_TRY_C(field_malloc((void**)&h->topo, h->field_bytes));
    if (field_zero(0, h->topo, h->field_bytes) != GPF_OK) { gpf_destroy(h); return GPF_ERR_HIP; }
    HIP_TRY_C(hipDeviceSynchronize());
    const size_t g1n = (size_t)3 * L.pitch + (size_t)3 * (L.Nx + 2);
    HIP_TRY_C(hipMalloc(&h->g1, g1n * sizeof(double)));
    HIP_TRY_C(hipMemset(h->g1, 0, g1n * sizeof(double)));
    HIP_TRY_C(hipMalloc(&h->st, sizeof(StepState)));
    HIP_TRY_C(hipMemset(h->st, 0, sizeof(StepState)));
    h->nghost_blocks = std::max(1, std::min(64, (2 * (L.Ny + 2) + 2 * L.Nx + 255) / 256));
    HIP_TRY_C(hipMalloc(&h->arrive, 4 * sizeof(unsigned int)));
    HIP_TRY_C(hipMemset(h->arrive, 0, 4 * sizeof(unsigned int)));
    HIP_TRY_C(hipMalloc(&h->block_partials, 1024 * sizeof(Partial)));
    h->nspart = 1024;
    HIP_TRY_C(hipMalloc(&h->spart, (size_t)(h->nspart + 8) * sizeof(ScalarPartial)));
    HIP_TRY_C(hipMemset(h->spart, 0, (size_t)(h->nspart + 8) * sizeof(ScalarPartial)));
    h->log_cap = 4096;
    HIP_TRY_C(hipMalloc(&h->log, (size_t)h->log_cap * sizeof(LogEntry)));
#undef HIP_TRY_C
    *out = h;
    return GPF_OK;
}

extern "C" int gpf_destroy(gpf_handle* h) {
    if (!h) return GPF_OK;
    hipSetDevice(h->cfg.device);
    for (void* p : {(void*)h->q[0], (void*)h->q[1], (void*)h->topo, (void*)h->plan_master}) field_free(p);
    void* ptrs[] = {h->topo_line, h->Ls, h->g1, h->seam, h->halo, h->beyond, h->st, h->partials, h->arrive, h->block_partials, h->spart,
                    h->log, h->stage, h->fields, h->work, h->st_trial, h->gpvar, h->gp_state_mean, h->gpscratch, h->gptile,
                    h->gp[0].Z, h->gp[0].alpha, h->gp[0].L, h->gp[1].Z, h->gp[1].alpha, h->gp[1].L,
                    h->gp[2].Z, h->gp[2].alpha, h->gp[2].L, h->gp[0].Linv, h->gp[1].Linv, h->gp[2].Linv, h->gp[0].W, h->gp[1].W, h->gp[2].W};
    if (h->blas && roclibs().ok) roclibs().destroy(h->blas);
    if (h->el.plan_f) fftlib().destroy(h->el.plan_f);
    if (h->el.plan_b) fftlib().destroy(h->el.plan_b);
    for (void* p : {(void*)h->el.greens, (void*)h->el.spec, (void*)h->el.dense, (void*)h->el.u_prev})     // h0, deformation live in u_prev's block
        if (p) hipFree(p);
    for (int r = 0; r < h->p2p.nranks; ++r)
        if (h->p2p.box[r] && r != h->p2p.rank) hipIpcCloseMemHandle(h->p2p.box[r]);
    if (h->p2p.mine) hipFree(h->p2p.mine);
    if (h->p2p.seq) hipFree(h->p2p.seq);
    for (void* p : ptrs)
        if (p) hipFree(p);
    delete h;
    return GPF_OK;
}

extern "C" int gpf_set_stream(gpf_handle* h, void* s) {
    if (!h) return fail(GPF_ERR_INVALID, "null handle");
    h->stream = (hipStream_t)s;
    return GPF_OK;
}

// ---------------------------------------------------------------------------------------------
static int field_ncomp(int field) {
    switch (field) {
    case GPF_FIELD_Q: return 3;
    case GPF_FIELD_TOPO: return 3;
    case GPF_FIELD_EXTRA: return 1;
    case GPF_FIELD_PRESSURE: return 1;
    case GPF_FIELD_TAU_AVG: return 3;
    case GPF_FIELD_WALL_LOWER: return 6;
    case GPF_FIELD_WALL_UPPER: return 6;
    case GPF_FIELD_PRESSURE_VAR: case GPF_FIELD_WALL_XZ_VAR: case GPF_FIELD_WALL_YZ_VAR: return 1;
    case GPF_FIELD_DEFORMATION: return 1;
    }
    return 0;
}

static int current_parity(gpf_handle* h, int* par) {
    StepState s;
    HIP_TRY(hipMemcpyAsync(&s, h->st, sizeof(s), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    *par = s.parity;
    return GPF_OK;
}

static int ensure_fields(gpf_handle* h) {
    if (h->fields) return GPF_OK;
    HIP_TRY(hipMalloc(&h->fields, (size_t)16 * h->L.plane * sizeof(double)));
    HIP_TRY(hipMemsetAsync(h->fields, 0, (size_t)16 * h->L.plane * sizeof(double), h->stream));
    return GPF_OK;
}

extern "C" int gpf_upload(gpf_handle* h, int field, const double* host, size_t count) {
    if (!h || !host) return fail(GPF_ERR_INVALID, "gpf_upload: null argument");
    GPF_TRY(enter(h));
    const Layout& L = h->L;
    const int nc = field_ncomp(field);
    if (field != GPF_FIELD_Q && field != GPF_FIELD_TOPO && field != GPF_FIELD_EXTRA)
        return fail(GPF_ERR_INVALID, "gpf_upload: only q, topography and the extra field are inputs");
    const size_t ncell = (size_t)(L.Nx + 2) * (L.Ny + 2);
    if (count != ncell * nc) return fail(GPF_ERR_INVALID, "gpf_upload: count does not match ncomp*(Nx+2)*(Ny+2)");
    double* dst = nullptr;
    if (field == GPF_FIELD_Q) {
        int par = 0;
        GPF_TRY(current_parity(h, &par));
        dst = h->q[par];
    } else if (field == GPF_FIELD_TOPO) {
        dst = h->topo;
    } else {
        bool nz = false;
        for (size_t i = 0; i < count && !nz; ++i) nz = host[i] != 0.0;
        if (!nz && !h->Ls) return GPF_OK;      // Ls == 0 everywhere: the HAS_LS=false kernels apply
        if (!h->Ls) { HIP_TRY(hipMalloc(&h->Ls, (size_t)L.plane * sizeof(double) + PLANE_PAD_BYTES)); h->plan2_valid = false; }
        HIP_TRY(hipMemsetAsync(h->Ls, 0, (size_t)L.plane * sizeof(double), h->stream));
        dst = h->Ls;
    }
    if (field == GPF_FIELD_TOPO) {
        // does the topography vary along one axis only?  (bitwise test on the host array)
        const int nx = L.Nx + 2, ny = L.Ny + 2;
        bool xonly = true, yonly = true;
        for (int c = 0; c < 3 && (xonly || yonly); ++c)
            for (int ix = 0; ix < nx && (xonly || yonly); ++ix) {
                const double* row = host + ((size_t)c * nx + ix) * ny;
                const double* row0 = host + (size_t)c * nx * ny;
                for (int iy = 0; iy < ny; ++iy) {
                    if (std::memcmp(&row[iy], &row[0], 8) != 0) xonly = false;
                    if (std::memcmp(&row[iy], &row0[iy], 8) != 0) yonly = false;
                }
            }
        const int mode = xonly ? 1 : (yonly ? 2 : 0);
        if (mode != h->topo_mode) { h->plan2_valid = false; }
        h->topo_mode = h->el.on ? 0 : mode;         // an elastic gap changes on the device: always read the planes
        if (mode) {
            const int n = mode == 1 ? nx : ny;
            std::vector<double> line((size_t)3 * n);
            for (int c = 0; c < 3; ++c)
                for (int i = 0; i < n; ++i)
                    line[(size_t)c * n + i] = mode == 1 ? host[((size_t)c * nx + i) * ny] : host[(size_t)c * nx * ny + i];
            bool hy0 = mode == 1;
            for (int i = 0; i < n && hy0; ++i) hy0 = line[(size_t)2 * n + i] == 0.0;
            if (hy0 != h->topo_hy0) h->plan2_valid = false;
            h->topo_hy0 = hy0;
            if (!h->topo_line) HIP_TRY(hipMalloc(&h->topo_line, (size_t)3 * (std::max(nx, ny)) * sizeof(double)));
            HIP_TRY(hipMemcpy(h->topo_line, line.data(), line.size() * sizeof(double), hipMemcpyHostToDevice));
        }
    }
    GPF_TRY(ensure_stage(h, count));
    HIP_TRY(hipMemcpyAsync(h->stage, host, count * sizeof(double), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_pack, dim3(blocks_for((long long)count)), dim3(256), 0, h->stream, h->stage, dst, L, nc);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (field == GPF_FIELD_Q) h->has_q = true;
    h->g1_ready = false;
    h->prev_state_valid = false;
    if (field == GPF_FIELD_TOPO) h->has_topo = true;
    return GPF_OK;
}

extern "C" int gpf_download(gpf_handle* h, int field, double* host, size_t count) {
    if (!h || !host) return fail(GPF_ERR_INVALID, "gpf_download: null argument");
    GPF_TRY(enter(h, true));
    const Layout& L = h->L;
    const int nc = field_ncomp(field);
    if (nc == 0) return fail(GPF_ERR_INVALID, "gpf_download: unknown field id");
    const size_t ncell = (size_t)(L.Nx + 2) * (L.Ny + 2);
    if (count != ncell * nc) return fail(GPF_ERR_INVALID, "gpf_download: count does not match ncomp*(Nx+2)*(Ny+2)");
    const double* src = nullptr;
    switch (field) {
    case GPF_FIELD_Q: {
        int par = 0;
        GPF_TRY(current_parity(h, &par));
        src = h->q[par];
        break;
    }
    case GPF_FIELD_TOPO: src = h->topo; break;
    case GPF_FIELD_EXTRA:
        if (!h->Ls) { std::memset(host, 0, count * sizeof(double)); return GPF_OK; }
        src = h->Ls; break;
    case GPF_FIELD_DEFORMATION:
        if (!h->el.on) { std::memset(host, 0, count * sizeof(double)); return GPF_OK; }
        src = h->el.deformation; break;
    case GPF_FIELD_PRESSURE_VAR: case GPF_FIELD_WALL_XZ_VAR: case GPF_FIELD_WALL_YZ_VAR:
        if (!h->gpvar) return fail(GPF_ERR_STATE, "gpf_download: no GP variance has been computed");
        src = h->gpvar + (size_t)(field - GPF_FIELD_PRESSURE_VAR) * L.plane;
        break;
    default:
        if (!h->fields) return fail(GPF_ERR_STATE, "gpf_download: derived fields requested before gpf_update_closures");
        src = h->fields + (field == GPF_FIELD_PRESSURE ? 0 : field == GPF_FIELD_TAU_AVG ? 1 : field == GPF_FIELD_WALL_LOWER ? 4 : 10) * L.plane;
    }
    GPF_TRY(ensure_stage(h, count));
    hipLaunchKernelGGL(k_unpack, dim3(blocks_for((long long)count)), dim3(256), 0, h->stream, src, h->stage, L, nc);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(host, h->stage, count * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return GPF_OK;
}

static FieldPtrs field_ptrs(gpf_handle* h) {
    FieldPtrs F;
    F.p = h->fields; F.tau = h->fields + h->L.plane; F.lower = h->fields + 4 * h->L.plane; F.upper = h->fields + 10 * h->L.plane;
    return F;
}

static int gp_launch_mean(gpf_handle* h, int which, const double* q, bool with_grad, double* c2_out);
static int read_state(gpf_handle* h, StepState& s);

// thinning on a slab: the beyond rows that belong to field q (the working field of an open step, or the state)
static const double* beyond_rows(gpf_handle* h, const double* q) {
    if (!h->beyond || (h->E.halo[0] != 1 && h->E.halo[1] != 1)) return nullptr;
    const bool working = h->step_open && q == h->q[h->open_parity ^ 1];
    return h->beyond + (working ? 2 * h->L.pitch : 0);
}

// `pressure_mean_cached`: the pressure surrogate's mean of this very field is in gp_state_mean (see gpf_handle)
static int launch_fields(gpf_handle* h, const double* q, bool pressure_mean_cached = false) {
    GPF_TRY(ensure_fields(h));
    const Layout& L = h->L;
    const long long n = (long long)(L.Nx + 2) * (L.Ny + 2);
    FieldPtrs F = field_ptrs(h);
    EOS_DISPATCH(h->cfg.eos, {
        if (h->cfg.thinning != GPF_THINNING_NONE) {
            hipLaunchKernelGGL((k_pressure<EOS_>), dim3(blocks_for(n)), dim3(256), 0, h->stream, q, F.p, L, h->P);
            if ((h->E.halo[0] == 1 || h->E.halo[1] == 1) && !h->beyond)
                return fail(GPF_ERR_STATE, "shear thinning on a slab: upload the rows beyond the halo first (gpf_upload_beyond)");
            const double* by = beyond_rows(h, q);
            if (h->Ls) hipLaunchKernelGGL((k_fields_thinning<EOS_, true>), dim3(blocks_for(n)), dim3(256), 0, h->stream, q, h->topo, h->Ls, F, L, h->P, h->cfg.dx, h->cfg.dy, h->E.halo[0], h->E.halo[1], by);
            else hipLaunchKernelGGL((k_fields_thinning<EOS_, false>), dim3(blocks_for(n)), dim3(256), 0, h->stream, q, h->topo, (const double*)nullptr, F, L, h->P, h->cfg.dx, h->cfg.dy, h->E.halo[0], h->E.halo[1], by);
        } else if (h->Ls) hipLaunchKernelGGL((k_fields<EOS_, true>), dim3(blocks_for(n)), dim3(256), 0, h->stream, q, h->topo, h->Ls, F, L, h->P);
        else hipLaunchKernelGGL((k_fields<EOS_, false>), dim3(blocks_for(n)), dim3(256), 0, h->stream, q, h->topo, (const double*)nullptr, F, L, h->P);
    });
    HIP_TRY(hipGetLastError());
    for (int w = 0; w < 3; ++w) {
        if (!h->gp[w].set) continue;
        if (w == 0 && pressure_mean_cached) {
            HIP_TRY(hipMemcpyAsync(F.p, h->gp_state_mean, (size_t)L.plane * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
            h->gp_reused += 1;
        } else GPF_TRY(gp_launch_mean(h, w, q, false, nullptr));
    }
    return GPF_OK;
}

extern "C" int gpf_update_closures(gpf_handle* h) {
    if (!h) return fail(GPF_ERR_INVALID, "null handle");
    if (!h->has_q || !h->has_topo) return fail(GPF_ERR_STATE, "gpf_update_closures: upload q and topography first");
    GPF_TRY(enter(h));
    int par = 0;
    GPF_TRY(current_parity(h, &par));
    if (h->prev_state_valid && !h->E.halo[0] && !h->E.halo[1]) {
        // What the reference's pressure / wall-stress / bulk-stress fields hold after update() are the closures of the CORRECTOR
        // stage, i.e. of the field the predictor left (problem.py:531-560; nothing re-evaluates them on the averaged state).  The
        // fused step keeps neither that field nor its closures, but the state it started from is still in the other buffer: the
        // predictor stage is run again on it, reference order, with the step size and sweep direction of that step, and the
        // closures are evaluated on the result.  (A slab would need its neighbours' predictor rows: it evaluates on the state.)
        const Layout& L = h->L;
        StepState s;
        GPF_TRY(read_state(h, s));
        double* w = h->q[par ^ 1];
        const long long n = (long long)(L.Nx + 2) * (L.Ny + 2);
        const int nb = blocks_for(n);
        if (!h->work) HIP_TRY(hipMalloc(&h->work, (size_t)9 * L.plane * sizeof(double)));
        GPF_TRY(launch_fields(h, w));
        double *fx = h->work, *fy = h->work + 3 * L.plane, *src = h->work + 6 * L.plane;
        FieldPtrs F = field_ptrs(h);
        const int mc = h->cfg.mc_order;
        const int sw = mc == 0 ? (((s.step - 1) % 2 == 0) ? 1 : -1) : mc;
        const int dir = ((sw + 1) / 2) ? 1 : -1;
        hipLaunchKernelGGL(k_fluxdiff, dim3(nb), dim3(256), 0, h->stream, w, F.p, F.tau, dir, fx, fy, L);
        hipLaunchKernelGGL(k_source, dim3(nb), dim3(256), 0, h->stream, w, h->topo, F.tau, F.lower, F.upper, src, L);
        hipLaunchKernelGGL(k_axpy, dim3(nb), dim3(256), 0, h->stream, w, fx, fy, src, (const double*)&h->st->dt_last, h->cfg.dx, h->cfg.dy, L);
        hipLaunchKernelGGL(k_bc_x, dim3((L.Ny + 2 + 255) / 256), dim3(256), 0, h->stream, w, L, h->E);
        hipLaunchKernelGGL(k_bc_y, dim3((L.Nx + 2 + 255) / 256), dim3(256), 0, h->stream, w, L, h->E);
        HIP_TRY(hipGetLastError());
        GPF_TRY(launch_fields(h, w));
        h->prev_state_valid = false;        // the buffer now holds the predictor's field; the closures stay in the derived fields
    } else {
        GPF_TRY(launch_fields(h, h->q[par]));
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    return GPF_OK;
}

// ---------------------------------------------------------------------------------------------
// scalars
// ---------------------------------------------------------------------------------------------
static int launch_scalars(gpf_handle* h, const double* q, ScalarPartial* total, bool need_sound_speed = true) {
    const Layout& L = h->L;
    // a slab leaves out copies of its neighbours' rows (kind 1); the domain's own ghost rows (kinds 0, 2) count
    const int row0 = h->E.halo[0] == 1 ? 1 : 0, row1 = h->E.halo[1] == 1 ? L.Nx : L.Nx + 1;
    const long long n = (long long)(row1 - row0 + 1) * (L.Ny + 2);
    const int nb = blocks_for(n, 256, h->nspart);
    EOS_DISPATCH(h->cfg.eos, {
        hipLaunchKernelGGL((k_scalars<EOS_>), dim3(nb), dim3(256), 0, h->stream, q, h->topo, L, h->P, row0, row1, h->spart);
    });
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_scalars_final, dim3(1), dim3(256), 0, h->stream, h->spart, nb, total);
    HIP_TRY(hipGetLastError());
    // with a pressure surrogate the sound speed is the steepest slope of the GP mean (stress.py:533-537)
    if (h->gp[0].set && need_sound_speed) GPF_TRY(gp_launch_mean(h, 0, q, true, &total->c2));
    return GPF_OK;
}

static void fill_scalars(const StepState& s, const ScalarPartial* sp, double dxdy, gpf_scalars_t* out) {
    out->step = s.step; out->simtime = s.simtime; out->dt = s.dt;
    out->ekin = sp ? sp->ekin : s.ekin; out->ekin_old = s.ekin_old; out->residual = s.residual;
    out->v_max = std::sqrt(sp ? sp->v2 : s.vmax2);
    out->v_sound = (sp && ((int)sp->flags & 4)) ? std::nan("") : std::sqrt(sp ? sp->c2 : s.c2max);
    out->mass = sp ? sp->mass * dxdy : 0.0;
    out->invalid = s.invalid; out->converged = s.converged;
}

extern "C" int gpf_scalars(gpf_handle* h, gpf_scalars_t* out) {
    if (!h || !out) return fail(GPF_ERR_INVALID, "gpf_scalars: null argument");
    if (!h->has_q || !h->has_topo) return fail(GPF_ERR_STATE, "gpf_scalars: upload q and topography first");
    GPF_TRY(enter(h, true));
    int par = 0;
    GPF_TRY(current_parity(h, &par));
    ScalarPartial* tot = h->spart + h->nspart;
    h->gp_state_mean_fresh = false;
    GPF_TRY(launch_scalars(h, h->q[par], tot));
    ScalarPartial sp; StepState s;
    HIP_TRY(hipMemcpyAsync(&sp, tot, sizeof(sp), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(&s, h->st, sizeof(s), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->gp_state_mean_fresh && s.parity == par) {        // the sound-speed pass ran on the committed state
        h->gp_state_mean_valid = true;
        h->gp_state_mean_step = s.step;
    }
    h->gp_state_mean_fresh = false;
    fill_scalars(s, &sp, h->cfg.dx * h->cfg.dy, out);
    return GPF_OK;
}

static int write_state(gpf_handle* h, const StepState& s) {
    HIP_TRY(hipMemcpyAsync(h->st, &s, sizeof(s), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return GPF_OK;
}
static int read_state(gpf_handle* h, StepState& s) {
    HIP_TRY(hipMemcpyAsync(&s, h->st, sizeof(s), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return GPF_OK;
}

extern "C" int gpf_set_ekin_old(gpf_handle* h, double v) {
    if (!h) return fail(GPF_ERR_INVALID, "null handle");
    StepState s;
    GPF_TRY(read_state(h, s));
    s.ekin_old = v;
    return write_state(h, s);
}

extern "C" int gpf_set_dt(gpf_handle* h, double dt) {
    if (!h) return fail(GPF_ERR_INVALID, "null handle");
    StepState s;
    GPF_TRY(read_state(h, s));
    s.dt = dt;
    h->g1_ready = false;
    return write_state(h, s);
}

// Problem._initialize's Ekin_old (problem.py:670) + _pre_run (problem.py:412-443)
extern "C" int gpf_pre_run(gpf_handle* h) {
    if (!h) return fail(GPF_ERR_INVALID, "null handle");
    if (!h->has_q || !h->has_topo) return fail(GPF_ERR_STATE, "gpf_pre_run: upload q and topography first");
    GPF_TRY(enter(h));
    gpf_scalars_t sc;
    GPF_TRY(gpf_scalars(h, &sc));
    StepState s;
    GPF_TRY(read_state(h, s));
    const gpf_config& c = h->cfg;
    s.hmin = std::min(c.dx, c.dy);
    s.tol = c.tol; s.CFL = c.CFL; s.adaptive = c.adaptive; s.mc_order = c.mc_order; s.max_it = c.max_it;
    s.step = 0; s.simtime = 0.0; s.residual = 1.0;
    s.rbuf[0] = 1.0; s.rcount = 1; s.rhead = 0;
    s.converged = (1.0 < c.tol) ? 1 : 0;
    s.invalid = 0;
    s.dt_last = 0.0;
    if (!h->pre_run_done) s.ekin_old = sc.ekin;     // otherwise keep a user-set kinetic_energy_old
    s.ekin = sc.ekin;
    s.vmax2 = sc.v_max * sc.v_max; s.c2max = sc.v_sound * sc.v_sound;
    const double dt_crit = s.hmin / (sc.v_max + sc.v_sound);
    s.dt = c.adaptive ? c.CFL * dt_crit : c.dt_fixed;
    GPF_TRY(write_state(h, s));
    h->pre_run_done = true;
    h->g1_ready = false;
    h->prev_state_valid = false;
    h->host_step = 0; h->next_step = 0;
    return GPF_OK;
}

// ---------------------------------------------------------------------------------------------
// the fused step
// ---------------------------------------------------------------------------------------------
// 0 planes, 1 profile over ix, 2 profile over iy, 3 profile over ix with dh/dy = 0 (the x-only-gap closure); the line
// modes only without slip-length field / piezo-viscosity.  GPF_TOPO_PLANES / GPF_TOPO_GENERIC switch the specialisations
// off for A/B runs.
static int topo_mode_of(const gpf_handle* h) {
    if (h->Ls != nullptr || h->cfg.piezo != 0 || std::getenv("GPF_TOPO_PLANES")) return 0;
    if (h->topo_mode == 1 && h->topo_hy0 && !std::getenv("GPF_TOPO_GENERIC")) return 3;
    return h->topo_mode;
}

// ---- two-columns-per-lane step kernel (step2_kernel.hip) ----
typedef void (*step2_kernel_t)(const Step2Args, const Phys);

static step2_kernel_t step2_kernel(int eos, bool has_ls, bool piezo, int D, int topo_mode) {
    step2_kernel_t k = nullptr;
    EOS_DISPATCH(eos, {
        if (piezo) {
            if (has_ls) k = D > 0 ? k_step2<EOS_, true, true, 1, 0> : k_step2<EOS_, true, true, -1, 0>;
            else k = D > 0 ? k_step2<EOS_, false, true, 1, 0> : k_step2<EOS_, false, true, -1, 0>;
        } else if (has_ls) {
            k = D > 0 ? k_step2<EOS_, true, false, 1, 0> : k_step2<EOS_, true, false, -1, 0>;
        } else if (topo_mode == 1) {
            k = D > 0 ? k_step2<EOS_, false, false, 1, 1> : k_step2<EOS_, false, false, -1, 1>;
        } else if (topo_mode == 2) {
            k = D > 0 ? k_step2<EOS_, false, false, 1, 2> : k_step2<EOS_, false, false, -1, 2>;
        } else if (topo_mode == 3) {
            k = D > 0 ? k_step2<EOS_, false, false, 1, 3> : k_step2<EOS_, false, false, -1, 3>;
        } else {
            k = D > 0 ? k_step2<EOS_, false, false, 1, 0> : k_step2<EOS_, false, false, -1, 0>;
        }
    });
    return k;
}

// Window geometry of k_step2 for predictor direction D (see Strip2Geom).  Logical column m sits at physical column
// iy = m (D > 0) or Ny+1-m (D < 0), i.e. at element off + iy of its row with off = 15: a lane's pair must start on an
// odd physical column.  D > 0: w0 odd.  D < 0: the pair (m, m+1) starts at iy = Ny - m, odd iff m and Ny differ in
// parity, so w0 is odd for even Ny and even for odd Ny.  Shifting w0 by -4 keeps the alignment and moves column Ny
// from window positions 123..126 (no room for the two wrap lanes behind the ghost column) to 1..4 of the next strip.
static Strip2Geom strip2_geom(const Layout& L, int D) {
    Strip2Geom G;
    const int e = (D < 0 && (L.Ny & 1)) ? 1 : 0;
    for (int k = 0; k < 2; ++k) {
        G.w0 = -1 - e - 4 * k;
        G.s_ghost = (L.Ny - G.w0 - 1) / STRIP2;
        const int p_ny = L.Ny - G.w0 - STRIP2 * G.s_ghost;       // 1..126
        G.lane_ghost = (p_ny + 1) >> 1; G.slot_ghost = (p_ny + 1) & 1;
        if (G.lane_ghost + 2 <= 63) break;
    }
    G.nstrips = G.s_ghost + 1;
    G.lane_wrap = G.lane_ghost + 1;
    const int p0 = -G.w0;                                       // window position of logical column 0 in strip 0
    G.wrap_ma = (p0 % 2 == 0) ? 0 : -1;
    const int d = 1 - G.wrap_ma;                                // logical column 1 relative to the first wrap lane
    G.wrap_src_lane = G.lane_wrap + d / 2; G.wrap_src_slot = d & 1;
    return G;
}

// Edge work inside k_step2 (ghost cells, stage-1 ghost data, slab message, reductions; the commit too unless the handle is a
// slab).  GPF_STEP_UNFUSED_EDGES=1 at gpf_create selects the older split form instead, kept as a cross-check.
static bool step2_fused(const gpf_handle* h) { return !h->split_edges; }

#include "api_plan.inc"

#ifndef GPF_SLAB_COMMIT_BLOCKS
#define GPF_SLAB_COMMIT_BLOCKS 24
#endif
constexpr int SLAB_COMMIT_BLOCKS = GPF_SLAB_COMMIT_BLOCKS;     // k_begin_slab without stage-1 work: a 6-row copy and the commit

static P2PArgs p2p_args(gpf_handle* h, bool on) {
    P2PArgs c;
    c.on = on ? 1 : 0; c.nranks = h->p2p.nranks; c.rank = h->p2p.rank; c.rank_lo = h->p2p.rank_lo; c.rank_hi = h->p2p.rank_hi;
    for (int r = 0; r < P2P_MAX_RANKS; ++r) c.box[r] = h->p2p.box[r];
    c.seq = h->p2p.seq; c.qa = h->q[0]; c.qb = h->q[1];
    return c;
}

static int ghost_args(gpf_handle* h, int honor_stop, GhostArgs& g) {
    const Layout& L = h->L;
    g.qa = h->q[0]; g.qb = h->q[1]; g.topo = h->topo; g.Ls = h->Ls;
    for (int e = 0; e < 2; ++e) {
        g.seam[e] = nullptr;
        if (h->E.halo[e] == 2) {
            if (!h->has_seam[e]) return fail(GPF_ERR_STATE, "periodic slab seam: call gpf_set_seam_topo for this edge first");
            g.seam[e] = h->seam + (size_t)e * 8 * L.pitch;
        }
    }
    g.g1x = h->g1; g.g1y = h->g1 + 3 * L.pitch;
    g.st = h->st; g.L = L; g.E = h->E; g.honor_stop = honor_stop;
    return GPF_OK;
}

static void fill_step2_args(gpf_handle* h, Step2Args& a2, int D, int honor_stop, long long log_base, double* slab_out, bool p2p) {
    const Layout& L = h->L;
    const bool fused = step2_fused(h);
    const Strip2Geom& G2 = h->geom2[D > 0 ? 0 : 1];
    a2.qa = h->q[0]; a2.qb = h->q[1]; a2.topo = h->topo; a2.topo_line = h->topo_line; a2.Ls = h->Ls;
    a2.g1x = h->g1; a2.g1y = h->g1 + 3 * L.pitch;
    a2.st = h->st; a2.partials = h->partials; a2.block_partials = h->block_partials; a2.arrive = h->arrive;
    a2.log = h->log; a2.log_base = log_base; a2.log_cap = h->log_cap;
    a2.L = L; a2.E = h->E; a2.G = G2; a2.nchunks = h->nchunks2; a2.fused = (fused ? 1 : 0) | (h->nt_policy2 >= 1 ? 2 : 0) | (h->nt_policy2 >= 2 ? 4 : 0); a2.honor_stop = honor_stop;
    const bool slab = slab_out != nullptr;
    for (int e = 0; e < 2; ++e) a2.seam[e] = (h->E.halo[e] == 2 && h->has_seam[e]) ? h->seam + (size_t)e * 8 * L.pitch : nullptr;
    a2.out = slab_out; a2.msg = (slab && !p2p) ? h->halo : nullptr; a2.p2p = p2p_args(h, p2p);
}

static int enqueue_step(gpf_handle* h, int honor_stop, long long log_base, double* slab_out,
                        hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr, bool p2p = false) {
    const Layout& L = h->L;
    const int mc = h->cfg.mc_order;
    const int D = mc == 0 ? ((h->next_step % 2 == 0) ? 1 : -1) : (((mc + 1) / 2) ? 1 : -1);
    GPF_TRY(plan_step2(h, D));
    h->next_step += 1;
    const bool fused = step2_fused(h);
    const Strip2Geom& G2 = h->geom2[D > 0 ? 0 : 1];
    const int np_step = G2.nstrips * h->nchunks2;
    GhostArgs g;
    GPF_TRY(ghost_args(h, honor_stop, g));
    Step2Args a2;
    fill_step2_args(h, a2, D, honor_stop, log_base, slab_out, p2p);
    const bool slab = slab_out != nullptr;
    FinishArgs f;
    f.partials = h->partials; f.st = h->st;
    f.log = h->log; f.log_base = log_base; f.log_cap = h->log_cap; f.out = slab_out; f.honor_stop = honor_stop;
    f.L = L; f.E = h->E;
    f.p2p = p2p_args(h, p2p);
    f.arrive = h->arrive; f.msg = slab ? h->halo : nullptr;
    f.nstep_partials = np_step; f.block_partials = h->block_partials;
    GhostFillArgs gf;
    gf.qa = h->q[0]; gf.qb = h->q[1]; gf.st = h->st;
    gf.L = L; gf.E = h->E; gf.honor_stop = honor_stop;

    // Launches per step: k_step2 alone (ghost cells, stage-1 ghost data, reductions and -- unless the handle is a slab --
    // the commit happen inside it).  A slab's k_step2 leaves its boundary rows and its record in the message / the peers'
    // mailboxes; after the exchange k_begin_slab scatters the neighbours' rows, reduces the records in rank order and
    // commits (launched here for the peer-to-peer transport, by gpf_step_commit for the all-gather).
    // The split form (GPF_STEP_UNFUSED_EDGES=1):
    //   k_ghost_stage1  stage-1 values on the downwind ghost row / column (needs the dt the previous step committed)
    //   k_step2         the fused predictor + corrector + average over the interior
    //   k_ghost_fill    ghost cells of the new field + the slab's boundary rows into its message / its peers' mailboxes;
    //                   its last block to finish reduces all records into this rank's record or commits
    //   k_begin_slab    (slabs, after the exchange) as above, plus k_ghost_stage1's job for the next step
    const int ntiles = (L.Nx + L.Ny + 63) / 64;                 // stage-1 ghost work: 64 items per block
    const dim3 ggrid(std::min(ntiles, 512));
    const int nsend = slab ? std::min((6 * L.pitch + 1023) / 1024, 256) : 0;        // block_partials holds 1024
    WaitArgs w;
    w.qa = h->q[0]; w.qb = h->q[1]; w.st = h->st; w.log = h->log; w.log_base = log_base; w.log_cap = h->log_cap;
    w.L = L; w.E = h->E; w.honor_stop = honor_stop; w.arrive = h->arrive + 1; w.timeout_ticks = h->p2p_timeout_ticks; w.p2p = f.p2p;
    w.gathered = nullptr; w.msg_len = 0; w.nranks = 0; w.rank_lo = w.rank_hi = -1; w.stage1 = fused ? 0 : 1;
    const bool has_ls = h->Ls != nullptr;
    if (fused) {
        // the whole step in one launch (step2_kernel.hip)
        const step2_kernel_t k2 = step2_kernel(h->cfg.eos, h->Ls != nullptr, h->cfg.piezo != 0, D, topo_mode_of(h));
        if (ev0) hipEventRecord(ev0, h->stream);
        hipLaunchKernelGGL(k2, dim3(h->nblocks2), dim3(256), 0, h->stream, a2, h->P);
        if (ev1) hipEventRecord(ev1, h->stream);
        if (p2p) {                          // wait for the peers, scatter their rows, commit
            const dim3 cgrid(SLAB_COMMIT_BLOCKS);
            EOS_DISPATCH(h->cfg.eos, {
                hipLaunchKernelGGL((k_begin_slab<EOS_, false, false, false, true>), cgrid, dim3(256), 0, h->stream, g, w, h->P);
            });
        }
        h->g1_ready = false;
        HIP_TRY(hipGetLastError());
        return GPF_OK;
    }
    EOS_DISPATCH(h->cfg.eos, {
        if (!h->g1_ready) {
            if (topo_mode_of(h) == 3) hipLaunchKernelGGL((k_ghost_stage1<EOS_, false, false, true>), ggrid, dim3(256), 0, h->stream, g, h->P);
            else LS_PIEZO_DISPATCH(has_ls, h->cfg.piezo != 0, hipLaunchKernelGGL((k_ghost_stage1<EOS_, LS_, PZ_, false>), ggrid, dim3(256), 0, h->stream, g, h->P));
        }
        if (ev0) hipEventRecord(ev0, h->stream);
        hipLaunchKernelGGL(step2_kernel(h->cfg.eos, h->Ls != nullptr, h->cfg.piezo != 0, D, topo_mode_of(h)), dim3(h->nblocks2), dim3(256), 0, h->stream, a2, h->P);
        if (ev1) hipEventRecord(ev1, h->stream);
        hipLaunchKernelGGL((k_ghost_fill<EOS_>), dim3(h->nghost_blocks + nsend), dim3(256), 0, h->stream, gf, f, h->nghost_blocks, h->P);
        if (p2p) {                          // wait for the peers, commit, stage-1 ghost data of the next step
            if (topo_mode_of(h) == 3) hipLaunchKernelGGL((k_begin_slab<EOS_, false, false, true, true>), ggrid, dim3(256), 0, h->stream, g, w, h->P);
            else LS_PIEZO_DISPATCH(has_ls, h->cfg.piezo != 0, hipLaunchKernelGGL((k_begin_slab<EOS_, LS_, PZ_, false, true>), ggrid, dim3(256), 0, h->stream, g, w, h->P));
        }
    });
    h->g1_ready = p2p;                  // k_begin_slab has prepared the next step's ghost data
    HIP_TRY(hipGetLastError());
    return GPF_OK;
}

// Problems that fit one workgroup's LDS advance whole batches of steps in one launch (small_kernel.hip).
static bool small_grid_eligible(gpf_handle* h) {
    static const bool off = getenv("GPF_SMALL_GRID") && atoi(getenv("GPF_SMALL_GRID")) == 0;
    const long long nc = (long long)(h->L.Nx + 2) * (h->L.Ny + 2);
    return !off && nc * SMALL_DOUBLES_PER_CELL * 8 <= 150 * 1024 && h->cfg.thinning == GPF_THINNING_NONE && !h->E.halo[0] && !h->E.halo[1] &&
           !h->gp[0].set && !h->gp[1].set && !h->gp[2].set && !h->el.on;
}

static int enqueue_small_steps(gpf_handle* h, int nsteps, int honor_stop, long long log_base) {
    const Layout& L = h->L;
    SmallArgs a;
    a.qa = h->q[0]; a.qb = h->q[1]; a.topo = h->topo; a.Ls = h->Ls; a.st = h->st;
    a.log = h->log; a.log_base = log_base; a.log_cap = h->log_cap; a.L = L; a.E = h->E; a.nsteps = nsteps; a.honor_stop = honor_stop;
    const size_t lds = (size_t)(L.Nx + 2) * (L.Ny + 2) * SMALL_DOUBLES_PER_CELL * 8;
    EOS_DISPATCH(h->cfg.eos, {
        if (h->Ls) {
            HIP_TRY(hipFuncSetAttribute((const void*)k_small_steps<EOS_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL((k_small_steps<EOS_, true>), dim3(1), dim3(512), lds, h->stream, a, h->P);
        } else {
            HIP_TRY(hipFuncSetAttribute((const void*)k_small_steps<EOS_, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL((k_small_steps<EOS_, false>), dim3(1), dim3(512), lds, h->stream, a, h->P);
        }
    });
    HIP_TRY(hipGetLastError());
    h->next_step += nsteps;
    h->g1_ready = false;
    return GPF_OK;
}

extern "C" int gpf_step(gpf_handle* h, int64_t n, int honor_stop, gpf_scalars_t* log, int64_t log_capacity,
                        int64_t* n_executed) {
    if (!h) return fail(GPF_ERR_INVALID, "null handle");
    if (!h->pre_run_done) return fail(GPF_ERR_STATE, "gpf_step: call gpf_pre_run first (Problem._pre_run, problem.py:412)");
    if (h->E.halo[0] || h->E.halo[1]) return fail(GPF_ERR_STATE, "gpf_step: this handle is a slab; use gpf_step_local / gpf_step_commit");
    if (h->cfg.thinning != GPF_THINNING_NONE)
        return fail(GPF_ERR_STATE, "gpf_step: shear thinning needs grad p (a wider stencil than the fused step has); use the stage-wise calls");
    if (n < 0) return fail(GPF_ERR_INVALID, "gpf_step: n < 0");
    GPF_TRY(enter(h));
    int64_t done = 0, logged = 0;
    const bool small = small_grid_eligible(h);
    while (done < n) {
        const int64_t batch = std::min<int64_t>(n - done, h->log_cap);
        const long long base = h->host_step;
        if (small) GPF_TRY(enqueue_small_steps(h, (int)batch, honor_stop, base));
        else for (int64_t i = 0; i < batch; ++i) GPF_TRY(enqueue_step(h, honor_stop, base, nullptr));
        StepState s;
        GPF_TRY(read_state(h, s));
        const long long ran = s.step - base;
        const long long entries = ran + ((s.invalid && ran < batch) ? 1 : 0);
        if (log && entries > 0) {
            const long long take = std::min<long long>(entries, log_capacity - logged);
            if (take > 0) {
                HIP_TRY(hipMemcpy(log + logged, h->log, (size_t)take * sizeof(LogEntry), hipMemcpyDeviceToHost));
                logged += take;
            }
        }
        h->host_step = s.step; h->next_step = s.step;
        h->prev_state_valid = ran >= 1 && !s.invalid;      // the other buffer: the state before the last step of this batch
        done += batch;
        if (ran < batch) break;     // stopped on the device (converged / max_it / invalid)
    }
    if (n_executed) *n_executed = h->host_step;
    return GPF_OK;
}

// n steps with a HIP event pair around every launch of the fused step kernel (on the handle's
// stream): *kernel_ms = summed duration of the n k_step launches, *total_ms = first event to last.
extern "C" int gpf_step_timed(gpf_handle* h, int64_t n, double* kernel_ms, double* total_ms) {
    if (!h || !kernel_ms || !total_ms) return fail(GPF_ERR_INVALID, "gpf_step_timed: null argument");
    if (!h->pre_run_done) return fail(GPF_ERR_STATE, "gpf_step_timed: call gpf_pre_run first");
    if (n < 1 || n > h->log_cap) return fail(GPF_ERR_INVALID, "gpf_step_timed: 1 <= n <= 4096");
    GPF_TRY(enter(h));
    std::vector<hipEvent_t> ev(2 * n + 2);
    for (auto& e : ev) HIP_TRY(hipEventCreate(&e));
    HIP_TRY(hipEventRecord(ev[2 * n], h->stream));
    int rc = GPF_OK;
    for (int64_t i = 0; i < n && rc == GPF_OK; ++i) rc = enqueue_step(h, 0, h->host_step, nullptr, ev[2 * i], ev[2 * i + 1]);
    HIP_TRY(hipEventRecord(ev[2 * n + 1], h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    double sum = 0.0;
    float ms = 0.f;
    for (int64_t i = 0; i < n; ++i) {
        HIP_TRY(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]));
        sum += ms;
    }
    HIP_TRY(hipEventElapsedTime(&ms, ev[2 * n], ev[2 * n + 1]));
    *kernel_ms = sum; *total_ms = ms;
    for (auto& e : ev) hipEventDestroy(e);
    StepState s;
    GPF_TRY(read_state(h, s));
    h->prev_state_valid = s.step > h->host_step && !s.invalid;
    h->host_step = s.step; h->next_step = s.step;
    return rc;
}

// The remaining entry points live in the files included below; they are parts of THIS translation unit (they share
// the handle and the static helpers above) and are split only for readability.
#include "api_unfused_step.inc"
#include "api_operators.inc"
#include "api_slab_allgather.inc"
#include "api_elastic.inc"
#include "api_slab_p2p.inc"
#include "api_gp.inc"
#include "api_stagewise.inc"
#include "api_slab_stagewise.inc"
