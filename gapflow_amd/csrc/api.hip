// C ABI of libgapflow_hip.so (see include/gapflow_hip.h for the contract of every entry point).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/gapflow_hip.h"
#include "step_kernel.hip"
#include "aux_kernels.hip"
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

struct gpf_handle {
    gpf_config cfg;
    Layout L;
    Edges E;
    Phys P;
    hipStream_t stream = nullptr;
    // device state
    double* q[2] = {nullptr, nullptr};      // ping-pong, 3 planes each
    double* topo = nullptr;                 // 3 planes
    double* topo_line = nullptr;            // [3][max(Nx,Ny)+2]: the profile when the topography varies along one axis only
    int topo_mode = 0;                      // 0: 2-D planes, 1: function of ix only, 2: function of iy only
    double* Ls = nullptr;                   // 1 plane (allocated on first non-zero upload)
    double* g1 = nullptr;                   // g1x [3][pitch], g1y [3][Nx+2]: stage-1 ghost values of the step about to run
    double* seam = nullptr;                 // [2 edges][2 rows][4: h,hx,hy,Ls][pitch]
    bool has_seam[2] = {false, false};
    // peer-to-peer slab transport (gpf_p2p_*): my mailbox, every rank's mailbox as mapped here, message counter
    struct { bool on = false; int nranks = 0, rank = 0, rank_lo = -1, rank_hi = -1; char* mine = nullptr;
             char* box[P2P_MAX_RANKS] = {}; unsigned long long* seq = nullptr; } p2p;
    double* halo = nullptr;                 // this slab's all-gather message: first row, last row (3 x pitch each), 8-double record
    StepState* st = nullptr;
    Partial* partials = nullptr;
    unsigned int* arrive = nullptr;         // [0] blocks of k_ghost_fill that are done (finish_step), [1] same for k_begin_slab
    bool g1_ready = false;                  // g1 already holds the next step's stage-1 ghost values (k_begin_slab wrote them)
    Partial* block_partials = nullptr;      // one record per edge-kernel block
    int npartials = 0, nstrips = 0, nchunks = 0, rows_per_chunk = 0, nghost_blocks = 0;
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
        double *Z = nullptr, *alpha = nullptr, *L = nullptr, *Linv = nullptr;
        int cap = 0;                            // training points the buffers can hold (grown in steps of 128)
        bool has_linv = false;
        double xscale0 = 1.0;
    } gp[3];
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
    long long host_step = 0;                // step count at the last sync
    long long next_step = 0;                // index of the next step to be enqueued (== device step unless halted)
    bool plan_valid = false;                // rows_per_chunk / nchunks fitted to the step kernel's residency
    int max_chunks = 0;
};

// ---------------------------------------------------------------------------------------------
static void make_phys(const gpf_config& c, Phys& P) {
    std::memset(&P, 0, sizeof(P));
    P.U = c.U; P.V = c.V; P.eta = c.eta; P.zeta = c.zeta;
    P.v1 = c.zeta + (4.0 / 3.0) * c.eta; P.v2 = c.zeta - (2.0 / 3.0) * c.eta;      // viscous.py:82-83
    P.inv_dx = 1.0 / c.dx; P.inv_dy = 1.0 / c.dy;
    P.eos = c.eos; P.piezo = c.piezo;
    const double* e = c.eos_par;
    switch (c.eos) {
    case GPF_EOS_DH:     // rho0, P0, C1, C2
        P.e[0] = e[0]; P.e[1] = e[1]; P.e[2] = e[2]; P.e[3] = e[3];
        P.e[4] = 0.99 * e[3] * e[0]; P.e[5] = 1.0 / e[0]; P.e[6] = e[2] * e[0] * (e[3] - 1.0); P.e[7] = e[3] * e[0];
        break;
    case GPF_EOS_PL:     // rho0, P0, alpha
        P.e[0] = e[0]; P.e[1] = e[1]; P.e[2] = e[2]; P.e[3] = 1.0 / (1.0 - 0.5 * e[2]);
        break;
    case GPF_EOS_VDW:    // M, T, a, b   (pressure.py:168-173)
        P.e[0] = 1000.0 / e[0]; P.e[1] = 8.31446261815324 * e[1]; P.e[2] = e[2] / 10.0; P.e[3] = e[3] / 1000.0;
        break;
    case GPF_EOS_MT:     // rho0, P0, K, n
    case GPF_EOS_CUBIC:  // a, b, c, d
        for (int i = 0; i < 4; ++i) P.e[i] = e[i];
        break;
    case GPF_EOS_BWR: {  // T, gamma; x1..x32 of Johnson, Zollweg & Gubbins (1993), pressure.py:255-272
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
    case GPF_EOS_BAYADA: {   // rho_l, rho_v, c_l, c_v (pressure.py:303-304)
        const double rl = e[0], rv = e[1], cl2 = e[2] * e[2], cv2 = e[3] * e[3];
        const double N = rv * cv2 * rl * cl2 * (rv - rl) / (rv * rv * cv2 - rl * rl * cl2);
        const double Pcav = rv * cv2 - N * std::log(rv * rv * cv2 / (rl * rl * cl2));
        P.e[0] = rl; P.e[1] = rv; P.e[2] = cl2; P.e[3] = cv2; P.e[4] = N; P.e[5] = Pcav; P.e[6] = 1.0 / (rv - rl);
        break;
    }
    }
    const double* z = c.piezo_par;
    switch (c.piezo) {
    case GPF_PIEZO_BARUS: P.pz[0] = z[0]; break;
    case GPF_PIEZO_ROELANDS: P.pz[0] = z[0]; P.pz[1] = z[1]; P.pz[2] = z[2]; P.pz[3] = std::log(c.eta / z[0]); break;
    case GPF_PIEZO_DUKLER:
    case GPF_PIEZO_MCADAMS: P.pz[0] = z[0]; P.pz[1] = z[1]; P.pz[2] = z[2]; break;
    }
    P.thinning = c.thinning;
    for (int i = 0; i < 4; ++i) P.th[i] = c.thinning_par[i];
}

static int blocks_for(long long n, int bs = 256, int cap = 4096) {
    long long b = (n + bs - 1) / bs;
    return (int)std::max<long long>(1, std::min<long long>(b, cap));
}

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
    L.plane = (long long)(cfg->Nx + 2) * L.pitch;
    for (int e = 0; e < 4; ++e) {
        for (int c = 0; c < 3; ++c) h->E.rule[e][c] = cfg->bc_rule[e][c];
        h->E.value[e] = cfg->bc_value[e];
    }
    h->E.halo[0] = cfg->halo_lo; h->E.halo[1] = cfg->halo_hi;
    make_phys(*cfg, h->P);

    // strips of the fused step; the split of the rows into chunks is fitted to the kernel's residency
    // on first use (plan_step), at most max_chunks of >= 2 rows
    h->nstrips = (L.Ny + STRIP - 1) / STRIP;
    h->max_chunks = std::max(1, (L.Nx + 1) / 2);
    h->npartials = h->nstrips * h->max_chunks;

    const size_t plane_b = (size_t)L.plane * sizeof(double);
    auto cleanup = [&](int code) { gpf_destroy(h); return code; };
#define HIP_TRY_C(expr)                                                                                 \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return cleanup(fail(GPF_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)));      \
    } while (0)
    for (int b = 0; b < 2; ++b) {
        HIP_TRY_C(hipMalloc(&h->q[b], 3 * plane_b));
        HIP_TRY_C(hipMemset(h->q[b], 0, 3 * plane_b));
    }
    HIP_TRY_C(hipMalloc(&h->topo, 3 * plane_b));
    HIP_TRY_C(hipMemset(h->topo, 0, 3 * plane_b));
    const size_t g1n = (size_t)3 * L.pitch + (size_t)3 * (L.Nx + 2);
    HIP_TRY_C(hipMalloc(&h->g1, g1n * sizeof(double)));
    HIP_TRY_C(hipMemset(h->g1, 0, g1n * sizeof(double)));
    HIP_TRY_C(hipMalloc(&h->st, sizeof(StepState)));
    HIP_TRY_C(hipMemset(h->st, 0, sizeof(StepState)));
    h->nghost_blocks = std::max(1, std::min(64, (2 * (L.Ny + 2) + 2 * L.Nx + 255) / 256));
    HIP_TRY_C(hipMalloc(&h->partials, (size_t)h->npartials * sizeof(Partial)));
    HIP_TRY_C(hipMemset(h->partials, 0, (size_t)h->npartials * sizeof(Partial)));
    HIP_TRY_C(hipMalloc(&h->arrive, 2 * sizeof(unsigned int)));
    HIP_TRY_C(hipMemset(h->arrive, 0, 2 * sizeof(unsigned int)));
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
    void* ptrs[] = {h->q[0], h->q[1], h->topo, h->topo_line, h->Ls, h->g1, h->seam, h->halo, h->st, h->partials, h->arrive, h->block_partials, h->spart,
                    h->log, h->stage, h->fields, h->work, h->gpvar, h->gpscratch, h->gptile,
                    h->gp[0].Z, h->gp[0].alpha, h->gp[0].L, h->gp[1].Z, h->gp[1].alpha, h->gp[1].L,
                    h->gp[2].Z, h->gp[2].alpha, h->gp[2].L, h->gp[0].Linv, h->gp[1].Linv, h->gp[2].Linv};
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
    HIP_TRY(hipSetDevice(h->cfg.device));
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
        if (!h->Ls) { HIP_TRY(hipMalloc(&h->Ls, (size_t)L.plane * sizeof(double))); h->plan_valid = false; }
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
        if (mode != h->topo_mode) h->plan_valid = false;
        h->topo_mode = h->el.on ? 0 : mode;         // an elastic gap changes on the device: always read the planes
        if (mode) {
            const int n = mode == 1 ? nx : ny;
            std::vector<double> line((size_t)3 * n);
            for (int c = 0; c < 3; ++c)
                for (int i = 0; i < n; ++i)
                    line[(size_t)c * n + i] = mode == 1 ? host[((size_t)c * nx + i) * ny] : host[(size_t)c * nx * ny + i];
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
    if (field == GPF_FIELD_TOPO) h->has_topo = true;
    return GPF_OK;
}

extern "C" int gpf_download(gpf_handle* h, int field, double* host, size_t count) {
    if (!h || !host) return fail(GPF_ERR_INVALID, "gpf_download: null argument");
    HIP_TRY(hipSetDevice(h->cfg.device));
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

static int launch_fields(gpf_handle* h, const double* q) {
    GPF_TRY(ensure_fields(h));
    const Layout& L = h->L;
    const long long n = (long long)(L.Nx + 2) * (L.Ny + 2);
    FieldPtrs F = field_ptrs(h);
    EOS_DISPATCH(h->cfg.eos, {
        if (h->cfg.thinning != GPF_THINNING_NONE) {
            hipLaunchKernelGGL((k_pressure<EOS_>), dim3(blocks_for(n)), dim3(256), 0, h->stream, q, F.p, L, h->P);
            if (h->Ls) hipLaunchKernelGGL((k_fields_thinning<EOS_, true>), dim3(blocks_for(n)), dim3(256), 0, h->stream, q, h->topo, h->Ls, F, L, h->P, h->cfg.dx, h->cfg.dy);
            else hipLaunchKernelGGL((k_fields_thinning<EOS_, false>), dim3(blocks_for(n)), dim3(256), 0, h->stream, q, h->topo, (const double*)nullptr, F, L, h->P, h->cfg.dx, h->cfg.dy);
        } else if (h->Ls) hipLaunchKernelGGL((k_fields<EOS_, true>), dim3(blocks_for(n)), dim3(256), 0, h->stream, q, h->topo, h->Ls, F, L, h->P);
        else hipLaunchKernelGGL((k_fields<EOS_, false>), dim3(blocks_for(n)), dim3(256), 0, h->stream, q, h->topo, (const double*)nullptr, F, L, h->P);
    });
    HIP_TRY(hipGetLastError());
    for (int w = 0; w < 3; ++w)
        if (h->gp[w].set) GPF_TRY(gp_launch_mean(h, w, q, false, nullptr));
    return GPF_OK;
}

extern "C" int gpf_update_closures(gpf_handle* h) {
    if (!h) return fail(GPF_ERR_INVALID, "null handle");
    if (!h->has_q || !h->has_topo) return fail(GPF_ERR_STATE, "gpf_update_closures: upload q and topography first");
    HIP_TRY(hipSetDevice(h->cfg.device));
    int par = 0;
    GPF_TRY(current_parity(h, &par));
    GPF_TRY(launch_fields(h, h->q[par]));
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
    HIP_TRY(hipSetDevice(h->cfg.device));
    int par = 0;
    GPF_TRY(current_parity(h, &par));
    ScalarPartial* tot = h->spart + h->nspart;
    GPF_TRY(launch_scalars(h, h->q[par], tot));
    ScalarPartial sp; StepState s;
    HIP_TRY(hipMemcpyAsync(&sp, tot, sizeof(sp), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(&s, h->st, sizeof(s), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
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
    HIP_TRY(hipSetDevice(h->cfg.device));
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
    if (!h->pre_run_done) s.ekin_old = sc.ekin;     // otherwise keep a user-set kinetic_energy_old
    s.ekin = sc.ekin;
    s.vmax2 = sc.v_max * sc.v_max; s.c2max = sc.v_sound * sc.v_sound;
    const double dt_crit = s.hmin / (sc.v_max + sc.v_sound);
    s.dt = c.adaptive ? c.CFL * dt_crit : c.dt_fixed;
    GPF_TRY(write_state(h, s));
    h->pre_run_done = true;
    h->g1_ready = false;
    h->host_step = 0; h->next_step = 0;
    return GPF_OK;
}

// ---------------------------------------------------------------------------------------------
// the fused step
// ---------------------------------------------------------------------------------------------
typedef void (*step_kernel_t)(const StepArgs, const Phys);

// topo_mode: 0 planes, 1 profile over ix, 2 profile over iy (only without slip-length field / piezo-viscosity)
static step_kernel_t step_kernel(int eos, bool has_ls, bool piezo, int D, int topo_mode) {
    step_kernel_t k = nullptr;
    EOS_DISPATCH(eos, {
        if (piezo) {
            if (has_ls) k = D > 0 ? k_step<EOS_, true, true, 1, 0> : k_step<EOS_, true, true, -1, 0>;
            else k = D > 0 ? k_step<EOS_, false, true, 1, 0> : k_step<EOS_, false, true, -1, 0>;
        } else if (has_ls) {
            k = D > 0 ? k_step<EOS_, true, false, 1, 0> : k_step<EOS_, true, false, -1, 0>;
        } else if (topo_mode == 1) {
            k = D > 0 ? k_step<EOS_, false, false, 1, 1> : k_step<EOS_, false, false, -1, 1>;
        } else if (topo_mode == 2) {
            k = D > 0 ? k_step<EOS_, false, false, 1, 2> : k_step<EOS_, false, false, -1, 2>;
        } else {
            k = D > 0 ? k_step<EOS_, false, false, 1, 0> : k_step<EOS_, false, false, -1, 0>;
        }
    });
    return k;
}

static int topo_mode_of(const gpf_handle* h) {
    if (h->Ls != nullptr || h->cfg.piezo != 0 || std::getenv("GPF_TOPO_PLANES")) return 0;
    return h->topo_mode;
}

// One wave marches over `rows_per_chunk` rows of one strip.  The chunks are sized so that the whole
// grid is resident at once (a single round of waves, no tail) when the problem is big enough.
static int plan_step(gpf_handle* h) {
    if (h->plan_valid) return GPF_OK;
    const Layout& L = h->L;
    int rows = 0;
    if (const char* s = std::getenv("GPF_ROWS_PER_CHUNK")) rows = std::atoi(s);
    if (rows <= 0) {
        int per_cu = 0, ncu = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)step_kernel(h->cfg.eos, h->Ls != nullptr, h->cfg.piezo != 0, 1, topo_mode_of(h)), 256, 0));
        HIP_TRY(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, h->cfg.device));
        const int blocks_per_chunk = (h->nstrips + 3) / 4;
        const int resident = std::max(1, per_cu * ncu);
        const int want_chunks = std::max(1, resident / blocks_per_chunk);
        rows = (L.Nx + want_chunks - 1) / want_chunks;
    }
    rows = std::max(4, std::min(rows, L.Nx));       // small grids: short chunks spread the rows over more CUs (256^2: 13.7 -> 10.8 us)
    if (rows > L.Nx) rows = L.Nx;
    h->rows_per_chunk = std::min(rows, std::max(L.Nx, 1));
    h->nchunks = (L.Nx + h->rows_per_chunk - 1) / h->rows_per_chunk;
    if (h->nchunks > h->max_chunks) return fail(GPF_ERR_INVALID, "plan_step: chunk count exceeds the partials buffer");
    h->plan_valid = true;
    return GPF_OK;
}

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

static int enqueue_step(gpf_handle* h, int honor_stop, long long log_base, double* slab_out,
                        hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr, bool p2p = false) {
    const Layout& L = h->L;
    GPF_TRY(plan_step(h));
    const int mc = h->cfg.mc_order;
    const int D = mc == 0 ? ((h->next_step % 2 == 0) ? 1 : -1) : (((mc + 1) / 2) ? 1 : -1);
    h->next_step += 1;
    const int np_step = h->nstrips * h->nchunks;
    StepArgs a;
    a.qa = h->q[0]; a.qb = h->q[1]; a.topo = h->topo; a.topo_line = h->topo_line; a.Ls = h->Ls;
    a.g1x = h->g1; a.g1y = h->g1 + 3 * L.pitch;
    a.st = h->st; a.partials = h->partials; a.L = L; a.E = h->E;
    a.rows_per_chunk = h->rows_per_chunk; a.nstrips = h->nstrips; a.honor_stop = honor_stop;
    GhostArgs g;
    GPF_TRY(ghost_args(h, honor_stop, g));
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

    // Launches per step:
    //   k_ghost_stage1  stage-1 values on the downwind ghost row / column (needs the dt the previous step committed)
    //   k_step          the fused predictor + corrector + average over the interior
    //   k_ghost_fill    ghost cells of the new field (+ a slab's boundary rows into its message / its peers'
    //                   mailboxes); its last block to finish reduces all records and commits dt, residual, step
    //   peer-to-peer slabs: k_begin_slab (wait for the peers' rows and records, commit, k_ghost_stage1's job for the
    //                   next step) takes the place of the next step's k_ghost_stage1
    const int ntiles = (L.Nx + L.Ny + 63) / 64;                 // stage-1 ghost work: 64 items per block
    const dim3 ggrid(std::min(ntiles, 512)), sgrid((h->nstrips + 3) / 4, h->nchunks);
    const step_kernel_t kstep = step_kernel(h->cfg.eos, h->Ls != nullptr, h->cfg.piezo != 0, D, topo_mode_of(h));
    const int nsend = slab ? std::min((6 * L.pitch + 1023) / 1024, 256) : 0;        // block_partials holds 1024
    WaitArgs w;
    w.qa = h->q[0]; w.qb = h->q[1]; w.st = h->st; w.log = h->log; w.log_base = log_base; w.log_cap = h->log_cap;
    w.L = L; w.E = h->E; w.honor_stop = honor_stop; w.arrive = h->arrive + 1; w.p2p = f.p2p;
    w.gathered = nullptr; w.msg_len = 0; w.nranks = 0; w.rank_lo = w.rank_hi = -1;
    const bool has_ls = h->Ls != nullptr;
    EOS_DISPATCH(h->cfg.eos, {
        if (!h->g1_ready) {
            if (has_ls) hipLaunchKernelGGL((k_ghost_stage1<EOS_, true>), ggrid, dim3(256), 0, h->stream, g, h->P);
            else hipLaunchKernelGGL((k_ghost_stage1<EOS_, false>), ggrid, dim3(256), 0, h->stream, g, h->P);
        }
        if (ev0) hipEventRecord(ev0, h->stream);
        hipLaunchKernelGGL(kstep, sgrid, dim3(256), 0, h->stream, a, h->P);
        if (ev1) hipEventRecord(ev1, h->stream);
        hipLaunchKernelGGL((k_ghost_fill<EOS_>), dim3(h->nghost_blocks + nsend), dim3(256), 0, h->stream, gf, f, h->nghost_blocks, h->P);
        if (p2p) {                          // wait for the peers, commit, stage-1 ghost data of the next step
            if (has_ls) hipLaunchKernelGGL((k_begin_slab<EOS_, true, true>), ggrid, dim3(256), 0, h->stream, g, w, h->P);
            else hipLaunchKernelGGL((k_begin_slab<EOS_, false, true>), ggrid, dim3(256), 0, h->stream, g, w, h->P);
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
    HIP_TRY(hipSetDevice(h->cfg.device));
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
    HIP_TRY(hipSetDevice(h->cfg.device));
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
    h->host_step = s.step; h->next_step = s.step;
    return rc;
}

// ---------------------------------------------------------------------------------------------
// the unfused, reference-ordered step (problem.py:509-586 line by line)
// ---------------------------------------------------------------------------------------------
extern "C" int gpf_open_step(gpf_handle* h);
extern "C" int gpf_stage_closures(gpf_handle* h);
extern "C" int gpf_stage_advance(gpf_handle* h, int stage);
extern "C" int gpf_close_step(gpf_handle* h, gpf_scalars_t* out);

extern "C" int gpf_step_unfused(gpf_handle* h) {
    if (!h) return fail(GPF_ERR_INVALID, "null handle");
    if (!h->pre_run_done) return fail(GPF_ERR_STATE, "gpf_step_unfused: call gpf_pre_run first");
    StepState s;
    GPF_TRY(read_state(h, s));
    if (s.invalid) return GPF_OK;
    GPF_TRY(gpf_open_step(h));
    for (int i = 0; i < 2; ++i) {
        GPF_TRY(gpf_stage_closures(h));
        GPF_TRY(gpf_stage_advance(h, i));
    }
    return gpf_close_step(h, nullptr);
}

// ---------------------------------------------------------------------------------------------
// stateless operators (integrate.py)
// ---------------------------------------------------------------------------------------------
static Layout dense_layout(int nx, int ny) {
    Layout L;
    L.Nx = nx - 2; L.Ny = ny - 2; L.pitch = ny; L.off = 0; L.plane = (long long)nx * ny;
    return L;
}

extern "C" int gpf_predictor_corrector(int nx, int ny, const double* q, const double* p, const double* tau,
                                       int direction, double* flux_x, double* flux_y) {
    if (!q || !p || !tau || !flux_x || !flux_y) return fail(GPF_ERR_INVALID, "gpf_predictor_corrector: null argument");
    if (nx < 1 || ny < 1 || (direction != 1 && direction != -1))
        return fail(GPF_ERR_INVALID, "gpf_predictor_corrector: nx, ny >= 1 and direction = +-1 required");
    if (gpf_device_count() == 0) return fail(GPF_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU path)");
    const Layout L = dense_layout(nx, ny);
    const size_t n = (size_t)nx * ny;
    double* d = nullptr;
    HIP_TRY(hipMalloc(&d, 13 * n * sizeof(double)));
    double *dq = d, *dp = d + 3 * n, *dt = d + 4 * n, *dfx = d + 7 * n, *dfy = d + 10 * n;
    int rc = GPF_OK;
    hipError_t e;
    if ((e = hipMemcpy(dq, q, 3 * n * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(dp, p, n * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(dt, tau, 3 * n * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess) {
        rc = fail(GPF_ERR_HIP, hipGetErrorString(e));
    } else {
        hipLaunchKernelGGL(k_fluxdiff, dim3(blocks_for((long long)n)), dim3(256), 0, 0, dq, dp, dt, direction, dfx, dfy, L);
        if ((e = hipGetLastError()) != hipSuccess ||
            (e = hipMemcpy(flux_x, dfx, 3 * n * sizeof(double), hipMemcpyDeviceToHost)) != hipSuccess ||
            (e = hipMemcpy(flux_y, dfy, 3 * n * sizeof(double), hipMemcpyDeviceToHost)) != hipSuccess)
            rc = fail(GPF_ERR_HIP, hipGetErrorString(e));
    }
    hipFree(d);
    return rc;
}

extern "C" int gpf_source(int nx, int ny, const double* q, const double* hh, const double* stress, const double* lower,
                          const double* upper, double* out) {
    if (!q || !hh || !stress || !lower || !upper || !out) return fail(GPF_ERR_INVALID, "gpf_source: null argument");
    if (nx < 1 || ny < 1) return fail(GPF_ERR_INVALID, "gpf_source: nx, ny >= 1 required");
    if (gpf_device_count() == 0) return fail(GPF_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU path)");
    const Layout L = dense_layout(nx, ny);
    const size_t n = (size_t)nx * ny;
    double* d = nullptr;
    HIP_TRY(hipMalloc(&d, 24 * n * sizeof(double)));
    double *dq = d, *dh = d + 3 * n, *ds = d + 6 * n, *dl = d + 9 * n, *du = d + 15 * n, *dout = d + 21 * n;
    int rc = GPF_OK;
    hipError_t e;
    if ((e = hipMemcpy(dq, q, 3 * n * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(dh, hh, 3 * n * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(ds, stress, 3 * n * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(dl, lower, 6 * n * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(du, upper, 6 * n * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess) {
        rc = fail(GPF_ERR_HIP, hipGetErrorString(e));
    } else {
        hipLaunchKernelGGL(k_source, dim3(blocks_for((long long)n)), dim3(256), 0, 0, dq, dh, ds, dl, du, dout, L);
        if ((e = hipGetLastError()) != hipSuccess ||
            (e = hipMemcpy(out, dout, 3 * n * sizeof(double), hipMemcpyDeviceToHost)) != hipSuccess)
            rc = fail(GPF_ERR_HIP, hipGetErrorString(e));
    }
    hipFree(d);
    return rc;
}

// models/viscous.py as functions of arrays: stress_bottom / stress_top / stress_avg in one call
extern "C" int gpf_viscous_stress(int64_t n, const double* q, const double* hh, const double* dqx, const double* dqy,
                                  const double* eta, const double* Ls, double U, double V, double zeta, int slip_both,
                                  double* bottom, double* top, double* avg) {
    if (!q || !hh || !eta || !Ls) return fail(GPF_ERR_INVALID, "gpf_viscous_stress: null argument");
    if (!bottom && !top && !avg) return fail(GPF_ERR_INVALID, "gpf_viscous_stress: no output requested");
    if (n < 1) return fail(GPF_ERR_INVALID, "gpf_viscous_stress: n >= 1 required");
    if (gpf_device_count() == 0) return fail(GPF_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU path)");
    const size_t N = (size_t)n;
    double* d = nullptr;
    HIP_TRY(hipMalloc(&d, 29 * N * sizeof(double)));
    ViscousArgs a;
    double* w = d;
    auto put = [&](const double* host, size_t count, const double** dev) -> hipError_t {
        *dev = nullptr;
        if (!host) return hipSuccess;
        *dev = w;
        hipError_t e = hipMemcpy(w, host, count * sizeof(double), hipMemcpyHostToDevice);
        w += count;
        return e;
    };
    hipError_t e;
    int rc = GPF_OK;
    if ((e = put(q, 3 * N, &a.q)) != hipSuccess || (e = put(hh, 3 * N, &a.h)) != hipSuccess ||
        (e = put(dqx, 3 * N, &a.dqx)) != hipSuccess || (e = put(dqy, 3 * N, &a.dqy)) != hipSuccess ||
        (e = put(eta, N, &a.eta)) != hipSuccess || (e = put(Ls, N, &a.Ls)) != hipSuccess) {
        rc = fail(GPF_ERR_HIP, hipGetErrorString(e));
    } else {
        a.U = U; a.V = V; a.zeta = zeta; a.slip_both = slip_both ? 1 : 0; a.n = n;
        a.out[0] = bottom ? w : nullptr; if (bottom) w += 6 * N;
        a.out[1] = top ? w : nullptr; if (top) w += 6 * N;
        a.out[2] = avg ? w : nullptr;
        hipLaunchKernelGGL(k_viscous, dim3(blocks_for((long long)n)), dim3(256), 0, 0, a);
        if ((e = hipGetLastError()) != hipSuccess ||
            (bottom && (e = hipMemcpy(bottom, a.out[0], 6 * N * sizeof(double), hipMemcpyDeviceToHost)) != hipSuccess) ||
            (top && (e = hipMemcpy(top, a.out[1], 6 * N * sizeof(double), hipMemcpyDeviceToHost)) != hipSuccess) ||
            (avg && (e = hipMemcpy(avg, a.out[2], 3 * N * sizeof(double), hipMemcpyDeviceToHost)) != hipSuccess))
            rc = fail(GPF_ERR_HIP, hipGetErrorString(e));
    }
    hipFree(d);
    return rc;
}

// models/pressure.py eos_pressure and models/sound.py eos_sound_velocity as functions of an array
extern "C" int gpf_eos(int eos, const double* eos_par, int64_t n, const double* rho, double* pressure, double* sound) {
    if (!eos_par || !rho || (!pressure && !sound)) return fail(GPF_ERR_INVALID, "gpf_eos: null argument");
    if (eos < 0 || eos > GPF_EOS_BAYADA) return fail(GPF_ERR_INVALID, "gpf_eos: unknown equation of state");
    if (n < 1) return fail(GPF_ERR_INVALID, "gpf_eos: n >= 1 required");
    if (gpf_device_count() == 0) return fail(GPF_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU path)");
    gpf_config c;
    std::memset(&c, 0, sizeof(c));
    c.eos = eos; c.dx = c.dy = 1.0;
    for (int i = 0; i < 8; ++i) c.eos_par[i] = eos_par[i];
    Phys P;
    make_phys(c, P);
    const size_t N = (size_t)n;
    double* d = nullptr;
    HIP_TRY(hipMalloc(&d, 3 * N * sizeof(double)));
    hipError_t e;
    int rc = GPF_OK;
    if ((e = hipMemcpy(d, rho, N * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess) {
        rc = fail(GPF_ERR_HIP, hipGetErrorString(e));
    } else {
        double* dp = pressure ? d + N : nullptr;
        double* dc = sound ? d + 2 * N : nullptr;
        EOS_DISPATCH(eos, { hipLaunchKernelGGL((k_eos<EOS_>), dim3(blocks_for((long long)n)), dim3(256), 0, 0, d, (long long)n, P, dp, dc); });
        if ((e = hipGetLastError()) != hipSuccess ||
            (pressure && (e = hipMemcpy(pressure, dp, N * sizeof(double), hipMemcpyDeviceToHost)) != hipSuccess) ||
            (sound && (e = hipMemcpy(sound, dc, N * sizeof(double), hipMemcpyDeviceToHost)) != hipSuccess))
            rc = fail(GPF_ERR_HIP, hipGetErrorString(e));
    }
    hipFree(d);
    return rc;
}

// models/viscosity.py as functions of arrays
extern "C" int gpf_viscosity(int kind, int law, const double* par, double mu0, int64_t n, const double* a0, const double* a1,
                             const double* a2, double u1, double u2, double* out) {
    if (!a0 || !out || (kind == 2 && (!a1 || !a2)) || (kind != 2 && !par)) return fail(GPF_ERR_INVALID, "gpf_viscosity: null argument");
    if (kind < 0 || kind > 2 || n < 1) return fail(GPF_ERR_INVALID, "gpf_viscosity: kind in 0..2 and n >= 1 required");
    if ((kind == 0 && (law < 0 || law > GPF_PIEZO_MCADAMS)) || (kind == 1 && (law < 0 || law > GPF_THINNING_CARREAU)))
        return fail(GPF_ERR_INVALID, "gpf_viscosity: unknown law");
    if (gpf_device_count() == 0) return fail(GPF_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU path)");
    gpf_config c;
    std::memset(&c, 0, sizeof(c));
    c.eos = GPF_EOS_DH; c.dx = c.dy = 1.0; c.eta = mu0;
    c.eos_par[0] = 1.0; c.eos_par[2] = 1.0; c.eos_par[3] = 2.0;
    if (kind == 0) { c.piezo = law; for (int i = 0; i < 4; ++i) c.piezo_par[i] = par[i]; }
    if (kind == 1) { c.thinning = law; for (int i = 0; i < 4; ++i) c.thinning_par[i] = par[i]; }
    Phys P;
    make_phys(c, P);
    const size_t N = (size_t)n;
    double* d = nullptr;
    HIP_TRY(hipMalloc(&d, 4 * N * sizeof(double)));
    hipError_t e = hipMemcpy(d, a0, N * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess && kind == 2) e = hipMemcpy(d + N, a1, N * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess && kind == 2) e = hipMemcpy(d + 2 * N, a2, N * sizeof(double), hipMemcpyHostToDevice);
    int rc = GPF_OK;
    if (e != hipSuccess) {
        rc = fail(GPF_ERR_HIP, hipGetErrorString(e));
    } else {
        hipLaunchKernelGGL(k_viscosity, dim3(blocks_for((long long)n)), dim3(256), 0, 0, kind, d, d + N, d + 2 * N, (long long)n,
                           mu0, u1, u2, P, d + 3 * N);
        if ((e = hipGetLastError()) != hipSuccess || (e = hipMemcpy(out, d + 3 * N, N * sizeof(double), hipMemcpyDeviceToHost)) != hipSuccess)
            rc = fail(GPF_ERR_HIP, hipGetErrorString(e));
    }
    hipFree(d);
    return rc;
}

// ---------------------------------------------------------------------------------------------
// slab decomposition
// ---------------------------------------------------------------------------------------------
static size_t halo_len(gpf_handle* h) { return (size_t)6 * h->L.pitch + 8; }

static int ensure_halo(gpf_handle* h) {
    if (h->halo) return GPF_OK;
    HIP_TRY(hipMalloc(&h->halo, halo_len(h) * sizeof(double)));
    HIP_TRY(hipMemsetAsync(h->halo, 0, halo_len(h) * sizeof(double), h->stream));
    return GPF_OK;
}

extern "C" int gpf_slab_message(gpf_handle* h, void** message, size_t* count) {
    if (!h || !message || !count) return fail(GPF_ERR_INVALID, "gpf_slab_message: null argument");
    HIP_TRY(hipSetDevice(h->cfg.device));
    GPF_TRY(ensure_halo(h));
    *message = h->halo;
    *count = halo_len(h);
    return GPF_OK;
}

static HaloArgs halo_args(gpf_handle* h, int honor_stop, const double* gathered, int rank_lo, int rank_hi) {
    HaloArgs a;
    a.qa = h->q[0]; a.qb = h->q[1];
    a.msg = h->halo; a.gathered = gathered; a.rank_lo = rank_lo; a.rank_hi = rank_hi;
    a.st = h->st; a.L = h->L; a.E = h->E; a.honor_stop = honor_stop;
    a.work_parity = -1;
    return a;
}

extern "C" int gpf_step_local(gpf_handle* h, int honor_stop) {
    if (!h) return fail(GPF_ERR_INVALID, "gpf_step_local: null handle");
    if (!h->pre_run_done) return fail(GPF_ERR_STATE, "gpf_step_local: call gpf_pre_run first");
    HIP_TRY(hipSetDevice(h->cfg.device));
    GPF_TRY(ensure_halo(h));
    double* rec = h->halo + (size_t)6 * h->L.pitch;
    GPF_TRY(enqueue_step(h, honor_stop, h->host_step, rec));      // leaves rows and record in the message
    return GPF_OK;
}

extern "C" int gpf_step_commit(gpf_handle* h, int honor_stop, const void* gathered, int nranks, int rank_lo, int rank_hi) {
    if (!h || !gathered || nranks < 1) return fail(GPF_ERR_INVALID, "gpf_step_commit: bad argument");
    if (rank_lo >= nranks || rank_hi >= nranks) return fail(GPF_ERR_INVALID, "gpf_step_commit: neighbour rank out of range");
    if (!h->halo) return fail(GPF_ERR_STATE, "gpf_step_commit without gpf_step_local");
    HIP_TRY(hipSetDevice(h->cfg.device));
    // one launch: scatter the neighbours' rows, reduce the records in rank order, commit, and prepare the next step's
    // stage-1 ghost data (so the next gpf_step_local starts with the stencil)
    const Layout& L = h->L;
    GhostArgs g;
    GPF_TRY(ghost_args(h, honor_stop, g));
    WaitArgs w;
    w.qa = h->q[0]; w.qb = h->q[1]; w.st = h->st; w.log = h->log; w.log_base = h->host_step; w.log_cap = h->log_cap;
    w.L = L; w.E = h->E; w.honor_stop = honor_stop; w.arrive = h->arrive + 1; w.p2p = p2p_args(h, false);
    w.gathered = (const double*)gathered; w.msg_len = (long long)halo_len(h); w.nranks = nranks; w.rank_lo = rank_lo; w.rank_hi = rank_hi;
    const dim3 ggrid(std::min((L.Nx + L.Ny + 63) / 64, 512));
    EOS_DISPATCH(h->cfg.eos, {
        if (h->Ls) hipLaunchKernelGGL((k_begin_slab<EOS_, true, false>), ggrid, dim3(256), 0, h->stream, g, w, h->P);
        else hipLaunchKernelGGL((k_begin_slab<EOS_, false, false>), ggrid, dim3(256), 0, h->stream, g, w, h->P);
    });
    HIP_TRY(hipGetLastError());
    h->g1_ready = true;
    return GPF_OK;
}

// ---------------------------------------------------------------------------------------------
// elastic deformation of the gap (topography.py:257-280, 327-437)
// ---------------------------------------------------------------------------------------------
extern "C" int gpf_elastic_setup(gpf_handle* h, int px, int py, const double* greens_ri, size_t count, double alpha,
                                 double force_scale, int relative) {
    if (!h || !greens_ri) return fail(GPF_ERR_INVALID, "gpf_elastic_setup: null argument");
    const Layout& L = h->L;
    if (px < L.Nx + 2 || py < L.Ny + 2) return fail(GPF_ERR_INVALID, "gpf_elastic_setup: the transform grid must hold the field incl. ghost cells");
    const size_t nspec = (size_t)px * (py / 2 + 1);
    if (count != 2 * nspec) return fail(GPF_ERR_INVALID, "gpf_elastic_setup: count must be 2 * px * (py/2 + 1) (real, imaginary)");
    if (!h->has_topo) return fail(GPF_ERR_STATE, "gpf_elastic_setup: upload the undeformed topography first");
    if (h->E.halo[0] || h->E.halo[1]) return fail(GPF_ERR_STATE, "gpf_elastic_setup: not available for slabs (the half-space couples the whole domain)");
    FftLib& F = fftlib();
    if (!F.ok) return fail(GPF_ERR_SOLVER, F.err);
    HIP_TRY(hipSetDevice(h->cfg.device));
    auto& e = h->el;
    if (e.on) return fail(GPF_ERR_STATE, "gpf_elastic_setup: already set up");
    HIP_TRY(hipMalloc((void**)&e.greens, nspec * sizeof(double2)));
    HIP_TRY(hipMalloc((void**)&e.spec, nspec * sizeof(double2)));
    HIP_TRY(hipMalloc((void**)&e.dense, (size_t)px * py * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&e.u_prev, (size_t)3 * L.plane * sizeof(double)));
    e.h0 = e.u_prev + L.plane; e.deformation = e.u_prev + 2 * L.plane;
    HIP_TRY(hipMemset(e.u_prev, 0, (size_t)3 * L.plane * sizeof(double)));
    HIP_TRY(hipMemcpy(e.greens, greens_ri, nspec * sizeof(double2), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(e.h0, h->topo, (size_t)L.plane * sizeof(double), hipMemcpyDeviceToDevice));
    if (F.plan2d(&e.plan_f, px, py, HIPFFT_D2Z_) != 0 || F.plan2d(&e.plan_b, px, py, HIPFFT_Z2D_) != 0)
        return fail(GPF_ERR_SOLVER, "hipfftPlan2d failed");
    e.px = px; e.py = py; e.alpha = alpha; e.relative = relative ? 1 : 0;
    e.scale = force_scale / ((double)px * (double)py);      // hipFFT's inverse is unnormalised
    e.on = true;
    h->topo_mode = 0;                                       // the gap now changes every step: read the planes
    return GPF_OK;
}

// Topography.update + update_gradients with the pressure of the last closure evaluation (problem.py:566 uses the
// stored field, i.e. stage 2's): h, dh/dx, dh/dy of the handle change in place.
extern "C" int gpf_elastic_update(gpf_handle* h) {
    if (!h) return fail(GPF_ERR_INVALID, "gpf_elastic_update: null handle");
    auto& e = h->el;
    if (!e.on) return fail(GPF_ERR_STATE, "gpf_elastic_update: call gpf_elastic_setup first");
    if (!h->fields) return fail(GPF_ERR_STATE, "gpf_elastic_update: no pressure field yet (gpf_stage_closures / gpf_update_closures)");
    FftLib& F = fftlib();
    HIP_TRY(hipSetDevice(h->cfg.device));
    const Layout& L = h->L;
    const long long nd = (long long)e.px * e.py, ns = (long long)e.px * (e.py / 2 + 1), nc = (long long)(L.Nx + 2) * (L.Ny + 2);
    F.set_stream(e.plan_f, h->stream); F.set_stream(e.plan_b, h->stream);
    hipLaunchKernelGGL(k_el_pack, dim3(blocks_for(nd)), dim3(256), 0, h->stream, h->fields, L, e.px, e.py, e.relative, e.dense);
    if (F.d2z(e.plan_f, e.dense, e.spec) != 0) return fail(GPF_ERR_SOLVER, "hipfftExecD2Z failed");
    hipLaunchKernelGGL(k_el_multiply, dim3(blocks_for(ns)), dim3(256), 0, h->stream, e.spec, e.greens, ns);
    if (F.z2d(e.plan_b, e.spec, e.dense) != 0) return fail(GPF_ERR_SOLVER, "hipfftExecZ2D failed");
    hipLaunchKernelGGL(k_el_relax, dim3(blocks_for(nc)), dim3(256), 0, h->stream, e.dense, e.py, e.scale, e.alpha, L, e.u_prev);
    hipLaunchKernelGGL(k_el_apply, dim3(blocks_for(nc)), dim3(256), 0, h->stream, e.u_prev, e.h0, e.relative, L, e.deformation, h->topo);
    hipLaunchKernelGGL(k_el_gradient, dim3(blocks_for(nc)), dim3(256), 0, h->stream, h->topo, L, 1.0 / h->cfg.dx, 1.0 / h->cfg.dy,
                       h->topo + L.plane, h->topo + 2 * L.plane);
    HIP_TRY(hipGetLastError());
    h->g1_ready = false;
    return GPF_OK;
}

// ---------------------------------------------------------------------------------------------
// peer-to-peer slab transport
// ---------------------------------------------------------------------------------------------
extern "C" int gpf_p2p_export(gpf_handle* h, void* ipc_handle, size_t handle_bytes) {
    if (!h || !ipc_handle) return fail(GPF_ERR_INVALID, "gpf_p2p_export: null argument");
    if (handle_bytes != sizeof(hipIpcMemHandle_t)) return fail(GPF_ERR_INVALID, "gpf_p2p_export: the handle buffer must hold 64 bytes");
    if (h->p2p.on) return fail(GPF_ERR_STATE, "gpf_p2p_export: already connected");
    HIP_TRY(hipSetDevice(h->cfg.device));
    const size_t bytes = p2p_mailbox_bytes(h->L.pitch);
    if (!h->p2p.mine) {
        // fine-grained: peers' stores become visible to a kernel that is already running here
        HIP_TRY(hipExtMallocWithFlags((void**)&h->p2p.mine, bytes, hipDeviceMallocFinegrained));
    }
    HIP_TRY(hipMemset(h->p2p.mine, 0, bytes));
    HIP_TRY(hipDeviceSynchronize());
    hipIpcMemHandle_t hm;
    HIP_TRY(hipIpcGetMemHandle(&hm, h->p2p.mine));
    std::memcpy(ipc_handle, &hm, sizeof(hm));
    return GPF_OK;
}

extern "C" int gpf_p2p_connect(gpf_handle* h, int rank, int nranks, const void* ipc_handles, int rank_lo, int rank_hi) {
    if (!h || !ipc_handles) return fail(GPF_ERR_INVALID, "gpf_p2p_connect: null argument");
    if (nranks < 1 || nranks > P2P_MAX_RANKS) return fail(GPF_ERR_INVALID, "gpf_p2p_connect: 1 <= nranks <= 16");
    if (rank < 0 || rank >= nranks || rank_lo >= nranks || rank_hi >= nranks) return fail(GPF_ERR_INVALID, "gpf_p2p_connect: rank out of range");
    if (!h->p2p.mine) return fail(GPF_ERR_STATE, "gpf_p2p_connect: call gpf_p2p_export first");
    if (h->p2p.on) return fail(GPF_ERR_STATE, "gpf_p2p_connect: already connected");
    HIP_TRY(hipSetDevice(h->cfg.device));
    const hipIpcMemHandle_t* hs = (const hipIpcMemHandle_t*)ipc_handles;
    for (int r = 0; r < nranks; ++r) {
        if (r == rank) { h->p2p.box[r] = h->p2p.mine; continue; }
        void* p = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&p, hs[r], hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            for (int k = 0; k < r; ++k)
                if (k != rank && h->p2p.box[k]) { hipIpcCloseMemHandle(h->p2p.box[k]); h->p2p.box[k] = nullptr; }
            return fail(GPF_ERR_HIP, std::string("gpf_p2p_connect: hipIpcOpenMemHandle(rank ") + std::to_string(r) + "): " + hipGetErrorString(e));
        }
        h->p2p.box[r] = (char*)p;
    }
    if (!h->p2p.seq) HIP_TRY(hipMalloc((void**)&h->p2p.seq, sizeof(unsigned long long)));
    HIP_TRY(hipMemset(h->p2p.seq, 0, sizeof(unsigned long long)));
    HIP_TRY(hipDeviceSynchronize());
    h->p2p.nranks = nranks; h->p2p.rank = rank; h->p2p.rank_lo = rank_lo; h->p2p.rank_hi = rank_hi;
    h->p2p.on = true;
    return GPF_OK;
}

// n whole steps, exchanges included, enqueued without touching the host in between
extern "C" int gpf_step_p2p(gpf_handle* h, int64_t n, int honor_stop) {
    if (!h) return fail(GPF_ERR_INVALID, "gpf_step_p2p: null handle");
    if (!h->pre_run_done) return fail(GPF_ERR_STATE, "gpf_step_p2p: call gpf_pre_run first");
    if (!h->p2p.on) return fail(GPF_ERR_STATE, "gpf_step_p2p: call gpf_p2p_export / gpf_p2p_connect first");
    if (n < 0) return fail(GPF_ERR_INVALID, "gpf_step_p2p: n < 0");
    HIP_TRY(hipSetDevice(h->cfg.device));
    GPF_TRY(ensure_halo(h));
    double* rec = h->halo + (size_t)6 * h->L.pitch;
    for (int64_t i = 0; i < n; ++i) GPF_TRY(enqueue_step(h, honor_stop, h->host_step, rec, nullptr, nullptr, true));
    return GPF_OK;
}

extern "C" int gpf_state(gpf_handle* h, gpf_scalars_t* out) {
    if (!h || !out) return fail(GPF_ERR_INVALID, "gpf_state: null argument");
    HIP_TRY(hipSetDevice(h->cfg.device));
    StepState s;
    GPF_TRY(read_state(h, s));
    fill_scalars(s, nullptr, 0.0, out);
    h->host_step = s.step; h->next_step = s.step;
    return GPF_OK;
}

extern "C" int gpf_set_seam_topo(gpf_handle* h, int side, const double* host, size_t count) {
    if (!h || !host || side < 0 || side > 1) return fail(GPF_ERR_INVALID, "gpf_set_seam_topo: bad argument");
    const Layout& L = h->L;
    if (count != (size_t)8 * (L.Ny + 2)) return fail(GPF_ERR_INVALID, "gpf_set_seam_topo: count must be 2*4*(Ny+2)");
    HIP_TRY(hipSetDevice(h->cfg.device));
    if (!h->seam) {
        HIP_TRY(hipMalloc(&h->seam, (size_t)16 * L.pitch * sizeof(double)));
        HIP_TRY(hipMemset(h->seam, 0, (size_t)16 * L.pitch * sizeof(double)));
    }
    std::vector<double> tmp((size_t)8 * L.pitch, 0.0);
    for (int r = 0; r < 8; ++r)
        for (int iy = 0; iy < L.Ny + 2; ++iy) tmp[(size_t)r * L.pitch + L.off + iy] = host[(size_t)r * (L.Ny + 2) + iy];
    HIP_TRY(hipMemcpy(h->seam + (size_t)side * 8 * L.pitch, tmp.data(), tmp.size() * sizeof(double), hipMemcpyHostToDevice));
    h->has_seam[side] = true;
    return GPF_OK;
}

// ---------------------------------------------------------------------------------------------
// GP surrogate closure (gp_kernels.hip)
// ---------------------------------------------------------------------------------------------
static int gp_blas(gpf_handle* h, void** out) {
    RocLibs& R = roclibs();
    if (!R.ok) return fail(GPF_ERR_SOLVER, R.err);
    if (!h->blas) {
        if (R.create(&h->blas) != 0) return fail(GPF_ERR_SOLVER, "rocblas_create_handle failed");
    }
    if (R.set_stream(h->blas, h->stream) != 0) return fail(GPF_ERR_SOLVER, "rocblas_set_stream failed");
    *out = h->blas;
    return GPF_OK;
}

// Builds K from kernel-coordinate inputs Z (device, [n][d]), factorises it in place (column-major lower),
// solves for alpha (device, column-major [n][m], holds Y on entry).  *info_host != 0: not positive definite.
static int gp_factorize(void* blas, hipStream_t stream, const double* Z, int n, int d, int m, double amp, double sigma,
                        double* K, double* alpha, int* info_host) {
    RocLibs& R = roclibs();
    DBG("gp_factorize: kernel matrix n=%d d=%d m=%d", n, d, m);
    hipLaunchKernelGGL(k_gp_matrix, dim3((n + 127) / 128, n), dim3(128), 0, stream, Z, n, d, amp, sigma * sigma, K);
    HIP_TRY(hipGetLastError());
    int* info = nullptr;
    HIP_TRY(hipMalloc(&info, sizeof(int)));
    int rc = 0;
    if (R.potrf) {      // GPF_USE_ROCSOLVER=1
        DBG("gp_factorize: rocsolver dpotrf");
        rc = R.potrf(blas, ROC_FILL_LOWER, n, K, n, info);
        if (rc == 0) rc = R.potrs(blas, ROC_FILL_LOWER, n, m, K, n, alpha, n);
    } else if (R.gemm && n > 96 && !getenv("GPF_GP_UNBLOCKED_POTRF")) {
        // right-looking blocked Cholesky: 64 x 64 diagonal blocks in one workgroup, the panel solve and the trailing update
        // on rocBLAS (dtrsm, dgemm on the matrix cores): 12 ms -> ~1 ms at n = 512
        const int NB = 64;
        const double one = 1.0, minus = -1.0;
        for (int j0 = 0; j0 < n && rc == 0; j0 += NB) {
            const int jb = std::min(NB, n - j0), rest = n - j0 - jb;
            double* A11 = K + j0 + (long long)j0 * n;
            hipLaunchKernelGGL(k_gp_potrf, dim3(1), dim3(1024), 0, stream, A11, jb, n, j0, info);
            if (rest > 0) {
                double* A21 = A11 + jb;
                double* A22 = A21 + (long long)jb * n;
                rc = R.trsm(blas, ROC_SIDE_RIGHT, ROC_FILL_LOWER, ROC_OP_TRANS, ROC_DIAG_NON_UNIT, rest, jb, &one, A11, n, A21, n);
                if (rc == 0) rc = R.gemm(blas, ROC_OP_NONE, ROC_OP_TRANS, rest, rest, jb, &minus, A21, n, A21, n, &one, A22, n);
            }
        }
        hipLaunchKernelGGL(k_gp_potrs, dim3(1), dim3(1024), 0, stream, K, n, m, alpha);
    } else {
        hipLaunchKernelGGL(k_gp_potrf, dim3(1), dim3(1024), 0, stream, K, n, n, 0, info);
        hipLaunchKernelGGL(k_gp_potrs, dim3(1), dim3(1024), 0, stream, K, n, m, alpha);
    }
    hipError_t e = hipMemcpyAsync(info_host, info, sizeof(int), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    hipFree(info);
    if (rc != 0) return fail(GPF_ERR_SOLVER, "rocBLAS / rocSOLVER returned status " + std::to_string(rc) + " in the Cholesky factorisation");
    if (e != hipSuccess) return fail(GPF_ERR_HIP, hipGetErrorString(e));
    hipLaunchKernelGGL(k_gp_clean_lower, dim3((n + 127) / 128, n), dim3(128), 0, stream, K, n);
    HIP_TRY(hipGetLastError());
    return GPF_OK;
}

extern "C" int gpf_gp_fit(int device, int n, int d, int m, const double* Xn, const double* Yn, double amp,
                          const double* inv_scale, double sigma, double* L_out, double* alpha_out, double* logdet) {
    if (!Xn || !Yn || !inv_scale) return fail(GPF_ERR_INVALID, "gpf_gp_fit: null argument");
    if (n < 1 || d < 1 || d > GP_MAX_D || m < 1 || m > 2) return fail(GPF_ERR_INVALID, "gpf_gp_fit: need n>=1, 1<=d<=4, 1<=m<=2");
    if (gpf_device_count() == 0) return fail(GPF_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU path)");
    HIP_TRY(hipSetDevice(device));
    DBG("gpf_gp_fit: dlopen rocBLAS/rocSOLVER");
    RocLibs& R = roclibs();
    if (!R.ok) return fail(GPF_ERR_SOLVER, R.err);
    DBG("gpf_gp_fit: libraries ready");
    std::vector<double> Z((size_t)n * d), Ycm((size_t)n * m);
    for (int i = 0; i < n; ++i) {
        for (int k = 0; k < d; ++k) Z[(size_t)i * d + k] = Xn[(size_t)i * d + k] * inv_scale[k];
        for (int o = 0; o < m; ++o) Ycm[(size_t)o * n + i] = Yn[(size_t)i * m + o];
    }
    double *dZ = nullptr, *dK = nullptr, *dA = nullptr, *dld = nullptr;
    void* blas = nullptr;
    int rc = GPF_OK, info = 0;
    auto done = [&](int code) {
        if (dZ) hipFree(dZ);
        if (dK) hipFree(dK);
        if (dA) hipFree(dA);
        if (dld) hipFree(dld);
        if (blas) R.destroy(blas);
        return code;
    };
    if (hipMalloc(&dZ, Z.size() * 8) != hipSuccess || hipMalloc(&dK, (size_t)n * n * 8) != hipSuccess ||
        hipMalloc(&dA, Ycm.size() * 8) != hipSuccess || hipMalloc(&dld, 8) != hipSuccess)
        return done(fail(GPF_ERR_HIP, "gpf_gp_fit: hipMalloc failed"));
    hipMemcpy(dZ, Z.data(), Z.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dA, Ycm.data(), Ycm.size() * 8, hipMemcpyHostToDevice);
    DBG("gpf_gp_fit: rocblas_create_handle");
    if (R.create(&blas) != 0) return done(fail(GPF_ERR_SOLVER, "rocblas_create_handle failed"));
    DBG("gpf_gp_fit: handle created");
    rc = gp_factorize(blas, nullptr, dZ, n, d, m, amp, sigma, dK, dA, &info);
    if (rc != GPF_OK) return done(rc);
    if (info != 0) return done(fail(GPF_ERR_SOLVER, "gpf_gp_fit: kernel matrix not positive definite (dpotrf info = " + std::to_string(info) + ")"));
    hipLaunchKernelGGL(k_gp_logdet, dim3(1), dim3(1), 0, 0, dK, n, dld);
    if (logdet) hipMemcpy(logdet, dld, 8, hipMemcpyDeviceToHost);
    if (alpha_out) {
        hipMemcpy(Ycm.data(), dA, Ycm.size() * 8, hipMemcpyDeviceToHost);
        for (int i = 0; i < n; ++i)
            for (int o = 0; o < m; ++o) alpha_out[(size_t)i * m + o] = Ycm[(size_t)o * n + i];
    }
    if (L_out) {
        std::vector<double> Lc((size_t)n * n);
        hipMemcpy(Lc.data(), dK, Lc.size() * 8, hipMemcpyDeviceToHost);
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) L_out[(size_t)i * n + j] = Lc[(size_t)j * n + i];      // row-major out
    }
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return done(fail(GPF_ERR_HIP, hipGetErrorString(e)));
    return done(GPF_OK);
}

extern "C" int gpf_gp_clear_model(gpf_handle* h, int which) {
    if (!h || which < 0 || which > 2) return fail(GPF_ERR_INVALID, "gpf_gp_clear_model: bad argument");
    auto& g = h->gp[which];
    HIP_TRY(hipSetDevice(h->cfg.device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (g.Z) hipFree(g.Z);
    if (g.alpha) hipFree(g.alpha);
    if (g.L) hipFree(g.L);
    if (g.Linv) hipFree(g.Linv);
    g.Z = g.alpha = g.L = g.Linv = nullptr;
    g.cap = 0;
    g.set = false;
    return GPF_OK;
}

extern "C" int gpf_gp_set_model(gpf_handle* h, int which, int n, int d, int m, const int32_t* dims, const double* x_scale,
                                const double* Xn, const double* Yn, double amp, const double* inv_scale, double sigma,
                                double yscale) {
    if (!h || !dims || !x_scale || !Xn || !Yn || !inv_scale) return fail(GPF_ERR_INVALID, "gpf_gp_set_model: null argument");
    if (which < 0 || which > 2) return fail(GPF_ERR_INVALID, "gpf_gp_set_model: which must be 0 (press), 1 (shear xz), 2 (shear yz)");
    if (n < 1 || d < 1 || d > GP_MAX_D) return fail(GPF_ERR_INVALID, "gpf_gp_set_model: need n >= 1 and 1 <= d <= 4");
    if (m != (which == 0 ? 1 : 2)) return fail(GPF_ERR_INVALID, "gpf_gp_set_model: pressure has 1 output, wall shear 2 (lower, upper)");
    for (int k = 0; k < d; ++k)
        if (dims[k] < 0 || dims[k] > 6) return fail(GPF_ERR_INVALID, "gpf_gp_set_model: feature index out of range 0..6");
    auto& g = h->gp[which];
    // active learning refits after every added point: keep the buffers while they are large enough
    if (n > g.cap) {
        GPF_TRY(gpf_gp_clear_model(h, which));
        const size_t cap = (size_t)((n + 127) / 128) * 128;
        HIP_TRY(hipMalloc(&g.Z, cap * GP_MAX_D * 8));
        HIP_TRY(hipMalloc(&g.alpha, cap * 2 * 8));
        HIP_TRY(hipMalloc(&g.L, cap * cap * 8));
        HIP_TRY(hipMalloc(&g.Linv, cap * cap * 8));
        g.cap = (int)cap;
    } else {
        HIP_TRY(hipStreamSynchronize(h->stream));      // nothing may still read the old model
    }
    g.set = false;
    void* blas = nullptr;
    GPF_TRY(gp_blas(h, &blas));
    std::vector<double> Z((size_t)n * d), Ycm((size_t)n * m);
    for (int i = 0; i < n; ++i) {
        for (int k = 0; k < d; ++k) Z[(size_t)i * d + k] = Xn[(size_t)i * d + k] * inv_scale[k];
        for (int o = 0; o < m; ++o) Ycm[(size_t)o * n + i] = Yn[(size_t)i * m + o];
    }
    HIP_TRY(hipMemcpyAsync(g.Z, Z.data(), Z.size() * 8, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(g.alpha, Ycm.data(), Ycm.size() * 8, hipMemcpyHostToDevice, h->stream));
    int info = 0;
    GPF_TRY(gp_factorize(blas, h->stream, g.Z, n, d, m, amp, sigma, g.L, g.alpha, &info));
    if (info != 0) {
        gpf_gp_clear_model(h, which);
        return fail(GPF_ERR_SOLVER, "gpf_gp_set_model: kernel matrix not positive definite (dpotrf info = " + std::to_string(info) + ")");
    }
    GpModelDev& D = g.dev;
    D.n = n; D.d = d; D.m = m; D.amp = amp; D.yscale = yscale;
    for (int k = 0; k < GP_MAX_D; ++k) { D.dims[k] = k < d ? dims[k] : 0; D.fscale[k] = k < d ? inv_scale[k] / x_scale[k] : 0.0; }
    D.Z = g.Z; D.alpha = g.alpha; D.L = g.L;
    g.xscale0 = x_scale[0];
    // L^-1 once per fit (n^3/3 flops): the predictive variance then is a dense product V = L^-1 Ks on the matrix cores
    // (rocBLAS dgemm) instead of a triangular solve per tile, 3-4x faster for a few hundred training points.
    // cond(L) = sqrt(cond(K)), so the explicit inverse costs no accuracy that the factorisation had.
    if (roclibs().gemm && !(getenv("GPF_GP_VARIANCE") && std::string(getenv("GPF_GP_VARIANCE")) == "trsm")) {
        RocLibs& R = roclibs();
        const double one = 1.0;
        hipLaunchKernelGGL(k_gp_identity, dim3((n + 127) / 128, n), dim3(128), 0, h->stream, g.Linv, n);
        hipLaunchKernelGGL(k_gp_clean_lower, dim3((n + 127) / 128, n), dim3(128), 0, h->stream, g.L, n);
        HIP_TRY(hipGetLastError());
        if (R.trsm(blas, ROC_SIDE_LEFT, ROC_FILL_LOWER, ROC_OP_NONE, ROC_DIAG_NON_UNIT, n, n, &one, g.L, n, g.Linv, n) != 0)
            return fail(GPF_ERR_SOLVER, "rocblas_dtrsm (inverse of the Cholesky factor) failed");
        hipLaunchKernelGGL(k_gp_clean_lower, dim3((n + 127) / 128, n), dim3(128), 0, h->stream, g.Linv, n);
        HIP_TRY(hipGetLastError());
        g.has_linv = true;
    } else {
        g.has_linv = false;
    }
    g.set = true;
    if (!h->gpvar) {
        HIP_TRY(hipMalloc(&h->gpvar, (size_t)3 * h->L.plane * 8));
        HIP_TRY(hipMemsetAsync(h->gpvar, 0, (size_t)3 * h->L.plane * 8, h->stream));
    }
    return GPF_OK;
}

static int gp_scratch(gpf_handle* h, int nblocks) {
    if (h->gpscratch_n >= nblocks) return GPF_OK;
    if (h->gpscratch) HIP_TRY(hipFree(h->gpscratch));
    h->gpscratch = nullptr;
    HIP_TRY(hipMalloc(&h->gpscratch, ((size_t)nblocks + 8) * 8));
    h->gpscratch_n = nblocks;
    return GPF_OK;
}

static GpFieldArgs gp_field_args(gpf_handle* h, int which, const double* q) {
    GpFieldArgs a;
    a.q = q; a.topo = h->topo; a.Ls = h->Ls; a.L = h->L;
    FieldPtrs F = field_ptrs(h);
    a.out0 = a.out1 = nullptr;
    if (which == 0) a.out0 = F.p;
    else {
        const int oi = which == 1 ? 4 : 3;          // Voigt xz / yz (stress.py:91)
        a.out0 = F.lower + (size_t)oi * h->L.plane;
        a.out1 = F.upper + (size_t)oi * h->L.plane;
    }
    a.blockmax = nullptr;
    return a;
}

#define GP_DISPATCH_D(d, ...)                                                                           \
    switch (d) {                                                                                        \
    case 1: { constexpr int D_ = 1; __VA_ARGS__; } break;                                               \
    case 2: { constexpr int D_ = 2; __VA_ARGS__; } break;                                               \
    case 3: { constexpr int D_ = 3; __VA_ARGS__; } break;                                               \
    default: { constexpr int D_ = 4; __VA_ARGS__; } break;                                              \
    }

// posterior mean of model `which` on field q into the derived-field planes; with_grad (pressure only):
// also *c2_out (device) = max_cells d mean/d rho * Yscale / X_scale[0]
static int gp_launch_mean(gpf_handle* h, int which, const double* q, bool with_grad, double* c2_out) {
    auto& g = h->gp[which];
    GPF_TRY(ensure_fields(h));
    const Layout& L = h->L;
    const long long ncell = (long long)(L.Nx + 2) * (L.Ny + 2);
    const int nb = (int)((ncell + 255) / 256);
    GpFieldArgs a = gp_field_args(h, which, q);
    if (with_grad) {
        GPF_TRY(gp_scratch(h, nb));
        a.blockmax = h->gpscratch;
        a.out0 = a.out1 = nullptr;
    }
    GP_DISPATCH_D(g.dev.d, {
        if (with_grad) hipLaunchKernelGGL((k_gp_mean<D_, 1, true>), dim3(nb), dim3(256), 0, h->stream, g.dev, a);
        else if (g.dev.m == 1) hipLaunchKernelGGL((k_gp_mean<D_, 1, false>), dim3(nb), dim3(256), 0, h->stream, g.dev, a);
        else hipLaunchKernelGGL((k_gp_mean<D_, 2, false>), dim3(nb), dim3(256), 0, h->stream, g.dev, a);
    });
    HIP_TRY(hipGetLastError());
    if (with_grad) {
        // dmean/dx_0 in normalised units carries s_0 = inv_scale_0: fscale_0 * X_scale_0; then * Yscale / X_scale_0
        const double scale = g.dev.fscale[0] * g.dev.yscale;
        hipLaunchKernelGGL(k_gp_maxreduce, dim3(1), dim3(256), 0, h->stream, h->gpscratch, nb, scale, c2_out);
        HIP_TRY(hipGetLastError());
    }
    return GPF_OK;
}

extern "C" int gpf_gp_variance(gpf_handle* h, int which, int on_open_step, double* max_var) {
    if (!h || which < 0 || which > 2) return fail(GPF_ERR_INVALID, "gpf_gp_variance: bad argument");
    auto& g = h->gp[which];
    if (!g.set) return fail(GPF_ERR_STATE, "gpf_gp_variance: model not set");
    if (on_open_step && !h->step_open) return fail(GPF_ERR_STATE, "gpf_gp_variance: no open step");
    HIP_TRY(hipSetDevice(h->cfg.device));
    void* blas = nullptr;
    GPF_TRY(gp_blas(h, &blas));
    RocLibs& R = roclibs();
    const Layout& L = h->L;
    int par = 0;
    GPF_TRY(current_parity(h, &par));
    const double* q = on_open_step ? h->q[par ^ 1] : h->q[par];
    const long long ncell = (long long)(L.Nx + 2) * (L.Ny + 2);
    const int n = g.dev.n;
    const long long tile = std::max<long long>(256, std::min<long long>(ncell, (64ll << 20) / (8ll * n)));   // <= 64 MiB of Ks
    const bool use_gemm = g.has_linv;
    if (h->gptile_doubles < (size_t)(tile * n) * 2) {        // Ks tile + the product tile
        if (h->gptile) HIP_TRY(hipFree(h->gptile));
        h->gptile = nullptr;
        HIP_TRY(hipMalloc(&h->gptile, (size_t)(tile * n) * 2 * 8));
        h->gptile_doubles = (size_t)(tile * n) * 2;
    }
    double* vtile = h->gptile + (size_t)(tile * n);
    const int nb_total = (int)((ncell + GP_VAR_COLS - 1) / GP_VAR_COLS) + (int)((ncell + tile - 1) / tile) + 8;
    GPF_TRY(gp_scratch(h, nb_total));
    GpFieldArgs a = gp_field_args(h, which, q);
    double* var_plane = h->gpvar + (size_t)which * L.plane;
    const double one = 1.0;
    int nbm = 0;
    for (long long c0 = 0; c0 < ncell; c0 += tile) {
        const int ncols = (int)std::min<long long>(tile, ncell - c0);
        GP_DISPATCH_D(g.dev.d, {
            hipLaunchKernelGGL((k_gp_ks_tile<D_>), dim3((n + 255) / 256, ncols), dim3(256), 0, h->stream, g.dev, a, c0, ncols, h->gptile);
        });
        HIP_TRY(hipGetLastError());
        const double* v = h->gptile;
        if (use_gemm) {     // V = L^-1 Ks
            // L^-1 is lower triangular: the upper row block needs only the first half of the columns -> 3/4 of the flops.
            // (Four row blocks would need 10/16, but 128-row products fill only half of the CUs: measured 56 ms per pass
            // against 40 ms with two blocks and 46 ms with one.)
            const double zero = 0.0;
            const int nblk = n >= 512 ? 2 : 1;
            const int rows = ((n + nblk - 1) / nblk + 15) / 16 * 16;
            for (int r0 = 0; r0 < n; r0 += rows) {
                const int mr = std::min(rows, n - r0), kk = std::min(n, r0 + mr);
                if (R.gemm(blas, ROC_OP_NONE, ROC_OP_NONE, mr, ncols, kk, &one, g.Linv + r0, n, h->gptile, n, &zero, vtile + r0, n) != 0)
                    return fail(GPF_ERR_SOLVER, "rocblas_dgemm failed");
            }
            v = vtile;
        } else if (R.trsm(blas, ROC_SIDE_LEFT, ROC_FILL_LOWER, ROC_OP_NONE, ROC_DIAG_NON_UNIT, n, ncols, &one, g.L, n, h->gptile, n) != 0) {
            return fail(GPF_ERR_SOLVER, "rocblas_dtrsm failed");
        }
        const int nb = (ncols + GP_VAR_COLS - 1) / GP_VAR_COLS;
        hipLaunchKernelGGL(k_gp_var_tile, dim3(nb), dim3(1024), 0, h->stream, v, n, ncols, g.dev.amp,
                           g.dev.yscale * g.dev.yscale, c0, L, var_plane, h->gpscratch + nbm);
        HIP_TRY(hipGetLastError());
        nbm += nb;
    }
    double* res = h->gpscratch + h->gpscratch_n;
    hipLaunchKernelGGL(k_gp_maxreduce, dim3(1), dim3(256), 0, h->stream, h->gpscratch, nbm, 1.0, res);
    HIP_TRY(hipGetLastError());
    double mv = 0.0;
    HIP_TRY(hipMemcpyAsync(&mv, res, 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (max_var) *max_var = mv;
    return GPF_OK;
}

// ---------------------------------------------------------------------------------------------
// the unfused step in pieces, for closures that need the host between stages (GP + active learning)
// ---------------------------------------------------------------------------------------------
extern "C" int gpf_open_step(gpf_handle* h) {
    if (!h) return fail(GPF_ERR_INVALID, "null handle");
    if (!h->pre_run_done) return fail(GPF_ERR_STATE, "gpf_open_step: call gpf_pre_run first");
    if (h->step_open) return fail(GPF_ERR_STATE, "gpf_open_step: a step is already open");
    HIP_TRY(hipSetDevice(h->cfg.device));
    const Layout& L = h->L;
    GPF_TRY(ensure_fields(h));
    if (!h->work) HIP_TRY(hipMalloc(&h->work, (size_t)9 * L.plane * sizeof(double)));
    StepState s;
    GPF_TRY(read_state(h, s));
    if (s.invalid) return fail(GPF_ERR_STATE, "gpf_open_step: the state is invalid (rolled back); nothing to advance");
    h->open_parity = s.parity;
    hipLaunchKernelGGL(k_copy3, dim3(blocks_for(3 * L.plane)), dim3(256), 0, h->stream, h->q[s.parity], h->q[s.parity ^ 1], 3 * L.plane);
    HIP_TRY(hipGetLastError());
    h->step_open = true;
    h->g1_ready = false;
    h->host_step = s.step; h->next_step = s.step;
    return GPF_OK;
}

extern "C" int gpf_stage_closures(gpf_handle* h) {
    if (!h || !h->step_open) return fail(GPF_ERR_STATE, "gpf_stage_closures: no open step");
    HIP_TRY(hipSetDevice(h->cfg.device));
    return launch_fields(h, h->q[h->open_parity ^ 1]);
}

extern "C" int gpf_stage_advance(gpf_handle* h, int stage) {
    if (!h || !h->step_open) return fail(GPF_ERR_STATE, "gpf_stage_advance: no open step");
    if (stage != 0 && stage != 1) return fail(GPF_ERR_INVALID, "gpf_stage_advance: stage must be 0 (predictor) or 1 (corrector)");
    HIP_TRY(hipSetDevice(h->cfg.device));
    const Layout& L = h->L;
    double* q = h->q[h->open_parity ^ 1];
    const long long n = (long long)(L.Nx + 2) * (L.Ny + 2);
    const int nb = blocks_for(n);
    double *fx = h->work, *fy = h->work + 3 * L.plane, *src = h->work + 6 * L.plane;
    FieldPtrs F = field_ptrs(h);
    const int mc = h->cfg.mc_order;
    const int sw = mc == 0 ? ((h->host_step % 2 == 0) ? 1 : -1) : mc;
    const int first = ((sw + 1) / 2) ? 1 : -1;
    const int dir = stage == 0 ? first : -first;
    hipLaunchKernelGGL(k_fluxdiff, dim3(nb), dim3(256), 0, h->stream, q, F.p, F.tau, dir, fx, fy, L);
    hipLaunchKernelGGL(k_source, dim3(nb), dim3(256), 0, h->stream, q, h->topo, F.tau, F.lower, F.upper, src, L);
    hipLaunchKernelGGL(k_axpy, dim3(nb), dim3(256), 0, h->stream, q, fx, fy, src, h->st, h->cfg.dx, h->cfg.dy, L);
    hipLaunchKernelGGL(k_bc_x, dim3((L.Ny + 2 + 255) / 256), dim3(256), 0, h->stream, q, L, h->E);
    hipLaunchKernelGGL(k_bc_y, dim3((L.Nx + 2 + 255) / 256), dim3(256), 0, h->stream, q, L, h->E);
    HIP_TRY(hipGetLastError());
    return GPF_OK;
}

extern "C" int gpf_close_step(gpf_handle* h, gpf_scalars_t* out) {
    if (!h || !h->step_open) return fail(GPF_ERR_STATE, "gpf_close_step: no open step");
    HIP_TRY(hipSetDevice(h->cfg.device));
    const Layout& L = h->L;
    double* q = h->q[h->open_parity ^ 1];
    const double* q0 = h->q[h->open_parity];
    const long long n = (long long)(L.Nx + 2) * (L.Ny + 2);
    hipLaunchKernelGGL(k_average, dim3(blocks_for(n)), dim3(256), 0, h->stream, q, q0, L);
    ScalarPartial* pre = h->spart + h->nspart;
    ScalarPartial* post = h->spart + h->nspart + 1;
    GPF_TRY(launch_scalars(h, q, pre, false));          // validity of the averaged field only (problem.py:565)
    hipLaunchKernelGGL(k_bc_x, dim3((L.Ny + 2 + 255) / 256), dim3(256), 0, h->stream, q, L, h->E);
    hipLaunchKernelGGL(k_bc_y, dim3((L.Nx + 2 + 255) / 256), dim3(256), 0, h->stream, q, L, h->E);
    GPF_TRY(launch_scalars(h, q, post));                // scalars after the ghost update (problem.py:576-578)
    hipLaunchKernelGGL(k_commit_unfused, dim3(1), dim3(1), 0, h->stream, h->st, pre, post);
    HIP_TRY(hipGetLastError());
    h->step_open = false;
    StepState s;
    GPF_TRY(read_state(h, s));
    h->host_step = s.step; h->next_step = s.step;
    if (out) fill_scalars(s, nullptr, 0.0, out);
    return GPF_OK;
}

// ---------------------------------------------------------------------------------------------
// stage-wise step of a slab (GP closures / shear thinning across several GPUs): the rows a neighbour needs
// travel after EACH stage, the scalars once per step
// ---------------------------------------------------------------------------------------------
extern "C" int gpf_stage_message(gpf_handle* h) {
    if (!h || !h->step_open) return fail(GPF_ERR_STATE, "gpf_stage_message: no open step");
    HIP_TRY(hipSetDevice(h->cfg.device));
    GPF_TRY(ensure_halo(h));
    HaloArgs a = halo_args(h, 0, nullptr, -1, -1);
    a.work_parity = h->open_parity ^ 1;
    hipLaunchKernelGGL(k_halo_pack, dim3((h->L.pitch + 255) / 256), dim3(256), 0, h->stream, a);
    HIP_TRY(hipGetLastError());
    return GPF_OK;
}

extern "C" int gpf_stage_absorb(gpf_handle* h, const void* gathered, int nranks, int rank_lo, int rank_hi) {
    if (!h || !gathered || nranks < 1) return fail(GPF_ERR_INVALID, "gpf_stage_absorb: bad argument");
    if (!h->step_open) return fail(GPF_ERR_STATE, "gpf_stage_absorb: no open step");
    if (rank_lo >= nranks || rank_hi >= nranks) return fail(GPF_ERR_INVALID, "gpf_stage_absorb: neighbour rank out of range");
    HIP_TRY(hipSetDevice(h->cfg.device));
    HaloArgs a = halo_args(h, 0, (const double*)gathered, rank_lo, rank_hi);
    a.work_parity = h->open_parity ^ 1;
    hipLaunchKernelGGL(k_halo_unpack, dim3((h->L.pitch + 255) / 256), dim3(256), 0, h->stream, a);
    HIP_TRY(hipGetLastError());
    return GPF_OK;
}

// average + validity + ghost rules + this slab's share of the scalars -> record in the message
extern "C" int gpf_close_step_local(gpf_handle* h) {
    if (!h || !h->step_open) return fail(GPF_ERR_STATE, "gpf_close_step_local: no open step");
    HIP_TRY(hipSetDevice(h->cfg.device));
    GPF_TRY(ensure_halo(h));
    const Layout& L = h->L;
    double* q = h->q[h->open_parity ^ 1];
    const double* q0 = h->q[h->open_parity];
    const long long n = (long long)(L.Nx + 2) * (L.Ny + 2);
    hipLaunchKernelGGL(k_average, dim3(blocks_for(n)), dim3(256), 0, h->stream, q, q0, L);
    ScalarPartial* pre = h->spart + h->nspart;
    ScalarPartial* post = h->spart + h->nspart + 1;
    GPF_TRY(launch_scalars(h, q, pre, false));
    hipLaunchKernelGGL(k_bc_x, dim3((L.Ny + 2 + 255) / 256), dim3(256), 0, h->stream, q, L, h->E);
    hipLaunchKernelGGL(k_bc_y, dim3((L.Nx + 2 + 255) / 256), dim3(256), 0, h->stream, q, L, h->E);
    GPF_TRY(launch_scalars(h, q, post));
    hipLaunchKernelGGL(k_record_unfused, dim3(1), dim3(1), 0, h->stream, pre, post, h->halo + (size_t)6 * L.pitch);
    HIP_TRY(hipGetLastError());
    return GPF_OK;
}

extern "C" int gpf_close_step_commit(gpf_handle* h, const void* gathered, int nranks, gpf_scalars_t* out) {
    if (!h || !gathered || nranks < 1) return fail(GPF_ERR_INVALID, "gpf_close_step_commit: bad argument");
    if (!h->step_open) return fail(GPF_ERR_STATE, "gpf_close_step_commit: no open step");
    HIP_TRY(hipSetDevice(h->cfg.device));
    hipLaunchKernelGGL(k_commit_gathered, dim3(1), dim3(1), 0, h->stream, h->st, (const double*)gathered,
                       (long long)halo_len(h), (long long)6 * h->L.pitch, nranks, (LogEntry*)nullptr, 0ll, 0ll, 0);
    HIP_TRY(hipGetLastError());
    h->step_open = false;
    StepState s;
    GPF_TRY(read_state(h, s));
    h->host_step = s.step; h->next_step = s.step;
    if (out) fill_scalars(s, nullptr, 0.0, out);
    return GPF_OK;
}

