/*
 * gapflow_hip.h -- C ABI of libgapflow_hip.so, the MI355X (gfx950) implementation of
 * GaPFlow's explicit time-integration hot path.
 *
 * The reference (hannes-holey/GaPFlow) has no FFI for this path: the boundary is its
 * Python API (GaPFlow/problem.py, GaPFlow/integrate.py).  Every entry point below names
 * the reference code it replaces (paths relative to the reference root).  The host side
 * (gapflow_amd/problem.py, gapflow_amd/integrate.py) binds these with ctypes; see
 * INTEGRATION.md for the stub a reference maintainer would add.
 *
 * Conventions
 *   - plain C types only; every function returns 0 on success or a negative gpf_status;
 *     nothing throws across the boundary; gpf_last_error() gives the message of the
 *     last failure on the calling thread.
 *   - host arrays are BORROWED for the duration of the call and use the reference's
 *     layout: C-contiguous double[ncomp][Nx+2][Ny+2], one ghost cell per side
 *     (problem.py:122-141).  The library owns all device memory.
 *   - a handle is not thread-safe; all work of a handle is issued on one HIP stream.
 *   - all arithmetic is IEEE binary64 ("f64").
 */
#ifndef GAPFLOW_HIP_H
#define GAPFLOW_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gpf_handle gpf_handle;

typedef enum {
    GPF_OK = 0,
    GPF_ERR_INVALID = -1,     /* bad argument / unsupported configuration */
    GPF_ERR_HIP = -2,         /* HIP runtime error (message has hipGetErrorString) */
    GPF_ERR_NO_DEVICE = -3,   /* no gfx950 device visible */
    GPF_ERR_SOLVER = -4,      /* rocSOLVER / rocBLAS failure, or matrix not positive definite */
    GPF_ERR_STATE = -5        /* call sequence error (e.g. step before upload) */
} gpf_status;

/* Equation of state ids (GaPFlow/models/pressure.py:51-71) */
enum { GPF_EOS_DH = 0, GPF_EOS_PL = 1, GPF_EOS_VDW = 2, GPF_EOS_MT = 3,
       GPF_EOS_CUBIC = 4, GPF_EOS_BWR = 5, GPF_EOS_BAYADA = 6 };

/* Ghost-cell rule per component (GaPFlow/problem.py:676-768) */
enum { GPF_BC_PERIODIC = 0, GPF_BC_DIRICHLET = 1, GPF_BC_NEUMANN = 2 };

/* Piezo-viscosity laws (GaPFlow/models/viscosity.py:34-66) */
enum { GPF_PIEZO_NONE = 0, GPF_PIEZO_BARUS = 1, GPF_PIEZO_ROELANDS = 2,
       GPF_PIEZO_DUKLER = 3, GPF_PIEZO_MCADAMS = 4 };

/* Shear-thinning laws (GaPFlow/models/viscosity.py:69-96) */
enum { GPF_THINNING_NONE = 0, GPF_THINNING_EYRING = 1, GPF_THINNING_CARREAU = 2 };

/* Field ids for gpf_upload / gpf_download */
enum {
    GPF_FIELD_Q = 0,          /* 3 comps: rho, jx, jy                 (problem.py:128)          */
    GPF_FIELD_TOPO = 1,       /* 3 comps: h, dh/dx, dh/dy             (topography.py:252-254)   */
    GPF_FIELD_EXTRA = 2,      /* 1 comp : slip length Ls              (problem.py:132-135)      */
    GPF_FIELD_PRESSURE = 3,   /* 1 comp : derived, stress.py:600-622                            */
    GPF_FIELD_TAU_AVG = 4,    /* 3 comps: xx, yy, xy, stress.py:427-459                         */
    GPF_FIELD_WALL_LOWER = 5, /* 6 comps Voigt = wall_stress_xz.lower + wall_stress_yz.lower    */
    GPF_FIELD_WALL_UPPER = 6, /* 6 comps Voigt (problem.py:554-555)                             */
    GPF_FIELD_PRESSURE_VAR = 7,   /* GP predictive variances (stress.py:97, 499), 1 comp each         */
    GPF_FIELD_WALL_XZ_VAR = 8,
    GPF_FIELD_WALL_YZ_VAR = 9,
    GPF_FIELD_DEFORMATION = 10  /* 1: elastic displacement added to the undeformed gap height (download only) */
};

/*
 * Static description of one problem (or of one x-slab of it).  Mirrors the sanitised
 * dicts of GaPFlow/io.py:128-394 after the host has resolved the boundary-condition
 * side quirks of problem.py:736-754 into "rule and target of each ghost edge".
 */
typedef struct {
    int32_t Nx, Ny;            /* interior cells of this slab; arrays are (Nx+2) x (Ny+2)          */
    double  dx, dy;
    double  U, V;              /* wall velocities, geometry dict                                    */
    double  eta, zeta;         /* shear / bulk viscosity, properties dict                           */
    int32_t eos;               /* GPF_EOS_*                                                         */
    double  eos_par[8];        /* DH: rho0,P0,C1,C2 | PL: rho0,P0,alpha | vdW: M,T,a,b |
                                  MT: rho0,P0,K,n | cubic: a,b,c,d | BWR: T,gamma |
                                  Bayada: rho_l,rho_v,c_l,c_v                                        */
    int32_t piezo;             /* GPF_PIEZO_*                                                       */
    double  piezo_par[4];      /* Barus: aB | Roelands: mu_inf,p_ref,z | Dukler/McAdams: eta_v,rho_l,rho_v */
    /* ghost edges: [0]=ix=0, [1]=ix=Nx+1, [2]=iy=0, [3]=iy=Ny+1; rule per component            */
    int32_t bc_rule[4][3];
    double  bc_value[4];       /* Dirichlet target of that edge (one scalar per edge)              */
    int32_t halo_lo, halo_hi;  /* kind of the rows ix=0 / ix=Nx+1 of this slab:
                                  0 physical ghost row (rule above applied locally),
                                  1 slab halo = copy of a neighbour's interior row (caller's exchange),
                                  2 periodic seam = the domain's periodic ghost row, filled by the
                                    caller's ring exchange instead of a local copy                    */
    /* time stepping (io.py:381-394, problem.py:412-443) */
    int32_t adaptive;
    double  CFL, dt_fixed, tol;
    int64_t max_it;
    int32_t mc_order;          /* +1, -1, or 0 = alternate by step parity (problem.py:521-522)     */
    int32_t device;            /* HIP device ordinal                                                */
    int32_t thinning;          /* GPF_THINNING_*; only the stage-wise step supports it (needs grad p)  */
    double  thinning_par[4];   /* Eyring: tauE | Carreau: mu_inf, lam, a, N                            */
} gpf_config;

/* Per-step scalars (problem.py:334-362, 571-586).  All sums/maxima run over the whole
 * array including ghost cells, as in the reference. */
typedef struct {
    int64_t step;
    double  simtime, dt;       /* dt = step size the NEXT step will use                             */
    double  ekin, ekin_old;    /* Ekin = sum (jx^2+jy^2)/rho/2                                      */
    double  residual;          /* |Ekin-Ekin_old|/Ekin_old/cfl                                      */
    double  v_max, v_sound;    /* max sqrt((jx^2+jy^2)/rho) (sic), max c(rho)                       */
    double  mass;              /* sum rho*h*dx*dy; filled by gpf_scalars only                       */
    int32_t invalid;           /* 1: NaN, 2: negative density (problem.py:319-332)                 */
    int32_t converged;         /* all of the last <=5 residuals < tol (problem.py:359-362)          */
} gpf_scalars_t;

/* ---- lifetime -------------------------------------------------------------------------- */
int gpf_create(const gpf_config* cfg, gpf_handle** out);      /* Problem.__init__, problem.py:77-151 */
int gpf_destroy(gpf_handle* h);
int gpf_set_stream(gpf_handle* h, void* hip_stream);          /* default: the null stream           */
const char* gpf_last_error(void);
int gpf_device_count(void);

/* ---- fields ---------------------------------------------------------------------------- */
int gpf_upload(gpf_handle* h, int field, const double* host, size_t count);     /* count = ncomp*(Nx+2)*(Ny+2) */
int gpf_download(gpf_handle* h, int field, double* host, size_t count);
/* Pressure/WallStress/BulkStress.update without time stepping: refreshes the derived
 * fields from the current q (stress.py:289-362, 427-459, 600-622). */
int gpf_update_closures(gpf_handle* h);

/* ---- time stepping --------------------------------------------------------------------- */
/* _pre_run (problem.py:412-443): step=0, residual=1, dt = CFL*dt_crit or dt_fixed, Ekin_old from q. */
int gpf_pre_run(gpf_handle* h);
/* n calls of Problem.update() (problem.py:509-586) enqueued back-to-back with no host
 * round trip.  If honor_stop != 0 the steps become no-ops on the device once `converged`
 * holds or step == max_it (the `while` of run(), problem.py:401); an invalid state
 * (NaN / rho<0) always rolls back to the pre-step field and stops (problem.py:565-610).
 * log (may be NULL): receives one gpf_scalars_t per executed step, at most log_capacity. */
int gpf_step(gpf_handle* h, int64_t n, int honor_stop, gpf_scalars_t* log, int64_t log_capacity,
             int64_t* n_executed);
/* Measurement aid: n updates with a HIP event pair around each launch of the fused step kernel
 * (recorded on the handle's stream).  *kernel_ms = sum of the n kernel durations, *total_ms =
 * elapsed device time of the whole sequence (stage-1 ghost data, fused step, ghost fill with the commit). */
int gpf_step_timed(gpf_handle* h, int64_t n, double* kernel_ms, double* total_ms);
/* How the fused step of this handle is laid out on the chip (row chunks per strip = waves per SIMD, cache policy of the stores)
 * and how that was decided: grids of a million cells and more time the candidates once on their own data before the first step
 * (GPF_PLAN_TUNE=0: rule of thumb; GPF_CHUNKS / GPF_NT_STORES pin a choice).  Empty until the first step has been planned.
 * The choice changes the order in which the kinetic energy is summed (last bits of the residual), nothing else. */
const char* gpf_plan_note(gpf_handle* h);
/* The reference-ordered, unfused stage pipeline (closures -> flux -> source -> axpy -> ghost),
 * one kernel per reference function; same results as gpf_step(h,1,...).  Kept for
 * cross-checking and for _finalize (problem.py:588-610). */
int gpf_step_unfused(gpf_handle* h);
/* scalars of the current state (mass, kinetic_energy, v_max, v_sound, dt_crit inputs) */
int gpf_scalars(gpf_handle* h, gpf_scalars_t* out);
int gpf_set_ekin_old(gpf_handle* h, double value);            /* kinetic_energy_old setter, tests/test_wave_decay.py:102 */
int gpf_set_dt(gpf_handle* h, double dt);

/* ---- slab decomposition (one process per GPU, x-slabs) ------------------------------------ */
/* One message per slab and step feeds a SINGLE all-gather: *message (device, *count = 6*pitch + 8 doubles) =
 * [first interior row | last interior row | 8-double record] of the field the current step has produced, where a
 * row is 3 components x the padded row and the record is [sum Ekin, max v^2, max c^2 (NaN as +inf), invalid flags,
 * 0...].  The address is fixed for the life of the handle (usable as an RCCL / torch.distributed buffer).
 *
 * gpf_step_local : stage-1 ghost data + fused stencil + local ghost rules + the message.
 * (caller)       : all-gather the messages of all slabs, rank order, into `gathered` (device, nranks * count).
 * gpf_step_commit: scatter the neighbours' rows into rows ix=0 / ix=Nx+1 (rank_lo: the rank whose LAST row is my
 *                  row 0, rank_hi: the rank whose FIRST row is my row Nx+1, -1 = physical edge), reduce the records
 *                  in rank order and advance dt / residual / step exactly as gpf_step does.
 * Everything is enqueued on the handle's stream; neither call synchronises with the host. */
int gpf_slab_message(gpf_handle* h, void** message, size_t* count);
int gpf_step_local(gpf_handle* h, int honor_stop);
int gpf_step_commit(gpf_handle* h, int honor_stop, const void* gathered, int nranks, int rank_lo, int rank_hi);
/* Elastic deformation of the gap under the film pressure (Topography.update, topography.py:257-280; ElasticDeformation,
 * topography.py:327-437).  gpf_elastic_setup receives the half-space Green's function in Fourier space on a px x py
 * transform grid (px >= Nx+2, py >= Ny+2; doubled along non-periodic axes), as numpy's rfft2 lays it out:
 * greens_ri[px][py/2+1][2] (real, imaginary), count = 2 px (py/2+1); alpha = under-relaxation factor; force_scale =
 * (force per cell / pressure) / (cell area the Green's function is normalised to); relative != 0: pressure and
 * displacement are taken relative to cell [0, 0] (every case but the fully periodic one).  The handle's current
 * topography is kept as the undeformed one.  gpf_elastic_update convolves the pressure of the last closure evaluation,
 * under-relaxes, and rewrites h, dh/dx, dh/dy (np.gradient stencil) in place; GPF_FIELD_DEFORMATION downloads the
 * displacement.  Undivided problems only. */
int gpf_elastic_setup(gpf_handle* h, int px, int py, const double* greens_ri, size_t count, double alpha,
                      double force_scale, int relative);
int gpf_elastic_update(gpf_handle* h);

/* Peer-to-peer slab transport (GPUs of one node, one process each).  Instead of a collective library the step's own
 * kernels store the two boundary rows and the 64-byte record straight into the peers' mailboxes (device memory mapped
 * through HIP IPC, xGMI underneath) and the receiving kernel polls a sequence flag -- no host and no extra launches
 * between steps, so gpf_step_p2p(n) enqueues n complete steps at once.
 *   gpf_p2p_export   allocate this slab's mailbox, return its 64-byte IPC handle (exchange them with any transport)
 *   gpf_p2p_connect  map every rank's mailbox; ipc_handles = nranks x 64 bytes in rank order
 *   gpf_step_p2p     n steps; a peer that stays silent for 30 s marks the state invalid (gpf_state: invalid == 3) --
 *                    decided once per launch: every block reports, the last one to arrive stops the handle if ANY block
 *                    missed a flag (no commit, no sequence advance), so a flag that lands at the deadline cannot split
 *                    the blocks of one launch
 *   gpf_p2p_set_timeout   the bound on that wait, in seconds (tests use a fraction of a second)
 * rank_lo / rank_hi as in gpf_step_commit. */
int gpf_p2p_export(gpf_handle* h, void* ipc_handle, size_t handle_bytes);
int gpf_p2p_connect(gpf_handle* h, int rank, int nranks, const void* ipc_handles, int rank_lo, int rank_hi);
int gpf_step_p2p(gpf_handle* h, int64_t n, int honor_stop);
int gpf_p2p_set_timeout(gpf_handle* h, double seconds);
/* Stage-wise step of a slab (GP closures, shear thinning): after each gpf_stage_advance the rows a neighbour needs
 * are packed from the working field (gpf_stage_message -> the same message buffer), all-gathered by the caller and
 * scattered (gpf_stage_absorb); gpf_close_step_local averages, applies the local ghost rules and leaves this slab's
 * record in the message, gpf_close_step_commit reduces the gathered records and advances dt / residual / step. */
/* Shear thinning (stress.py:170-192: eta depends on np.gradient(p)) gives the step a two-row reach in x: the viscosity of a
 * slab's halo row needs the pressure one row further into the neighbour.  With cfg.thinning != 0 the stage messages carry
 * two more rows (density of the second and second-to-last owned row; message length 8*pitch + 8) and the library keeps the
 * neighbours' copies; gpf_upload_beyond seeds them for the initial state (side 0: beyond row 0, 1: beyond row Nx+1;
 * Ny+2 densities).  Only the stage-wise calls serve such a slab (gpf_step_local / gpf_step_p2p refuse it). */
int gpf_upload_beyond(gpf_handle* h, int side, const double* rho_row, size_t count);
int gpf_stage_message(gpf_handle* h);
int gpf_stage_absorb(gpf_handle* h, const void* gathered, int nranks, int rank_lo, int rank_hi);
int gpf_close_step_local(gpf_handle* h);
int gpf_close_step_commit(gpf_handle* h, const void* gathered, int nranks, gpf_scalars_t* out);
/* Read back the run state after a batch of split steps (synchronises). */
int gpf_state(gpf_handle* h, gpf_scalars_t* out);
/* Topography across a periodic seam (halo kind 2): the rows the far side's first interior cell and
 * its upwind neighbour see, needed to reproduce problem.py:683/690 exactly.
 * side 0: edge ix=0, host = topo+Ls of global rows (Nx, Nx+1); side 1: edge ix=Nx+1, rows (1, 0).
 * Layout double[2][4][Ny+2] (h, dh/dx, dh/dy, Ls per row). */
int gpf_set_seam_topo(gpf_handle* h, int side, const double* host, size_t count);

/* ---- stateless operators: GaPFlow/integrate.py ------------------------------------------ */
/* predictor_corrector(q,p,tau,direction) -> flux_x, flux_y   (integrate.py:38-77) */
int gpf_predictor_corrector(int nx, int ny, const double* q, const double* p, const double* tau,
                            int direction, double* flux_x, double* flux_y);
/* source(q,h,stress,stress_lower,stress_upper) -> out        (integrate.py:80-130); h = first 3 comps of the topography */
int gpf_source(int nx, int ny, const double* q, const double* h, const double* stress,
               const double* lower, const double* upper, double* out);

/* ---- stateless operators: GaPFlow/models as functions of arrays ----------------------------------- */
/* stress_bottom / stress_top / stress_avg (models/viscous.py:37, 281, 612) for n points in one call.  Arrays are
 * component-major: q, hh (h, dh/dx, dh/dy), dqx, dqy [3][n] (dqx, dqy may be NULL = zero gradients, the solver's
 * case); eta, Ls [n] (viscosity and slip length per point).  slip_both = 0: slip="top" (only the upper wall slips,
 * what the solver uses); 1: the other branch of viscous.py (both walls; Ls = 0 is no slip).  Outputs, any of which
 * may be NULL: bottom, top [6][n] in Voigt order xx, yy, zz, yz, xz, xy; avg [3][n] = xx, yy, xy. */
int gpf_viscous_stress(int64_t n, const double* q, const double* hh, const double* dqx, const double* dqy,
                       const double* eta, const double* Ls, double U, double V, double zeta, int slip_both,
                       double* bottom, double* top, double* avg);
/* eos_pressure (models/pressure.py:35-76) and eos_sound_velocity (models/sound.py) of n densities; eos / eos_par as
 * in gpf_config; either output may be NULL. */
int gpf_eos(int eos, const double* eos_par, int64_t n, const double* rho, double* pressure, double* sound);
/* models/viscosity.py for n points.  kind 0: piezoviscosity(a0 = pressure, or density for the mixture laws; mu0;
 * law = gpf_config.piezo, par = piezo_par) (viscosity.py:34-66); kind 1: shear_thinning_factor(a0 = shear rate; mu0;
 * law = gpf_config.thinning, par = thinning_par) = mu/mu0 (viscosity.py:69-96); kind 2: shear_rate_avg(a0 = dp/dx,
 * a1 = dp/dy, a2 = h; wall speeds u1, u2; viscosity mu0) (viscosity.py:110-141; law, par unused). */
int gpf_viscosity(int kind, int law, const double* par, double mu0, int64_t n, const double* a0, const double* a1,
                  const double* a2, double u1, double u2, double* out);

/* ---- the unfused step in pieces ------------------------------------------------------------- */
/* For closures that need the host between stages (GP surrogates with active learning, gp.py:435-506):
 * gpf_open_step copies q0; per stage gpf_stage_closures evaluates Pressure/WallStress/BulkStress on the
 * working field (fixed-form laws, then the mean of every GP model that is set), gpf_stage_advance applies
 * flux + source + ghost rules (problem.py:543-560); gpf_close_step averages, checks validity, updates
 * dt / residual (problem.py:563-586).  gpf_step_unfused is exactly open, 2 x (closures, advance), close. */
int gpf_open_step(gpf_handle* h);
int gpf_stage_closures(gpf_handle* h);
int gpf_stage_advance(gpf_handle* h, int stage);
int gpf_close_step(gpf_handle* h, gpf_scalars_t* out);

/* ---- GP surrogate closure: GaPFlow/models/gp.py, models/stress.py ------------------------------ */
/* Stateless fit: K = A (1 + sqrt3 r) exp(-sqrt3 r) + sigma^2 I with r = ||inv_scale o (x - x')||
 * (gp.py:598-603; tinygp GaussianProcess(kernel, X, diag=yerr^2)), Cholesky K = L L^T and alpha = K^-1 Y
 * on the device: rocSOLVER's dpotrf / dpotrs (the copy that lies beside the rocBLAS the process uses); with GPF_USE_ROCSOLVER=0,
 * or where there is no such copy, the library's own blocked Cholesky.  Xn [n][d] and Yn [n][m] row-major, already normalised.  Outputs (host,
 * each may be NULL): L [n][n] row-major lower factor, alpha [n][m], logdet = log det K. */
/* Which factorisation serves this process: "rocsolver_dpotrf", "in-library blocked Cholesky", or the reason neither is available. */
const char* gpf_gp_factorisation(void);
int gpf_gp_fit(int device, int n, int d, int m, const double* Xn, const double* Yn, double amp,
               const double* inv_scale, double sigma, double* L, double* alpha, double* logdet);
/* Attach / replace a surrogate of this problem: which = 0 pressure (m = 1; writes the pressure field),
 * 1 wall shear xz, 2 wall shear yz (m = 2: lower, upper wall; Voigt index 4 / 3, stress.py:91, 356-357).
 * dims[d]: feature index of each active dimension in [rho, jx, jy, h, dh/dx, dh/dy, extra]
 * (gp.py:223-232); x_scale[d]: the database normalisers of those features (db.py:264-266);
 * yscale: output scale (stress.py:230-242, 562-564).  The model is factorised on the device. */
int gpf_gp_set_model(gpf_handle* h, int which, int n, int d, int m, const int32_t* dims, const double* x_scale,
                     const double* Xn, const double* Yn, double amp, const double* inv_scale, double sigma,
                     double yscale);
int gpf_gp_clear_model(gpf_handle* h, int which);
/* The reference normalises test inputs and outputs with the database's CURRENT max-abs scales (properties Xtest / Yscale,
 * stress.py:195-242, 542-564) while the factorised model and its cached alpha are those of the last fit (gp.py:343-349):
 * after the database has grown through another model and before this model's next fit, the mean is
 * Ks(x / x_scale_now)^T alpha_fit * yscale_now and the variance alike; the GP sound speed re-solves alpha from
 * Y_raw / yscale_now (stress.py:588, 533-535), so there only x_scale changes.  This call hands the current scales to an
 * attached model without refitting it. */
int gpf_gp_set_scales(gpf_handle* h, int which, const double* x_scale, double yscale);
/* Predictive variance A - ||L^-1 k(X, x*)||^2 (gp.py:509-522) of model `which` at every cell, written to
 * the GPF_FIELD_*_VAR field (times yscale^2); *max_var = its maximum (the active-learning criterion,
 * gp.py:408).  on_open_step != 0: evaluate on the working field of an open step. */
int gpf_gp_variance(gpf_handle* h, int which, int on_open_step, double* max_var);
/* Posterior-mean passes over the grid since gpf_create: launched[which] per model, and *reused = evaluations of the
 * pressure surrogate that were served by the pass that closed the previous step (stress.py:533-537 evaluates the
 * sound speed -- the slope of the mean -- on the state the next step's first closure update, stress.py:522-531, sees
 * again: one kernel launch yields both).  Introspection for tests and bench.py; GPF_GP_NO_STATE_MEAN=1 at gpf_create
 * turns the reuse off. */
int gpf_gp_pass_counts(gpf_handle* h, int64_t launched[3], int64_t* reused);

/* Hyper-parameter training on the device (replaces what tinygp + jax.grad + jaxopt evaluate per optimiser step,
 * gp.py:290-335, 576-603): -log p(Y | X, theta) and its gradient for theta = [log_amp, log_scale_1..d] of the
 * Matern-3/2 ARD kernel with fixed observation noise sigma.  A session keeps the training set (Xn: n x d row-major,
 * normalised inputs; Yn: n x m row-major) and its work space on the device; the optimiser (host, SciPy BFGS as
 * the reference's jaxopt.ScipyMinimize) calls gpf_gp_nll_eval once per step: kernel matrix from the raw inputs,
 * blocked Cholesky, alpha, log det, K^-1 = L^-T L^-1 (rocBLAS dtrsm + dgemm), and one pass over the n x n entries for the
 * 1 + d gradient sums.  *info != 0: K is not positive definite at this theta (the optimiser rejects the step). */
typedef struct gpf_nll gpf_nll;
int gpf_gp_nll_open(int device, int n, int d, int m, const double* Xn, const double* Yn, double sigma, gpf_nll** out);
int gpf_gp_nll_eval(gpf_nll* s, const double* theta, double* value, double* grad, int* info);
int gpf_gp_nll_close(gpf_nll* s);

/* Diagnostic: time of one pass of an elementwise kernel that reads `nin` and writes `nout` fp64 planes of
 * `doubles_per_plane` elements (16 bytes per lane, grid-stride): what THIS device streams for the byte count of a fused
 * step.  bench.py reports it beside the step kernel's HBM figure (no reference counterpart: the reference has no device). */
int gpf_stream_probe(int device, int nin, int nout, int64_t doubles_per_plane, int reps, double* ms_per_pass);

#ifdef __cplusplus
}
#endif
#endif /* GAPFLOW_HIP_H */
