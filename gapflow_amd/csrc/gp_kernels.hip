// Gaussian-process surrogate closure on the GPU (GaPFlow/models/gp.py, tinygp semantics).
//
//   k(x,x') = A (1 + sqrt3 r) exp(-sqrt3 r),  r = || s o (x - x') ||_2      gp.py:598-603 (Matern-3/2, ARD)
//   K = k(X,X) + sigma^2 I = L L^T                                           tinygp GaussianProcess(diag=yerr^2)
//   alpha = K^-1 Y ; mean(x*) = sum_i alpha_i k(x_i, x*)                     gp.py:525-535
//   var(x*) = A - || L^-1 k(X, x*) ||^2                                      gp.py:509-522
//   d mean / d x*_0 = sum_i alpha_i 3 A s_0 (s_0 (x_i0 - x*_0)) exp(-sqrt3 r)   stress.py:533-537
//
// The reference materialises Ks = k(X_train, X_test) for all cells (N_train x cells: 17 GB per GP
// at 2048^2 with 512 training points, gp.py:516,532).  Here the mean is a fused kernel: training
// inputs and alpha sit in LDS, every thread owns one cell and streams over the training set --
// fp64 VALU/transcendental bound, no HBM traffic beyond the fields themselves.  Only the variance
// needs a dense solve: cells are processed in tiles, Ks of a tile is built, rocBLAS dtrsm (the one
// MFMA-shaped operation of the path) forms L^-1 Ks and a column-norm kernel finishes.
// The Cholesky factorisation and alpha come from the in-library k_gp_potrf / k_gp_potrs (default) or from
// rocSOLVER dpotrf / dpotrs (the default; GPF_USE_ROCSOLVER=0: the in-library blocked Cholesky); rocBLAS / rocSOLVER are dlopen'ed so that a process
// which also hosts PyTorch binds to the copies already mapped.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdlib.h>
#include <string>
#include "device_types.hpp"

namespace gpf {

constexpr int GP_MAX_D = 4;
constexpr int GP_CHUNK = 512;        // training points staged in LDS per pass

struct GpModelDev {
    int n, d, m;
    int dims[GP_MAX_D];             // feature index of each active dimension: 0..2 q, 3..5 topography, 6 extra
    double fscale[GP_MAX_D];        // sqrt3 inv_scale / X_scale: raw feature -> kernel coordinate
    double amp, yscale;
    const double* Z;                // [n][d] training inputs in kernel coordinates (sqrt3 X_norm * inv_scale: |z - z'| IS sqrt3 r)
    const double* alpha;            // [m][n]
    const double* L;                // [n][n] column-major lower Cholesky factor
};

struct GpFieldArgs {
    const double* q; const double* topo; const double* Ls;
    Layout L;
    double* out0; double* out1;     // output planes (already offset to the component), out1 only for m == 2
    double* blockmax;               // per-block max of d mean/d x_0 (WITH_GRAD) or nullptr
};

// 2^(j/256), j = 0..255, correctly rounded: the exponential below is exp(x) = 2^m 2^(j/256) exp(f) with |f| <= ln2 / 512, a
// degree-4 polynomial (f^5 / 120 < 4e-17).  Kernels copy the table into LDS once (gp_exp_table_to_lds) and index it per lane.
constexpr int GP_EXP_N = 256;
constexpr double GP_SQRT3 = 1.7320508075688772;     // folded into the kernel coordinates (GpModelDev::Z, fscale) by the host
__device__ __constant__ double gp_exp2_tab[GP_EXP_N] = {
    1.0, 1.0027112750502025, 1.0054299011128027, 1.0081558981184175,
    1.0108892860517005, 1.0136300849514894, 1.016378314910953, 1.019133996077738,
    1.0218971486541166, 1.0246677928971357, 1.0274459491187637, 1.030231637686041,
    1.0330248790212284, 1.0358256936019572, 1.0386341019613787, 1.041450124688316,
    1.0442737824274138, 1.0471050958792898, 1.0499440858006872, 1.0527907730046264,
    1.0556451783605572, 1.0585073227945128, 1.061377227289262, 1.0642549128844645,
    1.0671404006768237, 1.0700337118202419, 1.0729348675259756, 1.075843889062791,
    1.0787607977571199, 1.0816856149932152, 1.0846183622133092, 1.0875590609177697,
    1.0905077326652577, 1.0934643990728858, 1.0964290818163769, 1.099401802630222,
    1.102382583307841, 1.1053714457017412, 1.1083684117236787, 1.1113735033448175,
    1.1143867425958924, 1.1174081515673693, 1.1204377524096067, 1.12347556733302,
    1.1265216186082418, 1.129575928566288, 1.1326385195987192, 1.1357094141578055,
    1.1387886347566916, 1.1418762039695616, 1.1449721444318042, 1.148076478840179,
    1.1511892299529827, 1.154310420590216, 1.1574400736337511, 1.1605782120274988,
    1.1637248587775775, 1.1668800369524817, 1.1700437696832502, 1.1732160801636373,
    1.1763969916502812, 1.1795865274628758, 1.182784710984341, 1.1859915656609938,
    1.189207115002721, 1.1924313825831512, 1.1956643920398273, 1.1989061670743806,
    1.202156731452703, 1.2054161090051239, 1.2086843236265816, 1.2119613992768012,
    1.215247359980469, 1.2185422298274085, 1.2218460329727576, 1.2251587936371455,
    1.22848053610687, 1.2318112847340759, 1.2351510639369334, 1.2384998981998165,
    1.241857812073484, 1.245224830175258, 1.2486009771892048, 1.2519862778663162,
    1.255380757024691, 1.2587844395497165, 1.2621973503942507, 1.2656195145788063,
    1.2690509571917332, 1.2724917033894028, 1.275941778396392, 1.2794012075056693,
    1.2828700160787783, 1.2863482295460256, 1.2898358734066657, 1.2933329732290895,
    1.2968395546510096, 1.3003556433796506, 1.3038812651919358, 1.3074164459346773,
    1.3109612115247644, 1.3145155879493546, 1.318079601266064, 1.3216532776031575,
    1.3252366431597413, 1.3288297242059544, 1.3324325470831615, 1.3360451382041458,
    1.339667524053303, 1.3432997311868353, 1.3469417862329458, 1.3505937158920345,
    1.3542555469368927, 1.3579273062129011, 1.3616090206382248, 1.365300717204012,
    1.3690024229745905, 1.3727141650876684, 1.3764359707545302, 1.380167867260238,
    1.383909881963832, 1.387662042298529, 1.3914243757719262, 1.3951969099662003,
    1.3989796725383112, 1.4027726912202048, 1.4065759938190154, 1.4103896082172707,
    1.4142135623730951, 1.4180478843204152, 1.4218926021691656, 1.4257477441054942,
    1.42961333839197, 1.433489413367789, 1.4373759974489824, 1.4412731191286257,
    1.4451808069770467, 1.449099089642035, 1.4530279958490526, 1.4569675544014438,
    1.460917794180647, 1.4648787441464057, 1.4688504333369818, 1.4728328908693675,
    1.4768261459394993, 1.4808302278224719, 1.4848451658727524, 1.488870989524397,
    1.4929077282912648, 1.4969554117672355, 1.5010140696264256, 1.5050837316234065,
    1.5091644275934228, 1.5132561874526098, 1.5173590411982147, 1.5214730189088146,
    1.5255981507445384, 1.529734466947287, 1.533881997840956, 1.5380407738316568,
    1.5422108254079407, 1.5463921831410214, 1.550584877685, 1.5547889397770887,
    1.559004400237837, 1.5632312899713576, 1.567469639965553, 1.5717194812923414,
    1.5759808451078865, 1.5802537626528246, 1.5845382652524937, 1.588834384317164,
    1.593142151342267, 1.597461597908627, 1.6017927556826934, 1.606135656416771,
    1.6104903319492543, 1.6148568142048607, 1.6192351351948637, 1.6236253270173289,
    1.6280274218573478, 1.632441451987275, 1.6368674497669644, 1.6413054476440063,
    1.645755478153965, 1.6502175739206177, 1.6546917676561943, 1.6591780921616162,
    1.6636765803267364, 1.6681872651305825, 1.6727101796415966, 1.6772453570178785,
    1.681792830507429, 1.6863526334483934, 1.6909247992693053, 1.6955093614893326,
    1.7001063537185235, 1.7047158096580513, 1.709337763100463, 1.713972247929926,
    1.718619298122478, 1.723278947746274, 1.7279512309618377, 1.732636182022311,
    1.7373338352737062, 1.7420442251551564, 1.746767386199169, 1.7515033530318782,
    1.7562521603732995, 1.761013843037584, 1.7657884359332727, 1.7705759740635547,
    1.7753764925265212, 1.7801900265154245, 1.785016611318935, 1.789856282321401,
    1.7947090750031072, 1.7995750249405351, 1.804454167806624, 1.809346539371032,
    1.8142521755003989, 1.8191711121586085, 1.8241033854070534, 1.8290490314048973,
    1.8340080864093424, 1.8389805867758937, 1.843966568958626, 1.8489660695104508,
    1.8539791250833855, 1.8590057724288205, 1.864046048397789, 1.8690999899412386,
    1.8741676341103, 1.8792490180565602, 1.8843441790323345, 1.8894531543909392,
    1.8945759815869656, 1.8997126981765553, 1.9048633418176741, 1.9100279502703899,
    1.9152065613971474, 1.9203992131630474, 1.925605943636125, 1.930826790987627,
    1.9360617934922943, 1.9413109895286405, 1.9465744175792332, 1.9518521162309783,
    1.9571441241754002, 1.9624504802089273, 1.9677712232331759, 1.9731063922552343,
    1.978456026387951, 1.9838201648502194, 1.9891988469672663, 1.9945921121709402};

// called by all threads of a block before the first matern_terms; `tab` is a __shared__ double[GP_EXP_LDS].  The entry behind the
// table holds 3/8, the one constant of the square root that the instruction encoding cannot carry: read from LDS once per kernel
// it lives in a register, written as a literal it costs two moves per evaluation.
constexpr int GP_EXP_LDS = GP_EXP_N + 1;
__device__ __forceinline__ void gp_exp_table_to_lds(double* tab) {
    for (int i = threadIdx.x; i < GP_EXP_N; i += blockDim.x) tab[i] = gp_exp2_tab[i];
    if (threadIdx.x == 0) tab[GP_EXP_N] = 0.375;
    __syncthreads();
}

// Matern-3/2 pieces for t = 3 r^2 >= 0:  s = sqrt(t),  e = exp(-s);  k = A (1 + s) e,  dk/ds ~ s e.
// These kernels are bound by the fp64 units (on gfx950 every vector operation costs four cycles per wave, the double-precision
// ones included), so the NUMBER of operations of this function is their speed:
//   * sqrt: v_rsq_f64 + one Halley step (5 operations) instead of the range-scaled library sqrt; callers start their sum of squares
//     at 1e-300 instead of 0, which keeps t = 0 away from rsq at no cost (k = A exactly there);
//   * exp (argument <= 0): reduction by ln2/256 with the rounding done by the adder (x c + 1.5 2^52: the integer sits in the low
//     word of the sum, no rint and no conversion), a 256-entry table of 2^(j/256) in LDS, a degree-4 polynomial and v_ldexp_f64 --
//     10 operations instead of the 12 of round 2's 64-entry / degree-5 form and the 19 of a degree-13 polynomial after a
//     reduction by ln 2 (and far fewer than the library exp with its overflow / underflow selects).
// Accurate to 1.5 ulp (exp) and 1.5 ulp (sqrt) on the range that matters (tools/matern_accuracy.hip); every kernel below (K, Ks,
// mean, variance, likelihood) uses this one function, so K and Ks stay consistent.
__device__ __forceinline__ void matern_terms(double t, double& s, double& e, const double* __restrict__ tab) {
    const double y = __builtin_amdgcn_rsq(t);               // ~24-bit seed (t >= 3e-300: the callers' sums start at 1e-300)
#ifdef GPF_GP_GOLDSCHMIDT_SQRT  // (A/B: two Goldschmidt steps, 7 operations, 1.1 ulp)
    double g = t * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    r = fma(-h, g, 0.5);
    s = fma(g, r, g);
#else
    // one Halley step around c = t y^2 = 1 + d, |d| < 1.1e-7: sqrt(t) = t y (1 - d/2 + 3/8 d^2) (1 + O(d^3)).  Written in d, with
    // the constants the instruction encoding holds (-1, -1/2, 1) and 3/8 from a register (tab[GP_EXP_N]; 1/2 in its place costs
    // 6 ulp: d^2 / 8 = 1.4e-15) -- the form 15/8 - 5/4 c + 3/8 c^2 spent two moves per evaluation on its constants.  5 operations.
    const double g = t * y;
    const double d = fma(g, y, -1.0);
    const double q = fma(d, fma(d, tab[GP_EXP_N], -0.5), 1.0);
    s = g * q;
#endif
    // exp(-800) = 0 in fp64 anyway; the clamp keeps the reduction exact (|n| < 2^19) and the integer in range for the absurd
    // distances a line-search probe of the training can produce (unclamped, t > 1e14 returned inf: tools/matern_accuracy.hip)
    const double x = -fmin(s, 800.0);
    const double magic = 6755399441055744.0;                // 1.5 * 2^52: x c + magic is rounded to an integer by the addition
    const double nm = fma(x, 369.32993046757463, magic);    // 256 / ln 2
    const double n = nm - magic;
    double f = fma(n, -0.0027076061733168900, x);           // ln2 / 256 = hi + lo; hi = (ln2 hi) / 256 keeps its 21 trailing zero bits
    f = fma(n, -7.4539645674632333e-13, f);
    double p = 4.1666666666666664e-02;                      // 1/4!
    p = fma(p, f, 1.6666666666666666e-01);                  // 1/3!
    p = fma(p, f, 0.5);
    p = fma(p, f, 1.0);
    p = fma(p, f, 1.0);
    const int ni = __double2loint(nm);                      // two's complement of n in the low word (|n| < 2^31)
    e = ldexp(tab[ni & (GP_EXP_N - 1)] * p, ni >> 8);       // arithmetic shift: floor(n / 256) for the negative n
}

__device__ __forceinline__ double gp_feature(const GpFieldArgs& a, int f, long long o) {
    if (f < 3) return a.q[o + f * a.L.plane];
    if (f < 6) return a.topo[o + (f - 3) * a.L.plane];
    return a.Ls ? a.Ls[o] : 0.0;
}

// K (column-major n x n) = k(Z, Z) + sigma^2 I, with Z already in kernel coordinates (the factor sqrt3 of the Matern argument
// folded into them: one multiplication less per evaluation in every kernel that works from them)
__global__ void k_gp_matrix(const double* Z, int n, int d, double amp, double sigma2, double* K) {
    __shared__ double exptab[GP_EXP_LDS];
    gp_exp_table_to_lds(exptab);
    const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
    if (i >= n) return;
    double r2 = 1e-300;      // (not 0: matern_terms takes rsq of it)
    for (int k = 0; k < d; ++k) {
        const double t = Z[i * d + k] - Z[j * d + k];
        r2 += t * t;
    }
    double r, e;                                      // r = sqrt3 * |dz|
    matern_terms(r2, r, e, exptab);
    K[i + (long long)j * n] = amp * (1.0 + r) * e + (i == j ? sigma2 : 0.0);
}

// posterior mean (and optionally d mean/d x_0) at every cell of the grid incl. ghost cells
template <int D, int M, bool WITH_GRAD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 4))) void k_gp_mean(const GpModelDev g, const GpFieldArgs a) {
    __shared__ double sZ[GP_CHUNK * D];
    __shared__ double sA[GP_CHUNK * M];
    __shared__ double sred[4];
    __shared__ double exptab[GP_EXP_LDS];
    gp_exp_table_to_lds(exptab);
    const long long w = a.L.Ny + 2, ncell = (long long)(a.L.Nx + 2) * w;
    const long long cell = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    const bool active = cell < ncell;
    long long o = 0;
    double z[D];
    if (active) {
        o = a.L.at((int)(cell / w), (int)(cell % w));
        for (int k = 0; k < D; ++k) z[k] = gp_feature(a, g.dims[k], o) * g.fscale[k];
    } else {
        for (int k = 0; k < D; ++k) z[k] = 0.0;
    }
    double acc[M], gacc = 0.0;
    for (int k = 0; k < M; ++k) acc[k] = 0.0;
    for (int base = 0; base < g.n; base += GP_CHUNK) {
        const int cnt = min(GP_CHUNK, g.n - base);
        __syncthreads();
        for (int t = threadIdx.x; t < cnt * D; t += blockDim.x) sZ[t] = g.Z[(long long)base * D + t];
        for (int t = threadIdx.x; t < cnt * M; t += blockDim.x) sA[t] = g.alpha[(long long)(t / cnt) * g.n + base + (t % cnt)];
        __syncthreads();
#pragma unroll 4
        for (int i = 0; i < cnt; ++i) {
            double r2 = 1e-300, d0 = 0.0;     // (not 0: matern_terms takes rsq of it)
            for (int k = 0; k < D; ++k) {
                const double t = sZ[i * D + k] - z[k];
                if (k == 0) d0 = t;
                r2 += t * t;
            }
            double r, e;
            matern_terms(r2, r, e, exptab);
            const double kv = fma(r, e, e);
            for (int k = 0; k < M; ++k) acc[k] += sA[k * cnt + i] * kv;
            if (WITH_GRAD) gacc += sA[i] * d0 * e;
        }
    }
    if (active) {
        if (a.out0) a.out0[o] = g.amp * acc[0] * g.yscale;
        if (M > 1 && a.out1) a.out1[o] = g.amp * acc[M - 1] * g.yscale;
    }
    if (WITH_GRAD) {
        // d mean/d x_0 = 3 A s_0 sum_i alpha_i s_0 (x_i0 - x*_0) e_i = A (sqrt3 s_0) sum_i alpha_i (z_i0 - z*_0) e_i in the sqrt3-scaled
        // coordinates; the host multiplies by fscale_0 = sqrt3 s_0 / X_scale_0
        double v = active ? g.amp * gacc : -__builtin_inf();
        for (int s = 32; s >= 1; s >>= 1) v = nanmax(v, __shfl_down(v, s));
        if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int i = 1; i < (int)(blockDim.x >> 6); ++i) v = nanmax(v, sred[i]);
            a.blockmax[blockIdx.x] = v;
        }
    }
}

__global__ void k_gp_maxreduce(const double* in, int n, double scale, double* out) {
    __shared__ double s[4];
    double v = -__builtin_inf();
    for (int i = threadIdx.x; i < n; i += blockDim.x) v = nanmax(v, in[i]);
    for (int k = 32; k >= 1; k >>= 1) v = nanmax(v, __shfl_down(v, k));
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < (int)(blockDim.x >> 6); ++i) v = nanmax(v, s[i]);
        out[0] = v * scale;
    }
}

// Ks tile: column j = k(Z, z*_j) for cells [cell0, cell0 + ncols), column-major n x ncols
template <int D>
__global__ __launch_bounds__(256) void k_gp_ks_tile(const GpModelDev g, const GpFieldArgs a, long long cell0, int ncols,
                                                    double* Ks) {
    __shared__ double exptab[GP_EXP_LDS];
    gp_exp_table_to_lds(exptab);
    const long long w = a.L.Ny + 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;      // training point
    const int j = blockIdx.y;                                 // column of the tile
    if (i >= g.n || j >= ncols) return;
    const long long cell = cell0 + j;
    const long long o = a.L.at((int)(cell / w), (int)(cell % w));
    double r2 = 1e-300;      // (not 0: matern_terms takes rsq of it)
    for (int k = 0; k < D; ++k) {
        const double t = g.Z[(long long)i * D + k] - gp_feature(a, g.dims[k], o) * g.fscale[k];
        r2 += t * t;
    }
    double r, e;
    matern_terms(r2, r, e, exptab);
    Ks[i + (long long)j * g.n] = g.amp * (1.0 + r) * e;
}

// var_j = (A - sum_i V_ij^2) * yscale^2 for a solved tile V = L^-1 Ks; per-block max for the AL criterion
constexpr int GP_VAR_COLS = 128;
__global__ __launch_bounds__(1024) void k_gp_var_tile(const double* V, int n, int ncols, double amp, double yscale2,
                                                     long long cell0, Layout L, double* var_plane, double* blockmax) {
    // variance of column j = amp - |V[:, j]|^2.  V is column-major, so a WAVE walks one column with 512-byte loads and
    // reduces across lanes (one thread per column read 4 KB apart from its neighbours: 69 us per tile instead of ~17)
    // a block of 16 waves owns GP_VAR_COLS = 128 columns, a wave eight of them, all in flight at once
    __shared__ double sred[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long w = L.Ny + 2;
    double best = -__builtin_inf();
    for (int c0 = 0; c0 < 8; c0 += 8) {
        const int j0 = blockIdx.x * GP_VAR_COLS + wave * 8 + c0;
        if (j0 >= ncols) break;                             // wave-uniform
        double s[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s[u] = 0.0;
            if (j0 + u < ncols) {
                const double* col = V + (long long)(j0 + u) * n;
                for (int i = lane; i < n; i += 64) { const double v = col[i]; s[u] = fma(v, v, s[u]); }
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u] += __shfl_down(s[u], d);
        }
        if (lane == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (j0 + u >= ncols) break;
                const double out = (amp - s[u]) * yscale2;
                const long long cell = cell0 + j0 + u;
                var_plane[L.at((int)(cell / w), (int)(cell % w))] = out;
                best = nanmax(best, out);
            }
        }
    }
    if (lane == 0) sred[wave] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 16; ++i) best = nanmax(best, sred[i]);
        blockmax[blockIdx.x] = best;
    }
}

// Predictive variance in ONE kernel, nothing through HBM: var_j = (A - |L^-1 k(Z, z*_j)|^2) * yscale^2 for n <= 512.
// The tiled path above writes a 64-MiB Ks tile, reads it back in the product, writes the product and reads that back for
// the column norms: 290 MB of HBM traffic per 16 K cells next to 2 x 17 MFLOP per cell.  Here a workgroup of 16 waves owns
// 64 cells:
//   * the Matern values k(Z_k, z*_j) of 32 training points x 64 cells at a time are produced on the vector ALUs straight
//     into LDS (each wave: its lane's cell against 2 wave-uniform training points, 2 evaluations per lane and batch),
//     double-buffered; batch p is multiplied in the same phase as batch 15 - p (the order of the columns of a sum is free),
//     so every phase holds the same work and no wave idles through the late, nearly empty batches; one barrier per phase;
//   * V = L^-1 Ks runs on the matrix cores (v_mfma_f64_16x16x4_f64; A operand = L^-1 rows straight from L2 -- the 2 MB
//     factor stays resident there --, B operand = the LDS tile, accumulators never leave the registers).  L^-1 is lower
//     triangular: a 16-row tile t has nothing to multiply beyond column 16 t + 15.  Wave w owns tiles w and 31 - w: 52 % of
//     the square product's flops, the same amount for every wave;
//   * the column norms are the sum of squares of the accumulators: over the registers, across the four lane groups that
//     hold different rows of the same column (layout of the f64 MFMA: col = lane & 15, row = (lane >> 4) + 4 reg), then
//     over the waves in a fixed order through LDS (deterministic).
// Lane maps of the operands: A[row = lane & 15][k = lane >> 4], B[k = lane >> 4][col = lane & 15], one f64 each.
// Measured (2048^2 cells x 512 points): 23.5 ms against 39.5 ms for the tiled path.  571 M MFMAs at 64 cycles each are
// 14.9 ms of matrix-pipe time (SQ_VALU_MFMA_BUSY_CYCLES); the rest cannot hide behind them: on this chip the f64 matrix
// instruction and the f64 vector ALU are the same units -- every fp64 VALU instruction between two MFMAs lengthens the pair
// by its own 4+ cycles (tools/mfma_f64_probe.hip: 78 TFLOP/s with nothing in between, 55 with six FMAs, 43 with twelve) --
// so the Matern values (~30 fp64 operations each) and the operand selects are paid in full.  Tried and dropped: 8 waves
// with four tiles each (250 registers; prefetched or not: 27-30 ms), 64-column batches (no change), unconditional
// prefetched A loads in straight-line half batches (exact s_waitcnt counts, but 128 registers per wave at 16 waves per
// workgroup do not hold them: scratch spills, 93 ms).
constexpr int GPV_CELLS = 64, GPV_KB = 32, GPV_MAX_N = 512;
constexpr int GPV_WAVES = 16;                       // waves per workgroup
constexpr int GPV_TPW = 32 / GPV_WAVES;             // 16-row tiles of L^-1 per wave
#ifndef GPV_KU
#define GPV_KU 4                                    // k-steps (of 4 columns) whose A operands are requested together: 1, 2 or 4
#endif                                              // (27.0 / 24.9 / 24.4 ms; with 4 the kernel sits at 112-118 of the 128
                                                    // registers a wave of a 16-wave workgroup can have -- tests/test_step_isa.py
                                                    // fails on any scratch use here)
typedef double gpv_acc __attribute__((ext_vector_type(4)));

template <int D>
__global__ __launch_bounds__(64 * GPV_WAVES) void k_gp_var_fused(const GpModelDev g, const double* __restrict__ Linv, const GpFieldArgs a,
                                                                 double yscale2, double* __restrict__ var_plane, double* __restrict__ blockmax) {
    // B tile of a batch, laid out the way the MFMA reads it: the 64 lanes of one read -- (k mod 4, cell mod 16) for a fixed
    // k-step and 16-cell tile -- are 64 consecutive doubles.  (Row-major [k][cell] put the four k rows of a read 512 bytes
    // apart, on the same banks: four-way conflicts on every read, 2 % of the kernel.)
    __shared__ double Bs[4][GPV_KB * GPV_CELLS];        // two phases (double buffer) x two batches per phase
    auto bs_index = [](int kk, int cell) { return (((kk >> 2) * 4 + (cell >> 4)) * 4 + (kk & 3)) * 16 + (cell & 15); };
    __shared__ double part[GPV_WAVES][GPV_CELLS];
    __shared__ double exptab[GP_EXP_LDS];
    gp_exp_table_to_lds(exptab);
    const int n = g.n;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long w = a.L.Ny + 2, ncell = (long long)(a.L.Nx + 2) * w;
    const long long cell = blockIdx.x * (long long)GPV_CELLS + lane;
    const bool cell_ok = cell < ncell;
    const long long o = cell_ok ? a.L.at((int)(cell / w), (int)(cell % w)) : a.L.at(0, 0);
    double z[D];
    for (int k = 0; k < D; ++k) z[k] = gp_feature(a, g.dims[k], o) * g.fscale[k];

    // This wave's 16-row tiles of L^-1: w and 31 - w.  Tile t has nothing to multiply beyond column 16 t + 15 (lower
    // triangular), so its work is proportional to t + 1 and every wave gets the same total.  Four waves share a SIMD: while
    // one waits for its operands (every k-step pays an L2 round trip -- hipcc drains the loads before their use, they sit
    // behind wave-uniform branches) the others keep the matrix pipe busy.
    int tile[GPV_TPW];
    for (int i = 0; i < GPV_TPW; ++i) tile[i] = (i & 1) ? 31 - wv - GPV_WAVES * (i >> 1) : wv + GPV_WAVES * (i >> 1);
    const int col16 = lane & 15, kq = lane >> 4;
    gpv_acc acc[GPV_TPW][4];
    for (int rt = 0; rt < GPV_TPW; ++rt)
        for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = gpv_acc{0.0, 0.0, 0.0, 0.0};

    // one Matern value per lane: this lane's cell against training point 32 b + wv + GPV_WAVES i (wave-uniform: a scalar operand)
    auto produce_one = [&](int b, int buf, int i) {
        const int kk = wv + GPV_WAVES * i;
        const int k = b * GPV_KB + kk;
        double val = 0.0;
#ifdef GPV_EXP_NOPRODUCE    // (timing experiment: wrong results)
        val = z[0] + k;
#else
        if (k < n) {
            double r2 = 1e-300;      // (not 0: matern_terms takes rsq of it)
            for (int d = 0; d < D; ++d) {
                const double t = g.Z[(long long)k * D + d] - z[d];
                r2 += t * t;
            }
            double r, e;
            matern_terms(r2, r, e, exptab);
            val = g.amp * (1.0 + r) * e;
        }
#endif
        Bs[buf][bs_index(kk, lane)] = val;
    };
    constexpr int NPROD = GPV_KB / GPV_WAVES;           // Matern values per lane and batch
    // tile rt meets the k-step that starts at column k0 (wave-uniform)
    auto meets = [&](int rt, int k0) { return k0 <= 16 * tile[rt] + 15 && 16 * tile[rt] < n && k0 < n; };
    const unsigned int lane_off = (unsigned int)col16 + (unsigned int)kq * (unsigned int)n;
    const int nbatch = (n + GPV_KB - 1) / GPV_KB;
    // Batch b has 32 - 2 b tiles at work: alone, the late batches would leave most waves idle and the few busy ones exposed
    // to the latency of their operand loads.  The order of the columns is free (a sum), so batch p is paired with batch
    // nbatch - 1 - p in one phase: every phase then has the same amount of work, and every wave two or three tile-batches
    // of it.  Slot 0 / 1 of the LDS buffer holds the B tile of the lower / upper batch of the phase.
    const int nphase = (nbatch + 1) / 2;
    auto batch_of = [&](int phase, int slot) { return slot == 0 ? phase : nbatch - 1 - phase; };
    auto paired = [&](int phase) { return nbatch - 1 - phase != phase; };          // the middle batch of an odd count is alone
    for (int i = 0; i < NPROD; ++i) produce_one(batch_of(0, 0), 0, i);
    if (paired(0)) for (int i = 0; i < NPROD; ++i) produce_one(batch_of(0, 1), 1, i);
    __syncthreads();
    // the products of one batch: GPV_KU k-steps at a time -- their A operands are requested together, then their MFMAs run
    // back to back (hipcc drains the loads before the first use, they sit behind wave-uniform branches: one L2 round trip
    // per group).  `next`, `nslot`: between the k-steps, the Matern values of a batch of the next phase are produced.
    auto consume = [&](int b, int buf, int next, int nbuf) {
#pragma unroll
        for (int ks = 0; ks < GPV_KB; ks += 4 * GPV_KU) {
            const int k0 = b * GPV_KB + ks;
            double av[GPV_TPW][GPV_KU];
#pragma unroll
            for (int rt = 0; rt < GPV_TPW; ++rt) {
                if (!meets(rt, k0)) continue;           // wave-uniform; the same for all k-steps of an aligned group of <= 16 columns
                // A operand from L2: wave-uniform base (tile row, column) + one 32-bit lane offset (row in the tile, column in the k-step)
                const bool rok = 16 * tile[rt] + col16 < n;
#pragma unroll
                for (int u = 0; u < GPV_KU; ++u) {
                    const double* base = Linv + (16 * tile[rt] + (long long)(k0 + 4 * u) * n);
                    av[rt][u] = (rok && k0 + 4 * u + kq < n) ? base[lane_off] : 0.0;
                }
            }
#pragma unroll
            for (int u = 0; u < GPV_KU; ++u) {
                double bv[4];
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) bv[ct] = Bs[buf][bs_index(ks + 4 * u + kq, ct * 16 + col16)];
#pragma unroll
                for (int rt = 0; rt < GPV_TPW; ++rt) {
                    if (!meets(rt, k0)) continue;
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct)
#ifdef GPV_EXP_NOMFMA       // (timing experiment: wrong results)
                        acc[rt][ct][0] += av[rt][u] * bv[ct];
#else
                        acc[rt][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[rt][u], bv[ct], acc[rt][ct], 0, 0, 0);
#endif
                }
                if (next >= 0 && (ks + 4 * u) % (GPV_KB / NPROD) == 0) produce_one(next, nbuf, (ks + 4 * u) / (GPV_KB / NPROD));
            }
        }
    };
    for (int ph = 0; ph < nphase; ++ph) {
        const int cur = 2 * (ph & 1), nxt = 2 * ((ph + 1) & 1);        // buffer pairs alternate between phases
        const bool more = ph + 1 < nphase;
        // (one copy of the batch body in the binary: two inlined copies overflow the 128 registers of a wave)
#pragma unroll 1
        for (int slot = 0; slot < (paired(ph) ? 2 : 1); ++slot) {
            const bool feeds = more && (slot == 0 || paired(ph + 1));
            consume(batch_of(ph, slot), cur + slot, feeds ? batch_of(ph + 1, slot) : -1, nxt + slot);
        }
        __syncthreads();
    }
    // |V[:, j]|^2: this wave's rows of column j = ct * 16 + (lane & 15)
    for (int ct = 0; ct < 4; ++ct) {
        double s = 0.0;
        for (int rt = 0; rt < GPV_TPW; ++rt)
            for (int r = 0; r < 4; ++r) s = fma(acc[rt][ct][r], acc[rt][ct][r], s);
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        if (lane < 16) part[wv][ct * 16 + lane] = s;
    }
    __syncthreads();
    if (wv == 0) {
        double s = 0.0;
        for (int i = 0; i < GPV_WAVES; ++i) s += part[i][lane];
        const double out = (g.amp - s) * yscale2;
        double best = -__builtin_inf();
        if (cell_ok) {
            var_plane[o] = out;
            best = out;
        }
        for (int d = 32; d >= 1; d >>= 1) best = nanmax(best, __shfl_down(best, d));
        if (lane == 0) blockmax[blockIdx.x] = best;
    }
}

// Gradient of the negative log marginal likelihood (gp.py:307-318 through jax.grad in the reference):
//   dL/dtheta_k = -1/2 tr(W dK/dtheta_k),  W = alpha alpha^T - m K^-1,
//   dK/dlog_amp = Kf (the kernel matrix without the noise diagonal),
//   dK/dlog_scale_j = 3 A exp(-r) s_j^2 d_j^2   (d_j the coordinate difference, s_j = exp(-log_scale_j)).
// One thread per matrix entry recomputes its kernel value from the RAW inputs X and accumulates its 1 + d products; per-block
// partial sums in a fixed order (deterministic), folded by the host.  Kinv: full symmetric n x n, column-major.
template <int D>
__global__ __launch_bounds__(256) void k_gp_nll_grad(const double* __restrict__ X, const double* __restrict__ alpha, const double* __restrict__ Kinv,
                                                     int n, int m, double amp, const double* __restrict__ inv_scale, double* __restrict__ partial) {
    __shared__ double red[4][1 + D];
    __shared__ double exptab[GP_EXP_LDS];
    gp_exp_table_to_lds(exptab);
    const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
    double acc[1 + D];
    for (int k = 0; k <= D; ++k) acc[k] = 0.0;
    if (i < n) {
        double t[D], r2 = 1e-300;       // (not 0: matern_terms takes rsq of it)
        for (int k = 0; k < D; ++k) {
            const double dz = (X[(long long)i * D + k] - X[(long long)j * D + k]) * inv_scale[k];
            t[k] = dz * dz;
            r2 += t[k];
        }
        double r, e;
        matern_terms(3.0 * r2, r, e, exptab);
        double w = -(double)m * Kinv[i + (long long)j * n];
        for (int o = 0; o < m; ++o) w = fma(alpha[i + (long long)o * n], alpha[j + (long long)o * n], w);
        acc[0] = w * (amp * (1.0 + r) * e);
        const double we = w * e;
        for (int k = 0; k < D; ++k) acc[1 + k] = we * t[k];       // t_k = s_k^2 d_k^2
    }
    for (int k = 0; k <= D; ++k) {
        double v = acc[k];
        for (int sft = 32; sft >= 1; sft >>= 1) v += __shfl_down(v, sft);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x <= D) {
        const int k = threadIdx.x;
        partial[((long long)blockIdx.y * gridDim.x + blockIdx.x) * (1 + D) + k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
    }
}

// kernel matrix from RAW inputs: K = A (1 + r) exp(-r) + sigma^2 I with r = sqrt(3 sum_k ((x_ik - x_jk) s_k)^2)
template <int D>
__global__ void k_gp_nll_matrix(const double* __restrict__ X, int n, double amp, const double* __restrict__ inv_scale, double sigma2,
                                double* __restrict__ K) {
    __shared__ double exptab[GP_EXP_LDS];
    gp_exp_table_to_lds(exptab);
    const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
    if (i >= n) return;
    double r2 = 1e-300;      // (not 0: matern_terms takes rsq of it)
    for (int k = 0; k < D; ++k) {
        const double dz = (X[(long long)i * D + k] - X[(long long)j * D + k]) * inv_scale[k];
        r2 += dz * dz;
    }
    double r, e;
    matern_terms(3.0 * r2, r, e, exptab);
    K[i + (long long)j * n] = amp * (1.0 + r) * e + (i == j ? sigma2 : 0.0);
}

// In-library Cholesky factorisation K = L L^T (lower, in place, column-major) for the few-hundred-point
// training sets of the surrogates: one 1024-thread workgroup, right-looking, the matrix stays in L2
// (512^2 doubles = 2 MB).  *info = 0, or j+1 if the leading minor of order j+1 is not positive definite
// (LAPACK dpotrf convention).  The fall-back of rocSOLVER's dpotrf (roclibs(): GPF_USE_ROCSOLVER=0, or no rocSOLVER beside the
// rocBLAS in use), and the second opinion the tests hold rocSOLVER against.
// A: n x n block with leading dimension lda, lower triangle factorised in place.  *info is written only on failure
// (first non-positive pivot, 1-based, offset by `row0` for a diagonal block of a larger matrix) or, with row0 == 0, reset.
__global__ __launch_bounds__(1024) void k_gp_potrf(double* A, int n, int lda, int row0, int* info) {
    __shared__ double diag;
    __shared__ int bad;
    const int tid = threadIdx.x, T = blockDim.x;
    const int tk = tid >> 5, ti = tid & 31;
    if (tid == 0) bad = 0;
    if (row0 > 0 && *info != 0) return;             // an earlier diagonal block already failed
    for (int j = 0; j < n; ++j) {
        __syncthreads();
        if (tid == 0) {
            double d = A[j + (long long)j * lda];
            if (!(d > 0.0)) { bad = row0 + j + 1; d = 1.0; } else d = sqrt(d);
            A[j + (long long)j * lda] = d;
            diag = d;
        }
        __syncthreads();
        if (bad) break;
        const double inv = 1.0 / diag;
        for (int i = j + 1 + tid; i < n; i += T) A[i + (long long)j * lda] *= inv;
        __syncthreads();
        const double* col = A + (long long)j * lda;
        for (int kk = j + 1; kk < n; kk += 32) {
            const int k = kk + tk;
            for (int ii = kk; ii < n; ii += 32) {
                const int i = ii + ti;
                if (k < n && i < n && i >= k) A[i + (long long)k * lda] -= col[i] * col[k];
            }
        }
    }
    __syncthreads();
    if (tid == 0 && (bad || row0 == 0)) *info = bad;
}

// Solves L L^T X = B in place for nrhs right-hand sides (B column-major n x nrhs), one workgroup
__global__ __launch_bounds__(1024) void k_gp_potrs(const double* Lm, int n, int nrhs, double* B) {
    __shared__ double piv;
    const int tid = threadIdx.x, T = blockDim.x;
    for (int r = 0; r < nrhs; ++r) {
        double* b = B + (long long)r * n;
        for (int j = 0; j < n; ++j) {                  // L y = b
            __syncthreads();
            if (tid == 0) { piv = b[j] / Lm[j + (long long)j * n]; b[j] = piv; }
            __syncthreads();
            const double y = piv;
            for (int i = j + 1 + tid; i < n; i += T) b[i] -= Lm[i + (long long)j * n] * y;
        }
        for (int j = n - 1; j >= 0; --j) {             // L^T x = y
            __syncthreads();
            if (tid == 0) { piv = b[j] / Lm[j + (long long)j * n]; b[j] = piv; }
            __syncthreads();
            const double x = piv;
            for (int i = tid; i < j; i += T) b[i] -= Lm[j + (long long)i * n] * x;
        }
    }
}

// log det K = 2 sum log L_ii
__global__ void k_gp_logdet(const double* Lm, int n, double* out) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += log(Lm[i + (long long)i * n]);
    out[0] = 2.0 * s;
}

// Panel solve of the blocked Cholesky: A21 <- A21 L11^-T by forward substitution, one thread per row of the panel, the jb x jb
// diagonal factor (jb <= 64) in LDS.  rocBLAS dtrsm solves through inverted diagonal blocks, whose error grows with cond(L11)^2;
// on the numerically singular kernel matrices a trained surrogate produces (noise 1e-5 of the amplitude: cond(K) ~ 1/eps) that
// made the blocked factorisation report a non-positive pivot where LAPACK and the unblocked kernel succeed.
__global__ __launch_bounds__(256) void k_gp_panel_solve(const double* __restrict__ L11, int jb, int lda, double* A21, int rest) {
    __shared__ double sl[64][65];
    for (int t = threadIdx.x; t < jb * jb; t += blockDim.x) sl[t / jb][t % jb] = L11[(t / jb) + (long long)(t % jb) * lda];    // sl[j][k] = L11(j, k)
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rest) return;
    // (the row's earlier results are read back from A21 itself -- this thread wrote them --: no private array, no scratch)
    for (int j = 0; j < jb; ++j) {
        double a = A21[i + (long long)j * lda];
        for (int k = 0; k < j; ++k) a = fma(-A21[i + (long long)k * lda], sl[j][k], a);
        A21[i + (long long)j * lda] = a / sl[j][j];
    }
}

// zero the strict upper triangle so L can be handed out as a clean lower-triangular matrix
// n x n identity, column-major
__global__ void k_gp_identity(double* I, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
    if (i < n) I[i + (long long)j * n] = i == j ? 1.0 : 0.0;
}

// B = A^T, n x n column-major (32 x 32 tiles through LDS; block 32 x 8)
__global__ void k_gp_transpose(const double* __restrict__ A, double* __restrict__ B, int n) {
    __shared__ double t[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int i = bx + threadIdx.x, j = by + r;
        if (i < n && j < n) t[r][threadIdx.x] = A[i + (long long)j * n];
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int i = by + threadIdx.x, j = bx + r;       // B(i, j) = A(j, i)
        if (i < n && j < n) B[i + (long long)j * n] = t[threadIdx.x][r];
    }
}

__global__ void k_gp_clean_lower(double* Lm, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
    if (i < n && i < j) Lm[i + (long long)j * n] = 0.0;
}

// ---------------------------------------------------------------------------------------------
// rocBLAS / rocSOLVER through dlopen (no link-time dependency)
// ---------------------------------------------------------------------------------------------
struct RocLibs {
    typedef int (*create_t)(void**);
    typedef int (*destroy_t)(void*);
    typedef int (*set_stream_t)(void*, hipStream_t);
    typedef int (*potrf_t)(void*, int, int, double*, int, int*);
    typedef int (*potrs_t)(void*, int, int, int, double*, int, double*, int);
    typedef int (*trsm_t)(void*, int, int, int, int, int, int, const double*, const double*, int, double*, int);
    typedef int (*gemm_t)(void*, int, int, int, int, int, const double*, const double*, int, const double*, int, const double*, double*, int);
    create_t create = nullptr; destroy_t destroy = nullptr; set_stream_t set_stream = nullptr;
    potrf_t potrf = nullptr; potrs_t potrs = nullptr; trsm_t trsm = nullptr; gemm_t gemm = nullptr;
    bool ok = false;
    const char* err = "";
};

// rocBLAS enum values (rocblas-types.h): fill lower = 122, side left = 141, op none = 111, diag non-unit = 131
enum { ROC_FILL_LOWER = 122, ROC_SIDE_LEFT = 141, ROC_SIDE_RIGHT = 142, ROC_OP_NONE = 111, ROC_OP_TRANS = 112, ROC_DIAG_NON_UNIT = 131 };

// ONE rocBLAS / rocSOLVER image per process, and from ONE ROCm release.  A PyTorch-ROCm wheel bundles its own ROCm (HIP runtime,
// rocBLAS, rocSOLVER, hipBLASLt, rocRoller ... with the system's SONAMEs); two rocBLAS images in one process crash in
// rocRoller's static component registry when the second one is initialised, and a rocSOLVER of another release than the HIP
// runtime / rocBLAS that serve the process stalled for minutes registering its code objects (profiles/r03_rocsolver/README.md:
// the cause of the hang and of the abort recorded in round 2).  Therefore:
//   * rocBLAS: the image ALREADY MAPPED wins (RTLD_NOLOAD by SONAME); only if there is none, GPF_ROCBLAS_PATH or the loader's
//     default is opened;
//   * rocSOLVER (GPF_USE_ROCSOLVER=1): an image already mapped, else the copy that lies NEXT TO the rocBLAS in use; never one
//     found through the search path, which may belong to another release.
inline std::string dir_of_symbol(void* sym) {
    Dl_info info;
    if (!sym || !dladdr(sym, &info) || !info.dli_fname) return "";
    std::string f = info.dli_fname;
    const size_t k = f.rfind('/');
    return k == std::string::npos ? "" : f.substr(0, k);
}

inline RocLibs& roclibs() {
    static RocLibs R;
    static bool tried = false;
    static std::string err_text;
    if (tried) return R;
    tried = true;
    void* hb = dlopen("librocblas.so.5", RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);         // whatever the process already uses
    if (!hb) if (const char* path = getenv("GPF_ROCBLAS_PATH")) hb = dlopen(path, RTLD_NOW | RTLD_GLOBAL);     // the copy PyTorch bundles
    if (!hb) hb = dlopen("librocblas.so.5", RTLD_NOW | RTLD_GLOBAL);
    if (!hb) hb = dlopen("librocblas.so", RTLD_NOW | RTLD_GLOBAL);
    if (!hb) hb = dlopen("/opt/rocm/lib/librocblas.so", RTLD_NOW | RTLD_GLOBAL);
    if (!hb) { R.err = "could not dlopen rocBLAS"; return R; }
    R.create = (RocLibs::create_t)dlsym(hb, "rocblas_create_handle");
    R.destroy = (RocLibs::destroy_t)dlsym(hb, "rocblas_destroy_handle");
    R.set_stream = (RocLibs::set_stream_t)dlsym(hb, "rocblas_set_stream");
    R.trsm = (RocLibs::trsm_t)dlsym(hb, "rocblas_dtrsm");
    R.gemm = (RocLibs::gemm_t)dlsym(hb, "rocblas_dgemm");
    R.ok = R.create && R.destroy && R.set_stream && R.trsm;
    if (!R.ok) { R.err = "rocBLAS symbols missing"; return R; }
    // rocSOLVER's dpotrf / dpotrs factorise the kernel matrices unless GPF_USE_ROCSOLVER=0 (then, or where no rocSOLVER lies next
    // to the rocBLAS in use, the in-library blocked Cholesky does: gp_cholesky in api_gp.inc).  GPF_USE_ROCSOLVER=1 makes a
    // missing rocSOLVER an error instead of a fall-back.
    const char* use = getenv("GPF_USE_ROCSOLVER");
    const bool required = use && use[0] == '1';
    if (!use || use[0] != '0') {
        void* hs = dlopen("librocsolver.so.0", RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
        const std::string dir = dir_of_symbol((void*)R.create);
        for (const char* name : {"/librocsolver.so.0", "/librocsolver.so"})
            if (!hs && !dir.empty()) hs = dlopen((dir + name).c_str(), RTLD_NOW | RTLD_GLOBAL);
        if (hs) {
            R.potrf = (RocLibs::potrf_t)dlsym(hs, "rocsolver_dpotrf");
            R.potrs = (RocLibs::potrs_t)dlsym(hs, "rocsolver_dpotrs");
        }
        if (!R.potrf || !R.potrs) R.potrf = nullptr, R.potrs = nullptr;
        if (!R.potrf && required) {
            R.ok = false;
            err_text = "GPF_USE_ROCSOLVER=1 but no rocSOLVER lies next to the rocBLAS in use (" + dir + "): a copy of another ROCm release is not loaded";
            R.err = err_text.c_str();
        }
    }
    return R;
}

}  // namespace gpf
