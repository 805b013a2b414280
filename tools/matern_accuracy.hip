// Accuracy of matern_terms (csrc/gp_kernels.hip) on the device against long-double sqrt / exp over 1e-300 .. 1e300:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -I gapflow_amd/csrc tools/matern_accuracy.hip -o tools/matern_accuracy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include "gp_kernels.hip"
using namespace gpf;
__global__ void k(const double* t, double* s, double* e, int n) {
    __shared__ double tab[GP_EXP_LDS];
    gp_exp_table_to_lds(tab);
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) matern_terms(t[i], s[i], e[i], tab);
}
int main() {
    const int n = 1 << 20;
    double *ht = new double[n], *hs = new double[n], *he = new double[n];
    srand(1);
    for (int i = 0; i < n; ++i) { double u = rand() / (double)RAND_MAX; ht[i] = (i % 4 == 0) ? pow(10.0, -299 + 598 * u) : (i % 4 == 1 ? 50 * u : (i % 4 == 2 ? 1e6 * u * u : u * 1e-6)); }
    ht[0] = 3e-300; ht[1] = 1e300; ht[2] = 5.6e5; ht[3] = 4e-300;      // (callers start their sums of squares at 1e-300: t >= 3e-300)
    double *dt, *ds, *de; (void)hipMalloc(&dt, n * 8); (void)hipMalloc(&ds, n * 8); (void)hipMalloc(&de, n * 8);
    hipMemcpy(dt, ht, n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dt, ds, de, n);
    hipMemcpy(hs, ds, n * 8, hipMemcpyDeviceToHost); hipMemcpy(he, de, n * 8, hipMemcpyDeviceToHost);
    double ms = 0, me = 0, mx = 0, m40 = 0; int bad = 0;      // mx: the exponential alone (against expl of the device's own s); m40: e for s <= 40
    for (int i = 0; i < n; ++i) {
        long double rs = sqrtl((long double)ht[i]), re = expl(-rs);
        if (!(hs[i] == hs[i]) || !(he[i] == he[i])) { if (bad++ < 5) printf("NaN at t=%g s=%g e=%g\n", ht[i], hs[i], he[i]); continue; }
        if (ht[i] > 1e-290) { double es = fabsl((hs[i] - rs) / rs); if (es > ms) ms = es; }
        if (re > 1e-290L) { double ee = fabsl((he[i] - re) / re); if (ee > me) me = ee; if (rs <= 40.0L && ee > m40) m40 = ee; }
        if (std::isfinite(hs[i]) && hs[i] < 700.0) { long double rx = expl(-(long double)hs[i]); double ex = fabsl((he[i] - rx) / rx); if (ex > mx) mx = ex; }
        else if (he[i] > 1e-280) { if (bad++ < 5) printf("underflow expected at t=%g: e=%g\n", ht[i], he[i]); }
    }
    printf("exponential alone %.3g (%.2f ulp); e for s <= 40: %.3g (%.2f ulp)\n", mx, mx / 2.22e-16, m40, m40 / 2.22e-16);
    printf("max rel err: sqrt %.3g (%.2f ulp), exp(-sqrt) %.3g (%.2f ulp), bad %d; t=3e-300 -> s=%g e=%g; t=1e300 -> s=%g e=%g\n", ms, ms / 2.22e-16, me, me / 2.22e-16, bad, hs[0], he[0], hs[1], he[1]);
    return 0;
}
