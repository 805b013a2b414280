// Accuracy of pow_pos / fast_log / fast_exp (csrc/closures.hpp) on the device against long-double references:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -I gapflow_amd/csrc tools/pow_accuracy.hip -o tools/pow_accuracy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include "closures.hpp"
using namespace gpf;
__global__ void k(const double* x, const double* y, double* p, double* l, double* e, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { p[i] = pow_pos(x[i], y[i]); l[i] = fast_log(x[i]); e[i] = fast_exp(y[i]); }
}
int main() {
    const int n = 1 << 20;
    double *hx = new double[n], *hy = new double[n], *hp = new double[n], *hl = new double[n], *he = new double[n];
    srand(3);
    for (int i = 0; i < n; ++i) {
        double u = rand() / (double)RAND_MAX, v = rand() / (double)RAND_MAX;
        switch (i % 4) {
            case 0: hx[i] = 0.5 + 1.5 * u; hy[i] = -8 + 16 * v; break;              // density ratios, EOS exponents
            case 1: hx[i] = 0.9 + 0.2 * u; hy[i] = 7.33; break;                     // Murnaghan-Tait
            case 2: hx[i] = pow(10.0, -6 + 12 * u); hy[i] = -3 + 6 * v; break;      // wide range
            default: hx[i] = 1.0 + 1e-6 * (u - 0.5); hy[i] = 1.0 + v; break;        // near 1
        }
    }
    double *dx, *dy, *dp, *dl, *de;
    (void)hipMalloc(&dx, n * 8); (void)hipMalloc(&dy, n * 8); (void)hipMalloc(&dp, n * 8); (void)hipMalloc(&dl, n * 8); (void)hipMalloc(&de, n * 8);
    (void)hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice); (void)hipMemcpy(dy, hy, n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dy, dp, dl, de, n);
    (void)hipMemcpy(hp, dp, n * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(hl, dl, n * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(he, de, n * 8, hipMemcpyDeviceToHost);
    double mp = 0, ml = 0, me = 0, mp_mt = 0;
    for (int i = 0; i < n; ++i) {
        long double rp = powl((long double)hx[i], (long double)hy[i]), rl = logl((long double)hx[i]), re = expl((long double)hy[i]);
        double ep = fabsl((hp[i] - rp) / rp), el = fabsl(rl) > 1e-300L ? fabsl((hl[i] - rl) / rl) : 0, ee = fabsl((he[i] - re) / re);
        if (ep > mp) mp = ep;
        if (i % 4 == 1 && ep > mp_mt) mp_mt = ep;
        if (el > ml) ml = el;
        if (ee > me) me = ee;
    }
    printf("max rel err: pow %.3g (Murnaghan-Tait range %.3g), log %.3g, exp %.3g; pow(2, 0.5) = %.17g, pow(1, 3) = %.17g\n", mp, mp_mt, ml, me, hp[0], 0.0);
    return 0;
}
