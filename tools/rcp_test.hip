// accuracy of v_rcp_f64 + k Newton steps against IEEE division (diagnostic, not part of the library)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdint>
#include <cstring>
__global__ void k(const double* x, double* r0, double* r1, double* r2, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = x[i];
    double r = __builtin_amdgcn_rcp(v);
    r0[i] = r;
    double e = fma(-v, r, 1.0); r = fma(r, e, r);
    r1[i] = r;
    e = fma(-v, r, 1.0); r = fma(r, e, r);
    r2[i] = r;
}
static double ulps(double a, double b) { int64_t x, y; memcpy(&x, &a, 8); memcpy(&y, &b, 8); return (double)llabs(x - y); }
int main() {
    const int n = 1 << 20;
    double* hx = new double[n];
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        double u = (s >> 11) * (1.0 / 9007199254740992.0);
        hx[i] = ldexp(1.0 + u, (int)(s % 120) - 60);
    }
    double *dx, *d0, *d1, *d2;
    hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
    hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, d0, d1, d2, n);
    double *h0 = new double[n], *h1 = new double[n], *h2 = new double[n];
    hipMemcpy(h0, d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(h1, d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(h2, d2, n * 8, hipMemcpyDeviceToHost);
    double m0 = 0, m1 = 0, m2 = 0, rel0 = 0;
    for (int i = 0; i < n; ++i) {
        double t = 1.0 / hx[i];
        m0 = fmax(m0, ulps(h0[i], t)); m1 = fmax(m1, ulps(h1[i], t)); m2 = fmax(m2, ulps(h2[i], t));
        rel0 = fmax(rel0, fabs(h0[i] - t) / t);
    }
    printf("v_rcp_f64: max rel err %.3e (%.0f ulp); +1 NR: %.0f ulp; +2 NR: %.0f ulp\n", rel0, m0, m1, m2);
    return 0;
}
