// fp64 FMA on one SIMD: dependent-issue latency and how many independent chains (per wave x waves per SIMD) it takes
// to fill the pipe.   hipcc --offload-arch=gfx950 -O3 tools/fma_probe.hip -o tools/fma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CHAINS>
__global__ void k(double* out, long long* t, int n) {
    double x[CHAINS];
    for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x * 1e-9 + 1.0 + c;
    const long long c0 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) x[c] = fma(x[c], 1.0000001, 1e-12);
    }
    const long long c1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = c1 - c0;
    double s = 0;
    for (int c = 0; c < CHAINS; ++c) s += x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CHAINS>
static void run(double* out, long long* t) {
    const int n = 200000;
    for (int waves_per_simd : {1, 2, 4}) {
        long long h = 0;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        // one workgroup per CU on all 256 CUs: 4 * waves_per_simd waves of 64 lanes
        hipLaunchKernelGGL(k<CHAINS>, dim3(256), dim3(256 * waves_per_simd), 0, 0, out, t, n);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<CHAINS>, dim3(256), dim3(256 * waves_per_simd), 0, 0, out, t, n);
        (void)hipEventRecord(e1, 0);
        (void)hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        const double ns_per_fma_simd = ms * 1e6 / ((double)n * CHAINS * waves_per_simd);
        printf("chains/wave %d, waves/SIMD %d: %6.2f ns per wave-FMA on the SIMD, %6.1f ns per loop trip per wave (clock64: %.1f ticks per trip) -> %5.1f TFLOP/s fp64 chip-wide\n",
               CHAINS, waves_per_simd, ns_per_fma_simd, ms * 1e6 / n, (double)h / n, 128.0 * 1024 / ns_per_fma_simd / 1e3);
    }
}
int main() {
    double* out; long long* t;
    (void)hipMalloc(&out, 8 * 1024 * 1024); (void)hipMalloc(&t, 16);
    run<1>(out, t); run<2>(out, t); run<4>(out, t); run<8>(out, t);
    return 0;
}
