// Shader clock seen by a tiny (one-workgroup) kernel vs a chip-filling one: hipcc --offload-arch=gfx950 -O2 tools/clock_probe.hip -o tools/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double* out, long long* t, int n) {
    double x = threadIdx.x * 1e-9 + 1.0;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < n; ++i) x = fma(x, 1.0000001, 1e-12);
    const long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
int main() {
    double* out; long long* t; long long h[2];
    hipMalloc(&out, 8 * 1024 * 4096); hipMalloc(&t, 16);
    for (int blocks : {1, 1, 4096}) {
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, t, 2000000);
            hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
            printf("blocks=%d: %lld shader cycles in %.2f ms -> %.0f MHz, %.1f cycles per dependent fp64 FMA\n", blocks, h[0], h[1] / 1e5,
                   h[0] / (h[1] / 100.0), h[0] / 2e6);
        }
    }
    return 0;
}
