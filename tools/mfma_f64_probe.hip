// What the f64 matrix pipe of gfx950 sustains: v_mfma_f64_16x16x4_f64 in a register-only loop, NACC independent accumulators
// per wave, W waves per SIMD.  hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_probe.hip -o tools/mfma_f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC, int VALU_PER_MFMA>
__global__ void k_probe(double* out, int iters, double a0, double b0) {
    v4d acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = v4d{0.0, 0.0, 0.0, 0.0};
    double a = a0 + threadIdx.x, b = b0 + threadIdx.x * 0.5, x = a0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < VALU_PER_MFMA; ++v) x = fma(x, 1.0000001, 0.5);     // independent fp64 VALU work in the MFMA's shadow
        }
    }
    double s = x;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, int V>
static void run(int waves_per_simd, double* out) {
    const int iters = 2000;
    const int threads = 256, blocks = 256 * waves_per_simd;         // 4 waves per block = 1 per SIMD
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_probe<NACC, V>), dim3(blocks), dim3(threads), 0, 0, out, 10, 1.0, 2.0);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_probe<NACC, V>), dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0, 2.0);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double mfmas = (double)blocks * 4 * iters * NACC;
    const double cyc_per = ms * 1e-3 * 2.4e9 / (mfmas / 1024.0);
    printf("accumulators %2d, %2d fp64 FMA between MFMAs, %d waves/SIMD: %7.2f TFLOP/s (matrix), ~%5.1f cycles per MFMA and SIMD at 2.4 GHz\n", NACC, V,
           waves_per_simd, mfmas * 2048.0 / (ms * 1e-3) / 1e12, cyc_per);
}

int main() {
    double* out; CK(hipMalloc(&out, 256 * 8 * 256 * 8));
    for (int w : {1, 2, 4}) { run<16, 0>(w, out); run<8, 0>(w, out); run<4, 0>(w, out); run<2, 0>(w, out); run<1, 0>(w, out); }
    for (int w : {1, 2, 4}) { run<8, 2>(w, out); run<8, 6>(w, out); run<8, 12>(w, out); }
    return 0;
}
