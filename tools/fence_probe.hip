// What does an agent-/system-scope fence cost in a small kernel that follows a big store-heavy one?
// hipcc --offload-arch=gfx950 -O2 tools/fence_probe.hip -o tools/fence_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_dirty(double* p, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = (double)i;
}
template <int MODE>
__global__ void k_fence(double* out, int reps, unsigned* ctr) {
    double v = threadIdx.x;
    for (int r = 0; r < reps; ++r) {
        out[blockIdx.x * 256 + threadIdx.x] = v + r;
        if (MODE == 1) __threadfence();
        if (MODE == 2) __threadfence_system();
        if (MODE == 3) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (MODE == 4) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (MODE == 5) { if (threadIdx.x == 0) atomicAdd(ctr, 1u); }
        if (MODE == 6) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    }
}
template <int MODE>
static int run(const char* name, double* big, long long n, double* out, unsigned* ctr, bool dirty) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int reps : {1, 9}) {
        float best = 1e9f;
        for (int it = 0; it < 5; ++it) {
            if (dirty) hipLaunchKernelGGL(k_dirty, dim3(4096), dim3(256), 0, 0, big, n);
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_fence<MODE>, dim3(34), dim3(256), 0, 0, out, reps, ctr);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("%-28s dirty=%d reps=%d: %.1f us\n", name, (int)dirty, reps, best * 1e3);
    }
    return 0;
}
int main() {
    const long long n = 50ll << 20;          // 400 MB
    double *big, *out; unsigned* ctr;
    CK(hipMalloc((void**)&big, n * 8)); CK(hipMalloc((void**)&out, 34 * 256 * 8)); CK(hipMalloc((void**)&ctr, 4));
    CK(hipMemset(ctr, 0, 4));
    for (int d = 0; d < 2; ++d) {
        run<0>("no fence", big, n, out, ctr, d);
        run<1>("__threadfence (agent)", big, n, out, ctr, d);
        run<2>("__threadfence_system", big, n, out, ctr, d);
        run<3>("release agent", big, n, out, ctr, d);
        run<4>("acquire agent", big, n, out, ctr, d);
        run<5>("atomicAdd", big, n, out, ctr, d);
        run<6>("release workgroup", big, n, out, ctr, d);
    }
    return 0;
}
