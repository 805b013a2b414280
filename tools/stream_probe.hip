// What does this device stream?  Elementwise kernels over fp64 planes of the benchmark's size (4096 x 4096 doubles = 134 MB each),
// NIN planes read and NOUT planes written, 16 bytes per lane and access, in every arrangement that could matter for the step
// kernel: bytes in flight per thread (UNROLL independent 16-byte loads per plane before the first use), grid size (resident
// waves), non-temporal loads / stores.  MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 COPY (1 in / 1 out); the fused step
// moves 3 in / 3 out (x-only gap) or 6 in / 3 out (2-D gap).  Output: one line per configuration, GB/s = (NIN + NOUT) x plane bytes / time.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/stream_probe tools/stream_probe.hip && tools/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));

template <int NIN, int NOUT, int U, bool NT>
__global__ __launch_bounds__(256) void k_stream(const d2* __restrict__ in, d2* __restrict__ out, long long n2) {
    const long long tile = 256ll * U;
    for (long long base = blockIdx.x * tile + threadIdx.x; base < n2; base += (long long)gridDim.x * tile) {
        d2 v[NIN][U];
#pragma unroll
        for (int p = 0; p < NIN; ++p)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const d2* a = in + p * n2 + base + u * 256;
                v[p][u] = NT ? __builtin_nontemporal_load(a) : *a;
            }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            d2 s = v[0][u];
#pragma unroll
            for (int p = 1; p < NIN; ++p) s += v[p][u];
#pragma unroll
            for (int p = 0; p < NOUT; ++p) {
                d2* a = out + p * n2 + base + u * 256;
                const d2 w = s + (double)p;
                if (NT) __builtin_nontemporal_store(w, a); else *a = w;
            }
        }
    }
}

template <int NIN, int NOUT, int U, bool NT>
static double run(const d2* in, d2* out, long long n2, int blocks, int reps) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_stream<NIN, NOUT, U, NT>), dim3(blocks), dim3(256), 0, 0, in, out, n2);
    CHECK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_stream<NIN, NOUT, U, NT>), dim3(blocks), dim3(256), 0, 0, in, out, n2);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0); hipEventDestroy(e1);
    return (NIN + NOUT) * (double)n2 * 16.0 / (ms / reps * 1e-3) / 1e9;
}

template <int NIN, int NOUT>
static void sweep(const d2* in, d2* out, long long n2, int ncu) {
    double best = 0.0;
    char best_cfg[96] = "";
    for (int per_cu : {2, 4, 8, 16, 32}) {
        const int blocks = ncu * per_cu;
        const double r[8] = {run<NIN, NOUT, 1, false>(in, out, n2, blocks, 10), run<NIN, NOUT, 2, false>(in, out, n2, blocks, 10),
                             run<NIN, NOUT, 4, false>(in, out, n2, blocks, 10), run<NIN, NOUT, 8, false>(in, out, n2, blocks, 10),
                             run<NIN, NOUT, 1, true>(in, out, n2, blocks, 10), run<NIN, NOUT, 2, true>(in, out, n2, blocks, 10),
                             run<NIN, NOUT, 4, true>(in, out, n2, blocks, 10), run<NIN, NOUT, 8, true>(in, out, n2, blocks, 10)};
        std::printf("%d in / %d out  blocks/CU %2d   plain U=1,2,4,8: %6.0f %6.0f %6.0f %6.0f   non-temporal: %6.0f %6.0f %6.0f %6.0f GB/s\n", NIN, NOUT, per_cu,
                    r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]);
        for (int i = 0; i < 8; ++i)
            if (r[i] > best) { best = r[i]; std::snprintf(best_cfg, sizeof best_cfg, "blocks/CU %d, U %d, %s", per_cu, 1 << (i & 3), i >= 4 ? "non-temporal" : "plain"); }
    }
    std::printf("%d in / %d out  BEST %6.0f GB/s = %.3f of 8 TB/s  (%s)\n\n", NIN, NOUT, best, best / 8000.0, best_cfg);
    std::fflush(stdout);
}

int main(int argc, char** argv) {
    const long long n = argc > 1 ? std::atoll(argv[1]) : 4096ll * 4096ll;      // doubles per plane
    const long long n2 = n / 2;
    int ncu = 0;
    CHECK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
    d2 *in = nullptr, *out = nullptr;
    CHECK(hipMalloc(&in, 6 * n2 * 16)); CHECK(hipMalloc(&out, 3 * n2 * 16));
    CHECK(hipMemset(in, 0, 6 * n2 * 16)); CHECK(hipMemset(out, 0, 3 * n2 * 16));
    std::printf("plane = %lld doubles (%.0f MB), %d CUs\n", n, n * 8 / 1e6, ncu);
    sweep<1, 1>(in, out, n2, ncu);
    sweep<3, 3>(in, out, n2, ncu);
    sweep<6, 3>(in, out, n2, ncu);
    sweep<1, 3>(in, out, n2, ncu);      // write-heavy
    sweep<6, 1>(in, out, n2, ncu);      // read-heavy
    return 0;
}
