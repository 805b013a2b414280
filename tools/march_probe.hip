// Memory-side probe of the step kernel's access pattern: waves that MARCH down the rows of a 4096^2 fp64 grid reading
// NIN planes and writing NOUT planes, with no arithmetic to speak of.  Which part of the pattern costs bandwidth?
//   hipcc --offload-arch=gfx950 -O3 tools/march_probe.hip -o tools/march_probe && tools/march_probe
// Every variant reports TB/s of algorithmic bytes ((NIN + NOUT) * 8 B * Nx * Ny per launch).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Args {
    const double* in; double* out;
    long long plane;        // doubles per plane
    int Nx, Ny, pitch;      // rows, columns, row pitch in doubles (multiple of 16)
    int col0;               // element offset of the first window in a row
    int stride;             // columns between the windows of neighbouring strips (126: overlapping by 2; 128: disjoint)
    int nstrips, nchunks;
    int xcd_map;            // 1: contiguous wave range per XCD
    int nt;                 // 1: nontemporal stores, 2: nontemporal loads, 3: both
};

// W = doubles per lane per access (2: 16-byte accesses, 1: 8-byte accesses), DEPTH rows in flight ahead
template <int NIN, int NOUT, int DEPTH, int W, int MINW>
__global__ __launch_bounds__(256, MINW) void k_march(const Args a) {
    const int nb = gridDim.x;
    const int lb = a.xcd_map ? (int)(blockIdx.x & 7) * (nb >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int w = lb * 4 + wv;
    const int strip = w % a.nstrips, chunk = w / a.nstrips;
    if (chunk >= a.nchunks) return;
    const int n0 = (int)(((long long)chunk * a.Nx) / a.nchunks), n1 = (int)(((long long)(chunk + 1) * a.Nx) / a.nchunks);
    int col = a.col0 + strip * a.stride + W * lane;
    const bool valid = col + W <= a.pitch;
    if (!valid) col = 0;
    const unsigned lane_bytes = (unsigned)col * 8u;
    typedef double vec __attribute__((ext_vector_type(W)));
    vec buf[DEPTH + 1][NIN];
    auto load = [&](int n, vec* r) {
        const long long rb = (long long)n * a.pitch;
        for (int p = 0; p < NIN; ++p) {
            const vec* src = reinterpret_cast<const vec*>(reinterpret_cast<const char*>(a.in + p * a.plane + rb) + lane_bytes);
            r[p] = (a.nt & 2) ? __builtin_nontemporal_load(src) : *src;
        }
    };
    auto row = [&](int n, vec* cur) {
        vec s = cur[0];
        for (int p = 1; p < NIN; ++p) s += cur[p];
        const long long rb = (long long)n * a.pitch;
        if (valid && n >= n0 + 1 && n <= n1) {          // halo rows n0 and n1+1 are read, not written
            for (int p = 0; p < NOUT; ++p) {
                vec* dst = reinterpret_cast<vec*>(reinterpret_cast<char*>(a.out + p * a.plane + rb) + lane_bytes);
                if (a.nt & 1) __builtin_nontemporal_store(s + (double)p, dst);
                else *dst = s + (double)p;
            }
        }
        if (n + DEPTH + 1 <= n1 + 1) load(n + DEPTH + 1, cur);
    };
    for (int d = 0; d <= DEPTH; ++d) load(n0 + d, buf[d]);
    for (int n = n0;;) {
        row(n, buf[0]); if (++n > n1 + 1) break;
        if (DEPTH >= 1) { row(n, buf[1 % (DEPTH + 1)]); if (++n > n1 + 1) break; }
        if (DEPTH >= 2) { row(n, buf[2 % (DEPTH + 1)]); if (++n > n1 + 1) break; }
        if (DEPTH >= 3) { row(n, buf[3 % (DEPTH + 1)]); if (++n > n1 + 1) break; }
    }
}

// plain streaming kernel with the same byte count: the reference point ("a += b" shape generalised)
template <int NIN, int NOUT>
__global__ __launch_bounds__(256) void k_stream(const Args a) {
    const long long n2 = a.plane / 2;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n2; i += (long long)gridDim.x * 256) {
        double2 s = reinterpret_cast<const double2*>(a.in)[i];
        for (int p = 1; p < NIN; ++p) { const double2 v = reinterpret_cast<const double2*>(a.in + p * a.plane)[i]; s.x += v.x; s.y += v.y; }
        for (int p = 0; p < NOUT; ++p) reinterpret_cast<double2*>(a.out + p * a.plane)[i] = make_double2(s.x + p, s.y + p);
    }
}

template <class F>
static double time_ms(F launch, int reps = 20) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

template <int NIN, int NOUT, int DEPTH, int W, int MINW>
static void run(const char* name, Args a, int waves_per_simd, int stride, int xcd, int nt = 0, int chunk_mult = 1) {
    a.stride = stride; a.xcd_map = xcd; a.nt = nt;
    const int cols_per_wave = 64 * W;
    a.nstrips = (a.Ny + 2 + (stride - 1)) / stride;
    if ((a.nstrips - 1) * stride + cols_per_wave > a.pitch - a.col0) a.nstrips = (a.pitch - a.col0 - cols_per_wave) / stride + 1;
    const int resident = 256 * 4 * waves_per_simd;
    a.nchunks = (resident / a.nstrips) * chunk_mult;
    const int nwaves = a.nstrips * a.nchunks;
    const int nblocks = (((nwaves + 3) / 4) + 7) / 8 * 8;
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)k_march<NIN, NOUT, DEPTH, W, MINW>, 256, 0));
    const double ms = time_ms([&] { hipLaunchKernelGGL((k_march<NIN, NOUT, DEPTH, W, MINW>), dim3(nblocks), dim3(256), 0, 0, a); });
    const double bytes = (double)(NIN + NOUT) * 8.0 * a.Nx * a.Ny;
    printf("%-58s strips %3d chunks %4d (%3d rows) occ %d blk/CU: %7.1f us  %5.2f TB/s\n", name, a.nstrips, a.nchunks, a.Nx / a.nchunks, occ,
           ms * 1e3, bytes / (ms * 1e-3) / 1e12);
    fflush(stdout);
}

int main() {
    Args a;
    a.Nx = 4096; a.Ny = 4096;
    a.pitch = ((a.Ny + 2 + 15 + 15) / 16) * 16;
    a.plane = (long long)(a.Nx + 2) * a.pitch;
    a.col0 = 14;
    double *in, *out;
    CK(hipMalloc(&in, 6 * a.plane * 8 + 4096)); CK(hipMalloc(&out, 3 * a.plane * 8 + 4096));
    CK(hipMemset(in, 0, 6 * a.plane * 8)); CK(hipMemset(out, 0, 3 * a.plane * 8));
    a.in = in; a.out = out;
    a.Nx = 4096;
    {
        const double ms = time_ms([&] { hipLaunchKernelGGL((k_stream<6, 3>), dim3(4096), dim3(256), 0, 0, a); });
        printf("%-58s %7.1f us  %5.2f TB/s\n", "stream 6 in / 3 out (grid-stride, 16 B per lane)", ms * 1e3, 9.0 * 8 * a.plane / (ms * 1e-3) / 1e12);
        const double ms3 = time_ms([&] { hipLaunchKernelGGL((k_stream<3, 3>), dim3(4096), dim3(256), 0, 0, a); });
        printf("%-58s %7.1f us  %5.2f TB/s\n", "stream 3 in / 3 out", ms3 * 1e3, 6.0 * 8 * a.plane / (ms3 * 1e-3) / 1e12);
    }
    // ---- the kernel's pattern and one change at a time ----
    run<6, 3, 2, 2, 2>("march 6/3, 16 B, depth 2, 2 waves/SIMD, stride 126, xcd", a, 2, 126, 1);
    run<6, 3, 2, 2, 2>("  ... no XCD-ordered numbering", a, 2, 126, 0);
    run<6, 3, 2, 2, 2>("  ... disjoint windows (stride 128, 128-B aligned)", [&] { Args b = a; b.col0 = 16; return b; }(), 2, 128, 1);
    run<6, 3, 1, 2, 2>("  ... depth 1", a, 2, 126, 1);
    run<6, 3, 3, 2, 2>("  ... depth 3", a, 2, 126, 1);
    run<6, 3, 2, 2, 4>("  ... 4 waves/SIMD (half as many rows per chunk)", a, 4, 126, 1);
    run<6, 3, 1, 2, 4>("  ... 4 waves/SIMD, depth 1", a, 4, 126, 1);
    run<6, 3, 2, 2, 8>("  ... 8 waves/SIMD", a, 8, 126, 1);
    run<6, 3, 2, 2, 2>("  ... nontemporal stores", a, 2, 126, 1, 1);
    run<6, 3, 2, 2, 2>("  ... nontemporal loads", a, 2, 126, 1, 2);
    run<6, 3, 2, 2, 2>("  ... nontemporal loads + stores", a, 2, 126, 1, 3);
    run<6, 3, 2, 2, 2>("  ... 2 rounds of waves (chunks x2)", a, 2, 126, 1, 0, 2);
    run<6, 3, 2, 2, 2>("  ... 4 rounds of waves (chunks x4)", a, 2, 126, 1, 0, 4);
    run<6, 3, 2, 1, 4>("march 6/3, 8 B per lane, depth 2, 4 waves/SIMD, stride 62", a, 4, 62, 1);
    run<6, 3, 2, 1, 4>("  ... no XCD-ordered numbering", a, 4, 62, 0);
    run<3, 3, 2, 2, 2>("march 3/3 (x-only gap), 16 B, depth 2, 2 waves/SIMD", a, 2, 126, 1);
    run<3, 3, 3, 2, 2>("  ... depth 3", a, 2, 126, 1);
    run<3, 3, 2, 2, 4>("  ... 4 waves/SIMD", a, 4, 126, 1);
    run<3, 3, 3, 2, 4>("  ... 4 waves/SIMD, depth 3", a, 4, 126, 1);
    run<3, 3, 2, 2, 2>("  ... nontemporal stores", a, 2, 126, 1, 1);
    return 0;
}
