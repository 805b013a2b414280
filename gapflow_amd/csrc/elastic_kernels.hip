// Elastic half-space deformation of the gap (GaPFlow/topography.py:257-280, 404-437): the convolution of the film
// pressure with the half-space Green's function on the device.  The Green's function arrives in Fourier space from the
// host (gapflow_amd/elastic.py); the transforms are hipFFT's (a plain library FFT, loaded with dlopen like rocBLAS), the
// rest are the small kernels below.  Not HBM- or MFMA-critical: two real-to-complex transforms of the (doubled) grid per
// time step next to a stage-wise step of ~15 passes.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>

namespace gpf {

struct FftLib {
    typedef int (*plan2d_t)(void**, int, int, int);
    typedef int (*exec_d2z_t)(void*, double*, double2*);
    typedef int (*exec_z2d_t)(void*, double2*, double*);
    typedef int (*set_stream_t)(void*, hipStream_t);
    typedef int (*destroy_t)(void*);
    plan2d_t plan2d = nullptr; exec_d2z_t d2z = nullptr; exec_z2d_t z2d = nullptr; set_stream_t set_stream = nullptr; destroy_t destroy = nullptr;
    bool ok = false;
    const char* err = "";
};
enum { HIPFFT_D2Z_ = 0x6a, HIPFFT_Z2D_ = 0x6c };     // hipfft.h: hipfftType

inline FftLib& fftlib() {
    static FftLib F;
    static bool tried = false;
    if (tried) return F;
    tried = true;
    void* hd = nullptr;
    hd = dlopen("libhipfft.so.0", RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);                         // an image already mapped wins (roclibs())
    if (!hd) if (const char* path = getenv("GPF_HIPFFT_PATH")) hd = dlopen(path, RTLD_NOW | RTLD_GLOBAL);   // the copy PyTorch bundles
    if (!hd) hd = dlopen("libhipfft.so.0", RTLD_NOW | RTLD_GLOBAL);
    if (!hd) hd = dlopen("libhipfft.so", RTLD_NOW | RTLD_GLOBAL);
    if (!hd) hd = dlopen("/opt/rocm/lib/libhipfft.so", RTLD_NOW | RTLD_GLOBAL);
    if (!hd) { F.err = "could not dlopen hipFFT"; return F; }
    F.plan2d = (FftLib::plan2d_t)dlsym(hd, "hipfftPlan2d");
    F.d2z = (FftLib::exec_d2z_t)dlsym(hd, "hipfftExecD2Z");
    F.z2d = (FftLib::exec_z2d_t)dlsym(hd, "hipfftExecZ2D");
    F.set_stream = (FftLib::set_stream_t)dlsym(hd, "hipfftSetStream");
    F.destroy = (FftLib::destroy_t)dlsym(hd, "hipfftDestroy");
    F.ok = F.plan2d && F.d2z && F.z2d && F.set_stream && F.destroy;
    if (!F.ok) F.err = "hipFFT symbols missing";
    return F;
}

// forces = (p - p_ref) on the (Nx+2) x (Ny+2) corner of the transform grid, zero elsewhere (the doubled part)
__global__ void k_el_pack(const double* p, Layout L, int px, int py, int relative, double* dense) {
    const long long n = (long long)px * py;
    const double pref = relative ? p[L.at(0, 0)] : 0.0;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int ix = (int)(i / py), iy = (int)(i % py);
        dense[i] = (ix < L.Nx + 2 && iy < L.Ny + 2) ? p[L.at(ix, iy)] - pref : 0.0;
    }
}

__global__ void k_el_multiply(double2* f, const double2* g, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const double2 a = f[i], b = g[i];
        f[i] = make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
    }
}

// u_relaxed = (1 - alpha) u_prev + alpha u_computed  (topography.py:419-437), u_computed = scale * inverse transform
__global__ void k_el_relax(const double* dense, int py, double scale, double alpha, Layout L, double* u_prev) {
    const long long w = L.Ny + 2, n = (long long)(L.Nx + 2) * w;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int ix = (int)(i / w), iy = (int)(i % w);
        const long long o = L.at(ix, iy);
        u_prev[o] = (1.0 - alpha) * u_prev[o] + alpha * (scale * dense[(long long)ix * py + iy]);
    }
}

// deformation = u_relaxed (minus its value at [0, 0] unless fully periodic); h = h_undeformed + deformation
__global__ void k_el_apply(const double* u_prev, const double* h0, int relative, Layout L, double* deformation, double* h) {
    const long long w = L.Ny + 2, n = (long long)(L.Nx + 2) * w;
    const double uref = relative ? u_prev[L.at(0, 0)] : 0.0;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long o = L.at((int)(i / w), (int)(i % w));
        const double d = u_prev[o] - uref;
        deformation[o] = d;
        h[o] = h0[o] + d;
    }
}

// np.gradient(h, axis) / spacing: central differences, one-sided at the two ends of the array (ghost cells included;
// topography.py:273-280).  A one-cell axis has no gradient in the reference either (np.gradient needs two points): the
// solver never gets there because Nx, Ny >= 1 means at least three points with the ghost cells.
__global__ void k_el_gradient(const double* h, Layout L, double inv_dx, double inv_dy, double* hx, double* hy) {
    const long long w = L.Ny + 2, n = (long long)(L.Nx + 2) * w;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int ix = (int)(i / w), iy = (int)(i % w);
        const int xm = ix > 0 ? ix - 1 : ix, xp = ix < L.Nx + 1 ? ix + 1 : ix;
        const int ym = iy > 0 ? iy - 1 : iy, yp = iy < L.Ny + 1 ? iy + 1 : iy;
        const long long o = L.at(ix, iy);
        hx[o] = (h[L.at(xp, iy)] - h[L.at(xm, iy)]) / (double)(xp - xm) * inv_dx;
        hy[o] = (h[L.at(ix, yp)] - h[L.at(ix, ym)]) / (double)(yp - ym) * inv_dy;
    }
}

}  // namespace gpf
