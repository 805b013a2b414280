// Feasibility probe: two processes on ONE GPU exchange data through IPC-mapped device memory, signalled by flags that
// a kernel polls (bounded).  Build: hipcc --offload-arch=gfx950 -O2 tools/ipc_probe.hip -o tools/ipc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <sys/wait.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "[%d] %s -> %s\n", getpid(), #x, hipGetErrorString(e_)); _exit(3); } } while (0)

struct Box { unsigned long long flag[2]; double data[2][1024]; };

// send: write payload + flag into the PEER's box; wait: poll MY box (bounded), copy out
__global__ void k_send(Box* peer, int slot, unsigned long long seq, double base) {
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) peer->data[slot][i] = base + i;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&peer->flag[slot], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_wait(Box* mine, int slot, unsigned long long seq, double* out, int* timeout) {
    __shared__ int ok;
    if (threadIdx.x == 0) {
        const long long t0 = wall_clock64();
        ok = 0;
        while (wall_clock64() - t0 < 500000000ll) {          // 5 s at 100 MHz
            if (__hip_atomic_load(&mine->flag[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) >= seq) { ok = 1; break; }
            __builtin_amdgcn_s_sleep(8);
        }
        if (!ok) *timeout = 1;
    }
    __syncthreads();
    if (ok) for (int i = threadIdx.x; i < 1024; i += blockDim.x) out[i] = mine->data[slot][i];
}

static int run(int me, int rfd, int wfd, int finegrained) {
    CK(hipSetDevice(0));
    Box* mine = nullptr;
    if (finegrained) CK(hipExtMallocWithFlags((void**)&mine, sizeof(Box), hipDeviceMallocFinegrained));
    else CK(hipMalloc((void**)&mine, sizeof(Box)));
    CK(hipMemset(mine, 0, sizeof(Box)));
    CK(hipDeviceSynchronize());
    hipIpcMemHandle_t hm, hp;
    CK(hipIpcGetMemHandle(&hm, mine));
    if (write(wfd, &hm, sizeof(hm)) != sizeof(hm)) return 4;
    if (read(rfd, &hp, sizeof(hp)) != sizeof(hp)) return 4;
    Box* peer = nullptr;
    CK(hipIpcOpenMemHandle((void**)&peer, hp, hipIpcMemLazyEnablePeerAccess));
    double* out; int* to;
    CK(hipMalloc((void**)&out, 8192)); CK(hipMalloc((void**)&to, 4)); CK(hipMemset(to, 0, 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    const int N = 2000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // ping-pong: both send seq k, both wait seq k
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0, s));
        for (int k = 1; k <= N; ++k) {
            const unsigned long long seq = (unsigned long long)rep * N + k;
            hipLaunchKernelGGL(k_send, dim3(1), dim3(256), 0, s, peer, k & 1, seq, (double)(me * 1000000 + seq));
            hipLaunchKernelGGL(k_wait, dim3(1), dim3(256), 0, s, mine, k & 1, seq, out, to);
        }
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        int h_to; double h_out[2];
        CK(hipMemcpy(&h_to, to, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(h_out, out, 16, hipMemcpyDeviceToHost));
        const double expect = (double)((1 - me) * 1000000 + (unsigned long long)rep * N + N);
        printf("[rank %d fine=%d rep %d] %d exchanges in %.2f ms = %.2f us each, timeout=%d, last payload %s (%.0f vs %.0f)\n",
               me, finegrained, rep, N, ms, ms * 1e3 / N, h_to, h_out[0] == expect ? "OK" : "WRONG", h_out[0], expect);
    }
    CK(hipIpcCloseMemHandle(peer));
    return 0;
}

int main(int argc, char** argv) {
    const int fine = argc > 1 ? atoi(argv[1]) : 1;
    int a[2], b[2];
    if (pipe(a) || pipe(b)) return 1;
    pid_t pid = fork();                 // before any HIP call
    if (pid == 0) _exit(run(1, a[0], b[1], fine));
    int rc = run(0, b[0], a[1], fine);
    int st = 0; waitpid(pid, &st, 0);
    return rc | (WIFEXITED(st) ? WEXITSTATUS(st) : 9);
}
