// Does HIP's virtual memory management keep buffers apart under the allocation pattern of gapflow_amd's scattered fields
// (csrc/api_fields.inc: field_malloc)?  Buffers = one reserved address range each, backed by separately created 16-MiB pieces mapped in
// a shuffled order; set A and set B are created and filled with patterns, A is freed, set C is created and filled; then B and C
// are read back.      hipcc --offload-arch=gfx950 -O2 tools/vmm_probe.hip -o /tmp/vmm_probe && /tmp/vmm_probe [unmap-whole]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>
struct Buf { void* va; size_t bytes, part; std::vector<hipMemGenericAllocationHandle_t> parts; unsigned long long seed; };
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("FAILED %s: %s\n", #x, hipGetErrorString(e_)); std::exit(2); } } while (0)
static Buf make(size_t bytes, size_t part, unsigned long long seed) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    part = (part + gran - 1) / gran * gran;
    const size_t n = (bytes + part - 1) / part;
    Buf b; b.bytes = n * part; b.part = part; b.seed = seed;
    CK(hipMemAddressReserve(&b.va, b.bytes, 0, nullptr, 0));
    b.parts.resize(n);
    for (size_t k = 0; k < n; ++k) CK(hipMemCreate(&b.parts[k], part, &prop, 0));
    std::vector<size_t> order(n);
    for (size_t k = 0; k < n; ++k) order[k] = k;
    unsigned long long x = 0x9e3779b97f4a7c15ull ^ seed;
    for (size_t k = n - 1; k > 0; --k) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; std::swap(order[k], order[x % (k + 1)]); }
    for (size_t k = 0; k < n; ++k) CK(hipMemMap((char*)b.va + k * part, part, 0, b.parts[order[k]], 0));
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(b.va, b.bytes, &acc, 1));
    return b;
}
static bool keep_va = false;
static void destroy(Buf& b, bool whole) {
    if (whole) CK(hipMemUnmap(b.va, b.bytes));
    else for (size_t k = 0; k < b.parts.size(); ++k) CK(hipMemUnmap((char*)b.va + k * b.part, b.part));
    for (auto& h : b.parts) CK(hipMemRelease(h));
    if (!keep_va) CK(hipMemAddressFree(b.va, b.bytes));
}
__global__ void fill(unsigned long long* p, size_t n, unsigned long long seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = seed * 0x100000001b3ull + i;
}
__global__ void check(const unsigned long long* p, size_t n, unsigned long long seed, unsigned long long* bad) {
    unsigned long long c = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += p[i] != seed * 0x100000001b3ull + i;
    if (c) atomicAdd(bad, c);
}
int main(int argc, char** argv) {
    bool whole = false;
    int rounds = 4;
    for (int i = 1; i < argc; ++i) {
        if (!std::strcmp(argv[i], "unmap-whole")) whole = true;
        if (!std::strcmp(argv[i], "keep-va")) keep_va = true;
    }
    const size_t bytes = 102ull << 20, part = 16ull << 20;
    unsigned long long* bad; CK(hipMalloc(&bad, 8));
    unsigned long long seed = 1;
    int nbad = 0, nchecked = 0;
    std::vector<Buf> live;                                                  // fields of the handles alive
    auto verify = [&](std::vector<Buf>& set, const char* name) {
        for (auto& b : set) {
            CK(hipMemset(bad, 0, 8));
            check<<<1024, 256>>>((const unsigned long long*)b.va, b.bytes / 8, b.seed, bad);
            unsigned long long h = 0; CK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
            ++nchecked;
            if (h) { std::printf("%s buffer at %p (seed %llu): %llu wrong words of %zu\n", name, b.va, b.seed, h, b.bytes / 8); ++nbad; }
        }
    };
    for (int r = 0; r < rounds; ++r) {
        std::vector<Buf> fields, pool;
        for (int k = 0; k < 3; ++k) fields.push_back(make(bytes, part, seed++));   // a new handle's fields
        for (auto& b : fields) fill<<<1024, 256>>>((unsigned long long*)b.va, b.bytes / 8, b.seed);
        // an older handle tunes: master and pool come and go
        for (int k = 0; k < 11; ++k) pool.push_back(make(bytes, part, seed++));
        for (auto& b : pool) fill<<<1024, 256>>>((unsigned long long*)b.va, b.bytes / 8, b.seed);
        CK(hipDeviceSynchronize());
        verify(pool, "pool");
        for (size_t k = 2; k < pool.size(); ++k) destroy(pool[k], whole);
        live.push_back(pool[0]); live.push_back(pool[1]);
        for (auto& b : fields) live.push_back(b);
        verify(live, "live");
    }
    std::printf("%s: %d of %d checks found a damaged buffer (%s, %s)\n", nbad ? "BROKEN" : "ok", nbad, nchecked,
                whole ? "one hipMemUnmap per buffer" : "one hipMemUnmap per piece", keep_va ? "address ranges kept" : "address ranges freed");
    return nbad != 0;
}
