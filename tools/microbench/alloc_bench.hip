// The ONE bounded allocation experiment of round 2, then the topic is closed.
// Round 1 found that the K1 store stream runs 5.9 .. 7.1 TB/s depending on which physical memory hipMalloc handed
// out for the output (profiles/r01_store_microbench_12_allocation_quality.log).  Question: does another ALLOCATOR give
// reliably fast buffers?  For each allocator, 10 output buffers (18.9 GB each: dist + mask of the headline shape) are
// allocated one after the other and all held at once; on each, K1's store stream as the product writes it (one
// contiguous 144 KB run per short-lived workgroup, XCD-contiguous map, 225 active lanes, 16 B per lane) is timed
// (min of 3 rounds of 3 launches).  Allocators:
//   hipMalloc                                      (what torch.empty reaches through PyTorch's caching allocator)
//   hipExtMallocWithFlags: default / uncached / contiguous
//   VMM: hipMemCreate + hipMemAddressReserve + hipMemMap, ONE physical handle for the whole buffer
//   VMM with one handle per chunk of the recommended granularity x 512 (~1 GB) and per recommended-granularity chunk
// Prints TB/s per buffer and min / median / max per allocator.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(256) void k_stream(u32x4* __restrict__ d, u32x4* __restrict__ m, unsigned n) {
    if (threadIdx.x >= 225) return;
    const unsigned w = blockIdx.x;
    const unsigned c = (w & 7u) * (n >> 3) + (w >> 3);   // XCD-contiguous map
    u32x4 v = {threadIdx.x, c, 7, 9};
    u32x4* o = d + (size_t)c * (225 * 32) + threadIdx.x;
#pragma unroll
    for (int g = 0; g < 32; ++g) o[g * 225] = v;
    u32x4* om = m + (size_t)c * (225 * 8) + threadIdx.x;
#pragma unroll
    for (int g = 0; g < 8; ++g) om[g * 225] = v;
}

struct Buf {
    char* p = nullptr;
    size_t bytes = 0;
    int kind = 0;
    std::vector<hipMemGenericAllocationHandle_t> handles;
    size_t reserved = 0;
};

static size_t round_up(size_t x, size_t g) { return (x + g - 1) / g * g; }

static bool vmm_alloc(Buf& b, size_t bytes, size_t chunk, int dev) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || !gran) return false;
    if (chunk == 0) chunk = round_up(bytes, gran);      // one handle for everything
    chunk = round_up(chunk, gran);
    b.reserved = round_up(bytes, chunk);
    void* va = nullptr;
    if (hipMemAddressReserve(&va, b.reserved, 0, nullptr, 0) != hipSuccess) return false;
    b.p = static_cast<char*>(va);
    for (size_t off = 0; off < b.reserved; off += chunk) {
        hipMemGenericAllocationHandle_t h;
        if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) return false;
        b.handles.push_back(h);
        if (hipMemMap(b.p + off, chunk, 0, h, 0) != hipSuccess) return false;
    }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if (hipMemSetAccess(b.p, b.reserved, &acc, 1) != hipSuccess) return false;
    b.bytes = bytes;
    b.kind = 2;
    return true;
}

static void release(Buf& b) {
    if (b.kind == 1) CK(hipFree(b.p));
    if (b.kind == 2) {
        CK(hipMemUnmap(b.p, b.reserved));
        for (auto h : b.handles) CK(hipMemRelease(h));
        CK(hipMemAddressFree(b.p, b.reserved));
    }
    b = Buf();
}

int main() {
    const size_t dist_bytes = 64ull * 512 * 512 * 900, mask_bytes = dist_bytes / 4, total = dist_bytes + mask_bytes;
    const unsigned n_wg = (unsigned)(dist_bytes / (3600 * 32));   // 131072
    int dev = 0;
    CK(hipGetDevice(&dev));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gmin = 0, grec = 0;
    hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum);
    hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended);
    printf("VMM granularity: minimum %zu B, recommended %zu B\n", gmin, grec);
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    struct Alloc { std::string name; int mode; unsigned flag; size_t chunk; };
    std::vector<Alloc> allocs = {
        {"hipMalloc", 0, 0, 0},
        {"hipExtMalloc default", 1, hipDeviceMallocDefault, 0},
        {"hipExtMalloc uncached", 1, hipDeviceMallocUncached, 0},
        {"hipExtMalloc contiguous", 1, hipDeviceMallocContiguous, 0},
        {"VMM one handle", 2, 0, 0},
        {"VMM 512 x recommended", 2, 0, grec * 512},
        {"VMM recommended granule", 2, 0, grec},
    };
    const int NB = 10;
    for (auto& al : allocs) {
        std::vector<Buf> bufs;
        bool ok = true;
        for (int i = 0; i < NB && ok; ++i) {
            Buf bf;
            if (al.mode == 0) {
                ok = hipMalloc(&bf.p, total) == hipSuccess; bf.kind = 1; bf.bytes = total;
            } else if (al.mode == 1) {
                ok = hipExtMallocWithFlags(reinterpret_cast<void**>(&bf.p), total, al.flag) == hipSuccess; bf.kind = 1; bf.bytes = total;
            } else {
                ok = vmm_alloc(bf, total, al.chunk, dev);
            }
            if (ok) bufs.push_back(bf);
            else { (void)hipGetLastError(); if (bf.kind == 2 && bf.p) { /* partial VMM mapping: leave it, process ends soon */ } }
        }
        if (bufs.empty()) { printf("%-26s allocation failed\n", al.name.c_str()); continue; }
        std::vector<float> best(bufs.size(), 1e30f);
        for (int i = 0; i < 20; ++i)     // warm-up: clocks
            k_stream<<<n_wg, 256>>>((u32x4*)bufs[0].p, (u32x4*)(bufs[0].p + dist_bytes), n_wg);
        CK(hipDeviceSynchronize());
        for (int round = 0; round < 3; ++round)
            for (size_t i = 0; i < bufs.size(); ++i) {
                u32x4* d = (u32x4*)bufs[i].p; u32x4* m = (u32x4*)(bufs[i].p + dist_bytes);
                k_stream<<<n_wg, 256>>>(d, m, n_wg);
                CK(hipEventRecord(a));
                for (int r = 0; r < 3; ++r) k_stream<<<n_wg, 256>>>(d, m, n_wg);
                CK(hipEventRecord(b));
                CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b));
                best[i] = std::min(best[i], ms / 3);
            }
        std::vector<float> tb;
        printf("%-26s", al.name.c_str());
        for (float ms : best) { tb.push_back(total / ms / 1e9); printf(" %5.2f", tb.back()); }
        std::sort(tb.begin(), tb.end());
        printf("   | min %.2f  median %.2f  max %.2f TB/s (%zu buffers)\n", tb.front(), tb[tb.size() / 2], tb.back(), tb.size());
        fflush(stdout);
        for (auto& bf : bufs) release(bf);
    }
    return 0;
}
