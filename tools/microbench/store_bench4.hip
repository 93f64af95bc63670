// Store-pattern microbenchmark 4: is there XCD <-> 4KB-chunk affinity?
// WG k (dealt round-robin to XCD k%8) writes 4 KB chunk c = (k & ~7) | ((k + S) & 7) for S = 0..7.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int GRAN>  // chunk = GRAN * 4 KB, rotation among 8 consecutive chunks
__global__ __launch_bounds__(256) void kX(u32x4* __restrict__ d, unsigned S) {
    const unsigned k = blockIdx.x;
    const unsigned c = (k & ~7u) | ((k + S) & 7u);
    u32x4 v = {threadIdx.x, k, 1, 2};
    u32x4* o = d + (size_t)c * (256 * GRAN) + threadIdx.x;
#pragma unroll
    for (int g = 0; g < GRAN; ++g) o[g * 256] = v;
}
// XCC id census: which XCD runs WG k?
__global__ void kCensus(unsigned* out) {
    if (threadIdx.x == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        out[blockIdx.x] = x & 0xF;
    }
}
// persistent, XCD-aware: each WG loops over the 4 KB chunks of "its" residue class mod 8
__global__ __launch_bounds__(256) void kPers(u32x4* __restrict__ d, unsigned nchunks, unsigned S) {
    const unsigned k = blockIdx.x, cls = (k + S) & 7u;
    u32x4 v = {threadIdx.x, k, 1, 2};
    for (unsigned n = k >> 3; n * 8 + cls < nchunks; n += gridDim.x >> 3) d[(size_t)(n * 8 + cls) * 256 + threadIdx.x] = v;
}
template <class F> float timeit(F f, int reps = 10) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f();
    CK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms / reps);
    }
    return *std::min_element(t.begin(), t.end());
}
int main() {
    const size_t dist_bytes = 64ull * 512 * 512 * 900;
    u32x4* d; CK(hipMalloc(&d, dist_bytes));
    printf("buffer base %p (mod 32 KB = %zu)\n", (void*)d, (size_t)((uintptr_t)d & 32767));
    unsigned* cen; CK(hipMalloc(&cen, 4096 * 4)); kCensus<<<4096, 64>>>(cen); std::vector<unsigned> h(4096); CK(hipMemcpy(h.data(), cen, 4096 * 4, hipMemcpyDeviceToHost));
    printf("XCC of WG 0..23:"); for (int i = 0; i < 24; ++i) printf(" %u", h[i]); printf("\n");
    int same = 0; for (int i = 8; i < 4096; ++i) same += (h[i] == h[i - 8]); printf("WG k and k-8 on same XCC: %d / %d\n", same, 4096 - 8);
    const size_t nchunks = dist_bytes / 4096;
    for (int round = 0; round < 2; ++round) {
        for (unsigned S = 0; S < 8; ++S) {
            float ms = timeit([&] { kX<1><<<(unsigned)nchunks, 256>>>(d, S); });
            printf("4KB chunks, rotation S=%u   %7.3f ms %6.2f TB/s\n", S, ms, dist_bytes / ms / 1e9); fflush(stdout);
        }
    }
    for (unsigned S = 0; S < 8; S += 1) {
        float ms = timeit([&] { kX<2><<<(unsigned)(nchunks / 2), 256>>>(d, S); });
        printf("8KB chunks, rotation S=%u   %7.3f ms %6.2f TB/s\n", S, ms, dist_bytes / ms / 1e9); fflush(stdout);
    }
    for (unsigned G : {2048u, 4096u, 8192u}) for (unsigned S : {0u, 1u, 4u}) {
        float ms = timeit([&] { kPers<<<G, 256>>>(d, (unsigned)nchunks, S); });
        printf("persistent XCD-class grid=%u S=%u  %7.3f ms %6.2f TB/s\n", G, S, ms, dist_bytes / ms / 1e9); fflush(stdout);
    }
    return 0;
}
