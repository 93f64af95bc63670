// Store-pattern microbenchmark 11: does the rate of the K1 store stream depend on WHERE the output sits?
// One allocation of dist + mask + 3 GB; the dist plane is placed at a series of byte offsets inside it (the mask
// plane right behind it), same kernel every time (K = 32 groups per workgroup, 225 lanes, XCD-contiguous map).
// Then the same for separate allocations made after a dummy allocation of varying size (different physical pages).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int K>
__global__ __launch_bounds__(256) void kP(u32x4* __restrict__ d, u32x4* __restrict__ m, unsigned n) {
    if (threadIdx.x >= 225) return;
    const unsigned w = blockIdx.x;
    const unsigned c = (w & 7u) * (n >> 3) + (w >> 3);
    u32x4 v = {threadIdx.x, c, 7, 9};
    u32x4* o = d + (size_t)c * (225 * K) + threadIdx.x;
#pragma unroll
    for (int g = 0; g < K; ++g) o[g * 225] = v;
    u32x4* om = m + (size_t)c * (225 * K / 4) + threadIdx.x;
#pragma unroll
    for (int g = 0; g < K / 4; ++g) om[g * 225] = v;
}

static float timeit(u32x4* d, u32x4* m, unsigned n) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 10; ++i) kP<32><<<n, 256>>>(d, m, n);
    CK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(a)); for (int i = 0; i < 5; ++i) kP<32><<<n, 256>>>(d, m, n); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms / 5);
    }
    std::sort(t.begin(), t.end());
    return t[2];
}

int main() {
    const size_t dist_bytes = 64ull * 512 * 512 * 900, mask_bytes = dist_bytes / 4;
    const unsigned n = (unsigned)(dist_bytes / 3600 / 32);
    if ((size_t)n * 32 * 3600 != dist_bytes || n % 8) { printf("size mismatch\n"); return 1; }
    const size_t slack = 3ull << 30;
    char* big; CK(hipMalloc(&big, dist_bytes + mask_bytes + slack));
    printf("one allocation at %p, dist plane at byte offset:\n", (void*)big);
    const size_t offs[] = {0, 4096, 65536, 1ull << 20, (2ull << 20) + 4096, 32ull << 20, (512ull << 20) + (1ull << 20), 1ull << 30, (2ull << 30) + 65536, 0};
    for (size_t off : offs) {
        u32x4* d = (u32x4*)(big + off);                         // off + dist + mask <= total (off <= slack)
        u32x4* m = (u32x4*)(big + off + dist_bytes);
        float ms = timeit(d, m, n);
        printf("  offset %12zu  %6.3f ms  %5.2f TB/s\n", off, ms, (dist_bytes + mask_bytes) / ms / 1e9);
    }
    CK(hipFree(big));
    printf("separate allocations after a dummy allocation of:\n");
    const size_t dummies[] = {0, 2ull << 20, 256ull << 20, 1ull << 30, 5ull << 30, 17ull << 30, 0};
    for (size_t du : dummies) {
        char* dummy = nullptr; if (du) CK(hipMalloc(&dummy, du));
        u32x4 *d, *m; CK(hipMalloc(&d, dist_bytes)); CK(hipMalloc(&m, mask_bytes));
        float ms = timeit(d, m, n);
        printf("  dummy %12zu  d=%p m=%p  %6.3f ms  %5.2f TB/s\n", du, (void*)d, (void*)m, ms, (dist_bytes + mask_bytes) / ms / 1e9);
        CK(hipFree(d)); CK(hipFree(m)); if (dummy) CK(hipFree(dummy));
    }
    return 0;
}
