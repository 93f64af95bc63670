// Store-pattern microbenchmark 16: PMC target.  Finds the slowest and the fastest of eight allocations for today's K1
// store stream, then launches the SAME kernel under two names -- kStream<0> on the slow buffers, kStream<1> on the fast
// ones -- so that rocprofv3 --pmc rows can be told apart by kernel name.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <functional>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int TAG>
__global__ __launch_bounds__(256) void kStream(u32x4* __restrict__ d, u32x4* __restrict__ m, unsigned n) {
    if (threadIdx.x >= 225) return;
    const unsigned w = blockIdx.x, c = (w & 7u) * (n >> 3) + (w >> 3);
    u32x4 v = {threadIdx.x, c, 7, (unsigned)TAG};
    u32x4* o = d + (size_t)c * (225 * 32) + threadIdx.x;
#pragma unroll
    for (int g = 0; g < 32; ++g) o[g * 225] = v;
    u32x4* om = m + (size_t)c * (225 * 8) + threadIdx.x;
#pragma unroll
    for (int g = 0; g < 8; ++g) om[g * 225] = v;
}

int main() {
    hipEvent_t ea, eb; CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    const size_t dist_bytes = 64ull * 512 * 512 * 900, mask_bytes = dist_bytes / 4, total = dist_bytes + mask_bytes;
    const unsigned n = (unsigned)(dist_bytes / 3600 / 32);
    if ((size_t)n * 32 * 3600 != dist_bytes || (size_t)n * 8 * 3600 != mask_bytes || n % 8) { printf("size mismatch\n"); return 1; }
    struct A { u32x4 *d, *m; float ms; };
    std::vector<A> al;
    for (int i = 0; i < 8; ++i) {
        A a; CK(hipMalloc(&a.d, dist_bytes)); CK(hipMalloc(&a.m, mask_bytes));
        for (int k = 0; k < 2; ++k) kStream<2><<<n, 256>>>(a.d, a.m, n);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(ea)); for (int k = 0; k < 3; ++k) kStream<2><<<n, 256>>>(a.d, a.m, n); CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb));
        CK(hipEventElapsedTime(&a.ms, ea, eb)); a.ms /= 3;
        al.push_back(a);
    }
    std::sort(al.begin(), al.end(), [](const A& x, const A& y) { return x.ms > y.ms; });
    printf("hunt (kStream<2>): slowest %.3f ms = %.2f TB/s, fastest %.3f ms = %.2f TB/s\n", al.front().ms,
           total / al.front().ms / 1e9, al.back().ms, total / al.back().ms / 1e9);
    for (int rep = 0; rep < 3; ++rep) {
        kStream<0><<<n, 256>>>(al.front().d, al.front().m, n);
        kStream<1><<<n, 256>>>(al.back().d, al.back().m, n);
    }
    CK(hipDeviceSynchronize());
    return 0;
}
