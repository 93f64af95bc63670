// Store-pattern microbenchmark 15: fewer, fatter workgroups.  Same 144 KB run per workgroup as today's K1 stream, but
// written by 512 or 1024 lanes (2 or 4 groups of 3600 B per step), so that 2x / 4x fewer runs are open at any instant.
// On the slowest and the fastest of eight allocations.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <functional>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// W = 256-lane groups per workgroup (1, 2, 4); K = 32 dist groups + 8 mask groups per run
template <int W>
__global__ __launch_bounds__(256 * W) void kFat(u32x4* __restrict__ d, u32x4* __restrict__ m, unsigned n) {
    const unsigned lane = threadIdx.x & 255u, sub = threadIdx.x >> 8;
    if (lane >= 225) return;
    const unsigned w = blockIdx.x, c = (w & 7u) * (n >> 3) + (w >> 3);
    u32x4 v = {lane, c, 7, 9};
    u32x4* o = d + (size_t)c * (225 * 32) + lane;
#pragma unroll
    for (int g = 0; g < 32 / W; ++g) o[(g * W + sub) * 225] = v;
    u32x4* om = m + (size_t)c * (225 * 8) + lane;
#pragma unroll
    for (int g = 0; g < 8 / W; ++g) om[(g * W + sub) * 225] = v;
}

static hipEvent_t ea, eb;
static float tbps(const std::function<void()>& f, size_t bytes) {
    for (int i = 0; i < 3; ++i) f();
    CK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(ea)); for (int i = 0; i < 4; ++i) f(); CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb));
        float ms; CK(hipEventElapsedTime(&ms, ea, eb)); t.push_back(ms / 4);
    }
    std::sort(t.begin(), t.end());
    return bytes / t[1] / 1e9;
}

int main() {
    CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    const size_t dist_bytes = 64ull * 512 * 512 * 900, mask_bytes = dist_bytes / 4, total = dist_bytes + mask_bytes;
    const unsigned n = (unsigned)(dist_bytes / 3600 / 32);
    if ((size_t)n * 32 * 3600 != dist_bytes || (size_t)n * 8 * 3600 != mask_bytes || n % 8) { printf("size mismatch\n"); return 1; }
    struct A { u32x4 *d, *m; float r; };
    std::vector<A> al;
    for (int i = 0; i < 8; ++i) {
        A a; CK(hipMalloc(&a.d, dist_bytes)); CK(hipMalloc(&a.m, mask_bytes));
        a.r = tbps([=] { kFat<1><<<n, 256>>>(a.d, a.m, n); }, total);
        al.push_back(a);
    }
    std::sort(al.begin(), al.end(), [](const A& x, const A& y) { return x.r < y.r; });
    printf("today's stream on the eight allocations:");
    for (auto& a : al) printf(" %.2f", a.r);
    printf(" TB/s\n");
    for (const A& a : {al.front(), al[al.size() / 2], al.back()}) {
        u32x4 *d = a.d, *m = a.m;
        printf("allocation at %.2f:  256 lanes %.2f   512 lanes %.2f   1024 lanes %.2f TB/s\n", a.r,
               tbps([=] { kFat<1><<<n, 256>>>(d, m, n); }, total), tbps([=] { kFat<2><<<n, 512>>>(d, m, n); }, total),
               tbps([=] { kFat<4><<<n, 1024>>>(d, m, n); }, total));
    }
    return 0;
}
