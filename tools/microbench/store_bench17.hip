// Store-pattern microbenchmark 17: on a SLOW allocation, does the relative placement of the eight XCD streams matter?
// (If a slow allocation were one whose physical frames make the eight streams collide on DRAM banks, skewing the
// streams against each other or interleaving the XCD regions in blocks should change the rate.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <functional>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// MODE 0: XCD-contiguous; 1: skewed by S runs per XCD inside its eighth; 2: blocked map with block M (= S)
template <int MODE>
__global__ __launch_bounds__(256) void kS(u32x4* __restrict__ d, u32x4* __restrict__ m, unsigned n, unsigned S) {
    if (threadIdx.x >= 225) return;
    const unsigned w = blockIdx.x;
    unsigned c;
    if (MODE == 0) c = (w & 7u) * (n >> 3) + (w >> 3);
    else if (MODE == 1) c = (w & 7u) * (n >> 3) + ((w >> 3) + (w & 7u) * S) % (n >> 3);
    else { const unsigned blk = w / (8u * S), r = w - blk * 8u * S; c = blk * 8u * S + (r & 7u) * S + (r >> 3); }
    u32x4 v = {threadIdx.x, c, 7, 9};
    u32x4* o = d + (size_t)c * (225 * 32) + threadIdx.x;
#pragma unroll
    for (int g = 0; g < 32; ++g) o[g * 225] = v;
    u32x4* om = m + (size_t)c * (225 * 8) + threadIdx.x;
#pragma unroll
    for (int g = 0; g < 8; ++g) om[g * 225] = v;
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
    const unsigned n = (unsigned)(dist_bytes / 3600 / 32);   // 131072 runs
    if ((size_t)n * 32 * 3600 != dist_bytes || n % (8 * 4096)) { printf("size mismatch\n"); return 1; }
    struct A { u32x4 *d, *m; float r; };
    std::vector<A> al;
    for (int i = 0; i < 8; ++i) {
        A a; CK(hipMalloc(&a.d, dist_bytes)); CK(hipMalloc(&a.m, mask_bytes));
        a.r = tbps([=] { kS<0><<<n, 256>>>(a.d, a.m, n, 0); }, total);
        al.push_back(a);
    }
    std::sort(al.begin(), al.end(), [](const A& x, const A& y) { return x.r < y.r; });
    printf("today's stream on the eight allocations:");
    for (auto& a : al) printf(" %.2f", a.r);
    printf(" TB/s\n");
    for (const A& a : {al.front(), al.back()}) {
        u32x4 *d = a.d, *m = a.m;
        printf("--- allocation at %.2f TB/s\n", a.r);
        printf("  XCD-contiguous                 %.2f\n", tbps([=] { kS<0><<<n, 256>>>(d, m, n, 0); }, total));
        for (unsigned S : {1u, 3u, 37u, 1031u, 8191u})
            printf("  streams skewed by %4u runs    %.2f\n", S, tbps([=] { kS<1><<<n, 256>>>(d, m, n, S); }, total));
        for (unsigned M : {16u, 128u, 1024u, 4096u})
            printf("  blocked map, block %4u runs   %.2f\n", M, tbps([=] { kS<2><<<n, 256>>>(d, m, n, M); }, total));
    }
    return 0;
}
