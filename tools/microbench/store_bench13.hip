// Store-pattern microbenchmark 13: hunt for a "slow" allocation (store_bench11: the K1 stream runs 5.97 .. 7.09 TB/s
// depending on which physical memory the buffers got), then run a battery of patterns on the slowest and on the
// fastest allocation found, to see which properties make a pattern immune.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <functional>
#include <string>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// ACT active lanes (225: K1's 3600-byte groups; 256: aligned 4 KB groups), K groups per workgroup;
// XMAP 0 plain (adjacent runs on different XCDs), 1 XCD-contiguous
template <int ACT, int K, int XMAP>
__global__ __launch_bounds__(256) void kG(u32x4* __restrict__ d, unsigned n) {   // host: n * K * ACT * 16 B <= bytes
    if (threadIdx.x >= ACT) return;
    const unsigned w = blockIdx.x;
    const unsigned c = XMAP ? (w & 7u) * (n >> 3) + (w >> 3) : w;
    u32x4 v = {threadIdx.x, c, 7, 9};
    u32x4* o = d + (size_t)c * (ACT * K) + threadIdx.x;
#pragma unroll
    for (int g = 0; g < K; ++g) o[g * ACT] = v;
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
    const size_t bytes = 64ull * 512 * 512 * 900;            // one plane: 15.1 GB = 2^22 groups of 3600 B
    const size_t g225 = bytes / 3600, g256 = bytes / 4096;   // g256 = 3686400 = 2^14 * 225
    std::vector<u32x4*> held;
    std::vector<float> rate;
    for (int i = 0; i < 10; ++i) {                           // 10 x 15.1 GB = 151 GB held at once
        u32x4* d; CK(hipMalloc(&d, bytes));
        const unsigned n = g225 / 32;
        float r = tbps([=] { kG<225, 32, 1><<<n, 256>>>(d, n); }, bytes);
        printf("allocation %d at %p: K1 stream (225 lanes, K=32, XCD-contiguous) %.2f TB/s\n", i, (void*)d, r);
        held.push_back(d); rate.push_back(r);
    }
    const int worst = std::min_element(rate.begin(), rate.end()) - rate.begin();
    const int best = std::max_element(rate.begin(), rate.end()) - rate.begin();
    for (int which : {worst, best}) {
        u32x4* d = held[which];
        printf("--- battery on allocation %d (%.2f TB/s above)\n", which, rate[which]);
#define ROW(ACT, K, XMAP, G)                                                                      \
        { const unsigned n = (unsigned)((G) / (K));                                                \
          printf("  %3d lanes, %2d stores/lane (%6d B/WG), %-14s %.2f TB/s\n", ACT, K, ACT * 16 * K, \
                 XMAP ? "XCD-contiguous" : "plain map", tbps([=] { kG<ACT, K, XMAP><<<n, 256>>>(d, n); }, bytes)); }
        ROW(256, 1, 0, g256) ROW(256, 1, 1, g256) ROW(256, 2, 0, g256) ROW(256, 2, 1, g256) ROW(256, 4, 0, g256)
        ROW(256, 4, 1, g256) ROW(256, 16, 1, g256) ROW(256, 32, 0, g256) ROW(256, 32, 1, g256)
        ROW(225, 1, 0, g225) ROW(225, 1, 1, g225) ROW(225, 4, 0, g225) ROW(225, 4, 1, g225) ROW(225, 32, 0, g225)
        ROW(225, 32, 1, g225)
    }
    return 0;
}
