// Store-pattern microbenchmark 10: does the ORDER in which a long-lived workgroup walks its 144 KB run matter?
// K1 pattern store stream (K = 32 groups of 3600 B dist + 8 groups of 3600 B mask per workgroup, 225 lanes,
// XCD-contiguous map).  Variants: natural order; start group rotated by the workgroup id (neighbouring workgroups
// are then never at the same phase of their runs); descending order; mask groups interleaved after every 4th
// dist group instead of at the end; the eight XCD streams skewed against each other.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <functional>
#include <string>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int K, int MODE>
__global__ __launch_bounds__(256) void kP(u32x4* __restrict__ d, u32x4* __restrict__ m, unsigned n) {
    if (threadIdx.x >= 225) return;
    const unsigned w = blockIdx.x;
    unsigned c = (w & 7u) * (n >> 3) + (w >> 3);
    if (MODE >= 5) {   // skew the eight XCD streams against each other inside their eighths (partition camping?)
        const unsigned S = MODE == 5 ? 37u : (MODE == 6 ? 1031u : 5u);
        c = (w & 7u) * (n >> 3) + ((w >> 3) + (w & 7u) * S) % (n >> 3);
    }
    u32x4 v = {threadIdx.x, c, 7, 9};
    u32x4* o = d + (size_t)c * (225 * K) + threadIdx.x;     // host checks n * K * 3600 == dist bytes
    u32x4* om = m + (size_t)c * (225 * K / 4) + threadIdx.x;
    const unsigned rot = MODE == 1 ? (c % K) : (MODE == 2 ? (c * 7u) % K : 0u);
    if (MODE == 4) {
#pragma unroll
        for (int g = 0; g < K; ++g) {
            o[g * 225] = v;
            if ((g & 3) == 3) om[(g >> 2) * 225] = v;
        }
        return;
    }
#pragma unroll
    for (int g = 0; g < K; ++g) {
        const unsigned gg = MODE == 3 ? (unsigned)(K - 1 - g) : ((unsigned)g + rot) % K;
        o[gg * 225] = v;
    }
#pragma unroll
    for (int g = 0; g < K / 4; ++g) {
        const unsigned gg = MODE == 3 ? (unsigned)(K / 4 - 1 - g) : ((unsigned)g + rot) % (K / 4);
        om[gg * 225] = v;
    }
}

int main() {
    const size_t dist_bytes = 64ull * 512 * 512 * 900, mask_bytes = dist_bytes / 4;
    u32x4 *d, *m; CK(hipMalloc(&d, dist_bytes)); CK(hipMalloc(&m, mask_bytes));
    constexpr int K = 32;
    const unsigned n = (unsigned)(dist_bytes / 3600 / K);
    if ((size_t)n * K * 3600 != dist_bytes || (size_t)n * (K / 4) * 3600 != mask_bytes || n % 8) { printf("size mismatch\n"); return 1; }
    std::vector<std::pair<std::string, std::function<void()>>> v;
    v.push_back({"natural order", [=] { kP<K, 0><<<n, 256>>>(d, m, n); }});
    v.push_back({"start rotated by run index", [=] { kP<K, 1><<<n, 256>>>(d, m, n); }});
    v.push_back({"start rotated by 7 * run index", [=] { kP<K, 2><<<n, 256>>>(d, m, n); }});
    v.push_back({"descending order", [=] { kP<K, 3><<<n, 256>>>(d, m, n); }});
    v.push_back({"mask group after every 4th dist group", [=] { kP<K, 4><<<n, 256>>>(d, m, n); }});
    v.push_back({"XCD streams skewed by 37 runs each", [=] { kP<K, 5><<<n, 256>>>(d, m, n); }});
    v.push_back({"XCD streams skewed by 1031 runs each", [=] { kP<K, 6><<<n, 256>>>(d, m, n); }});
    v.push_back({"XCD streams skewed by 5 runs each", [=] { kP<K, 7><<<n, 256>>>(d, m, n); }});
    std::vector<std::vector<float>> t(v.size());
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int warm = 0; warm < 30; ++warm) v[0].second();
    CK(hipDeviceSynchronize());
    for (int round = 0; round < 5; ++round)
        for (size_t i = 0; i < v.size(); ++i) {
            v[i].second(); CK(hipDeviceSynchronize());
            CK(hipEventRecord(a)); for (int r = 0; r < 5; ++r) v[i].second(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); t[i].push_back(ms / 5);
        }
    for (size_t i = 0; i < v.size(); ++i) {
        std::sort(t[i].begin(), t[i].end());
        printf("%-42s min %6.3f med %6.3f ms  %5.2f TB/s (med)\n", v[i].first.c_str(), t[i][0], t[i][2], (dist_bytes + mask_bytes) / t[i][2] / 1e9);
    }
    return 0;
}
