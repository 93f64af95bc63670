// Store-pattern microbenchmark 9: cache-policy bits on the K1 pattern kernel's store stream (K = 32 groups of
// 3600 B + 8 groups of 900 B per workgroup, 225 active lanes, XCD-contiguous workgroup -> run map).
// gfx942/gfx950 stores take sc0 / sc1 / nt; which combination does the write path like best?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <functional>
#include <string>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int POL>
__device__ __forceinline__ void st(u32x4* p, u32x4 v) {
    if (POL == 0) *p = v;
    else if (POL == 1) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    else if (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else if (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    else if (POL == 4) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
    else if (POL == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" ::"v"(p), "v"(v) : "memory");
    else if (POL == 6) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
}

template <int K, int POL>
__global__ __launch_bounds__(256) void kP(u32x4* __restrict__ d, u32x4* __restrict__ m, unsigned n) {
    if (threadIdx.x >= 225) return;
    const unsigned w = blockIdx.x;
    const unsigned c = (w & 7u) * (n >> 3) + (w >> 3);
    u32x4 v = {threadIdx.x, c, 7, 9};
    u32x4* o = d + (size_t)c * (225 * K) + threadIdx.x;   // host: n * 225 * K * 16 B == dist_bytes exactly
#pragma unroll
    for (int g = 0; g < K; ++g) st<POL>(o + g * 225, v);
    u32x4* om = m + (size_t)c * (225 * K / 4) + threadIdx.x;
#pragma unroll
    for (int g = 0; g < K / 4; ++g) st<POL>(om + g * 225, v);
}

int main() {
    const size_t dist_bytes = 64ull * 512 * 512 * 900, mask_bytes = dist_bytes / 4;
    u32x4 *d, *m; CK(hipMalloc(&d, dist_bytes)); CK(hipMalloc(&m, mask_bytes));
    const size_t groups = dist_bytes / 3600;  // 4194304
    constexpr int K = 32;
    const unsigned n = (unsigned)(groups / K);  // 131072 workgroups, multiple of 8; n * K * 3600 == dist_bytes
    if ((size_t)n * K * 3600 != dist_bytes || (size_t)n * (K / 4) * 3600 != mask_bytes) { printf("size mismatch\n"); return 1; }
    std::vector<std::pair<std::string, std::function<void()>>> v;
    const char* names[8] = {"default", "sc0", "sc1", "sc0 sc1", "nt", "sc0 nt", "sc1 nt", "sc0 sc1 nt"};
    v.push_back({names[0], [=] { kP<K, 0><<<n, 256>>>(d, m, n); }});
    v.push_back({names[1], [=] { kP<K, 1><<<n, 256>>>(d, m, n); }});
    v.push_back({names[2], [=] { kP<K, 2><<<n, 256>>>(d, m, n); }});
    v.push_back({names[3], [=] { kP<K, 3><<<n, 256>>>(d, m, n); }});
    v.push_back({names[4], [=] { kP<K, 4><<<n, 256>>>(d, m, n); }});
    v.push_back({names[5], [=] { kP<K, 5><<<n, 256>>>(d, m, n); }});
    v.push_back({names[6], [=] { kP<K, 6><<<n, 256>>>(d, m, n); }});
    v.push_back({names[7], [=] { kP<K, 7><<<n, 256>>>(d, m, n); }});
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
        printf("store policy %-12s min %6.3f med %6.3f ms  %5.2f TB/s (med)\n", v[i].first.c_str(), t[i][0], t[i][2], (dist_bytes + mask_bytes) / t[i][2] / 1e9);
    }
    return 0;
}
