// Store-pattern microbenchmark 8: small misaligned pattern chunks (225 lanes x K groups of 3600 B) with the
// workgroup -> chunk map either plain (adjacent chunks on different XCDs) or XCD-contiguous (WG w handles chunk
// (w % 8) * (n / 8) + w / 8, so each XCD sweeps its own eighth of the buffer and neighbours share an L2).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <functional>
#include <string>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int K, bool REMAP>
__global__ __launch_bounds__(256) void kP(u32x4* __restrict__ d, u32x4* __restrict__ m, unsigned n) {
    if (threadIdx.x >= 225) return;
    const unsigned w = blockIdx.x;
    const unsigned c = REMAP ? (w & 7u) * (n >> 3) + (w >> 3) : w;
    u32x4 v = {threadIdx.x, c, 7, 9};
    u32x4* o = d + (size_t)c * (225 * K) + threadIdx.x;
#pragma unroll
    for (int g = 0; g < K; ++g) o[g * 225] = v;
    if (K % 4 == 0) {  // mask plane: K/4 groups of 225 slots
        u32x4* om = m + (size_t)c * (225 * K / 4) + threadIdx.x;
#pragma unroll
        for (int g = 0; g < K / 4; ++g) om[g * 225] = v;
    }
}
// blocked XCD map: inside every block of 8*M consecutive runs, XCD x (= w % 8) takes runs [x*M, (x+1)*M)
template <int K>
__global__ __launch_bounds__(256) void kPB(u32x4* __restrict__ d, u32x4* __restrict__ m, unsigned n, unsigned M) {
    if (threadIdx.x >= 225) return;
    const unsigned w = blockIdx.x;
    const unsigned blk = w / (8u * M), r = w - blk * 8u * M;   // r in [0, 8M): dealt round-robin -> XCD r % 8, slot r / 8
    const unsigned c = blk * 8u * M + (r & 7u) * M + (r >> 3);
    u32x4 v = {threadIdx.x, c, 7, 9};
    u32x4* o = d + (size_t)c * (225 * K) + threadIdx.x;
#pragma unroll
    for (int g = 0; g < K; ++g) o[g * 225] = v;
    u32x4* om = m + (size_t)c * (225 * K / 4) + threadIdx.x;
#pragma unroll
    for (int g = 0; g < K / 4; ++g) om[g * 225] = v;
}

int main() {
    const size_t dist_bytes = 64ull * 512 * 512 * 900, mask_bytes = dist_bytes / 4;
    u32x4 *d, *m; CK(hipMalloc(&d, dist_bytes)); CK(hipMalloc(&m, mask_bytes));
    const size_t groups = dist_bytes / 3600;  // 4194304
    std::vector<std::pair<std::string, std::function<void()>>> v;
    auto add = [&](auto kc) {
        constexpr int K = decltype(kc)::value;
        unsigned n = (unsigned)(groups / K);  // multiple of 8
        char nm[96];
        snprintf(nm, 96, "pattern K=%2d (%6d+%5d B/WG) plain", K, K * 3600, K * 900); v.push_back({nm, [=] { kP<K, false><<<n, 256>>>(d, m, n); }});
        snprintf(nm, 96, "pattern K=%2d (%6d+%5d B/WG) XCD-contiguous", K, K * 3600, K * 900); v.push_back({nm, [=] { kP<K, true><<<n, 256>>>(d, m, n); }});
    };
    add(std::integral_constant<int, 4>{}); add(std::integral_constant<int, 8>{}); add(std::integral_constant<int, 16>{}); add(std::integral_constant<int, 32>{});
    for (unsigned M : {1u, 4u, 16u, 64u, 256u, 1024u, 4096u}) {
        unsigned n = (unsigned)(groups / 32);   // K = 32: today's 115.2 + 28.8 KB runs; n = 131072 is a multiple of 8*M for all M above
        char nm[96]; snprintf(nm, 96, "pattern K=32 blocked XCD map M=%u", M);
        v.push_back({nm, [=] { kPB<32><<<n, 256>>>(d, m, n, M); }});
    }
    for (unsigned M : {4u, 64u, 1024u}) {
        unsigned n = (unsigned)(groups / 16);
        char nm[96]; snprintf(nm, 96, "pattern K=16 blocked XCD map M=%u", M);
        v.push_back({nm, [=] { kPB<16><<<n, 256>>>(d, m, n, M); }});
    }
    std::vector<std::vector<float>> t(v.size());
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int round = 0; round < 5; ++round)
        for (size_t i = 0; i < v.size(); ++i) {
            v[i].second(); CK(hipDeviceSynchronize());
            CK(hipEventRecord(a)); for (int r = 0; r < 5; ++r) v[i].second(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); t[i].push_back(ms / 5);
        }
    for (size_t i = 0; i < v.size(); ++i) {
        std::sort(t[i].begin(), t[i].end());
        printf("%-52s min %6.3f med %6.3f ms  %5.2f TB/s (med)\n", v[i].first.c_str(), t[i][0], t[i][2], (dist_bytes + mask_bytes) / t[i][2] / 1e9);
    }
    return 0;
}
