// Store-pattern microbenchmark 6: the "structure resident in LDS" K1 candidate.
// Persistent grid (G workgroups of BD lanes); workgroup w serves structure b = w / P, part = w % P and sweeps
// that structure's output in chunks c = part + P*n of BD float4 slots (dist plane) + BD/4 slots (mask plane).
// VALU = dummy fma chain of `valu` instructions per dist slot to mimic the arithmetic.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <functional>
#include <string>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int BD, int VALU>
__global__ __launch_bounds__(BD) void kR(u32x4* __restrict__ d, u32x4* __restrict__ m, unsigned P, unsigned chunks_per_b,
                                         size_t d_slots_per_b, size_t m_slots_per_b) {
    const unsigned w = blockIdx.x, b = w / P, part = w % P, t = threadIdx.x;
    u32x4* db = d + (size_t)b * d_slots_per_b;
    u32x4* mb = m + (size_t)b * m_slots_per_b;
    float acc = (float)t;
    for (unsigned c = part; c < chunks_per_b; c += P) {
#pragma unroll
        for (int k = 0; k < VALU; ++k) acc = acc * 1.0001f + 0.5f;
        u32x4 v = {__float_as_uint(acc), c, t, 7};
        db[(size_t)c * BD + t] = v;
        if (t < BD / 4) mb[(size_t)c * (BD / 4) + t] = v;
    }
}
// reference pattern: today's K1 (one WG per 57.6 KB + 14.4 KB run, 256 lanes, 225 active), with the same dummy VALU
template <int VALU>
__global__ __launch_bounds__(256) void kP(u32x4* __restrict__ d, u32x4* __restrict__ m) {
    const unsigned t = threadIdx.x;
    if (t >= 225) return;
    float acc = (float)t;
    u32x4* o = d + (size_t)blockIdx.x * 3600 + t;
    for (int g = 0; g < 16; ++g) {
#pragma unroll
        for (int k = 0; k < VALU; ++k) acc = acc * 1.0001f + 0.5f;
        u32x4 v = {__float_as_uint(acc), (unsigned)g, t, 7};
        o[g * 225] = v;
    }
    u32x4* om = m + (size_t)blockIdx.x * 900 + t;
    for (int g = 0; g < 4; ++g) { u32x4 v = {__float_as_uint(acc), (unsigned)g, t, 9}; om[g * 225] = v; }
}
int main() {
    const unsigned B = 64, N = 512;
    const size_t d_slots_per_b = (size_t)N * N * 225 / 4, m_slots_per_b = (size_t)N * N * 225 / 16;
    const size_t dist_bytes = B * d_slots_per_b * 16, mask_bytes = B * m_slots_per_b * 16;
    u32x4 *d, *m; CK(hipMalloc(&d, dist_bytes)); CK(hipMalloc(&m, mask_bytes));
    std::vector<std::pair<std::string, std::function<void()>>> v;
    auto addR = [&](auto bdc, auto vc, unsigned P) {
        constexpr int BD = decltype(bdc)::value; constexpr int VA = decltype(vc)::value;
        unsigned chunks = (unsigned)(d_slots_per_b / BD);
        if ((size_t)chunks * BD != d_slots_per_b) { printf("skip BD=%d\n", BD); return; }
        char nm[128]; snprintf(nm, 128, "resident BD=%4d P=%u grid=%u valu=%d", BD, P, B * P, VA);
        v.push_back({nm, [=] { kR<BD, VA><<<B * P, BD>>>(d, m, P, chunks, d_slots_per_b, m_slots_per_b); }});
    };
    using I = std::integral_constant<int, 0>;
    addR(std::integral_constant<int, 1024>{}, std::integral_constant<int, 0>{}, 4);
    addR(std::integral_constant<int, 1024>{}, std::integral_constant<int, 64>{}, 4);
    addR(std::integral_constant<int, 1024>{}, std::integral_constant<int, 100>{}, 4);
    addR(std::integral_constant<int, 512>{}, std::integral_constant<int, 0>{}, 4);
    addR(std::integral_constant<int, 512>{}, std::integral_constant<int, 64>{}, 4);
    addR(std::integral_constant<int, 512>{}, std::integral_constant<int, 0>{}, 8);
    addR(std::integral_constant<int, 512>{}, std::integral_constant<int, 64>{}, 8);
    addR(std::integral_constant<int, 256>{}, std::integral_constant<int, 0>{}, 4);
    addR(std::integral_constant<int, 256>{}, std::integral_constant<int, 0>{}, 16);
    addR(std::integral_constant<int, 256>{}, std::integral_constant<int, 64>{}, 16);
    addR(std::integral_constant<int, 1024>{}, std::integral_constant<int, 0>{}, 8);
    addR(std::integral_constant<int, 1024>{}, std::integral_constant<int, 64>{}, 8);
    v.push_back({"today: 57.6+14.4 KB per WG, 225 lanes, valu=0", [=] { kP<0><<<B * N * 8, 256>>>(d, m); }});
    v.push_back({"today: 57.6+14.4 KB per WG, 225 lanes, valu=64", [=] { kP<64><<<B * N * 8, 256>>>(d, m); }});
    v.push_back({"today: 57.6+14.4 KB per WG, 225 lanes, valu=100", [=] { kP<100><<<B * N * 8, 256>>>(d, m); }});
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
