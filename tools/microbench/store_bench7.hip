// Store-pattern microbenchmark 7: single-wave workgroups (64 lanes) writing 4 KB (dist) + 1 KB (mask),
// with a K1-like amount of staging (2 float4 global loads per lane -> wave-private LDS -> 8 LDS reads per slot),
// a per-slot table load from a small L1-resident global table, and VALU dummy arithmetic per slot.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <functional>
#include <string>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int VALU, bool STAGE, bool TABLE>
__global__ __launch_bounds__(64) void kW(u32x4* __restrict__ d, u32x4* __restrict__ m, const float4* __restrict__ src,
                                         const uint4* __restrict__ tab) {
    __shared__ float4 lds[128];
    const unsigned t = threadIdx.x, w = blockIdx.x;
    float acc = (float)t;
    if (STAGE) {
        lds[t] = src[(w * 6u + t) & 0x3FFFF];
        lds[t + 64] = src[(w * 6u + 64u + t) & 0x3FFFF];
        // single wave: no barrier needed, the compiler inserts the waits
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned ls = k * 64 + t;
        unsigned off = ls & 127u;
        if (TABLE) { uint4 e = tab[(w * 31u + ls) % 225u]; off = (e.x + e.y + e.z + e.w) & 127u; }
        if (STAGE) {
#pragma unroll
            for (int r = 0; r < 8; ++r) { float4 q = lds[(off + r * 5) & 127]; acc += q.x * q.y + q.z; }
        }
#pragma unroll
        for (int v = 0; v < VALU; ++v) acc = acc * 1.0001f + 0.5f;
        u32x4 val = {__float_as_uint(acc), (unsigned)k, t, w};
        d[(size_t)w * 256 + ls] = val;
    }
    u32x4 mv = {__float_as_uint(acc), 1, t, w};
    m[(size_t)w * 64 + t] = mv;
}
// generalised: BD lanes (BD/64 independent waves, wave-private LDS, NO barrier), K dist slots per lane,
// chunk per WG = BD*K*16 B dist + BD*K*4 B mask (lanes whose slot index < BD*K/4 write one mask slot)
template <int BD, int K, int VALU>
__global__ __launch_bounds__(BD) void kG(u32x4* __restrict__ d, u32x4* __restrict__ m, const float4* __restrict__ src,
                                         const uint4* __restrict__ tab) {
    __shared__ float4 lds[(BD / 64) * 128];
    const unsigned t = threadIdx.x, w = blockIdx.x, wave = t >> 6, l = t & 63;
    float4* my = lds + wave * 128;
    float acc = (float)t;
    my[l] = src[(w * 6u + t) & 0x3FFFF];
    my[l + 64] = src[(w * 6u + 64u + t) & 0x3FFFF];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const unsigned ls = k * BD + t;
        uint4 e = tab[(w * 31u + ls) % 225u];
        unsigned off = (e.x + e.y + e.z + e.w + ls) & 127u;
#pragma unroll
        for (int r = 0; r < 8; ++r) { float4 q = my[(off + r * 5) & 127]; acc += q.x * q.y + q.z; }
#pragma unroll
        for (int v = 0; v < VALU; ++v) acc = acc * 1.0001f + 0.5f;
        u32x4 val = {__float_as_uint(acc), (unsigned)k, t, w};
        d[(size_t)w * (BD * K) + ls] = val;
    }
    if (t < BD * K / 4) { u32x4 mv = {__float_as_uint(acc), 1, t, w}; m[(size_t)w * (BD * K / 4) + t] = mv; }
}

int main() {
    const size_t dist_bytes = 64ull * 512 * 512 * 900, mask_bytes = dist_bytes / 4;
    u32x4 *d, *m; float4* src; uint4* tab;
    CK(hipMalloc(&d, dist_bytes)); CK(hipMalloc(&m, mask_bytes)); CK(hipMalloc(&src, (1 << 18) * 16)); CK(hipMalloc(&tab, 225 * 16));
    CK(hipMemset(src, 0, (1 << 18) * 16)); CK(hipMemset(tab, 0, 225 * 16));
    const unsigned nwg = (unsigned)(dist_bytes / 4096);
    std::vector<std::pair<std::string, std::function<void()>>> v = {
        {"wave-WG 4+1 KB  no staging, valu 0", [&] { kW<0, false, false><<<nwg, 64>>>(d, m, src, tab); }},
        {"wave-WG 4+1 KB  no staging, valu 64", [&] { kW<64, false, false><<<nwg, 64>>>(d, m, src, tab); }},
        {"wave-WG 4+1 KB  staging, valu 0", [&] { kW<0, true, false><<<nwg, 64>>>(d, m, src, tab); }},
        {"wave-WG 4+1 KB  staging+table, valu 0", [&] { kW<0, true, true><<<nwg, 64>>>(d, m, src, tab); }},
        {"wave-WG 4+1 KB  staging+table, valu 40", [&] { kW<40, true, true><<<nwg, 64>>>(d, m, src, tab); }},
        {"wave-WG 4+1 KB  staging+table, valu 64", [&] { kW<64, true, true><<<nwg, 64>>>(d, m, src, tab); }},
        {"wave-WG 4+1 KB  staging+table, valu 90", [&] { kW<90, true, true><<<nwg, 64>>>(d, m, src, tab); }},
    };
    auto addG = [&](auto bdc, auto kc) {
        constexpr int BD = decltype(bdc)::value; constexpr int K = decltype(kc)::value;
        unsigned n = (unsigned)(dist_bytes / (BD * K * 16));
        char nm[96]; snprintf(nm, 96, "BD=%4d K=%d (%5d+%4d B/WG) staged, valu 90", BD, K, BD * K * 16, BD * K * 4);
        v.push_back({nm, [=] { kG<BD, K, 90><<<n, BD>>>(d, m, src, tab); }});
    };
    using std::integral_constant;
    addG(integral_constant<int, 64>{}, integral_constant<int, 4>{});
    addG(integral_constant<int, 64>{}, integral_constant<int, 8>{});
    addG(integral_constant<int, 64>{}, integral_constant<int, 16>{});
    addG(integral_constant<int, 128>{}, integral_constant<int, 2>{});
    addG(integral_constant<int, 128>{}, integral_constant<int, 4>{});
    addG(integral_constant<int, 256>{}, integral_constant<int, 1>{});
    addG(integral_constant<int, 256>{}, integral_constant<int, 2>{});
    addG(integral_constant<int, 256>{}, integral_constant<int, 4>{});
    addG(integral_constant<int, 512>{}, integral_constant<int, 1>{});
    addG(integral_constant<int, 1024>{}, integral_constant<int, 1>{});
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
        printf("%-44s min %6.3f med %6.3f ms  %5.2f TB/s (med)\n", v[i].first.c_str(), t[i][0], t[i][2], (dist_bytes + mask_bytes) / t[i][2] / 1e9);
    }
    return 0;
}
