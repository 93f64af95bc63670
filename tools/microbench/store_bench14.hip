// Store-pattern microbenchmark 14: can LONG-LIVED workgroups keep the written window small?
// Persistent grid: XCD x (= w % 8) owns the x-th eighth of both planes; its Q workgroups sweep that region in an
// interleaved order -- in step k workgroup q writes chunk k*Q + q -- so the region being written at any instant is
// about Q chunks wide.  A chunk is what one K1 workgroup-iteration would produce from 16 pairs: 4 dist groups of 3600 B
// (225 lanes) + 1 mask group.  Compared, on the slowest and fastest of ten allocations, with today's stream (32
// groups per short-lived workgroup, XCD-contiguous) and with short-lived one-chunk workgroups.
// VALU: dummy fma chain per dist slot; BAR: two workgroup barriers per chunk (what staging through LDS would need).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <functional>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int K>
__global__ __launch_bounds__(256) void kToday(u32x4* __restrict__ d, u32x4* __restrict__ m, unsigned n) {
    if (threadIdx.x >= 225) return;
    const unsigned w = blockIdx.x, c = (w & 7u) * (n >> 3) + (w >> 3);
    u32x4 v = {threadIdx.x, c, 7, 9};
    u32x4* o = d + (size_t)c * (225 * K) + threadIdx.x;
#pragma unroll
    for (int g = 0; g < K; ++g) o[g * 225] = v;
    u32x4* om = m + (size_t)c * (225 * K / 4) + threadIdx.x;
#pragma unroll
    for (int g = 0; g < K / 4; ++g) om[g * 225] = v;
}

// chunks_per_xcd = total chunks / 8 (host checks divisibility); grid = 8 * Q
template <int VALU, bool BAR>
__global__ __launch_bounds__(256) void kSweep(u32x4* __restrict__ d, u32x4* __restrict__ m, unsigned chunks_per_xcd,
                                              unsigned Q, float seed) {
    const unsigned w = blockIdx.x, x = w & 7u, q = w >> 3;
    const bool act = threadIdx.x < 225;
    for (unsigned c = q; c < chunks_per_xcd; c += Q) {
        const size_t chunk = (size_t)x * chunks_per_xcd + c;
        if (BAR) __syncthreads();
        if (act) {
            u32x4* o = d + chunk * (4 * 225) + threadIdx.x;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float a = seed + g;
#pragma unroll
                for (int i = 0; i < VALU; ++i) a = __builtin_fmaf(a, 1.0001f, 0.5f);
                u32x4 v = {threadIdx.x, (unsigned)chunk, __float_as_uint(a), 9};
                o[g * 225] = v;
            }
            u32x4 v = {threadIdx.x, (unsigned)chunk, 7, 9};
            m[chunk * 225 + threadIdx.x] = v;
        }
        if (BAR) __syncthreads();
    }
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
    const size_t groups = dist_bytes / 3600;              // 2^22
    const unsigned n32 = groups / 32;
    const unsigned chunks = groups / 4, cpx = chunks / 8;  // 2^20 chunks, 2^17 per XCD
    if ((size_t)cpx * 8 * 4 * 3600 != dist_bytes || (size_t)cpx * 8 * 3600 != mask_bytes) { printf("size mismatch\n"); return 1; }
    struct A { u32x4 *d, *m; float r; };
    std::vector<A> al;
    for (int i = 0; i < 8; ++i) {                          // 8 x 18.9 GB = 151 GB held at once
        A a; CK(hipMalloc(&a.d, dist_bytes)); CK(hipMalloc(&a.m, mask_bytes));
        a.r = tbps([=] { kToday<32><<<n32, 256>>>(a.d, a.m, n32); }, total);
        printf("allocation %d: today's stream %.2f TB/s\n", i, a.r);
        al.push_back(a);
    }
    auto cmp = [](const A& x, const A& y) { return x.r < y.r; };
    const A worst = *std::min_element(al.begin(), al.end(), cmp), best = *std::max_element(al.begin(), al.end(), cmp);
    for (const A& a : {worst, best}) {
        u32x4 *d = a.d, *m = a.m;
        printf("--- allocation with today's stream at %.2f TB/s\n", a.r);
        printf("  today (32 groups per short-lived WG)          %.2f\n", tbps([=] { kToday<32><<<n32, 256>>>(d, m, n32); }, total));
        printf("  short-lived one-chunk WGs (4 groups + mask)   %.2f\n", tbps([=] { kToday<4><<<chunks, 256>>>(d, m, chunks); }, total));
        for (unsigned Q : {128u, 256u, 512u}) {
            printf("  sweep Q=%3u/XCD  stores only                  %.2f\n", Q, tbps([=] { kSweep<0, false><<<8 * Q, 256>>>(d, m, cpx, Q, 1.f); }, total));
            printf("  sweep Q=%3u/XCD  + 2 barriers per chunk       %.2f\n", Q, tbps([=] { kSweep<0, true><<<8 * Q, 256>>>(d, m, cpx, Q, 1.f); }, total));
            printf("  sweep Q=%3u/XCD  + barriers + 40 VALU/slot    %.2f\n", Q, tbps([=] { kSweep<40, true><<<8 * Q, 256>>>(d, m, cpx, Q, 1.f); }, total));
        }
    }
    return 0;
}
