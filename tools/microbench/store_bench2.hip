// Store-pattern microbenchmark 2: "one store per lane" workgroups of various sizes, two planes, with/without
// an LDS staging phase + barrier in front (what K1 needs).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// one dist store per lane (BD lanes -> BD*16 bytes), one mask store for the first BD/4 lanes
template <int BD, bool MASK, bool STAGE>
__global__ __launch_bounds__(BD) void k1s(u32x4* __restrict__ d, u32x4* __restrict__ m, const float* __restrict__ src) {
    __shared__ float lds[2048];
    const unsigned t = threadIdx.x;
    u32x4 v = {t, blockIdx.x, 1, 2};
    if (STAGE) {  // ~1 global load per lane, LDS write, barrier, 4 LDS reads (like staging ~20 residues)
        lds[t] = src[(blockIdx.x * 61u + t) & 0xFFFFF];
        if (BD < 1024) lds[t + BD] = src[(blockIdx.x * 67u + t) & 0xFFFFF];
        __syncthreads();
        v.x = __float_as_uint(lds[(t * 7) & (BD - 1)] + lds[(t * 3 + 1) & (BD - 1)]);
        v.y = __float_as_uint(lds[(t * 5 + 2) & (BD - 1)] * lds[(t + 9) & (BD - 1)]);
    }
    d[(size_t)blockIdx.x * BD + t] = v;
    if (MASK && t < BD / 4) m[(size_t)blockIdx.x * (BD / 4) + t] = v;
}
// K stores per lane, wave-interleaved (lane stride BD), all issued back to back (unrolled)
template <int BD, int K>
__global__ __launch_bounds__(BD) void kK(u32x4* __restrict__ d) {
    const unsigned t = threadIdx.x;
    u32x4 v = {t, blockIdx.x, 1, 2};
    u32x4* o = d + (size_t)blockIdx.x * BD * K + t;
#pragma unroll
    for (int k = 0; k < K; ++k) o[k * BD] = v;
}

template <class F> float timeit(F f, int reps = 10) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f();
    CK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms / reps);
    }
    return *std::min_element(t.begin(), t.end());
}
template <int BD, bool MASK, bool STAGE> void run1(u32x4* d, u32x4* m, float* src, size_t n16, const char* tag) {
    unsigned nb = (unsigned)(n16 / BD); size_t bytes = (size_t)nb * BD * 16 + (MASK ? (size_t)nb * (BD / 4) * 16 : 0);
    float ms = timeit([&] { k1s<BD, MASK, STAGE><<<nb, BD>>>(d, m, src); });
    printf("1-store/lane BD=%4d mask=%d stage=%d %-10s %7.3f ms %6.2f TB/s\n", BD, MASK, STAGE, tag, ms, bytes / ms / 1e9); fflush(stdout);
}
template <int BD, int K> void runK(u32x4* d, size_t n16) {
    unsigned nb = (unsigned)(n16 / (BD * K)); size_t bytes = (size_t)nb * BD * K * 16;
    float ms = timeit([&] { kK<BD, K><<<nb, BD>>>(d); });
    printf("K-stores unrolled BD=%4d K=%2d (%6d B/WG)      %7.3f ms %6.2f TB/s\n", BD, K, BD * K * 16, ms, bytes / ms / 1e9); fflush(stdout);
}
int main() {
    const size_t dist_bytes = 64ull * 512 * 512 * 900, mask_bytes = 64ull * 512 * 512 * 225;
    u32x4 *d, *m; float* src; CK(hipMalloc(&d, dist_bytes)); CK(hipMalloc(&m, mask_bytes)); CK(hipMalloc(&src, 4 << 20)); CK(hipMemset(src, 0, 4 << 20));
    const size_t n16 = dist_bytes / 16;
    run1<64, false, false>(d, m, src, n16, ""); run1<128, false, false>(d, m, src, n16, ""); run1<256, false, false>(d, m, src, n16, "");
    run1<512, false, false>(d, m, src, n16, ""); run1<1024, false, false>(d, m, src, n16, "");
    run1<256, true, false>(d, m, src, n16, ""); run1<512, true, false>(d, m, src, n16, ""); run1<1024, true, false>(d, m, src, n16, "");
    run1<256, false, true>(d, m, src, n16, ""); run1<512, false, true>(d, m, src, n16, ""); run1<1024, false, true>(d, m, src, n16, "");
    run1<256, true, true>(d, m, src, n16, ""); run1<512, true, true>(d, m, src, n16, ""); run1<1024, true, true>(d, m, src, n16, "");
    runK<256, 1>(d, n16); runK<256, 2>(d, n16); runK<256, 3>(d, n16); runK<256, 4>(d, n16); runK<256, 8>(d, n16); runK<256, 16>(d, n16);
    runK<64, 4>(d, n16); runK<64, 16>(d, n16); runK<128, 2>(d, n16); runK<128, 8>(d, n16); runK<512, 2>(d, n16); runK<1024, 2>(d, n16);
    return 0;
}
