// Store-pattern microbenchmark 3: the shapes the K1 "pattern" trick allows (multiples of 225 slots).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// ACT active lanes of BD; each writes K slots: t, t+ACT, ...; WG chunk = ACT*K slots
template <int BD, int ACT, int K, bool STAGE>
__global__ __launch_bounds__(BD) void kP(u32x4* __restrict__ d, const float* __restrict__ src) {
    __shared__ float lds[2048];
    const unsigned t = threadIdx.x;
    u32x4 v = {t, blockIdx.x, 1, 2};
    if (STAGE) {
        lds[t] = src[(blockIdx.x * 61u + t) & 0xFFFFF];
        __syncthreads();
        v.x = __float_as_uint(lds[(t * 7) & (BD - 1)] + lds[(t * 3 + 1) & (BD - 1)]);
    }
    if (t >= ACT) return;
    u32x4* o = d + (size_t)blockIdx.x * (ACT * K) + t;
#pragma unroll
    for (int k = 0; k < K; ++k) o[k * ACT] = v;
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
template <int BD, int ACT, int K, bool STAGE> void run(u32x4* d, float* src, size_t n16) {
    unsigned nb = (unsigned)(n16 / (ACT * K)); size_t bytes = (size_t)nb * ACT * K * 16;
    float ms = timeit([&] { kP<BD, ACT, K, STAGE><<<nb, BD>>>(d, src); });
    printf("BD=%4d active=%4d K=%2d chunk=%6d B stage=%d  %7.3f ms %6.2f TB/s\n", BD, ACT, K, ACT * K * 16, STAGE, ms, bytes / ms / 1e9); fflush(stdout);
}
int main() {
    const size_t dist_bytes = 64ull * 512 * 512 * 900;
    u32x4* d; float* src; CK(hipMalloc(&d, dist_bytes)); CK(hipMalloc(&src, 4 << 20)); CK(hipMemset(src, 0, 4 << 20));
    const size_t n16 = dist_bytes / 16;
    run<1024, 900, 1, false>(d, src, n16); run<1024, 900, 2, false>(d, src, n16); run<1024, 900, 4, false>(d, src, n16);
    run<512, 450, 2, false>(d, src, n16); run<512, 450, 4, false>(d, src, n16); run<512, 450, 8, false>(d, src, n16);
    run<256, 225, 4, false>(d, src, n16); run<256, 225, 8, false>(d, src, n16); run<256, 225, 16, false>(d, src, n16);
    run<1024, 900, 1, true>(d, src, n16); run<1024, 900, 2, true>(d, src, n16); run<1024, 900, 4, true>(d, src, n16);
    run<512, 450, 2, true>(d, src, n16); run<512, 450, 4, true>(d, src, n16); run<512, 450, 8, true>(d, src, n16);
    run<256, 225, 4, true>(d, src, n16); run<256, 225, 8, true>(d, src, n16); run<256, 225, 16, true>(d, src, n16);
    run<1024, 1024, 1, true>(d, src, n16); run<1024, 1024, 2, true>(d, src, n16); run<512, 512, 2, true>(d, src, n16); run<256, 256, 4, true>(d, src, n16);
    return 0;
}
