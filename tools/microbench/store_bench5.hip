// Store-pattern microbenchmark 5: dwordx3 (12 B/lane) full-lane stores vs dwordx4 variants, interleaved A/B.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <functional>
#include <string>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// X3: WG chunk = nchunk wave-chunks of 768 B (64 lanes x 12 B); wave w takes wave-chunks w, w+4, ...
__global__ __launch_bounds__(256) void kX3(unsigned* __restrict__ d, unsigned nchunk) {
    const unsigned w = threadIdx.x >> 6, l = threadIdx.x & 63;
    unsigned* o = d + (size_t)blockIdx.x * nchunk * 192;
    u32x3 v = {threadIdx.x, blockIdx.x, 7};
    for (unsigned c = w; c < nchunk; c += 4) *reinterpret_cast<u32x3*>(o + c * 192 + l * 3) = v;
}
// X4: same with 16 B/lane: wave-chunks of 1 KB
__global__ __launch_bounds__(256) void kX4(unsigned* __restrict__ d, unsigned nchunk) {
    const unsigned w = threadIdx.x >> 6, l = threadIdx.x & 63;
    unsigned* o = d + (size_t)blockIdx.x * nchunk * 256;
    u32x4 v = {threadIdx.x, blockIdx.x, 7, 9};
    for (unsigned c = w; c < nchunk; c += 4) *reinterpret_cast<u32x4*>(o + c * 256 + l * 4) = v;
}
// P225: 225 active lanes, 3600-byte groups (current K1 pattern kernel)
__global__ __launch_bounds__(256) void kP225(unsigned* __restrict__ d, unsigned ngroups) {
    if (threadIdx.x >= 225) return;
    unsigned* o = d + (size_t)blockIdx.x * ngroups * 900 + threadIdx.x * 4;
    u32x4 v = {threadIdx.x, blockIdx.x, 7, 9};
    for (unsigned g = 0; g < ngroups; ++g) *reinterpret_cast<u32x4*>(o + g * 900) = v;
}
// X2: 8 B/lane (dwordx2) for reference
__global__ __launch_bounds__(256) void kX2(unsigned* __restrict__ d, unsigned nchunk) {
    const unsigned w = threadIdx.x >> 6, l = threadIdx.x & 63;
    unsigned* o = d + (size_t)blockIdx.x * nchunk * 128;
    uint2 v = {threadIdx.x, blockIdx.x};
    for (unsigned c = w; c < nchunk; c += 4) *reinterpret_cast<uint2*>(o + c * 128 + l * 2) = v;
}
int main() {
    const size_t bytes = 64ull * 512 * 512 * 900;
    unsigned* d; CK(hipMalloc(&d, bytes));
    const unsigned nwg = 64 * 512 * 8;  // one WG per 57600-byte run
    std::vector<std::pair<std::string, std::function<void()>>> v = {
        {"X3 dwordx3 full-lane 57600 B/WG (75 x 768)", [&] { kX3<<<nwg, 256>>>(d, 75); }},
        {"X4 dwordx4 full-lane 57344 B/WG (56 x 1024)", [&] { kX4<<<nwg, 256>>>(d, 56); }},
        {"P225 dwordx4 225-lane 57600 B/WG (16 x 3600)", [&] { kP225<<<nwg, 256>>>(d, 16); }},
        {"X2 dwordx2 full-lane 57344 B/WG (112 x 512)", [&] { kX2<<<nwg, 256>>>(d, 112); }},
        {"X3 dwordx3 full-lane 14592 B/WG (19 x 768)", [&] { kX3<<<nwg * 3, 256>>>(d, 19); }},
        {"X4 dwordx4 full-lane 16384 B/WG (16 x 1024)", [&] { kX4<<<nwg * 3, 256>>>(d, 16); }},
        {"X4 dwordx4 full-lane 4096 B/WG (4 x 1024)", [&] { kX4<<<nwg * 14, 256>>>(d, 4); }},
        {"X3 dwordx3 full-lane 3072 B/WG (4 x 768)", [&] { kX3<<<nwg * 18, 256>>>(d, 4); }},
    };
    std::vector<size_t> nb = {(size_t)nwg * 75 * 768, (size_t)nwg * 56 * 1024, (size_t)nwg * 57600, (size_t)nwg * 112 * 512,
                              (size_t)nwg * 3 * 19 * 768, (size_t)nwg * 3 * 16384, (size_t)nwg * 14 * 4096, (size_t)nwg * 18 * 3072};
    for (size_t i = 0; i < nb.size(); ++i)  // host-side bounds check: every configuration must stay inside the buffer
        if (nb[i] > bytes) { printf("config %zu would write %zu > %zu bytes -- refusing to launch\n", i, nb[i], bytes); return 1; }
    std::vector<std::vector<float>> t(v.size());
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int round = 0; round < 5; ++round)
        for (size_t i = 0; i < v.size(); ++i) {
            v[i].second(); v[i].second(); CK(hipDeviceSynchronize());
            CK(hipEventRecord(a)); for (int r = 0; r < 10; ++r) v[i].second(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); t[i].push_back(ms / 10);
        }
    for (size_t i = 0; i < v.size(); ++i) {
        std::sort(t[i].begin(), t[i].end());
        printf("%-50s min %6.3f med %6.3f ms  %5.2f TB/s (med)\n", v[i].first.c_str(), t[i][0], t[i][2], nb[i] / t[i][2] / 1e9);
    }
    return 0;
}
