// Store-pattern microbenchmark 12: which store patterns are ROBUST against where the buffers happen to land?
// (store_bench11: the same kernel runs 5.97 .. 7.09 TB/s on buffers allocated at different moments of one process.)
// For each of several fresh allocations, interleaved: the K1 stream as written today (K = 32 groups per workgroup,
// XCD-contiguous), small pattern workgroups (K = 4, 8) under XCD-contiguous and blocked-XCD maps, and the fill-like
// reference (one aligned 16-byte store per lane, 4 KB per short-lived workgroup, natural order).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <functional>
#include <string>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// MAP 0: XCD-contiguous; MAP M > 0: blocked, XCD x takes runs [x*M, (x+1)*M) of every block of 8*M runs
template <int K>
__global__ __launch_bounds__(256) void kP(u32x4* __restrict__ d, u32x4* __restrict__ m, unsigned n, unsigned M) {
    if (threadIdx.x >= 225) return;
    const unsigned w = blockIdx.x;
    unsigned c;
    if (M == 0) c = (w & 7u) * (n >> 3) + (w >> 3);
    else {
        const unsigned blk = w / (8u * M), r = w - blk * 8u * M;
        c = blk * 8u * M + (r & 7u) * M + (r >> 3);
    }
    u32x4 v = {threadIdx.x, c, 7, 9};
    u32x4* o = d + (size_t)c * (225 * K) + threadIdx.x;
#pragma unroll
    for (int g = 0; g < K; ++g) o[g * 225] = v;
    u32x4* om = m + (size_t)c * (225 * K / 4) + threadIdx.x;
#pragma unroll
    for (int g = 0; g < K / 4; ++g) om[g * 225] = v;
}
__global__ __launch_bounds__(256) void kFill(u32x4* __restrict__ p) {   // grid * 256 * 16 B == bytes exactly
    u32x4 v = {threadIdx.x, blockIdx.x, 7, 9};
    p[(size_t)blockIdx.x * 256 + threadIdx.x] = v;
}

int main() {
    const size_t dist_bytes = 64ull * 512 * 512 * 900, mask_bytes = dist_bytes / 4;
    const size_t groups = dist_bytes / 3600;  // 4194304 = 2^22
    if (dist_bytes % 4096 || mask_bytes % 4096) { printf("size mismatch\n"); return 1; }
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const size_t dummies[] = {0, 2ull << 20, 256ull << 20, 1ull << 30, 5ull << 30, 17ull << 30, 0, 3ull << 30};
    printf("%-14s", "allocation");
    const char* names[] = {"K32 xcd", "K8 xcd", "K4 xcd", "K8 blk16", "K8 blk128", "K4 blk16", "K4 blk128", "K4 blk1024", "fill 4KB"};
    for (auto nm : names) printf("%11s", nm);
    printf("   (TB/s)\n");
    for (size_t du : dummies) {
        char* dummy = nullptr; if (du) CK(hipMalloc(&dummy, du));
        u32x4 *d, *m; CK(hipMalloc(&d, dist_bytes)); CK(hipMalloc(&m, mask_bytes));
        const unsigned n32 = groups / 32, n8 = groups / 8, n4 = groups / 4;   // all multiples of 8 * 1024
        std::vector<std::function<void()>> v = {
            [=] { kP<32><<<n32, 256>>>(d, m, n32, 0); }, [=] { kP<8><<<n8, 256>>>(d, m, n8, 0); },
            [=] { kP<4><<<n4, 256>>>(d, m, n4, 0); },    [=] { kP<8><<<n8, 256>>>(d, m, n8, 16); },
            [=] { kP<8><<<n8, 256>>>(d, m, n8, 128); },  [=] { kP<4><<<n4, 256>>>(d, m, n4, 16); },
            [=] { kP<4><<<n4, 256>>>(d, m, n4, 128); },  [=] { kP<4><<<n4, 256>>>(d, m, n4, 1024); },
            [=] { kFill<<<(unsigned)(dist_bytes / 4096), 256>>>(d); kFill<<<(unsigned)(mask_bytes / 4096), 256>>>(m); }};
        std::vector<std::vector<float>> t(v.size());
        for (int i = 0; i < 10; ++i) v[0]();
        CK(hipDeviceSynchronize());
        for (int round = 0; round < 3; ++round)
            for (size_t i = 0; i < v.size(); ++i) {
                v[i](); CK(hipDeviceSynchronize());
                CK(hipEventRecord(a)); for (int r = 0; r < 4; ++r) v[i](); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b)); t[i].push_back(ms / 4);
            }
        printf("dummy %5.1f GB", du / 1073741824.0);
        for (size_t i = 0; i < v.size(); ++i) {
            std::sort(t[i].begin(), t[i].end());
            printf("%11.2f", (dist_bytes + mask_bytes) / t[i][1] / 1e9);
        }
        printf("\n");
        CK(hipFree(d)); CK(hipFree(m)); if (dummy) CK(hipFree(dummy));
    }
    return 0;
}
