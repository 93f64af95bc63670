// Store-pattern microbenchmark (experiment, not product): what write patterns reach the HBM ceiling?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <bool NT> __device__ __forceinline__ void st(u32x4* p, u32x4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// A: WG writes a contiguous chunk of `chunk16` 16-byte slots, 256 lanes interleaved (4 KB per iteration)
template <bool NT> __global__ __launch_bounds__(256) void kA(u32x4* out, unsigned chunk16) {
    u32x4* o = out + (size_t)blockIdx.x * chunk16;
    u32x4 v = {threadIdx.x, blockIdx.x, 1, 2};
    for (unsigned q = threadIdx.x; q < chunk16; q += 256) st<NT>(o + q, v);
}
// B: 225 active lanes, 3600-byte groups (the K1 pattern kernel's dist plane), ngroups per WG
template <bool NT> __global__ __launch_bounds__(256) void kB(u32x4* out, unsigned ngroups) {
    if (threadIdx.x >= 225) return;
    u32x4* o = out + (size_t)blockIdx.x * ngroups * 225 + threadIdx.x;
    u32x4 v = {threadIdx.x, blockIdx.x, 1, 2};
    for (unsigned g = 0; g < ngroups; ++g) st<NT>(o + g * 225, v);
}
// C: each WAVE writes its own contiguous quarter of the WG chunk
template <bool NT> __global__ __launch_bounds__(256) void kC(u32x4* out, unsigned chunk16) {
    const unsigned w = threadIdx.x >> 6, l = threadIdx.x & 63, per = chunk16 / 4;
    u32x4* o = out + (size_t)blockIdx.x * chunk16 + w * per;
    u32x4 v = {threadIdx.x, blockIdx.x, 1, 2};
    for (unsigned q = l; q < per; q += 64) st<NT>(o + q, v);
}
// D: grid-stride fill (persistent-ish): grid = G blocks, each sweeps the whole buffer with stride G*256
template <bool NT> __global__ __launch_bounds__(256) void kD(u32x4* out, size_t n16) {
    u32x4 v = {threadIdx.x, blockIdx.x, 1, 2};
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n16; q += (size_t)gridDim.x * 256) st<NT>(out + q, v);
}
// E: like A but each lane writes 4 consecutive slots per iteration group (unrolled x4, 16 KB per WG iteration)
template <bool NT> __global__ __launch_bounds__(256) void kE(u32x4* out, unsigned chunk16) {
    u32x4* o = out + (size_t)blockIdx.x * chunk16;
    u32x4 v = {threadIdx.x, blockIdx.x, 1, 2};
    unsigned q = threadIdx.x;
    for (; q + 768 < chunk16; q += 1024) { st<NT>(o + q, v); st<NT>(o + q + 256, v); st<NT>(o + q + 512, v); st<NT>(o + q + 768, v); }
    for (; q < chunk16; q += 256) st<NT>(o + q, v);
}
// F: two planes like K1: dist chunk (3600 slots = 57600 B) + mask chunk (900 slots = 14400 B) per WG, 256 lanes
template <bool NT> __global__ __launch_bounds__(256) void kF(u32x4* out, u32x4* out2) {
    u32x4 v = {threadIdx.x, blockIdx.x, 1, 2};
    u32x4* o = out + (size_t)blockIdx.x * 3600;
    for (unsigned q = threadIdx.x; q < 3600; q += 256) st<NT>(o + q, v);
    u32x4* m = out2 + (size_t)blockIdx.x * 900;
    for (unsigned q = threadIdx.x; q < 900; q += 256) st<NT>(m + q, v);
}

// G: A with at most `K` stores outstanding per wave (s_waitcnt vmcnt)
template <int K> __global__ __launch_bounds__(256) void kG(u32x4* out, unsigned chunk16) {
    u32x4* o = out + (size_t)blockIdx.x * chunk16;
    u32x4 v = {threadIdx.x, blockIdx.x, 1, 2};
    for (unsigned q = threadIdx.x; q < chunk16; q += 256) {
        o[q] = v;
        if (K == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (K == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        if (K == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        if (K == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
}
// H: persistent chunk loop: WG w handles chunks w, w+G, w+2G, ... (chunk = chunk16 slots), compact moving window
__global__ __launch_bounds__(256) void kH(u32x4* out, unsigned chunk16, unsigned nchunks) {
    u32x4 v = {threadIdx.x, blockIdx.x, 1, 2};
    for (unsigned c = blockIdx.x; c < nchunks; c += gridDim.x) {
        u32x4* o = out + (size_t)c * chunk16;
        for (unsigned q = threadIdx.x; q < chunk16; q += 256) o[q] = v;
    }
}
// F2: two planes with a configurable granule: dist chunk d16 slots + mask chunk d16/4 slots per WG
__global__ __launch_bounds__(256) void kF2(u32x4* out, u32x4* out2, unsigned d16) {
    u32x4 v = {threadIdx.x, blockIdx.x, 1, 2};
    u32x4* o = out + (size_t)blockIdx.x * d16;
    for (unsigned q = threadIdx.x; q < d16; q += 256) o[q] = v;
    u32x4* m = out2 + (size_t)blockIdx.x * (d16 / 4);
    for (unsigned q = threadIdx.x; q < d16 / 4; q += 256) m[q] = v;
}
// F3: persistent two-plane: WG loops over (dist chunk, mask chunk) pairs with grid stride
__global__ __launch_bounds__(256) void kF3(u32x4* out, u32x4* out2, unsigned d16, unsigned nchunks) {
    u32x4 v = {threadIdx.x, blockIdx.x, 1, 2};
    for (unsigned c = blockIdx.x; c < nchunks; c += gridDim.x) {
        u32x4* o = out + (size_t)c * d16;
        for (unsigned q = threadIdx.x; q < d16; q += 256) o[q] = v;
        u32x4* m = out2 + (size_t)c * (d16 / 4);
        for (unsigned q = threadIdx.x; q < d16 / 4; q += 256) m[q] = v;
    }
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

int main() {
    const size_t dist_bytes = 64ull * 512 * 512 * 900, mask_bytes = 64ull * 512 * 512 * 225;
    u32x4 *d, *m; CK(hipMalloc(&d, dist_bytes)); CK(hipMalloc(&m, mask_bytes));
    const size_t n16 = dist_bytes / 16;
    auto rep = [&](const char* name, float ms, size_t bytes) { printf("%-44s %7.3f ms  %6.2f TB/s\n", name, ms, bytes / ms / 1e9); fflush(stdout); };
    for (unsigned chunk16 : {1024u, 3584u, 3600u, 14336u, 57344u}) {
        unsigned nb = (unsigned)(n16 / chunk16); size_t bytes = (size_t)nb * chunk16 * 16; char nm[96];
        snprintf(nm, 96, "A interleaved  chunk=%6u B", chunk16 * 16); rep(nm, timeit([&] { kA<false><<<nb, 256>>>(d, chunk16); }), bytes);
        snprintf(nm, 96, "A interleaved  chunk=%6u B nt", chunk16 * 16); rep(nm, timeit([&] { kA<true><<<nb, 256>>>(d, chunk16); }), bytes);
        snprintf(nm, 96, "E unroll4      chunk=%6u B", chunk16 * 16); rep(nm, timeit([&] { kE<false><<<nb, 256>>>(d, chunk16); }), bytes);
        if (chunk16 % 256 == 0) { snprintf(nm, 96, "C wave-contig  chunk=%6u B", chunk16 * 16); rep(nm, timeit([&] { kC<false><<<nb, 256>>>(d, chunk16); }), bytes);
                                  snprintf(nm, 96, "C wave-contig  chunk=%6u B nt", chunk16 * 16); rep(nm, timeit([&] { kC<true><<<nb, 256>>>(d, chunk16); }), bytes); }
    }
    for (unsigned ng : {4u, 16u, 64u}) {
        unsigned nb = (unsigned)(n16 / (ng * 225)); size_t bytes = (size_t)nb * ng * 225 * 16; char nm[96];
        snprintf(nm, 96, "B 225-lane groups=%u (%u B/WG)", ng, ng * 3600); rep(nm, timeit([&] { kB<false><<<nb, 256>>>(d, ng); }), bytes);
        snprintf(nm, 96, "B 225-lane groups=%u nt", ng); rep(nm, timeit([&] { kB<true><<<nb, 256>>>(d, ng); }), bytes);
    }
    for (unsigned G : {256u, 512u, 1024u, 2048u, 4096u, 8192u}) {
        char nm[96]; snprintf(nm, 96, "D grid-stride grid=%u", G); rep(nm, timeit([&] { kD<false><<<G, 256>>>(d, n16); }), dist_bytes);
        snprintf(nm, 96, "D grid-stride grid=%u nt", G); rep(nm, timeit([&] { kD<true><<<G, 256>>>(d, n16); }), dist_bytes);
    }
    { unsigned nb = 64 * 512 * 8; rep("F two planes 57600+14400 B/WG", timeit([&] { kF<false><<<nb, 256>>>(d, m); }), dist_bytes + mask_bytes);
      rep("F two planes nt", timeit([&] { kF<true><<<nb, 256>>>(d, m); }), dist_bytes + mask_bytes); }
    for (unsigned chunk16 : {64u, 128u, 256u, 512u}) {
        unsigned nb = (unsigned)(n16 / chunk16); size_t bytes = (size_t)nb * chunk16 * 16; char nm[96];
        snprintf(nm, 96, "A small chunk=%6u B", chunk16 * 16); rep(nm, timeit([&] { kA<false><<<nb, 256>>>(d, chunk16); }), bytes);
    }
    { unsigned chunk16 = 3600, nb = (unsigned)(n16 / chunk16); size_t bytes = (size_t)nb * chunk16 * 16;
      rep("G chunk=57600 vmcnt(0) after each store", timeit([&] { kG<0><<<nb, 256>>>(d, chunk16); }), bytes);
      rep("G chunk=57600 vmcnt(1)", timeit([&] { kG<1><<<nb, 256>>>(d, chunk16); }), bytes);
      rep("G chunk=57600 vmcnt(2)", timeit([&] { kG<2><<<nb, 256>>>(d, chunk16); }), bytes);
      rep("G chunk=57600 vmcnt(4)", timeit([&] { kG<4><<<nb, 256>>>(d, chunk16); }), bytes);
      for (unsigned G : {256u, 512u, 1024u, 2048u}) { char nm[96]; snprintf(nm, 96, "H persistent chunk=57600 grid=%u", G);
        rep(nm, timeit([&] { kH<<<G, 256>>>(d, chunk16, nb); }), bytes); }
      for (unsigned G : {256u, 512u, 1024u, 2048u}) { char nm[96]; snprintf(nm, 96, "H persistent chunk=14400 grid=%u", G);
        rep(nm, timeit([&] { kH<<<G, 256>>>(d, 900, nb * 4); }), bytes); }
    }
    for (unsigned d16 : {900u, 1800u, 3600u, 7200u}) { unsigned nb = (unsigned)(n16 / d16); char nm[96];
      snprintf(nm, 96, "F2 two planes granule %u+%u B", d16 * 16, d16 * 4); rep(nm, timeit([&] { kF2<<<nb, 256>>>(d, m, d16); }), dist_bytes + mask_bytes);
      for (unsigned G : {256u, 512u, 1024u}) { snprintf(nm, 96, "F3 persistent two planes %u B grid=%u", d16 * 16, G);
        rep(nm, timeit([&] { kF3<<<G, 256>>>(d, m, d16, nb); }), dist_bytes + mask_bytes); } }
    rep("hipMemsetAsync", timeit([&] { CK(hipMemsetAsync(d, 1, dist_bytes, 0)); }), dist_bytes);
    return 0;
}
