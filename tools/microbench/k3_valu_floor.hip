// What does K3's arithmetic alone cost?  The dihedral / planar-angle bodies of ps_common.hpp on register operands (no
// loads, no stores, no LDS), 4 columns x 2 rows per trip as in the strip kernels, at 1 / 2 / 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -o tools/microbench/k3_valu_floor tools/microbench/k3_valu_floor.hip
#include "../../protstruc_amd/csrc/ps_common.hpp"
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE>   // 0: dihedral (2,2)   1: planar (2,1)   2: dihedral without atan2 (x + y)   3: atan2 only
__global__ __launch_bounds__(1024) void k(float* out, int trips, float seed) {
    f3 cj[4], dj[4];
    for (int c = 0; c < 4; ++c) {
        cj[c] = mk3(seed + threadIdx.x * 0.37f + c, seed * 0.5f - c, 0.25f * threadIdx.x + c);
        dj[c] = mk3(seed * 1.5f - threadIdx.x * 0.11f, seed + c * 0.3f, 1.0f + c);
    }
    f3v a = mk3v(mk3(seed, 1.f, 2.f), mk3(0.5f, seed, 1.5f)), b = mk3v(mk3(1.f, seed, 0.f), mk3(2.f, 0.3f, seed));
    f32x2 acc = {0.f, 0.f};
    for (int t = 0; t < trips; ++t) {
        // new row points every trip (uniform, cheap): the compiler cannot hoist the row-only part
        a.x += f32x2{0.001f, 0.002f}; b.y += f32x2{0.003f, 0.001f};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f3v cv = mk3v(cj[c], cj[c]), dv = mk3v(dj[c], dj[c]);
            f32x2 v;
            if (MODE == 0) v = dihedral4v_k3(a, b, cv, dv);
            else if (MODE == 1) v = angle3v(a, b, dv);
            else if (MODE == 2) {
                const f3v b0 = sub3v(a, b), b1 = sub3v(cv, b), b2n = sub3v(cv, dv);
                const f3v n1 = cross3v(b0, b1), n2 = cross3v(b1, b2n);
                const f32x2 nn = (b1.x * b1.x + b1.y * b1.y) + b1.z * b1.z;
                v = dot3v(n1, n2) * f32x2{__builtin_amdgcn_rsqf(nn.x), __builtin_amdgcn_rsqf(nn.y)} + dot3v(n1, b2n);
            } else v = atan2_k3_v(a.x + cv.x, b.y + dv.y);
            acc += v;
        }
    }
    if (acc.x + acc.y == 12345.678f) out[threadIdx.x] = acc.x;
}

// the column-interleaved atan2 (atan2_k3_vn) is the product's, ps_common.hpp
__global__ __launch_bounds__(1024) void k_il(float* out, int trips, float seed) {
    f3 cj[4], dj[4];
    for (int c = 0; c < 4; ++c) {
        cj[c] = mk3(seed + threadIdx.x * 0.37f + c, seed * 0.5f - c, 0.25f * threadIdx.x + c);
        dj[c] = mk3(seed * 1.5f - threadIdx.x * 0.11f, seed + c * 0.3f, 1.0f + c);
    }
    f3v a = mk3v(mk3(seed, 1.f, 2.f), mk3(0.5f, seed, 1.5f)), b = mk3v(mk3(1.f, seed, 0.f), mk3(2.f, 0.3f, seed));
    f32x2 acc = {0.f, 0.f};
    for (int t = 0; t < trips; ++t) {
        a.x += f32x2{0.001f, 0.002f}; b.y += f32x2{0.003f, 0.001f};
        f32x2 x[4], y[4], r[4];
        const f3v b0 = sub3v(a, b);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f3v cv = mk3v(cj[c], cj[c]), dv = mk3v(dj[c], dj[c]);
            const f3v b1 = sub3v(cv, b), b2n = sub3v(cv, dv);
            const f3v n1 = cross3v(b0, b1), n2 = cross3v(b1, b2n);
            const f32x2 nn = (b1.x * b1.x + b1.y * b1.y) + b1.z * b1.z;
            x[c] = dot3v(n1, n2) * f32x2{__builtin_amdgcn_rsqf(nn.x), __builtin_amdgcn_rsqf(nn.y)};
            y[c] = dot3v(n1, b2n);
        }
        atan2_k3_vn<4>(y, x, r);
        for (int c = 0; c < 4; ++c) acc += r[c];
    }
    if (acc.x + acc.y == 12345.678f) out[threadIdx.x] = acc.x;
}

template <int MODE>
void run(const char* name, float* out) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount, trips = 2000;
    printf("%-40s", name);
    for (int wps : {1, 2, 4}) {
        const int threads = 256 * wps;
        if (MODE == 9) hipLaunchKernelGGL(k_il, dim3(cus), dim3(threads), 0, 0, out, 10, 1.0f);
        else hipLaunchKernelGGL(k<MODE>, dim3(cus), dim3(threads), 0, 0, out, 10, 1.0f);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        if (MODE == 9) hipLaunchKernelGGL(k_il, dim3(cus), dim3(threads), 0, 0, out, trips, 1.0f);
        else hipLaunchKernelGGL(k<MODE>, dim3(cus), dim3(threads), 0, 0, out, trips, 1.0f);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        // one trip of one wave = 8 pairs per lane; per SIMD `wps` waves x trips
        printf("  %d w/SIMD: %.3f us per wave-trip per SIMD (config 3 = 64 trips per SIMD -> %.1f us)", wps, ms * 1e3 / (trips * wps), ms * 1e3 / (trips * wps) * 64);
    }
    printf("\n");
}

int main() {
    float* out; CK(hipMalloc(&out, 4096));
    run<0>("dihedral (2,2), 4 cols x 2 rows", out);
    run<9>("dihedral (2,2), atan2 chains interleaved", out);
    run<2>("  without atan2", out);
    run<3>("  atan2 alone", out);
    run<1>("planar (2,1), 4 cols x 2 rows", out);
    return 0;
}
