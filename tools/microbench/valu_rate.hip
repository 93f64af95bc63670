// VALU issue / execution rates on gfx950 that K3's floor depends on: cycles per wave-instruction per SIMD for
// v_fma_f32, v_pk_fma_f32, v_rcp_f32 and a 1:1 mix, at 1, 2, 4 and 8 waves per SIMD, independent and dependent chains.
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench/valu_rate tools/microbench/valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));

// MODE 0: 8 independent v_fma_f32; 1: 8 independent v_pk_fma_f32; 2: 8 independent v_rcp_f32; 3: 4 + 4 mix fma / pk_fma;
// 4: ONE dependent v_fma_f32 chain; 5: ONE dependent v_pk_fma_f32 chain; 6: 2 dependent pk chains; 7: 4 dependent pk chains
template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, int iters, float a, float b) {
    float x[8];
    f32x2 y[8];
    for (int k = 0; k < 8; ++k) { x[k] = a + threadIdx.x + k; y[k] = f32x2{a + k, b + threadIdx.x}; }
    const f32x2 a2 = {a, a}, b2 = {b, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) {
#pragma unroll
                for (int k = 0; k < 8; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(a), "v"(b));
            } else if (MODE == 1) {
#pragma unroll
                for (int k = 0; k < 8; ++k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[k]) : "v"(a2), "v"(b2));
            } else if (MODE == 2) {
#pragma unroll
                for (int k = 0; k < 8; ++k) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[k]));
            } else if (MODE == 3) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(a), "v"(b));
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[k]) : "v"(a2), "v"(b2));
                }
            } else if (MODE == 4) {
#pragma unroll
                for (int k = 0; k < 8; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[0]) : "v"(a), "v"(b));
            } else if (MODE == 5) {
#pragma unroll
                for (int k = 0; k < 8; ++k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n s_nop 0" : "+v"(y[0]) : "v"(a2), "v"(b2));
            } else if (MODE == 6) {
#pragma unroll
                for (int k = 0; k < 8; ++k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[k & 1]) : "v"(a2), "v"(b2));
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[k & 3]) : "v"(a2), "v"(b2));
            }
        }
    }
    float s = 0;
    for (int k = 0; k < 8; ++k) s += x[k] + y[k].x + y[k].y;
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, float* out) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount, iters = 4000;
    printf("%-44s", name);
    for (int wps : {1, 2, 4, 8}) {       // waves per SIMD: one workgroup of 4 * wps waves per CU
        const int threads = 256 * wps;
        if (threads > 1024) {           // two workgroups of 1024
            hipLaunchKernelGGL(k<MODE>, dim3(cus * 2), dim3(1024), 0, 0, out, 10, 1.0f, 0.5f);
        } else hipLaunchKernelGGL(k<MODE>, dim3(cus), dim3(threads), 0, 0, out, 10, 1.0f, 0.5f);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        if (threads > 1024) hipLaunchKernelGGL(k<MODE>, dim3(cus * 2), dim3(1024), 0, 0, out, iters, 1.0f, 0.5f);
        else hipLaunchKernelGGL(k<MODE>, dim3(cus), dim3(threads), 0, 0, out, iters, 1.0f, 0.5f);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double insts_per_simd = (double)iters * 64 * wps;   // wave-instructions issued on one SIMD
        printf("  %d w/SIMD: %.3f ns/inst", wps, ms * 1e6 / insts_per_simd);
    }
    printf("\n");
}

int main() {
    float* out; CK(hipMalloc(&out, 4096));
    printf("ns per wave-instruction per SIMD (x clock in GHz = cycles; the chip runs ~2.1-2.4 GHz)\n");
    run<0>("8 independent v_fma_f32", out);
    run<1>("8 independent v_pk_fma_f32", out);
    run<2>("8 independent v_rcp_f32", out);
    run<3>("4 v_fma_f32 + 4 v_pk_fma_f32 interleaved", out);
    run<4>("1 dependent v_fma_f32 chain", out);
    run<5>("1 dependent v_pk_fma_f32 chain (+ s_nop 0)", out);
    run<6>("2 dependent v_pk_fma_f32 chains", out);
    run<7>("4 dependent v_pk_fma_f32 chains", out);
    return 0;
}
