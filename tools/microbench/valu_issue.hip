// VALU issue cost per instruction class on gfx950, measured.  Every lane runs `iters` trips over 8 independent accumulators
// (no dependent-issue stall inside a wave), 1 / 2 / 4 waves per SIMD on all CUs; HIP events over the launch -> ns per
// wave-instruction per SIMD (min of 5 launches).  The last column is cycles at 2.4 GHz with four waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench/valu_issue.hip -o tools/microbench/valu_issue
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f32x2 __attribute__((ext_vector_type(2)));

#define SPIN_KERNEL(NAME, PER_TRIP, BODY)                                                                              \
    __global__ __launch_bounds__(1024) void NAME(int iters, float* out) {                                              \
        float a[8];                                                                                                    \
        f32x2 p[8];                                                                                                    \
        for (int k = 0; k < 8; ++k) { a[k] = 1.0f + 0.001f * (float)(threadIdx.x + k); p[k] = f32x2{a[k], a[k] + 0.5f}; } \
        const float c0 = 0.999f, c1 = 0.001f;                                                                          \
        const f32x2 q0 = {0.999f, 0.998f}, q1 = {0.001f, 0.002f};                                                      \
        (void)c0; (void)c1; (void)q0; (void)q1;                                                                        \
        unsigned long long sm = 0x5555555555555555ull;                                                                 \
        asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[0]), "v"(c0) : "vcc");                                       \
        asm volatile("" : "+s"(sm));                                                                                   \
        for (int i = 0; i < iters; ++i) {                                                                              \
            _Pragma("unroll") for (int k = 0; k < 8; ++k) { BODY; }                                                    \
        }                                                                                                              \
        float s = 0.0f;                                                                                                \
        for (int k = 0; k < 8; ++k) s += a[k] + p[k].x + p[k].y;                                                       \
        if (s == 12345.678f) out[0] = s;                                                                               \
    }                                                                                                                  \
    static const double NAME##_per_trip = PER_TRIP;

SPIN_KERNEL(k_fma, 1, asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(c0), "v"(c1)))
SPIN_KERNEL(k_mul, 1, asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c0)))
SPIN_KERNEL(k_add, 1, asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c1)))
SPIN_KERNEL(k_max, 1, asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c1)))
SPIN_KERNEL(k_mov, 1, asm volatile("v_mov_b32 %0, %1" : "+v"(a[k]) : "v"(c1)))
SPIN_KERNEL(k_and, 1, asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[k]) : "v"(c1)))
SPIN_KERNEL(k_bfi, 1, asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[k]) : "v"(c0), "v"(c1)))
SPIN_KERNEL(k_pk_fma, 1, asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[k]) : "v"(q0), "v"(q1)))
SPIN_KERNEL(k_pk_mul, 1, asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[k]) : "v"(q0)))
SPIN_KERNEL(k_pk_add, 1, asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[k]) : "v"(q1)))
SPIN_KERNEL(k_pk_mov, 1, asm volatile("v_pk_mov_b32 %0, %1, %1" : "+v"(p[k]) : "v"(q1)))
SPIN_KERNEL(k_rsq, 1, asm volatile("v_rsq_f32 %0, %0" : "+v"(a[k])))
SPIN_KERNEL(k_sqrt, 1, asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[k])))
SPIN_KERNEL(k_rcp, 1, asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k])))
SPIN_KERNEL(k_frexp, 1, asm volatile("v_frexp_mant_f32 %0, %0" : "+v"(a[k])))
SPIN_KERNEL(k_ldexp, 1, asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(a[k]) : "v"(1)))
SPIN_KERNEL(k_cnd_vcc, 1, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(c0)))
SPIN_KERNEL(k_cnd_sgpr, 1, asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(c0), "s"(sm)))
SPIN_KERNEL(k_cmp_vcc, 1, asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[k]), "v"(c0) : "vcc"))
SPIN_KERNEL(k_cmp_cnd, 2, asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(c0) : "vcc"))
SPIN_KERNEL(k_cmp_sgpr_cnd, 2, { unsigned long long m; asm volatile("v_cmp_lt_f32_e64 %1, %0, %2\n v_cndmask_b32_e64 %0, %0, %2, %1" : "+v"(a[k]), "=&s"(m) : "v"(c0)); })
SPIN_KERNEL(k_div_scale, 1, asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(a[k]) : "v"(c0) : "vcc"))
SPIN_KERNEL(k_div_fmas, 1, asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(c0), "v"(c1) : "vcc"))
SPIN_KERNEL(k_div_fixup, 1, asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(c0), "v"(c1)))
SPIN_KERNEL(k_fma_dep, 1, asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(c0), "v"(c1)))          // one dependent chain
SPIN_KERNEL(k_pk_fma_dep, 1, asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[0]) : "v"(q0), "v"(q1)))
SPIN_KERNEL(k_rsq_dep, 1, asm volatile("v_rsq_f32 %0, %0" : "+v"(a[0])))
SPIN_KERNEL(k_mix, 4.125, { asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n v_pk_mul_f32 %0, %0, %1\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_add_f32 %0, %0, %2" : "+v"(p[k]) : "v"(q0), "v"(q1)); if (k == 7) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[0])); })

static int g_cus;
static float* g_out;

template <typename K>
void bench(const char* name, K kernel, double per_trip) {
    const int iters = 20000;
    double t[3];
    const int w[3] = {1, 2, 4};
    for (int k = 0; k < 3; ++k) {
        const int block = 256 * w[k], grid = g_cus;
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        kernel<<<grid, block>>>(iters / 8, g_out);
        (void)hipDeviceSynchronize();
        float best = 1e30f;
        for (int r = 0; r < 5; ++r) {
            (void)hipEventRecord(e0);
            kernel<<<grid, block>>>(iters, g_out);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        t[k] = (double)best * 1e6 / ((double)iters * 8.0 * per_trip * w[k]);
    }
    printf("%-34s %10.3f %10.3f %10.3f   %6.2f\n", name, t[0], t[1], t[2], t[2] * 2.4);
}

#define BENCH(NAME, LABEL) bench(LABEL, NAME, NAME##_per_trip)

int main() {
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    g_cus = prop.multiProcessorCount;
    (void)hipMalloc(&g_out, 64);
    printf("%d CUs; ns per wave-instruction per SIMD (min of 5 launches, 160 000 instructions per lane)\n", g_cus);
    printf("%-34s %10s %10s %10s   cycles at 2.4 GHz, 4 waves\n", "", "1 wave/SIMD", "2", "4");
    BENCH(k_fma, "v_fma_f32"); BENCH(k_mul, "v_mul_f32"); BENCH(k_add, "v_add_f32"); BENCH(k_max, "v_max_f32"); BENCH(k_mov, "v_mov_b32");
    BENCH(k_and, "v_and_b32"); BENCH(k_bfi, "v_bfi_b32");
    BENCH(k_pk_fma, "v_pk_fma_f32"); BENCH(k_pk_mul, "v_pk_mul_f32"); BENCH(k_pk_add, "v_pk_add_f32"); BENCH(k_pk_mov, "v_pk_mov_b32");
    BENCH(k_rsq, "v_rsq_f32"); BENCH(k_sqrt, "v_sqrt_f32"); BENCH(k_rcp, "v_rcp_f32"); BENCH(k_frexp, "v_frexp_mant_f32"); BENCH(k_ldexp, "v_ldexp_f32");
    BENCH(k_cnd_vcc, "v_cndmask_b32 (vcc)"); BENCH(k_cnd_sgpr, "v_cndmask_b32 (sgpr pair)"); BENCH(k_cmp_vcc, "v_cmp_lt_f32 -> vcc");
    BENCH(k_cmp_cnd, "v_cmp -> vcc; v_cndmask (per instr)"); BENCH(k_cmp_sgpr_cnd, "v_cmp -> sgpr; v_cndmask (per instr)");
    BENCH(k_div_scale, "v_div_scale_f32"); BENCH(k_div_fmas, "v_div_fmas_f32"); BENCH(k_div_fixup, "v_div_fixup_f32");
    BENCH(k_fma_dep, "v_fma_f32, one dependent chain"); BENCH(k_pk_fma_dep, "v_pk_fma_f32, one dependent chain"); BENCH(k_rsq_dep, "v_rsq_f32, one dependent chain");
    BENCH(k_mix, "mix: 32 packed : 1 v_rsq_f32");
    return hipDeviceSynchronize() == hipSuccess ? 0 : 2;
}
