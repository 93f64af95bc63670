// The restatements of the device library's atan2f / acosf and of the compiler's IEEE division (ps_common.hpp: the scalar
// atan2_lib / acos_lib of the one-column kernels and the packed atan2_lib_vn / acos_lib_vn / div_ieee_vn of the FAITHFUL
// sweep kernels of K3) against the library calls themselves, bit for bit:
//   acosf:   every one of the 2^32 float bit patterns;
//   atan2f:  2^32 (y, x) pairs -- a quarter with both words uniformly random bit patterns (NaN, inf, subnormals, huge exponent
//            gaps included), a quarter with |y| / |x| within 2^-8 .. 2^8 (where the polynomial works), a quarter with one
//            operand among {+-0, +-inf, NaN, +-FLT_MIN, +-FLT_MAX, subnormal} and the rest random normal pairs at any exponent;
//   a / b:   2^32 pairs, same mixture.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -I protstruc_amd/csrc tools/microbench/libm_identity.hip -o tools/microbench/libm_identity
// Prints the mismatch count of each (must be 0) and the first few mismatching inputs.  (The same flags as the library:
// -ffp-contract=off is part of the contract.)
#include "ps_common.hpp"

#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ uint32_t mix(uint32_t a) {   // lowbias32
    a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16;
    return a;
}

__device__ __forceinline__ float special(uint32_t k) {
    const uint32_t t[12] = {0x00000000u, 0x80000000u, 0x7f800000u, 0xff800000u, 0x7fc00000u, 0x00800000u, 0x80800000u, 0x7f7fffffu,
                            0xff7fffffu, 0x00000001u, 0x807fffffu, 0x3f800000u};
    return __uint_as_float(t[k % 12u]);
}

// the (y, x) pair of global index `i` (0 .. 2^32 - 1)
__device__ __forceinline__ void pair_of(uint32_t i, float& y, float& x) {
    const uint32_t a = mix(i), b = mix(i ^ 0x9e3779b9u), cls = i >> 30;
    if (cls == 0) { y = __uint_as_float(a); x = __uint_as_float(b); }
    else if (cls == 1) {   // comparable magnitudes
        x = __uint_as_float((b & 0x807fffffu) | (((b >> 23) % 200u + 27u) << 23));
        const int e = (int)((__float_as_uint(x) >> 23) & 255u) + (int)(a >> 28) - 8;
        y = __uint_as_float((a & 0x807fffffu) | ((uint32_t)min(max(e, 1), 254) << 23));
    } else if (cls == 2) {
        y = (a & 1u) ? special(a >> 1) : __uint_as_float(a);
        x = (a & 1u) ? __uint_as_float(b) : special(b >> 1);
        if ((a & 6u) == 6u) { y = special(a >> 3); x = special(b >> 3); }
    } else {
        y = __uint_as_float((a & 0x807fffffu) | (((a >> 23) % 254u + 1u) << 23));
        x = __uint_as_float((b & 0x807fffffu) | (((b >> 23) % 254u + 1u) << 23));
    }
}

__device__ __forceinline__ bool same_bits(float a, float b) {
    return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b);   // any NaN equals any NaN (payloads are not part of the contract)
}

// what: 0 acos (exhaustive), 1 atan2, 2 division
__global__ __launch_bounds__(256) void check(int what, unsigned long long* bad, uint32_t* first, unsigned chunk) {
    const uint32_t base = ((uint32_t)blockIdx.x * 256u + threadIdx.x) * (2u * chunk);
    unsigned long long n_bad = 0;
    for (uint32_t k = 0; k < chunk; ++k) {
        const uint32_t i0 = base + 2u * k, i1 = i0 + 1u;
        f32x2 got[1], want;
        if (what == 0) {
            const f32x2 x[1] = {f32x2{__uint_as_float(i0), __uint_as_float(i1)}};
            acos_lib_vn<1>(x, got);
            want = f32x2{acosf(x[0].x), acosf(x[0].y)};
        } else {
            float y0, x0, y1, x1;
            pair_of(i0, y0, x0); pair_of(i1, y1, x1);
            const f32x2 y[1] = {f32x2{y0, y1}}, x[1] = {f32x2{x0, x1}};
            if (what == 1) {
                atan2_lib_vn<1>(y, x, got);
                want = f32x2{atan2f(y0, x0), atan2f(y1, x1)};
            } else {
                div_ieee_vn<1>(y, x, got);
                want = f32x2{y0 / x0, y1 / x1};
            }
        }
        // the scalar restatements (what the one-column kernels run) against the same calls
        f32x2 sc;
        if (what == 0) sc = f32x2{acos_lib(__uint_as_float(i0)), acos_lib(__uint_as_float(i1))};
        else {
            float y0, x0, y1, x1;
            pair_of(i0, y0, x0); pair_of(i1, y1, x1);
            sc = what == 1 ? f32x2{atan2_lib(y0, x0), atan2_lib(y1, x1)} : want;
        }
        const bool b0 = !same_bits(got[0].x, want.x) || !same_bits(sc.x, want.x), b1 = !same_bits(got[0].y, want.y) || !same_bits(sc.y, want.y);
        if (b0 || b1) {
            if (n_bad == 0 && atomicAdd(reinterpret_cast<unsigned*>(first), 1u) < 8u) {
                const unsigned slot = atomicAdd(reinterpret_cast<unsigned*>(first) + 1, 1u);
                if (slot < 8u) first[2 + slot] = b0 ? i0 : i1;
            }
            n_bad += (b0 ? 1 : 0) + (b1 ? 1 : 0);
        }
    }
    if (n_bad) atomicAdd(bad, n_bad);
}

// two columns: the interleaved form used by the kernels (NC = 2), spot-checked on 2^28 pairs against NC = 1
__global__ __launch_bounds__(256) void check_nc2(unsigned long long* bad) {
    const uint32_t t = (uint32_t)blockIdx.x * 256u + threadIdx.x;
    unsigned long long n_bad = 0;
    for (uint32_t k = 0; k < 64u; ++k) {
        const uint32_t i = (t * 64u + k) * 4u;
        float yy[4], xx[4];
        for (int q = 0; q < 4; ++q) pair_of(i * 16u + (uint32_t)q * 0x40000001u, yy[q], xx[q]);
        const f32x2 y[2] = {f32x2{yy[0], yy[1]}, f32x2{yy[2], yy[3]}}, x[2] = {f32x2{xx[0], xx[1]}, f32x2{xx[2], xx[3]}};
        f32x2 a[2], d[2], c[2];
        atan2_lib_vn<2>(y, x, a);
        div_ieee_vn<2>(y, x, d);
        acos_lib_vn<2>(y, c);
        for (int q = 0; q < 4; ++q) {
            const float ga = (q & 1) ? a[q >> 1].y : a[q >> 1].x, gd = (q & 1) ? d[q >> 1].y : d[q >> 1].x, gc = (q & 1) ? c[q >> 1].y : c[q >> 1].x;
            n_bad += !same_bits(ga, atan2f(yy[q], xx[q])) + !same_bits(gd, yy[q] / xx[q]) + !same_bits(gc, acosf(yy[q]));
        }
    }
    if (n_bad) atomicAdd(bad, n_bad);
}

int main() {
    unsigned long long* bad;
    uint32_t* first;
    hipMalloc(&bad, 8);
    hipMalloc(&first, 64);
    const char* names[3] = {"acosf  (all 2^32 arguments)", "atan2f (2^32 pairs)", "a / b  (2^32 pairs)"};
    int rc = 0;
    for (int what = 0; what < 3; ++what) {
        hipMemset(bad, 0, 8);
        hipMemset(first, 0, 64);
        const unsigned chunk = 512;                               // 2 * 512 arguments per thread
        const unsigned threads = (unsigned)((1ull << 32) / (2ull * chunk));
        check<<<threads / 256u, 256>>>(what, bad, first, chunk);
        unsigned long long h = 0;
        uint32_t hf[16];
        hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
        hipMemcpy(hf, first, 64, hipMemcpyDeviceToHost);
        printf("%-30s mismatches: %llu", names[what], h);
        for (unsigned k = 0; k < 8 && k < hf[1]; ++k) printf(" [i=0x%08x]", hf[2 + k]);
        printf("\n");
        rc |= h != 0;
    }
    hipMemset(bad, 0, 8);
    check_nc2<<<(1u << 22) / 256u, 256>>>(bad);
    unsigned long long h = 0;
    hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
    printf("%-30s mismatches: %llu\n", "two-column forms (2^28 x 3)", h);
    rc |= h != 0;
    if (hipDeviceSynchronize() != hipSuccess) { printf("device error\n"); return 2; }
    return rc;
}
