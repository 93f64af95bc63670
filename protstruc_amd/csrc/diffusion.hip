// K5 / K6 -- the per-coordinate ops of the diffusion loop.
//   K5  diffuse_xyz   (reference protstruc.py:864-878)
//   K6  standardize   (reference protstruc.py:696-734) and the affine map of
//       unstandardize (reference protstruc.py:736-744)
// All three update xyz IN PLACE so a hipGraph-captured loop can keep one static
// coordinate buffer; at the BASELINE sizes (<= 18 MB) the buffer is
// L2 / Infinity-Cache resident and the kernels are launch-latency bound.
#include "ps_common.hpp"

namespace {

// ---------------- Philox4x32-10 counter-based generator ----------------
struct u32x4 {
    uint32_t x, y, z, w;
};

__device__ __forceinline__ u32x4 philox4x32_10(u32x4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = u32x4{hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

// 24 random bits -> uniform in (0,1), never 0 or 1
__device__ __forceinline__ float u01(uint32_t r) { return ((float)(r >> 8) + 0.5f) * (1.0f / 16777216.0f); }

// four standard normals for float4 group `group` of draw number `offset`
__device__ __forceinline__ void normal4(uint64_t seed, uint64_t offset, uint64_t group, float z[4]) {
    u32x4 r = philox4x32_10(u32x4{(uint32_t)group, (uint32_t)(group >> 32), (uint32_t)offset, (uint32_t)(offset >> 32)},
                            (uint32_t)seed, (uint32_t)(seed >> 32));
    const float r0 = sqrtf(-2.0f * logf(u01(r.x)));
    const float r1 = sqrtf(-2.0f * logf(u01(r.z)));
    float s0, c0, s1, c1;
    sincosf(6.283185307179586f * u01(r.y), &s0, &c0);
    sincosf(6.283185307179586f * u01(r.w), &s1, &c1);
    z[0] = r0 * c0;
    z[1] = r0 * s0;
    z[2] = r1 * c1;
    z[3] = r1 * s1;
}

// K5: one lane per group of four consecutive coordinates
__global__ __launch_bounds__(256) void k5_diffuse(float* __restrict__ xyz, const float* __restrict__ beta,
                                                  size_t n_total, unsigned n_per_struct,
                                                  const uint64_t* __restrict__ rng_state,
                                                  const float* __restrict__ noise) {
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t e0 = g * 4;
    if (e0 >= n_total) return;
    float eps[4];
    const bool full = e0 + 4 <= n_total;
    if (noise) {
        if (full) {
            float4 t = *reinterpret_cast<const float4*>(noise + e0);
            eps[0] = t.x; eps[1] = t.y; eps[2] = t.z; eps[3] = t.w;
        } else {
            for (int k = 0; k < 4; ++k) eps[k] = (e0 + k < n_total) ? noise[e0 + k] : 0.f;
        }
    } else {
        normal4(rng_state[0], rng_state[1], g, eps);
    }
    float x[4];
    if (full) {
        float4 t = *reinterpret_cast<const float4*>(xyz + e0);
        x[0] = t.x; x[1] = t.y; x[2] = t.z; x[3] = t.w;
    } else {
        for (int k = 0; k < 4; ++k) x[k] = (e0 + k < n_total) ? xyz[e0 + k] : 0.f;
    }
    // structure index of each coordinate (a group may straddle two structures)
    size_t b = e0 / n_per_struct;
    size_t next = (b + 1) * (size_t)n_per_struct;
    float bt = beta[b];
    float keep = sqrtf(1.0f - bt), add = sqrtf(bt);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (e0 + k >= next && e0 + k < n_total) {
            ++b;
            next += n_per_struct;
            bt = beta[b];
            keep = sqrtf(1.0f - bt);
            add = sqrtf(bt);
        }
        const float scaled = eps[k] * add;   // noise = randn * beta.sqrt()
        x[k] = keep * x[k] + scaled;         // (1 - beta).sqrt() * xyz + noise
    }
    if (full) {
        *reinterpret_cast<float4*>(xyz + e0) = make_float4(x[0], x[1], x[2], x[3]);
    } else {
        for (int k = 0; k < 4; ++k)
            if (e0 + k < n_total) xyz[e0 + k] = x[k];
    }
}

__global__ void k5_advance(uint64_t* rng_state) { rng_state[1] += 1; }

// nan_to_num(0.0): NaN -> 0, +-inf -> +-FLT_MAX
__device__ __forceinline__ float nan_to_num0(float v) {
    if (v != v) return 0.f;
    if (v == __builtin_huge_valf()) return 3.4028234663852886e38f;
    if (v == -__builtin_huge_valf()) return -3.4028234663852886e38f;
    return v;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = PS_WAVE / 2; o > 0; o >>= 1) v += __shfl_down(v, o);
    return v;
}

// block-wide sum of four doubles; result valid in every thread
__device__ __forceinline__ void block_sum4(double v[4], double* red /* [4][16] + [4] */) {
    const int lane = threadIdx.x & (PS_WAVE - 1), wave = threadIdx.x / PS_WAVE, nw = blockDim.x / PS_WAVE;
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = wave_sum(v[k]);
    __syncthreads();
    if (lane == 0)
        for (int k = 0; k < 4; ++k) red[k * 16 + wave] = v[k];
    __syncthreads();
    if (threadIdx.x < 4) {
        double s = 0;
        for (int w = 0; w < nw; ++w) s += red[threadIdx.x * 16 + w];
        red[64 + threadIdx.x] = s;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = red[64 + k];
}

// K6: one workgroup per structure, three sweeps of the (cache-resident) structure
__global__ __launch_bounds__(1024) void k6_standardize(float* __restrict__ xyz, const uint8_t* __restrict__ amask,
                                                       float* __restrict__ mu_out, float* __restrict__ std_out,
                                                       int n_atoms) {
    __shared__ double red[68];
    const int b = blockIdx.x;
    float* x = xyz + (size_t)b * n_atoms * 3;
    const uint8_t* m = amask ? amask + (size_t)b * n_atoms : nullptr;

    double acc[4] = {0, 0, 0, 0};  // sum x, sum y, sum z, count
    for (int a = threadIdx.x; a < n_atoms; a += blockDim.x) {
        const float w = m ? (m[a] ? 1.f : 0.f) : 1.f;
        acc[0] += (double)nan_to_num0(x[a * 3 + 0] * w);
        acc[1] += (double)nan_to_num0(x[a * 3 + 1] * w);
        acc[2] += (double)nan_to_num0(x[a * 3 + 2] * w);
        acc[3] += (double)w;
    }
    block_sum4(acc, red);
    const float cnt = (float)acc[3];
    const float mu[3] = {(float)acc[0] / cnt, (float)acc[1] / cnt, (float)acc[2] / cnt};

    double sq[4] = {0, 0, 0, 0};
    for (int a = threadIdx.x; a < n_atoms; a += blockDim.x) {
        const float w = m ? (m[a] ? 1.f : 0.f) : 1.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float d = nan_to_num0(x[a * 3 + k]) - mu[k];
            sq[k] += (double)((d * d) * w);
        }
    }
    block_sum4(sq, red);
    const float sd[3] = {sqrtf((float)sq[0] / cnt), sqrtf((float)sq[1] / cnt), sqrtf((float)sq[2] / cnt)};

    for (int a = threadIdx.x; a < n_atoms; a += blockDim.x) {
#pragma unroll
        for (int k = 0; k < 3; ++k) x[a * 3 + k] = (x[a * 3 + k] - mu[k]) / sd[k];
    }
    if (threadIdx.x < 3) {
        mu_out[b * 3 + threadIdx.x] = mu[threadIdx.x];
        std_out[b * 3 + threadIdx.x] = sd[threadIdx.x];
    }
}

__global__ __launch_bounds__(256) void k6_affine(float* __restrict__ xyz, const float* __restrict__ scale,
                                                 const float* __restrict__ shift, unsigned n_atoms, size_t total_atoms) {
    const size_t a = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (a >= total_atoms) return;
    const size_t b = a / n_atoms;
    float* p = xyz + a * 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) p[k] = p[k] * scale[b * 3 + k] + shift[b * 3 + k];
}

}  // namespace

extern "C" int ps_diffuse_f32(float* xyz, const float* beta, int B, int n_per_struct, uint64_t* rng_state,
                              const float* noise, void* stream) {
    if (!xyz || !beta || B < 0 || n_per_struct < 0 || (!rng_state && !noise)) return (int)hipErrorInvalidValue;
    if ((reinterpret_cast<uintptr_t>(xyz) & 15) || (noise && (reinterpret_cast<uintptr_t>(noise) & 15)))
        return (int)hipErrorInvalidValue;
    const size_t n_total = (size_t)B * n_per_struct;
    if (n_total == 0) return 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const size_t groups = (n_total + 3) / 4;
    hipLaunchKernelGGL(k5_diffuse, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s, xyz, beta, n_total,
                       (unsigned)n_per_struct, rng_state, noise);
    int rc = ps_check_launch();
    if (rc) return rc;
    if (!noise) {
        hipLaunchKernelGGL(k5_advance, dim3(1), dim3(1), 0, s, rng_state);
        rc = ps_check_launch();
    }
    return rc;
}

extern "C" int ps_standardize_f32(float* xyz, const uint8_t* atom_mask, float* mu, float* std, int B, int N, int A,
                                  void* stream) {
    if (!xyz || !mu || !std || B < 0 || N < 0 || A <= 0) return (int)hipErrorInvalidValue;
    if (B == 0 || N == 0) return 0;
    hipLaunchKernelGGL(k6_standardize, dim3(B), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), xyz, atom_mask,
                       mu, std, N * A);
    return ps_check_launch();
}

extern "C" int ps_affine_f32(float* xyz, const float* scale, const float* shift, int B, int n_atoms_per_struct,
                             void* stream) {
    if (!xyz || !scale || !shift || B < 0 || n_atoms_per_struct < 0) return (int)hipErrorInvalidValue;
    const size_t total = (size_t)B * n_atoms_per_struct;
    if (total == 0) return 0;
    hipLaunchKernelGGL(k6_affine, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), xyz, scale, shift, (unsigned)n_atoms_per_struct, total);
    return ps_check_launch();
}
