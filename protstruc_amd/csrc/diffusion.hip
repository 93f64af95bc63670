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
        // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32) instead of a v_mul_hi_u32 / v_mul_lo_u32 pair: the
        // quarter-rate integer multiplies are what the sampler's time goes into, and this halves their number
        const uint64_t p0 = (uint64_t)0xD2511F53u * c.x, p1 = (uint64_t)0xCD9E8D57u * c.z;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        c = u32x4{hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

// 24 random bits -> uniform in (0,1), never 0 or 1
__device__ __forceinline__ float u01(uint32_t r) { return ((float)(r >> 8) + 0.5f) * (1.0f / 16777216.0f); }

// four standard normals for float4 group `group` of draw number `offset` (Box-Muller).
// The hardware transcendentals are used directly: v_log_f32 is log2, and v_sin_f32 / v_cos_f32
// take their argument in revolutions, so cos(2*pi*u) is one instruction.  Their ~1e-6 absolute error
// is irrelevant for a sampler and keeps K5 bandwidth-bound instead of VALU-bound.
__device__ __forceinline__ void normal4(uint64_t seed, uint64_t offset, uint64_t group, float z[4]) {
    u32x4 r = philox4x32_10(u32x4{(uint32_t)group, (uint32_t)(group >> 32), (uint32_t)offset, (uint32_t)(offset >> 32)},
                            (uint32_t)seed, (uint32_t)(seed >> 32));
    const float kNeg2Ln2 = -1.3862943611198906f;  // -2 ln 2: -2 ln u = (-2 ln 2) * log2 u
    const float r0 = __builtin_amdgcn_sqrtf(kNeg2Ln2 * __builtin_amdgcn_logf(u01(r.x)));
    const float r1 = __builtin_amdgcn_sqrtf(kNeg2Ln2 * __builtin_amdgcn_logf(u01(r.z)));
    const float t0 = u01(r.y), t1 = u01(r.w);
    z[0] = r0 * __builtin_amdgcn_cosf(t0);
    z[1] = r0 * __builtin_amdgcn_sinf(t0);
    z[2] = r1 * __builtin_amdgcn_cosf(t1);
    z[3] = r1 * __builtin_amdgcn_sinf(t1);
}

// sqrt(1-beta), sqrt(beta) for the structures a workgroup's contiguous element span touches, cached in
// LDS so that no lane divides or takes a square root per group.  The span [e_begin, e_begin+len) starts
// in structure b0 = e_begin / nps (one uniform division); lane k < n_span fills entry k.  All index math
// is 32-bit (the launchers reject >= 2^32 coordinates).
#define PS_MAX_SPAN 64  // structures cached per workgroup; longer spans (tiny structures) use the slow path

struct BetaSpan {
    unsigned b0, first_end, nps;  // first_end: first element index that belongs to structure b0 + 1
    bool cached;
};

__device__ __forceinline__ BetaSpan beta_span_fill(const float* __restrict__ beta, unsigned e_begin, unsigned len,
                                                   unsigned nps, unsigned n_struct, float2* lds_keep_add) {
    BetaSpan sp;
    sp.nps = nps;
    sp.b0 = e_begin / nps;
    sp.first_end = (sp.b0 + 1u) * nps;
    const unsigned n_span = (e_begin + len - 1u) / nps - sp.b0 + 1u;
    sp.cached = n_span <= PS_MAX_SPAN;
    if (sp.cached && threadIdx.x < n_span && sp.b0 + threadIdx.x < n_struct) {
        const float bt = beta[sp.b0 + threadIdx.x];
        lds_keep_add[threadIdx.x] = make_float2(sqrtf(1.0f - bt), sqrtf(bt));
    }
    return sp;
}

// (keep, add) of element e >= span start
__device__ __forceinline__ float2 beta_of(const BetaSpan& sp, const float* __restrict__ beta, unsigned e,
                                          const float2* lds_keep_add) {
    if (sp.cached) {
        unsigned k = 0;
        if (e >= sp.first_end) k = 1u + (e - sp.first_end) / sp.nps;   // rarely taken: most spans sit in one structure
        return lds_keep_add[k];
    }
    const float bt = beta[e / sp.nps];
    return make_float2(sqrtf(1.0f - bt), sqrtf(bt));
}

// Advance the draw counter without a second launch.  Every workgroup takes a ticket when it is done
// (its read of rng_state[1] happened-before: the value fed its stores); the last one bumps the offset.
// A single ticket word would serialise ~13 ns per workgroup (56 us for 4320 workgroups), so tickets are
// two-level: 32 sub-counters on their own 128-byte lines (workgroup k uses line k % 32), whose last
// arrivers take a ticket on the top word.  Every counter is reset by its last arriver.
// rng_state (uint64 words): [0] seed, [1] offset, [2] top ticket, [16 + 16*r] sub ticket r (r < 32).
#define PS_RNG_STATE_WORDS 528

__device__ __forceinline__ void rng_advance_by_last_block(uint64_t* rng_state, uint64_t off) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned r = blockIdx.x & 31u;
        const unsigned long long members = (gridDim.x - r + 31u) >> 5;  // workgroups with blockIdx % 32 == r
        unsigned long long* sub = reinterpret_cast<unsigned long long*>(rng_state + 16 + 16 * r);
        if (atomicAdd(sub, 1ull) == members - 1ull) {
            *sub = 0;
            const unsigned long long groups = gridDim.x < 32u ? gridDim.x : 32u;
            unsigned long long* top = reinterpret_cast<unsigned long long*>(rng_state + 2);
            if (atomicAdd(top, 1ull) == groups - 1ull) {
                *top = 0;
                rng_state[1] = off + 1;
            }
        }
    }
}

// The same bookkeeping with the ticket taken EARLY (round 4).  rng_advance_by_last_block takes its ticket at the very end of a
// workgroup, and the workgroup cannot retire before the returning atomic has come back (1-3 us with every CU issuing one):
// at config 5's shape that was 3 of K5's 11.5 us -- the injected-noise form, which reads 50 % more bytes but takes no ticket,
// ran 8.2 us (profiles/r04_k5_trace_stats.csv).  A ticket only has to follow the workgroup's READ of the offset, so:
// thread 0 alone reads (seed, offset) from memory, hands them to the other waves through LDS (they pick them up after the
// barrier the kernels have anyway) and takes the ticket at once -- the atomic's operand is made to depend on the loaded offset,
// so it cannot be performed before the read has returned -- and looks at the ticket's value only at the end, after the stores.
struct RngTicket {
    unsigned long long sub_ticket;
};

__device__ __forceinline__ RngTicket rng_read_and_take_ticket(uint64_t* rng_state, uint64_t (&rng_sh)[2]) {
    RngTicket t{0ull};
    if (threadIdx.x == 0) {
        const uint64_t seed = rng_state[0], off = rng_state[1];
        rng_sh[0] = seed;
        rng_sh[1] = off;
        unsigned zero;
        asm volatile("v_and_b32 %0, 0, %1" : "=v"(zero) : "v"((unsigned)off));   // 0, but only once `off` has arrived
        const unsigned r = blockIdx.x & 31u;
        unsigned long long* sub = reinterpret_cast<unsigned long long*>(rng_state + 16 + 16 * r);
        t.sub_ticket = atomicAdd(sub, 1ull + (unsigned long long)zero);
    }
    return t;
}

// The ticket alone, for kernels whose waves read (seed, offset) themselves and have USED them before a workgroup barrier:
// called by every thread right after that barrier, thread 0 takes the ticket.
__device__ __forceinline__ RngTicket rng_take_ticket(uint64_t* rng_state) {
    RngTicket t{0ull};
    if (threadIdx.x == 0) {
        const unsigned r = blockIdx.x & 31u;
        t.sub_ticket = atomicAdd(reinterpret_cast<unsigned long long*>(rng_state + 16 + 16 * r), 1ull);
    }
    return t;
}

__device__ __forceinline__ void rng_finish(uint64_t* rng_state, uint64_t new_off, RngTicket t) {
    if (threadIdx.x == 0) {
        const unsigned r = blockIdx.x & 31u;
        const unsigned long long members = (gridDim.x - r + 31u) >> 5;  // workgroups with blockIdx % 32 == r
        if (t.sub_ticket == members - 1ull) {                            // every one of them has read the offset
            unsigned long long* sub = reinterpret_cast<unsigned long long*>(rng_state + 16 + 16 * r);
            *sub = 0;
            const unsigned long long groups = gridDim.x < 32u ? gridDim.x : 32u;
            unsigned long long* top = reinterpret_cast<unsigned long long*>(rng_state + 2);
            if (atomicAdd(top, 1ull) == groups - 1ull) {
                *top = 0;
                rng_state[1] = new_off;
            }
        }
    }
}

// K5: a lane owns GPL = 2 groups of four consecutive coordinates, 1024 coordinates apart (a workgroup covers 2048): both
// loads are requested up front, and the grid is half the waves -- one round of the chip at config 5's shape instead of 2.1
// (round 4).  Group g draws from Philox counter g whatever the lane layout, so the noise stream is the one the fused and the
// trajectory kernels draw.
constexpr int K5_GPL = 2;

__global__ __launch_bounds__(256) void k5_diffuse(float* __restrict__ xyz, const float* __restrict__ beta,
                                                  unsigned n_total, unsigned nps, unsigned n_struct,
                                                  uint64_t* __restrict__ rng_state, const float* __restrict__ noise) {
    __shared__ float2 keep_add[PS_MAX_SPAN];
    uint64_t seed = 0, off = 0;
    if (!noise) {
        seed = rng_state[0];
        off = rng_state[1];
    }
    const unsigned blk_begin = blockIdx.x * (1024u * K5_GPL);
    // The coordinates (and the injected noise) are requested FIRST: they depend on nothing but the lane's index, so their
    // memory latency overlaps the beta loads, the square roots and the barrier of beta_span_fill and the Philox rounds.
    unsigned g[K5_GPL], e0[K5_GPL];
    bool live[K5_GPL], full[K5_GPL];
    float eps[K5_GPL][4], x[K5_GPL][4];
#pragma unroll
    for (int u = 0; u < K5_GPL; ++u) {
        g[u] = blockIdx.x * (256u * K5_GPL) + 256u * u + threadIdx.x;
        e0[u] = g[u] * 4u;
        live[u] = e0[u] < n_total;
        full[u] = e0[u] + 4u <= n_total;
#pragma unroll
        for (int k = 0; k < 4; ++k) eps[u][k] = x[u][k] = 0.f;
        if (live[u]) {
            if (full[u]) {
                const float4 t = *reinterpret_cast<const float4*>(xyz + e0[u]);
                x[u][0] = t.x; x[u][1] = t.y; x[u][2] = t.z; x[u][3] = t.w;
            } else {
                for (int k = 0; k < 4; ++k) x[u][k] = (e0[u] + k < n_total) ? xyz[e0[u] + k] : 0.f;
            }
            if (noise) {
                if (full[u]) {
                    const float4 t = *reinterpret_cast<const float4*>(noise + e0[u]);
                    eps[u][0] = t.x; eps[u][1] = t.y; eps[u][2] = t.z; eps[u][3] = t.w;
                } else {
                    for (int k = 0; k < 4; ++k) eps[u][k] = (e0[u] + k < n_total) ? noise[e0[u] + k] : 0.f;
                }
            }
        }
    }
    const BetaSpan sp = beta_span_fill(beta, blk_begin, min(1024u * K5_GPL, n_total - blk_begin), nps, n_struct, keep_add);
    // the Philox rounds (the vector unit's quarter-rate 32 x 32 multiplies: what the sampler's time is) run while the
    // coordinate loads are in flight; every wave has USED its (seed, offset) before it reaches the barrier ...
    if (!noise) {
#pragma unroll
        for (int u = 0; u < K5_GPL; ++u) normal4(seed, off, g[u], eps[u]);
        // whatever the scheduler does with the arithmetic: the two words are IN registers here, i.e. this wave's read of the
        // offset has returned before it arrives at the barrier
        asm volatile("" ::"s"((unsigned)off), "s"((unsigned)seed));
    }
    __syncthreads();
    // ... so the workgroup's ticket may be taken here, mid-kernel: the returning atomic (1-3 us with every CU issuing one)
    // overlaps the wait for the coordinates and the stores instead of holding the finished workgroup on its CU
    RngTicket ticket{0ull};
    if (!noise) ticket = rng_take_ticket(rng_state);
#pragma unroll
    for (int u = 0; u < K5_GPL; ++u) {
        if (!live[u]) continue;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float2 ka = beta_of(sp, beta, min(e0[u] + k, n_total - 1u), keep_add);
            const float scaled = eps[u][k] * ka.y;      // noise = randn * beta.sqrt()
            x[u][k] = ka.x * x[u][k] + scaled;          // (1 - beta).sqrt() * xyz + noise
        }
        if (full[u]) {
            *reinterpret_cast<float4*>(xyz + e0[u]) = make_float4(x[u][0], x[u][1], x[u][2], x[u][3]);
        } else {
            for (int k = 0; k < 4; ++k)
                if (e0[u] + k < n_total) xyz[e0[u] + k] = x[u][k];
        }
    }
    if (!noise) rng_finish(rng_state, off + 1, ticket);
}

// K5+K4 fused diffusion step: a workgroup owns RB consecutive residues (RB*A*3 contiguous floats).
// Phase 1 sweeps them as float4 groups (coalesced; same Philox counters as k5_diffuse, so the noise
// stream is identical to the unfused call), writes the new coordinates to HBM and keeps them in LDS;
// phase 2 builds one frame per residue from the LDS copy.
template <int RB>
__global__ __launch_bounds__(256) void k54_diffuse_frames(float* __restrict__ xyz, const float* __restrict__ beta,
                                                          unsigned n_res, unsigned N, unsigned A,
                                                          uint64_t* __restrict__ rng_state,
                                                          const float* __restrict__ noise, float* __restrict__ rot,
                                                          float* __restrict__ trans, int a1, int a2, int a3,
                                                          int t_atom) {
    extern __shared__ __attribute__((aligned(16))) float lds54[];
    __shared__ float2 keep_add[PS_MAX_SPAN];
    const unsigned rf = A * 3;                                // floats per residue
    const unsigned r0 = blockIdx.x * RB;                      // first residue of the block
    const unsigned nr = min((unsigned)RB, n_res - r0);
    const unsigned e_begin = r0 * rf, e_end = e_begin + nr * rf, n_total = n_res * rf;
    const unsigned g_begin = e_begin >> 2, g_end = (e_end + 3u) >> 2;  // float4 groups touching the block
    const unsigned nps = N * rf;                              // >= 9 >= 4 here (A >= 3)
    __shared__ uint64_t rng_sh[2];
    RngTicket ticket{0ull};
    if (!noise) ticket = rng_read_and_take_ticket(rng_state, rng_sh);   // early ticket: see the helper
    const BetaSpan sp = beta_span_fill(beta, e_begin, e_end - e_begin, nps, n_res / N, keep_add);
    __syncthreads();
    uint64_t seed = 0, off = 0;
    if (!noise) {
        seed = rng_sh[0];
        off = rng_sh[1];
    }
    for (unsigned g = g_begin + threadIdx.x; g < g_end; g += 256) {
        const unsigned e0 = g * 4u;
        const bool inner = e0 >= e_begin && e0 + 4u <= e_end;  // whole group owned by this block
        float eps[4], x[4];
        if (noise) {
#pragma unroll
            for (int k = 0; k < 4; ++k) eps[k] = (e0 + k < n_total) ? noise[e0 + k] : 0.f;
        } else {
            normal4(seed, off, g, eps);
        }
        if (inner) {
            float4 t = *reinterpret_cast<const float4*>(xyz + e0);
            x[0] = t.x; x[1] = t.y; x[2] = t.z; x[3] = t.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) x[k] = (e0 + k >= e_begin && e0 + k < e_end) ? xyz[e0 + k] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned e = min(max(e0 + k, e_begin), e_end - 1u);  // clamp: out-of-span lanes are discarded below
            const float2 ka = beta_of(sp, beta, e, keep_add);
            const float scaled = eps[k] * ka.y;
            x[k] = ka.x * x[k] + scaled;
        }
        if (inner) {
            *reinterpret_cast<float4*>(xyz + e0) = make_float4(x[0], x[1], x[2], x[3]);
#pragma unroll
            for (int k = 0; k < 4; ++k) lds54[e0 + k - e_begin] = x[k];
        } else {  // a group at the block edge is shared with the neighbour block: each writes its own part
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned e = e0 + k;
                if (e >= e_begin && e < e_end) {
                    xyz[e] = x[k];
                    lds54[e - e_begin] = x[k];
                }
            }
        }
    }
    __syncthreads();
    for (unsigned r = threadIdx.x; r < nr; r += 256) {
        const float* p = lds54 + r * rf;
        const size_t gr = (size_t)r0 + r;
        if (rot) {
            f3 e1, e2, e3;
            gram_schmidt3(load3(p + a1 * 3), load3(p + a2 * 3), load3(p + a3 * 3), e1, e2, e3);
            float* o = rot + gr * 9;
            o[0] = e1.x; o[1] = e2.x; o[2] = e3.x;
            o[3] = e1.y; o[4] = e2.y; o[5] = e3.y;
            o[6] = e1.z; o[7] = e2.z; o[8] = e3.z;
        }
        if (trans) {
            f3 t = load3(p + t_atom * 3);
            float* o = trans + gr * 3;
            o[0] = t.x; o[1] = t.y; o[2] = t.z;
        }
    }
    if (!noise) rng_finish(rng_state, off + 1, ticket);
}

// K55 -- the whole diffusion loop of BASELINE config 5 in ONE launch.  Kernel boundaries write back and
// invalidate the per-XCD L2s, so a step-per-launch loop re-streams the coordinates (35 MB per step at
// B=256, N=384) through Infinity Cache / HBM every step.  Residues never interact in this loop, so a
// workgroup can simply keep its RB residues in LDS for all T steps: per step it draws the noise
// (same Philox counters as T calls of ps_diffuse_f32: offset + t), updates the LDS copy, builds the
// frames and writes only those (and, if asked, the coordinates of that step).  The final coordinates go
// back to xyz once.  Bit-identical to T calls of ps_diffuse_frames_f32.
template <int RB>
__global__ __launch_bounds__(256) void k55_diffusion_trajectory(
    float* __restrict__ xyz, const float* __restrict__ betas /* [T][B] */, unsigned T, unsigned n_res, unsigned N,
    unsigned A, uint64_t* __restrict__ rng_state, float* __restrict__ rot /* [T][n_res][9] */,
    float* __restrict__ trans /* [T][n_res][3] */, float* __restrict__ xyz_traj /* [T][n_res*A*3] */, int a1, int a2,
    int a3, int t_atom) {
    extern __shared__ __attribute__((aligned(16))) float lds55[];
    __shared__ float2 keep_add[2][PS_MAX_SPAN];
    const unsigned rf = A * 3;
    const unsigned r0 = blockIdx.x * RB;
    const unsigned nr = min((unsigned)RB, n_res - r0);
    const unsigned e_begin = r0 * rf, len = nr * rf, e_end = e_begin + len, n_total = n_res * rf;
    const unsigned g_begin = e_begin >> 2, g_end = (e_end + 3u) >> 2;
    const unsigned nps = N * rf, n_struct = n_res / N;
    const uint64_t seed = rng_state[0], off = rng_state[1];
    const bool vec4 = ((e_begin | len | nps | n_total) & 3u) == 0 &&
                      (xyz_traj == nullptr || (reinterpret_cast<uintptr_t>(xyz_traj) & 15) == 0);

    for (unsigned k = threadIdx.x; k < len; k += 256) lds55[k] = xyz[e_begin + k];
    BetaSpan sp = beta_span_fill(betas, e_begin, len, nps, n_struct, keep_add[0]);
    __syncthreads();

    for (unsigned t = 0; t < T; ++t) {
        const float2* ka_t = keep_add[t & 1];
        const float* beta_t = betas + (size_t)t * n_struct;
        // ---- phase 1: x <- sqrt(1-beta) x + sqrt(beta) eps on the LDS copy ----
        if (vec4) {
            // aligned span: every float4 group lies inside the span and inside one structure
            for (unsigned g = g_begin + threadIdx.x; g < g_end; g += 256) {
                const unsigned e0 = g * 4u;
                float eps[4];
                normal4(seed, off + t, g, eps);
                const float2 ka = beta_of(sp, beta_t, e0, ka_t);
                float4* px = reinterpret_cast<float4*>(lds55 + (e0 - e_begin));
                float4 x = *px;
                const float s0 = eps[0] * ka.y, s1 = eps[1] * ka.y, s2 = eps[2] * ka.y, s3 = eps[3] * ka.y;
                x.x = ka.x * x.x + s0;
                x.y = ka.x * x.y + s1;
                x.z = ka.x * x.z + s2;
                x.w = ka.x * x.w + s3;
                *px = x;
                if (xyz_traj) *reinterpret_cast<float4*>(xyz_traj + (size_t)t * n_total + e0) = x;
            }
        } else {
            for (unsigned g = g_begin + threadIdx.x; g < g_end; g += 256) {
                const unsigned e0 = g * 4u;
                float eps[4];
                normal4(seed, off + t, g, eps);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned e = e0 + k;
                    if (e >= e_begin && e < e_end) {
                        const float2 ka = beta_of(sp, beta_t, e, ka_t);
                        const float scaled = eps[k] * ka.y;
                        const float v = ka.x * lds55[e - e_begin] + scaled;
                        lds55[e - e_begin] = v;
                        if (xyz_traj) xyz_traj[(size_t)t * n_total + e] = v;
                    }
                }
            }
        }
        // next step's sqrt(1-beta), sqrt(beta) into the other buffer (nobody reads it during this step)
        if (t + 1 < T) beta_span_fill(betas + (size_t)(t + 1) * n_struct, e_begin, len, nps, n_struct, keep_add[(t + 1) & 1]);
        __syncthreads();
        // ---- phase 2: frames of the new coordinates ----
        for (unsigned r = threadIdx.x; r < nr; r += 256) {
            const float* p = lds55 + r * rf;
            const size_t gr = (size_t)t * n_res + r0 + r;
            if (rot) {
                f3 e1, e2, e3;
                gram_schmidt3(load3(p + a1 * 3), load3(p + a2 * 3), load3(p + a3 * 3), e1, e2, e3);
                float* o = rot + gr * 9;
                o[0] = e1.x; o[1] = e2.x; o[2] = e3.x;
                o[3] = e1.y; o[4] = e2.y; o[5] = e3.y;
                o[6] = e1.z; o[7] = e2.z; o[8] = e3.z;
            }
            if (trans) {
                f3 tt = load3(p + t_atom * 3);
                float* o = trans + gr * 3;
                o[0] = tt.x; o[1] = tt.y; o[2] = tt.z;
            }
        }
        __syncthreads();  // phase 1 of the next step overwrites what phase 2 just read
    }
    for (unsigned k = threadIdx.x; k < len; k += 256) xyz[e_begin + k] = lds55[k];
    // advance the draw counter by T (same two-level ticket as the single-step kernels)
    rng_advance_by_last_block(rng_state, off + T - 1);
}

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

// K6, LDS-resident form: the structure (n_atoms * 3 floats; 69 KB at the config-5 shape) is read from HBM ONCE into
// LDS, both statistics sweeps and the normalisation run on the LDS copy, and the result is written once -- 1 read +
// 1 write of the coordinates instead of 3 + 1.  Per thread the accumulation visits the same atoms in the same order
// as k6_standardize and the block reductions are the same, so mu, std and the output are bit-identical to it
// (tests/test_gpu_parity.py::test_k6_lds_resident_equals_streaming).  Used when the structure fits (<= 150 KB).
__global__ __launch_bounds__(1024) void k6_standardize_lds(float* __restrict__ xyz, const uint8_t* __restrict__ amask,
                                                           float* __restrict__ mu_out, float* __restrict__ std_out,
                                                           int n_atoms) {
    extern __shared__ __attribute__((aligned(16))) float sx[];   // n_atoms * 3 coordinates
    __shared__ double red[68];
    const int b = blockIdx.x;
    const int nf = n_atoms * 3;
    float* x = xyz + (size_t)b * nf;
    const uint8_t* m = amask ? amask + (size_t)b * n_atoms : nullptr;

    // coalesced copy in: float4 where the structure starts 16-byte aligned, dwords otherwise / for the tail
    const bool v4 = ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
    const int n4 = v4 ? nf >> 2 : 0;
    for (int q = threadIdx.x; q < n4; q += blockDim.x)
        reinterpret_cast<float4*>(sx)[q] = reinterpret_cast<const float4*>(x)[q];
    for (int f = 4 * n4 + threadIdx.x; f < nf; f += blockDim.x) sx[f] = x[f];
    __syncthreads();

    double acc[4] = {0, 0, 0, 0};  // sum x, sum y, sum z, count
    for (int a = threadIdx.x; a < n_atoms; a += blockDim.x) {
        const float w = m ? (m[a] ? 1.f : 0.f) : 1.f;
        acc[0] += (double)nan_to_num0(sx[a * 3 + 0] * w);
        acc[1] += (double)nan_to_num0(sx[a * 3 + 1] * w);
        acc[2] += (double)nan_to_num0(sx[a * 3 + 2] * w);
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
            const float d = nan_to_num0(sx[a * 3 + k]) - mu[k];
            sq[k] += (double)((d * d) * w);
        }
    }
    block_sum4(sq, red);
    const float sd[3] = {sqrtf((float)sq[0] / cnt), sqrtf((float)sq[1] / cnt), sqrtf((float)sq[2] / cnt)};

    // normalise in LDS (atom-major, so every thread knows its axis without a division), then stream out coalesced
    for (int a = threadIdx.x; a < n_atoms; a += blockDim.x) {
#pragma unroll
        for (int k = 0; k < 3; ++k) sx[a * 3 + k] = (sx[a * 3 + k] - mu[k]) / sd[k];
    }
    __syncthreads();
    for (int q = threadIdx.x; q < n4; q += blockDim.x)
        reinterpret_cast<float4*>(x)[q] = reinterpret_cast<const float4*>(sx)[q];
    for (int f = 4 * n4 + threadIdx.x; f < nf; f += blockDim.x) x[f] = sx[f];
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
    if (n_total >= 0xFFFFFFF0ull) return (int)hipErrorInvalidValue;  // 32-bit index math (17 GB of coordinates)
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const size_t groups = (n_total + 3) / 4, per_wg = 256 * (size_t)K5_GPL;
    return ps_launch(k5_diffuse, dim3((unsigned)((groups + per_wg - 1) / per_wg)), dim3(256), 0, s, xyz, beta,
                       (unsigned)n_total, (unsigned)n_per_struct, (unsigned)B, rng_state, noise);
}

// Dispatch: LDS-resident kernel when one structure fits in 150 KB of LDS, the three-sweep kernel otherwise.
// `force_streaming` (ps_standardize_variant_f32 only) exists so that the two can be compared bit for bit.
static int standardize_dispatch(float* xyz, const uint8_t* atom_mask, float* mu, float* std, int B, int N, int A,
                                bool force_streaming, void* stream) {
    if (!xyz || !mu || !std || B < 0 || N < 0 || A <= 0) return (int)hipErrorInvalidValue;
    if (B == 0 || N == 0) return 0;
    const size_t lds = (size_t)N * A * 3 * sizeof(float);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (!force_streaming && lds <= 150 * 1024)
        return ps_launch(k6_standardize_lds, dim3(B), dim3(1024), lds, s, xyz, atom_mask, mu, std, N * A);
    return ps_launch(k6_standardize, dim3(B), dim3(1024), 0, s, xyz, atom_mask, mu, std, N * A);
}

extern "C" int ps_standardize_f32(float* xyz, const uint8_t* atom_mask, float* mu, float* std, int B, int N, int A,
                                  void* stream) {
    return standardize_dispatch(xyz, atom_mask, mu, std, B, N, A, false, stream);
}

extern "C" int ps_standardize_variant_f32(float* xyz, const uint8_t* atom_mask, float* mu, float* std, int B, int N,
                                          int A, int variant, void* stream) {
    if (variant != 0 && variant != 1) return (int)hipErrorInvalidValue;
    return standardize_dispatch(xyz, atom_mask, mu, std, B, N, A, variant == 1, stream);
}

extern "C" int ps_affine_f32(float* xyz, const float* scale, const float* shift, int B, int n_atoms_per_struct,
                             void* stream) {
    if (!xyz || !scale || !shift || B < 0 || n_atoms_per_struct < 0) return (int)hipErrorInvalidValue;
    const size_t total = (size_t)B * n_atoms_per_struct;
    if (total == 0) return 0;
    return ps_launch(k6_affine, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), xyz, scale, shift, (unsigned)n_atoms_per_struct, total);
}

extern "C" int ps_diffuse_frames_f32(float* xyz, const float* beta, int B, int N, int A, uint64_t* rng_state,
                                     const float* noise, float* rot, float* trans, int a1, int a2, int a3, int t_atom,
                                     void* stream) {
    if (!xyz || !beta || B < 0 || N < 0 || A < 3 || (!rng_state && !noise) || (!rot && !trans))
        return (int)hipErrorInvalidValue;
    if ((size_t)B * N * A * 3 >= 0xFFFFFFF0ull) return (int)hipErrorInvalidValue;  // 32-bit index math
    if (reinterpret_cast<uintptr_t>(xyz) & 15) return (int)hipErrorInvalidValue;
    if (rot && (a1 < 0 || a1 >= A || a2 < 0 || a2 >= A || a3 < 0 || a3 >= A)) return (int)hipErrorInvalidValue;
    if (trans && (t_atom < 0 || t_atom >= A)) return (int)hipErrorInvalidValue;
    const size_t n_res = (size_t)B * N;
    if (n_res == 0) return 0;
    constexpr int RB = 128;  // 128 residues * 45 floats = 23 KB of LDS at A = 15
    const size_t lds = (size_t)RB * A * 3 * sizeof(float);
    if (lds > 160 * 1024) return (int)hipErrorInvalidValue;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    return ps_launch(k54_diffuse_frames<RB>, dim3((unsigned)((n_res + RB - 1) / RB)), dim3(256), lds, s, xyz, beta,
                       (unsigned)n_res, (unsigned)N, (unsigned)A, rng_state, noise, rot, trans, a1, a2, a3, t_atom);
}

extern "C" int ps_diffusion_trajectory_f32(float* xyz, const float* betas, int T, int B, int N, int A,
                                           uint64_t* rng_state, float* rot, float* trans, float* xyz_traj, int a1,
                                           int a2, int a3, int t_atom, void* stream) {
    if (!xyz || !betas || !rng_state || T < 0 || B < 0 || N < 0 || A < 3) return (int)hipErrorInvalidValue;
    if (rot && (a1 < 0 || a1 >= A || a2 < 0 || a2 >= A || a3 < 0 || a3 >= A)) return (int)hipErrorInvalidValue;
    if (trans && (t_atom < 0 || t_atom >= A)) return (int)hipErrorInvalidValue;
    if ((size_t)B * N * A * 3 >= 0xFFFFFFF0ull) return (int)hipErrorInvalidValue;  // 32-bit index math
    const size_t n_res = (size_t)B * N;
    if (n_res == 0 || T == 0) return 0;
    constexpr int RB = 128;
    const size_t lds = (size_t)RB * A * 3 * sizeof(float);
    if (lds > 150 * 1024) return (int)hipErrorInvalidValue;
    return ps_launch(k55_diffusion_trajectory<RB>, dim3((unsigned)((n_res + RB - 1) / RB)), dim3(256), lds,
                       reinterpret_cast<hipStream_t>(stream), xyz, betas, (unsigned)T, (unsigned)n_res, (unsigned)N,
                       (unsigned)A, rng_state, rot, trans, xyz_traj, a1, a2, a3, t_atom);
}
