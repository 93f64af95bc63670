// K3 design-space probe (round 4): candidate sweeps for pairwise_dihedrals / pairwise_planar_angles against the product
// kernel, in one process, bit-compared with it.  The product source is included verbatim, so "base" IS the product.
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -I include -o tools/microbench/k3_variants \
//         tools/microbench/k3_variants.hip && tools/microbench/k3_variants [B N reps]
//
// Knobs of the candidate kernel k3_strip<NP, SRC, NC, POL>:
//   NC   consecutive column residues per lane (2: 8-byte stores, 4: 16-byte stores)
//   POL  cache policy of the output stores: 0 plain, 2 nt, 16 sc1 (write-through), 17 sc0|sc1, 18 sc1|nt
//   IR   rows per wave (run-time); a wave owns 64 * NC columns x IR rows, the row scalars of the NEXT row pair are
//        requested before the current pair is computed
#include "../../protstruc_amd/csrc/pairwise_angles.hip"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ unsigned long long* g_stamps = nullptr;   // diagnostic: {start, end} of every wave-task in s_memrealtime ticks (10 ns)
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2v __attribute__((ext_vector_type(2)));

// The same sweep with the wave's row-side points staged once in LDS (wave-private region, pair-interleaved: one
// broadcast ds_read_b64 delivers {row 2r, row 2r + 1} of one component, already in the packed layout) instead of
// scalar loads per trip: no s_waitcnt on SMEM inside the loop, no SGPR -> VGPR moves, no constant-bus limits.
template <int NP, int SRC, int NC, int POL>
__global__ __launch_bounds__(1024) void k3_strip_lds(const float* __restrict__ xyz, float* __restrict__ out, int N, int A,
                                                    AtomSel sel, int row_begin, int row_end, int out_rows,
                                                    int out_row_origin, int IR, int n_strips, int n_chunks,
                                                    unsigned n_tasks) {
    static_assert(NC == 2 || NC == 4, "columns per lane");
    constexpr int NPI = NP - __builtin_popcount(SRC & ((1 << NP) - 1));   // points taken from the row residue
    constexpr int MAXIR = 64;
    __shared__ f32x2 rowbuf[16][(MAXIR / 2) * (NPI > 0 ? NPI : 1) * 3];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned wpw = blockDim.x >> 6;                                           // waves per workgroup
    const unsigned task = min(blockIdx.x * wpw + (unsigned)wave, n_tasks - 1u);   // surplus waves repeat the last task
    unsigned long long* stamps = g_stamps;
    const unsigned long long t_start = stamps ? __builtin_amdgcn_s_memrealtime() : 0ull;
    const unsigned strip = task % (unsigned)n_strips, rest = task / (unsigned)n_strips;
    const int b = (int)(rest / (unsigned)n_chunks);
    const int i0 = row_begin + (int)(rest % (unsigned)n_chunks) * IR;
    const int i1 = min(i0 + IR, row_end);
    const int j0 = ((int)strip * 64 + lane) * NC;
    const bool live = j0 < N;
    const int jc = live ? j0 : N - NC;
    const float* xb = xyz + (size_t)b * N * (size_t)A * 3;   // uniform
    if constexpr (NPI > 0) {
        int amap[NPI];
        {
            int q = 0;
#pragma unroll
            for (int k = 0; k < NP; ++k)
                if (!((SRC >> k) & 1)) amap[q++] = sel.atom[k];
        }
        float* rb = reinterpret_cast<float*>(rowbuf[wave]);
        for (int e = lane; e < IR * NPI * 3; e += 64) {
            const int row = e / (NPI * 3), rem = e - row * (NPI * 3), q = rem / 3, c = rem - q * 3;
            int at = amap[0];
#pragma unroll
            for (int t = 1; t < NPI; ++t) at = (q == t) ? amap[t] : at;
            const int ii = min(i0 + row, N - 1);
            rb[((((row >> 1) * NPI + q) * 3 + c) << 1) + (row & 1)] = xb[(size_t)ii * (size_t)A * 3 + at * 3 + c];
        }
    }
    f3 pj[NC][NP];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const float* sj = xb + (size_t)(jc + c) * (size_t)A * 3;
#pragma unroll
        for (int k = 0; k < NP; ++k) pj[c][k] = ((SRC >> k) & 1) ? load3(sj + sel.atom[k] * 3) : mk3(0.f, 0.f, 0.f);
    }
    __syncthreads();
    float* obase = out + ((size_t)b * out_rows + (size_t)(i0 - out_row_origin)) * N;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(obase, 0, 0xFFFFFFFFu, 0x00020000u);
    const int lane_off = j0 * 4;
    const int row_bytes = N * 4;
    const f32x2* rp = rowbuf[wave];
    auto rows = [&](int r, f3v (&p)[NP]) {
        int q = 0;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            if ((SRC >> k) & 1) {
                p[k] = mk3v(mk3(0.f, 0.f, 0.f), mk3(0.f, 0.f, 0.f));
            } else {
                p[k] = f3v{rp[(r * NPI + q) * 3 + 0], rp[(r * NPI + q) * 3 + 1], rp[(r * NPI + q) * 3 + 2]};
                ++q;
            }
        }
    };
    int i = i0, r = 0;
    for (; i + 1 < i1; i += 2, ++r) {
        f3v cur[NP];
        rows(r, cur);
        f32x2 v[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            f3v p[NP];
#pragma unroll
            for (int k = 0; k < NP; ++k) p[k] = ((SRC >> k) & 1) ? mk3v(pj[c][k], pj[c][k]) : cur[k];
            if constexpr (NP == 4)
                v[c] = dihedral4v_k3(p[0], p[1], p[2], p[3]);
            else
                v[c] = angle3v(p[0], p[1], p[2]);
        }
        if (live) {
            const int so = (i - i0) * row_bytes;
            if constexpr (NC == 4) {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, f32x4{v[0].x, v[1].x, v[2].x, v[3].x}), rsrc, lane_off, so, POL);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, f32x4{v[0].y, v[1].y, v[2].y, v[3].y}), rsrc, lane_off, so + row_bytes, POL);
            } else {
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2v, f32x2{v[0].x, v[1].x}), rsrc, lane_off, so, POL);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2v, f32x2{v[0].y, v[1].y}), rsrc, lane_off, so + row_bytes, POL);
            }
        }
    }
    if (i < i1) {   // odd last row
        f3v cur[NP];
        rows(r, cur);
        float v[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            f3 p[NP];
#pragma unroll
            for (int k = 0; k < NP; ++k) p[k] = ((SRC >> k) & 1) ? pj[c][k] : mk3(cur[k].x.x, cur[k].y.x, cur[k].z.x);
            if constexpr (NP == 4)
                v[c] = dihedral4_k3(p[0], p[1], p[2], p[3]);
            else
                v[c] = angle3(p[0], p[1], p[2]);
        }
        if (live) {
            const int so = (i - i0) * row_bytes;
            if constexpr (NC == 4)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, f32x4{v[0], v[1], v[2], v[3]}), rsrc, lane_off, so, POL);
            else
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2v, f32x2{v[0], v[1]}), rsrc, lane_off, so, POL);
        }
    }
    if (stamps) {
        __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0): the stores have left the wave... (acknowledged)
        const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) {
            unsigned long long* o = stamps + 4 * (size_t)(blockIdx.x * wpw + (unsigned)wave);
            o[0] = t_start; o[1] = t_end;
            o[2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_ID: wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13]
            o[3] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // XCC_ID
        }
    }
}

template <int NP, int SRC, int NC, int POL>
int launch_strip_lds(const float* xyz, float* out, int B, int N, int A, const AtomSel& sel, int IR, hipStream_t s,
                     int threads = 256, int per_cu = 0) {
    if (IR > 64) return (int)hipErrorInvalidValue;
    const int n_strips = (N + 64 * NC - 1) / (64 * NC), n_chunks = (N + IR - 1) / IR;
    const unsigned long long n_tasks = (unsigned long long)n_strips * n_chunks * B;
    const unsigned wpw = threads / 64;
    size_t dyn = 0;
    if (per_cu > 0) {   // idle LDS so that exactly per_cu workgroups fit a CU (160 KB)
        static hipFuncAttributes fa;
        static bool have = false;
        if (!have) { CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k3_strip_lds<NP, SRC, NC, POL>))); have = true; }
        const size_t want = 163840 / per_cu;
        dyn = want > fa.sharedSizeBytes ? want - fa.sharedSizeBytes : 0;
        if (dyn > 65536) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k3_strip_lds<NP, SRC, NC, POL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    }
    return ps_launch(k3_strip_lds<NP, SRC, NC, POL>, dim3((unsigned)((n_tasks + wpw - 1) / wpw)), dim3(threads), dyn, s, xyz, out, N,
                     A, sel, 0, N, N, 0, IR, n_strips, n_chunks, (unsigned)n_tasks);
}

// One workgroup per CU (1024 threads = 4 waves per SIMD; the dynamic LDS request keeps a second workgroup off the CU),
// each owning a contiguous range of the flat task list  t = (b * n_strips + strip) * n_chunks + chunk  (a task = CH rows x
// 64 * NC columns).  The rows a (b, strip) segment needs are staged once in LDS, pair-interleaved; the 16 waves then PULL
// tasks from an LDS counter: the SIMD arbiter favours its oldest wave, so with equal static shares the waves of a SIMD finish
// one after the other and the last one runs alone at half the issue rate -- pulled tasks let the fast waves take more.
template <int NP, int SRC, int NC, int POL>
__global__ __launch_bounds__(1024) void k3_cu(const float* __restrict__ xyz, float* __restrict__ out, int N, int A,
                                              AtomSel sel, int row_begin, int row_end, int out_rows, int out_row_origin,
                                              int CH, int n_strips, int n_chunks, unsigned n_tasks, unsigned tasks_per_wg) {
    static_assert(NC == 2 || NC == 4, "columns per lane");
    constexpr int NPI = NP - __builtin_popcount(SRC & ((1 << NP) - 1));   // points taken from the row residue
    constexpr int NPIq = NPI > 0 ? NPI : 1;
    extern __shared__ f32x2 rowbuf[];            // [row pair][row point][xyz] of the current segment
    __shared__ unsigned next_task;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int n_waves = (int)(blockDim.x >> 6);
    unsigned long long* stamps = g_stamps;
    const unsigned long long t_start = stamps ? __builtin_amdgcn_s_memrealtime() : 0ull;
    const unsigned t0 = blockIdx.x * tasks_per_wg, t1 = min(t0 + tasks_per_wg, n_tasks);
    if (t0 >= t1) return;
    int amap[NPIq];
    {
        int q = 0;
#pragma unroll
        for (int k = 0; k < NP; ++k)
            if (!((SRC >> k) & 1)) amap[q++] = sel.atom[k];
    }
    const int row_bytes = N * 4;
    int staged_b = -1, staged_lo = -1, staged_hi = -1;
    for (unsigned g = t0 / (unsigned)n_chunks; g <= (t1 - 1u) / (unsigned)n_chunks; ++g) {   // g = b * n_strips + strip
        const int c_lo = (int)(max(t0, g * (unsigned)n_chunks) - g * (unsigned)n_chunks);
        const int c_hi = (int)(min(t1, (g + 1u) * (unsigned)n_chunks) - g * (unsigned)n_chunks);
        const int b = (int)(g / (unsigned)n_strips), strip = (int)(g % (unsigned)n_strips);
        const int r_lo = row_begin + c_lo * CH, r_hi = min(row_begin + c_hi * CH, row_end);
        const float* xb = xyz + (size_t)b * N * (size_t)A * 3;   // uniform
        __syncthreads();                                          // the previous segment's readers are done
        if (NPI > 0 && (b != staged_b || r_lo != staged_lo || r_hi != staged_hi)) {
            float* rb = reinterpret_cast<float*>(rowbuf);
            const int n_el = (r_hi - r_lo) * NPI * 3;
            for (int e = (int)threadIdx.x; e < n_el; e += (int)blockDim.x) {
                const int row = e / (NPI * 3), rem = e - row * (NPI * 3), q = rem / 3, c = rem - q * 3;
                int at = amap[0];
#pragma unroll
                for (int t = 1; t < NPI; ++t) at = (q == t) ? amap[t] : at;
                rb[((((row >> 1) * NPI + q) * 3 + c) << 1) + (row & 1)] = xb[(size_t)(r_lo + row) * (size_t)A * 3 + at * 3 + c];
            }
            staged_b = b; staged_lo = r_lo; staged_hi = r_hi;
        }
        if (threadIdx.x == 0) next_task = (unsigned)(c_lo + n_waves);
        const int j0 = (strip * 64 + lane) * NC;
        const bool live = j0 < N;
        const int jc = live ? j0 : N - NC;
        f3 pj[NC][NP];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float* sj = xb + (size_t)(jc + c) * (size_t)A * 3;
#pragma unroll
            for (int k = 0; k < NP; ++k) pj[c][k] = ((SRC >> k) & 1) ? load3(sj + sel.atom[k] * 3) : mk3(0.f, 0.f, 0.f);
        }
        __syncthreads();
        float* obase = out + ((size_t)b * out_rows + (size_t)(r_lo - out_row_origin)) * N;   // row r_lo of the segment
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(obase, 0, 0xFFFFFFFFu, 0x00020000u);
        const int lane_off = j0 * 4;
        auto rows = [&](int r, f3v (&p)[NP]) {                    // r = row pair index inside the segment
            int q = 0;
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                if ((SRC >> k) & 1) {
                    p[k] = mk3v(mk3(0.f, 0.f, 0.f), mk3(0.f, 0.f, 0.f));
                } else {
                    p[k] = f3v{rowbuf[(r * NPI + q) * 3 + 0], rowbuf[(r * NPI + q) * 3 + 1], rowbuf[(r * NPI + q) * 3 + 2]};
                    ++q;
                }
            }
        };
        int c = c_lo + wave;
        while (c < c_hi) {
            const int i0 = (c - c_lo) * CH;                           // rows relative to r_lo (CH even: pairs stay aligned)
            const int i1 = min(i0 + CH, r_hi - r_lo);
            int i = i0;
            for (; i + 1 < i1; i += 2) {
                f3v cur[NP];
                rows(i >> 1, cur);
                f32x2 v[NC];
#pragma unroll
                for (int cc = 0; cc < NC; ++cc) {
                    f3v p[NP];
#pragma unroll
                    for (int k = 0; k < NP; ++k) p[k] = ((SRC >> k) & 1) ? mk3v(pj[cc][k], pj[cc][k]) : cur[k];
                    if constexpr (NP == 4)
                        v[cc] = dihedral4v_k3(p[0], p[1], p[2], p[3]);
                    else
                        v[cc] = angle3v(p[0], p[1], p[2]);
                }
                if (live) {
                    const int so = i * row_bytes;
                    if constexpr (NC == 4) {
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, f32x4{v[0].x, v[1].x, v[2].x, v[3].x}), rsrc, lane_off, so, POL);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, f32x4{v[0].y, v[1].y, v[2].y, v[3].y}), rsrc, lane_off, so + row_bytes, POL);
                    } else {
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2v, f32x2{v[0].x, v[1].x}), rsrc, lane_off, so, POL);
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2v, f32x2{v[0].y, v[1].y}), rsrc, lane_off, so + row_bytes, POL);
                    }
                }
            }
            if (i < i1) {   // odd last row of the row range
                f3v cur[NP];
                rows(i >> 1, cur);
                float v[NC];
#pragma unroll
                for (int cc = 0; cc < NC; ++cc) {
                    f3 p[NP];
#pragma unroll
                    for (int k = 0; k < NP; ++k) p[k] = ((SRC >> k) & 1) ? pj[cc][k] : mk3(cur[k].x.x, cur[k].y.x, cur[k].z.x);
                    if constexpr (NP == 4)
                        v[cc] = dihedral4_k3(p[0], p[1], p[2], p[3]);
                    else
                        v[cc] = angle3(p[0], p[1], p[2]);
                }
                if (live) {
                    const int so = i * row_bytes;
                    if constexpr (NC == 4)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, f32x4{v[0], v[1], v[2], v[3]}), rsrc, lane_off, so, POL);
                    else
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2v, f32x2{v[0], v[1]}), rsrc, lane_off, so, POL);
                }
            }
            unsigned nx = 0;
            if (lane == 0) nx = atomicAdd(&next_task, 1u);
            c = __builtin_amdgcn_readfirstlane((int)nx);
        }
    }
    if (stamps) {
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) {
            unsigned long long* o = stamps + 4 * (size_t)(blockIdx.x * (unsigned)n_waves + (unsigned)wave);
            o[0] = t_start; o[1] = t_end;
            o[2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
            o[3] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        }
    }
}

int g_cus = 0;
template <int NP, int SRC, int NC, int POL>
int launch_cu(const float* xyz, float* out, int B, int N, int A, const AtomSel& sel, int CH, hipStream_t s, int threads = 1024,
              int wg_per_cu = 1, int rounds = 1) {
    constexpr int NPI = NP - __builtin_popcount(SRC & ((1 << NP) - 1));
    if (!g_cus) { hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0)); g_cus = p.multiProcessorCount; }
    const int n_strips = (N + 64 * NC - 1) / (64 * NC), n_chunks = (N + CH - 1) / CH;
    const unsigned long long n_tasks = (unsigned long long)n_strips * n_chunks * B;
    const unsigned G = (unsigned)(g_cus * wg_per_cu * rounds);
    const unsigned tasks_per_wg = (unsigned)((n_tasks + G - 1) / G);
    const unsigned grid = (unsigned)((n_tasks + tasks_per_wg - 1) / tasks_per_wg);
    size_t need = (size_t)((N + 2) / 2) * (NPI > 0 ? NPI : 1) * 3 * 8;            // one structure's rows, pair-interleaved
    size_t dyn = std::max(need, (size_t)(163840 / wg_per_cu) - 64);                // idle LDS: exactly wg_per_cu workgroups per CU
    if (dyn > 160 * 1024 - 64) return (int)hipErrorInvalidValue;
    static size_t granted = 0;
    if (dyn > granted) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k3_cu<NP, SRC, NC, POL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn)); granted = dyn; }
    return ps_launch(k3_cu<NP, SRC, NC, POL>, dim3(grid), dim3(threads), dyn, s, xyz, out, N, A, sel, 0, N, N, 0, CH, n_strips,
                     n_chunks, (unsigned)n_tasks, tasks_per_wg);
}

hipEvent_t ea, eb;
unsigned g_wpw = 4;
struct T { float mean_us, min_us, lo_us; };
T time_it(const std::function<void()>& f, int reps) {
    for (int i = 0; i < 3; ++i) f();
    CK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(ea)); f(); CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb));
        float ms; CK(hipEventElapsedTime(&ms, ea, eb)); t.push_back(ms * 1e3f);
    }
    // back-to-back: what a stream of launches costs per launch
    CK(hipEventRecord(ea)); for (int r = 0; r < reps; ++r) f(); CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb));
    float ms; CK(hipEventElapsedTime(&ms, ea, eb));
    float mn = *std::min_element(t.begin(), t.end());
    (void)mn;
    std::sort(t.begin(), t.end());
    return T{ms * 1e3f / reps, t[t.size() / 2], t[0]};
}

}  // namespace

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 128, N = argc > 2 ? atoi(argv[2]) : 512, reps = argc > 3 ? atoi(argv[3]) : 30;
    const int A = 15;
    CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> h((size_t)B * N * A * 3);
    for (auto& x : h) x = nd(rng);
    float *xyz, *ref, *out;
    const size_t ob = (size_t)B * N * N * 4;
    CK(hipMalloc(&xyz, h.size() * 4)); CK(hipMalloc(&ref, ob)); CK(hipMalloc(&out, ob));
    CK(hipMemcpy(xyz, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    std::vector<uint32_t> hr(ob / 4), ho(ob / 4);
    printf("K3 variants  B=%d N=%d A=%d  output %.1f MB  reps=%d   (us: back-to-back mean / single-launch median / single-launch min)\n", B, N, A, ob / 1e6, reps);

    struct Feature { const char* name; int np; int src[4]; int atom[4]; };
    const Feature feats_all[] = {{"dihedral(2,2) CA,CB|CA,CB", 4, {0, 0, 1, 1}, {1, 4, 1, 4}},
                             {"dihedral(3,1) N,CA,CB|CB", 4, {0, 0, 0, 1}, {0, 1, 4, 4}},
                             {"planar(2,1) CA,CB|CB", 3, {0, 0, 1, 0}, {1, 4, 4, 0}}};
    // fill: the floor of writing the output at all
    {
        T t = time_it([&] { CK(hipMemsetAsync(out, 0, ob, 0)); }, reps);
        printf("%-34s %8.1f / %8.1f\n", "hipMemsetAsync of the output", t.mean_us, t.min_us);
    }
    const int nf = argc > 4 ? atoi(argv[4]) : 3;
    for (int fi = 0; fi < nf; ++fi) {
        const Feature& f = feats_all[fi];
        AtomSel sel{};
        for (int k = 0; k < f.np; ++k) sel.atom[k] = f.atom[k];
        printf("---- %s\n", f.name);
        CK(hipMemset(ref, 0xFF, ob));
        T t0 = time_it([&] { ps_pairwise_angles_f32(xyz, ref, B, N, A, f.np, f.src, f.atom, 0, N, N, 0, 0, nullptr); }, reps);
        printf("%-34s %8.1f / %8.1f / %8.1f\n", "base (product)", t0.mean_us, t0.min_us, t0.lo_us);
        CK(hipMemcpy(hr.data(), ref, ob, hipMemcpyDeviceToHost));
        auto run = [&](const char* name, const std::function<int()>& fn) {
            CK(hipMemset(out, 0xFF, ob));
            int rc = fn();
            if (rc) { printf("%-34s launch error %d\n", name, rc); return; }
            T t = time_it([&] { fn(); }, reps);
            CK(hipMemcpy(ho.data(), out, ob, hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (size_t k = 0; k < ho.size(); ++k) bad += ho[k] != hr[k];
            printf("%-34s %8.1f / %8.1f / %8.1f  %s (%zu differ)\n", name, t.mean_us, t.min_us, t.lo_us, bad ? "MISMATCH" : "same bits", bad);
        };
        auto stamped = [&](const char* name, unsigned n_waves, const std::function<int()>& fn) {
            unsigned long long* d;
            CK(hipMalloc(&d, (size_t)n_waves * 32));
            CK(hipMemset(d, 0, (size_t)n_waves * 32));
            CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &d, sizeof(d)));
            for (int w = 0; w < 3; ++w) fn();
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(ea)); fn(); CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb));
            float ms; CK(hipEventElapsedTime(&ms, ea, eb));
            std::vector<unsigned long long> hs((size_t)n_waves * 4);
            CK(hipMemcpy(hs.data(), d, hs.size() * 8, hipMemcpyDeviceToHost));
            unsigned long long* nul = nullptr;
            CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &nul, sizeof(nul)));
            CK(hipFree(d));
            std::vector<double> st, en, du;
            unsigned long long t0 = ~0ull;
            for (unsigned w = 0; w < n_waves; ++w) if (hs[4 * w]) t0 = std::min(t0, hs[4 * w]);
            std::map<unsigned, std::vector<double>> by_simd, by_xcc;
            std::map<unsigned, int> by_cu;
            for (unsigned w = 0; w < n_waves; ++w) {
                if (!hs[4 * w]) continue;
                const double dur = (hs[4 * w + 1] - hs[4 * w]) * 0.01;
                st.push_back((hs[4 * w] - t0) * 0.01); en.push_back((hs[4 * w + 1] - t0) * 0.01); du.push_back(dur);
                const unsigned hw = (unsigned)hs[4 * w + 2], xcc = (unsigned)hs[4 * w + 3] & 15u;
                const unsigned simd = (hw >> 4) & 3u, cu = (hw >> 8) & 15u, sh = (hw >> 12) & 1u, se = (hw >> 13) & 7u;
                const unsigned cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu;
                by_simd[cuid * 4 + simd].push_back(dur); by_xcc[xcc].push_back(dur); by_cu[cuid]++;
            }
            std::sort(st.begin(), st.end()); std::sort(en.begin(), en.end()); std::sort(du.begin(), du.end());
            auto q = [](const std::vector<double>& v, double f) { return v[(size_t)(f * (v.size() - 1))]; };
            printf("%s: events %.1f us | waves %zu | start p50 %.1f p100 %.1f | end p0 %.1f p50 %.1f p90 %.1f p100 %.1f | "
                   "wave us p10 %.1f p50 %.1f p90 %.1f\n", name, ms * 1e3, st.size(), q(st, .5), q(st, 1),
                   q(en, 0), q(en, .5), q(en, .9), q(en, 1), q(du, .1), q(du, .5), q(du, .9));
            std::map<size_t, std::pair<int, double>> hist;    // waves on a SIMD -> (#SIMDs, sum of mean durations)
            for (auto& kv : by_simd) { double m = 0; for (double x : kv.second) m += x; auto& h = hist[kv.second.size()]; h.first++; h.second += m / kv.second.size(); }
            printf("    SIMDs used %zu, CUs used %zu | waves per SIMD -> #SIMDs (mean wave us):", by_simd.size(), by_cu.size());
            for (auto& kv : hist) printf("  %zu -> %d (%.1f)", kv.first, kv.second.first, kv.second.second / kv.second.first);
            {   // who is slow?  mean wave time by SIMD id, by CU id, by SE, by wave slot, by wave index in the workgroup
                double sm[4] = {0}, cm[16] = {0}, em[8] = {0}, wm[16] = {0}, im[16] = {0};
                int sn[4] = {0}, cn[16] = {0}, en_[8] = {0}, wn[16] = {0}, in_[16] = {0};
                const unsigned wpwg = n_waves / (unsigned)by_cu.size() > 0 ? 0 : 0; (void)wpwg;
                for (unsigned w = 0; w < n_waves; ++w) {
                    if (!hs[4 * w]) continue;
                    const double dur = (hs[4 * w + 1] - hs[4 * w]) * 0.01;
                    const unsigned hw = (unsigned)hs[4 * w + 2];
                    sm[(hw >> 4) & 3] += dur; sn[(hw >> 4) & 3]++;
                    cm[(hw >> 8) & 15] += dur; cn[(hw >> 8) & 15]++;
                    em[(hw >> 13) & 7] += dur; en_[(hw >> 13) & 7]++;
                    wm[hw & 15] += dur; wn[hw & 15]++;
                    im[w % g_wpw] += dur; in_[w % g_wpw]++;
                }
                printf("\n    by SIMD:"); for (int k = 0; k < 4; ++k) if (sn[k]) printf(" %d:%.1f", k, sm[k] / sn[k]);
                printf("\n    by CU id:"); for (int k = 0; k < 16; ++k) if (cn[k]) printf(" %d:%.1f(%d)", k, cm[k] / cn[k], cn[k]);
                printf("\n    by SE:"); for (int k = 0; k < 8; ++k) if (en_[k]) printf(" %d:%.1f(%d)", k, em[k] / en_[k], en_[k]);
                printf("\n    by wave slot:"); for (int k = 0; k < 16; ++k) if (wn[k]) printf(" %d:%.1f(%d)", k, wm[k] / wn[k], wn[k]);
                printf("\n    by wave index in workgroup:"); for (int k = 0; k < 16; ++k) if (in_[k]) printf(" %d:%.1f", k, im[k] / in_[k]);
                // spread inside one SIMD: mean of (max - min) over SIMDs
                double spread = 0; for (auto& kv : by_simd) { auto mm = std::minmax_element(kv.second.begin(), kv.second.end()); spread += *mm.second - *mm.first; }
                printf("\n    mean (max - min) wave time inside one SIMD: %.1f us", spread / by_simd.size());
                // per CU: the time its last wave ends
                std::map<unsigned, double> cu_end;
                for (unsigned w = 0; w < n_waves; ++w) {
                    if (!hs[4 * w]) continue;
                    const unsigned hw = (unsigned)hs[4 * w + 2], xcc = (unsigned)hs[4 * w + 3] & 15u;
                    const unsigned cuid = ((xcc * 8 + ((hw >> 13) & 7u)) * 2 + ((hw >> 12) & 1u)) * 16 + ((hw >> 8) & 15u);
                    cu_end[cuid] = std::max(cu_end[cuid], (hs[4 * w + 1] - t0) * 0.01);
                }
                std::vector<double> ce; for (auto& kv : cu_end) ce.push_back(kv.second); std::sort(ce.begin(), ce.end());
                printf("\n    last wave of a CU ends at: p0 %.1f p10 %.1f p50 %.1f p90 %.1f p100 %.1f", ce[0], ce[ce.size() / 10], ce[ce.size() / 2], ce[ce.size() * 9 / 10], ce.back());
            }
            printf("\n    per XCC waves (mean us):");
            for (auto& kv : by_xcc) { double m = 0; for (double x : kv.second) m += x; printf("  %u: %zu (%.1f)", kv.first, kv.second.size(), m / kv.second.size()); }
            printf("\n");
        };
#define STAMPL(NPv, SRCv, NCv, POLv, IRv, THR, CAP) \
        g_wpw = THR / 64; stamped("stamps NC=" #NCv " pol=" #POLv " IR=" #IRv " thr=" #THR " cap=" #CAP, (unsigned)(((size_t)((N + 64 * NCv - 1) / (64 * NCv)) * ((N + IRv - 1) / IRv) * B + THR / 64 - 1) / (THR / 64) * (THR / 64)), \
                [&] { return launch_strip_lds<NPv, SRCv, NCv, POLv>(xyz, out, B, N, A, sel, IRv, nullptr, THR, CAP); })
#define STAMPC(NPv, SRCv, NCv, POLv, CHv, THR, WPC, RND) \
        g_wpw = THR / 64; stamped("stamps cu NC=" #NCv " pol=" #POLv " CH=" #CHv " thr=" #THR " wg/cu=" #WPC " rounds=" #RND, (unsigned)(256 * WPC * RND * (THR / 64)), \
                [&] { return launch_cu<NPv, SRCv, NCv, POLv>(xyz, out, B, N, A, sel, CHv, nullptr, THR, WPC, RND); })
#define STAMPS(NPv, SRCv)
        if (f.np == 4 && f.src[2] == 1) { STAMPS(4, 12); } else if (f.np == 4) { STAMPS(4, 8); } else { STAMPS(3, 4); }
#define STRIPL(NPv, SRCv, NCv, POLv, IRv, THR, CAP) \
        run("strip_lds NC=" #NCv " pol=" #POLv " IR=" #IRv " thr=" #THR " cap=" #CAP, [&] { return launch_strip_lds<NPv, SRCv, NCv, POLv>(xyz, out, B, N, A, sel, IRv, nullptr, THR, CAP); })
#define CU(NPv, SRCv, NCv, POLv, CHv, THR, WPC, RND) \
        run("cu NC=" #NCv " pol=" #POLv " CH=" #CHv " thr=" #THR " wg/cu=" #WPC " rounds=" #RND, [&] { return launch_cu<NPv, SRCv, NCv, POLv>(xyz, out, B, N, A, sel, CHv, nullptr, THR, WPC, RND); })
#define SWEEP(NPv, SRCv) \
        CU(NPv, SRCv, 4, 16, 8, 1024, 1, 1); CU(NPv, SRCv, 4, 16, 4, 1024, 1, 1); CU(NPv, SRCv, 4, 16, 16, 1024, 1, 1); CU(NPv, SRCv, 2, 16, 8, 1024, 1, 1); \
        run("product again", [&] { return ps_pairwise_angles_f32(xyz, out, B, N, A, f.np, f.src, f.atom, 0, N, N, 0, 0, nullptr); })
        if (f.np == 4 && f.src[2] == 1) { SWEEP(4, 12); }
        else if (f.np == 4) { SWEEP(4, 8); }
        else { SWEEP(3, 4); }
    }
    return 0;
}
