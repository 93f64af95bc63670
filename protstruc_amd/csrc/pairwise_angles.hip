// K3 -- inter-residue dihedrals and planar angles over all residue pairs, and the fused featuriser.
// Replaces StructureBatch.pairwise_dihedrals / pairwise_planar_angles and the
// (B, N*N, n_i+n_j, 3) gather of _pairwise_xyz (reference protstruc.py:589-660);
// geometry.dihedral / geometry.angle (geometry.py:39-124) are evaluated per pair
// in registers, so the 1.6 GB gather of the reference is never materialised.
//
// Which point comes from which side is a template parameter (SRC bit k = 1: point k
// from the column residue j), so nothing is selected at run time and what depends on
// one side only is shared or hoisted.  4 bytes are written per pair against ~90 flops
// and an atan2 / acos: the kernels are bound by the vector unit, not by HBM.
// Two arithmetics (exact_angles bit 0, ps_common.hpp): the fast one spends the 1e-5
// parity tolerance (triple-product dihedral, polynomial atan2 / acos), the faithful one
// is the reference's order of operations; every kernel below exists in both.
//
// Kernels, in the order of this file (DESIGN.md section 4 has the dispatch table and
// the numbers; ps_k3_plan_f32 / ps_featuriser_plan_f32 name the one a launch takes):
//   k3_pairwise_angles   one column per lane, rows by scalar loads: the layout-free twin
//                        every other kernel is held to bit for bit, and the fallback
//   k3_small             N <= 32: one wave per structure
//   k3_flat              33..99 residues (and wherever a sweep would idle its lanes): a lane's
//                        element is a 2 x 2 tile (row pairs x columns), several structures staged per pass
//   k3_sweep             >= 100 residues: one workgroup per CU, rows staged in LDS, tasks
//                        pulled from an LDS counter, NC column residues per lane
//   k3_inter_residue_geometry, k3_featurise, k3_featurise_tiles   the fused featuriser: one-column / sweep /
//                        the tile map of k3_flat over all nine planes (8..96 residues and where a sweep would idle lanes)
#include "ps_common.hpp"
#include "../../include/protstruc_hip.h"

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <type_traits>

namespace {

struct AtomSel {
    int atom[4];
};

typedef float k3_f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t k3_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t k3_u32x2 __attribute__((ext_vector_type(2)));

// The device a launch on `stream` runs on: the stream's own device; the calling thread's current device for the null stream
// (a C-ABI caller may launch on a stream of device 1 while device 0 is current: grid size and the LDS attribute are that
// device's business, not the current one's)
inline int k3_device_of(hipStream_t stream) {
    int dev = 0;
    if (stream) {
        hipDevice_t d;
        if (hipStreamGetDevice(stream, &d) == hipSuccess) return (int)d;
    }
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    return dev;
}

// CUs of that device (cached per ordinal; a constant of the device, not library state)
inline int k3_cu_count(hipStream_t stream) {
    static int cached[64] = {0};
    const int dev = k3_device_of(stream);
    if (dev < 0 || dev >= 64) return 256;
    int n = __atomic_load_n(&cached[dev], __ATOMIC_RELAXED);
    if (n <= 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        __atomic_store_n(&cached[dev], n, __ATOMIC_RELAXED);
    }
    return n;
}

// More than 64 KB of dynamic LDS has to be allowed once per kernel AND per device (function attributes belong to the module
// a device loaded; hipFuncSetAttribute acts on the CURRENT device, so the stream's device is made current around it when it
// is another one).  `done` is the kernel instantiation's own table; idempotent, so a race between two threads is harmless.
template <typename K>
inline int k3_allow_big_lds(K kernel, unsigned long long (&done)[1], hipStream_t stream) {
    const int dev = k3_device_of(stream);
    if (dev < 0 || dev >= 64) return (int)hipErrorInvalidDevice;
    const unsigned long long bit = 1ull << dev;
    if (__atomic_load_n(&done[0], __ATOMIC_ACQUIRE) & bit) return 0;
    int cur = dev;
    (void)hipGetDevice(&cur);
    if (cur != dev && hipSetDevice(dev) != hipSuccess) return (int)hipErrorInvalidDevice;
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)(160 * 1024 - 256));
    if (cur != dev) (void)hipSetDevice(cur);
    if (e != hipSuccess) return (int)e;
    __atomic_fetch_or(&done[0], bit, __ATOMIC_RELEASE);
    return 0;
}

constexpr bool K3_FEATURISE_NC4 = true;           // four columns per lane: 512-thread workgroups (two waves per SIMD, 256 VGPRs each)
// the faithful sweeps take four columns per lane for every split (at most 247 of the 256 VGPRs of their 512-thread workgroups,
// nothing spilled: tools/kernel_resources.sh; same-box A/B at config 3 against two columns at 1024 threads: 80 / 66 / 43 us against
// 83 / 74 / 48, profiles/r05_k3_modes_first.log)
constexpr bool K3_FAITHFUL_NC4(int, int) { return true; }
// threads of a full-width sweep workgroup: four waves per SIMD with 128 VGPRs each -- two with 256 for the faithful chains at
// four columns per lane (the library-order chains keep ~2x the values live)
constexpr int k3_sweep_threads(int NC, bool FAITHFUL) { return FAITHFUL && NC == 4 ? 512 : 1024; }
#ifdef PS_K3_AB
// tools only (tools/k3f_probe.py): 2 = every K3 / featuriser kernel computes but stores out of range (the arithmetic and the store
// issue alone); 1 = k3_featurise_tiles stores constants instead of computing (the store pattern alone)
__device__ int k3f_probe;
#define K3_PROBE_RECORDS(n) (k3f_probe == 2 ? 0u : (unsigned)(n))
#else
#define K3_PROBE_RECORDS(n) (n)
#endif
constexpr size_t K3_LDS_MAX = 160 * 1024 - 256;   // the most dynamic LDS a workgroup of the sweep kernels asks for
constexpr size_t K3_LDS_ONE_PER_CU = 80 * 1024;   // with its few static bytes on top, two such workgroups do not fit a CU
constexpr size_t K3_LDS_TWO_PER_CU = 80 * 1024 - 256;   // exactly two such workgroups fit a CU (three would need 240 KB)

template <int NP, int SRC, bool FAITHFUL>
__global__ __launch_bounds__(256) void k3_pairwise_angles(const float* __restrict__ xyz, float* __restrict__ out,
                                                          int N, int A, AtomSel sel, int row_begin, int row_end,
                                                          int out_rows, int out_row_origin, int IR, int n_tiles,
                                                          int n_chunks) {
    // 1-D grid (column tile fastest, then row chunk, then structure): no 65 535 limit on any axis
    const unsigned w = blockIdx.x;
    const unsigned tile = w % (unsigned)n_tiles, rest = w / (unsigned)n_tiles;
    const int b = (int)(rest / (unsigned)n_chunks);
    const int j = (int)tile * (int)blockDim.x + threadIdx.x;   // a workgroup is 64, 128, 192 or 256 lanes wide (short chains)
    const int i0 = row_begin + (int)(rest % (unsigned)n_chunks) * IR;
    const int i1 = min(i0 + IR, row_end);
    const bool live = j < N;
    const int jc = live ? j : N - 1;

    const float* sj = xyz + ((size_t)b * N + jc) * (size_t)A * 3;
    f3 pj[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) pj[k] = ((SRC >> k) & 1) ? load3(sj + sel.atom[k] * 3) : mk3(0.f, 0.f, 0.f);

    int i = i0;
    if constexpr (FAITHFUL) {   // the reference's order of operations (dihedral4_ref / angle3_ref), one row per trip
        for (; i < i1; ++i) {
            const float* si = xyz + ((size_t)b * N + i) * (size_t)A * 3;  // wave-uniform
            f3 p[NP];
#pragma unroll
            for (int k = 0; k < NP; ++k) p[k] = ((SRC >> k) & 1) ? pj[k] : load3(si + sel.atom[k] * 3);
            float v;
            if constexpr (NP == 4)
                v = dihedral4_ref(p[0], p[1], p[2], p[3]);
            else
                v = angle3_ref(p[0], p[1], p[2]);
            if (live) out[((size_t)b * out_rows + (size_t)(i - out_row_origin)) * N + j] = v;
        }
        return;
    }
    // two rows per trip in the two halves of float2 registers -> packed v_pk_* math (bit-identical per element);
    // dihedrals and planar angles alike
    for (; i + 1 < i1; i += 2) {
        const float* s0 = xyz + ((size_t)b * N + i) * (size_t)A * 3;  // wave-uniform
        const float* s1 = s0 + (size_t)A * 3;
        f3v p[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k)
            p[k] = ((SRC >> k) & 1) ? mk3v(pj[k], pj[k]) : mk3v(load3(s0 + sel.atom[k] * 3), load3(s1 + sel.atom[k] * 3));
        f32x2 v;
        if constexpr (NP == 4)
            v = dihedral4v_k3(p[0], p[1], p[2], p[3]);
        else
            v = angle3v(p[0], p[1], p[2]);
        if (live) {
            float* o = out + ((size_t)b * out_rows + (size_t)(i - out_row_origin)) * N + j;
            o[0] = v.x;
            o[N] = v.y;
        }
    }
    for (; i < i1; ++i) {
        const float* si = xyz + ((size_t)b * N + i) * (size_t)A * 3;  // wave-uniform
        f3 p[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) p[k] = ((SRC >> k) & 1) ? pj[k] : load3(si + sel.atom[k] * 3);
        float v;
        if constexpr (NP == 4)
            v = dihedral4_k3(p[0], p[1], p[2], p[3]);
        else
            v = angle3(p[0], p[1], p[2]);
        if (live) out[((size_t)b * out_rows + (size_t)(i - out_row_origin)) * N + j] = v;
    }
}

// Short chains (N <= 64; peptide batches).  A 64-lane wave of the one-column kernel covers 64 column residues, so a
// 16-residue chain uses a quarter of it.  Here a wave owns one structure and its lanes are (row group, column): Npad = 16, 32
// or 64 columns x G = 64 / Npad rows at a time; the row-side points are per-lane vector loads (the Npad lanes of a row group
// share an address), four rows per trip (two packed pairs, their chains interleaved), and a store instruction writes G
// consecutive rows of the structure.  Same arithmetic per pair as the one-column kernel: same bits.
template <int NP, int SRC, bool FAITHFUL = false>
__global__ __launch_bounds__(256) void k3_small(const float* __restrict__ xyz, float* __restrict__ out, int B, int N, int A,
                                                AtomSel sel, int row_begin, int row_end, int out_rows, int out_row_origin,
                                                int lg_npad) {
    const int lane = threadIdx.x & 63;
    const int b = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);   // one wave per structure
    if (b >= B) return;
    const int npad = 1 << lg_npad, G = 64 >> lg_npad;
    const int j = lane & (npad - 1), ri = lane >> lg_npad;
    const bool live_j = j < N;
    const float* xb = xyz + (size_t)b * N * (size_t)A * 3;
    const float* sj = xb + (size_t)min(j, N - 1) * (size_t)A * 3;
    f3 pj[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) pj[k] = ((SRC >> k) & 1) ? load3(sj + sel.atom[k] * 3) : mk3(0.f, 0.f, 0.f);
    float* ob = out + (size_t)b * out_rows * N;
    for (int i0 = row_begin + ri; i0 < row_end; i0 += 4 * G) {     // rows i0, i0 + G, i0 + 2G, i0 + 3G of this lane
        f3v P[NP][2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float* sa = xb + (size_t)min(i0 + 2 * h * G, N - 1) * (size_t)A * 3;
            const float* sb2 = xb + (size_t)min(i0 + (2 * h + 1) * G, N - 1) * (size_t)A * 3;
#pragma unroll
            for (int k = 0; k < NP; ++k)
                P[k][h] = ((SRC >> k) & 1) ? mk3v(pj[k], pj[k]) : mk3v(load3(sa + sel.atom[k] * 3), load3(sb2 + sel.atom[k] * 3));
        }
        f32x2 v[2];
        if constexpr (NP == 4 && FAITHFUL)
            dihedral4v_ref_n<2>(P[0], P[1], P[2], P[3], v);
        else if constexpr (NP == 4)
            dihedral4v_k3_n<2>(P[0], P[1], P[2], P[3], v);
        else if constexpr (FAITHFUL)
            angle3v_ref_n<2>(P[0], P[1], P[2], v);
        else
            angle3v_n<2>(P[0], P[1], P[2], v);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ia = i0 + 2 * h * G, ib = ia + G;
            if (live_j && ia < row_end) ob[(size_t)(ia - out_row_origin) * N + j] = v[h].x;
            if (live_j && ib < row_end) ob[(size_t)(ib - out_row_origin) * N + j] = v[h].y;
        }
    }
}

// Chains of 33 .. ~100 residues, and longer ones wherever a sweep would idle its lanes (round 5): several structures per
// staging pass, every lane busy whatever N is.
// The per-CU sweep gives a wave a strip of 64 * NC columns of ONE structure: below ~100 residues most lanes of a strip have no
// column, a (structure, strip) segment has fewer tasks than the workgroup has waves, and every segment costs two barriers and
// a round trip to L2 (N = 64: 143 us at 2^25 pairs, which is why such chains stayed with the one-column kernel: 85 us).  Here
//   * the selected atoms of KS structures are staged at once (column-side atoms as {x, y, z, -} per (atom, residue), row-side
//     atoms pair-interleaved as {x0, x1, y0, y1}, {z0, z1, -, -} per (row pair, atom): every read in the loop is 16 bytes),
//     so the two barriers and the load latency of a pass are paid once per KS * N * N pairs instead of once per 4 096;
//   * TWO 512-thread workgroups share a CU and a workgroup's share takes four or more passes: while one workgroup waits for
//     its next structures the other computes;
//   * both sides of a pair come from LDS per lane (lanes of one row pair read the same address: a broadcast);
//   * tasks are pulled from an LDS counter, as in the sweep; stores are unconditional (a dead element's offset lies beyond the
//     buffer descriptor's range and the hardware drops it), so the four chains stay in one basic block and interleave.
// Lane map: a lane's element is a 2 x 2 TILE -- two row pairs x two adjacent columns, four chains -- so that what depends on
// the row pair alone is shared by the tile's two columns and what depends on the column alone by its two row pairs (as in the
// sweep), the index arithmetic is paid once per tile and two columns are one 8-byte store: ~300 VALU instructions per 8 pairs,
// 57-64 us per 2^25 pairs from 64 to 220 residues.  Measured and removed (profiles/r05_k3_flat_lane_maps.log, NOTES.md): four
// elements 64 apart in the flat index (408 instructions per 8 pairs, 68-77 us), a lane per column with four consecutive row
// pairs (N <= 64: 324 instructions, 69 us at N = 64), a plain coalesced copy of the coordinates instead of the gather of the
// selected atoms (88 us: it reads all 15 atoms), one 1024-thread workgroup staging its whole share at once (74 us).
// Same arithmetic per pair as the one-column kernel: same bits.
template <int NP, int SRC, bool FAITHFUL>
__global__ __launch_bounds__(1024) void k3_flat(const float* __restrict__ xyz, float* __restrict__ out, int N, int A,
                                                AtomSel sel, int row_begin, int row_end, int out_rows,
                                                int out_row_origin, int KS, unsigned tps, unsigned n_tasks,
                                                unsigned tasks_per_wg, unsigned rcpN, int col_vec4, int slot_vec4,
                                                unsigned rcpTC, int vec) {
    constexpr int NPI = NP - __builtin_popcount(SRC & ((1 << NP) - 1)), NPJ = NP - NPI;
    constexpr int NPIq = NPI > 0 ? NPI : 1, NPJq = NPJ > 0 ? NPJ : 1;
    // [structure slot][column part: (atom, residue) -> {x, y, z, -} | row part: (row pair, atom) -> {x0, x1, y0, y1}, {z0, z1, -, -}]
    extern __shared__ __attribute__((aligned(16))) k3_f32x4 k3_flatbuf[];
    __shared__ unsigned next_task;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned n_waves = blockDim.x >> 6;
    const unsigned t0 = blockIdx.x * tasks_per_wg, t1 = min(t0 + tasks_per_wg, n_tasks);
    if (t0 >= t1) return;                         // whole workgroup
    int amap_i[NPIq], amap_j[NPJq];
    {
        int qi = 0, qj = 0;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            if ((SRC >> k) & 1) amap_j[qj++] = sel.atom[k];
            else amap_i[qi++] = sel.atom[k];
        }
        if (NPI == 0) amap_i[0] = 0;
        if (NPJ == 0) amap_j[0] = 0;
    }
    const int rows = row_end - row_begin, n_rp = (rows + 1) >> 1;
    const unsigned b_first = t0 / tps, b_last = (t1 - 1u) / tps;
    for (unsigned bs = b_first; bs <= b_last; bs += (unsigned)KS) {
        const unsigned ks = min((unsigned)KS, b_last - bs + 1u);
        __syncthreads();                                          // the previous pass's readers are done
        for (unsigned it = threadIdx.x; it < ks * (unsigned)N; it += blockDim.x) {   // one residue of one structure per thread
            unsigned s = __umulhi(it, rcpN), r = it - s * (unsigned)N;
            if (r >= (unsigned)N) ++s, r -= (unsigned)N;
            const float* pr = xyz + ((size_t)(bs + s) * N + r) * (size_t)A * 3;
            k3_f32x4* slot = k3_flatbuf + (size_t)s * slot_vec4;
            float vj[NPJq * 3], vi[NPIq * 3];
#pragma unroll
            for (int q = 0; q < NPJ; ++q) { vj[q * 3] = pr[amap_j[q] * 3]; vj[q * 3 + 1] = pr[amap_j[q] * 3 + 1]; vj[q * 3 + 2] = pr[amap_j[q] * 3 + 2]; }
            const int rr = (int)r - row_begin;
            const bool in_rows = NPI > 0 && rr >= 0 && rr < rows;
            if (in_rows) {
#pragma unroll
                for (int q = 0; q < NPI; ++q) { vi[q * 3] = pr[amap_i[q] * 3]; vi[q * 3 + 1] = pr[amap_i[q] * 3 + 1]; vi[q * 3 + 2] = pr[amap_i[q] * 3 + 2]; }
            }
#pragma unroll
            for (int q = 0; q < NPJ; ++q) slot[q * N + (int)r] = k3_f32x4{vj[q * 3], vj[q * 3 + 1], vj[q * 3 + 2], 0.0f};
            if (in_rows) {
                float* rb = reinterpret_cast<float*>(slot + col_vec4) + (size_t)(rr >> 1) * (NPI * 8) + (rr & 1);
#pragma unroll
                for (int q = 0; q < NPI; ++q) { rb[q * 8] = vi[q * 3]; rb[q * 8 + 2] = vi[q * 3 + 1]; rb[q * 8 + 4] = vi[q * 3 + 2]; }
            }
        }
        const unsigned seg_t0 = max(t0, bs * tps), seg_t1 = min(t1, (bs + ks) * tps);
        if (threadIdx.x == 0) next_task = seg_t0 + n_waves;       // the first n_waves tasks are pre-assigned
        __syncthreads();
        unsigned t = seg_t0 + (unsigned)wave;
        while (t < seg_t1) {
            const unsigned b = t / tps, chunk = t - b * tps;      // (uniform)
            const k3_f32x4* slot = k3_flatbuf + (size_t)(b - bs) * slot_vec4;
            const k3_f32x4* rowp = slot + col_vec4;
            float* obase = out + ((size_t)b * out_rows + (size_t)(row_begin - out_row_origin)) * N;
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(obase, 0, (int)K3_PROBE_RECORDS(rows * N * 4), 0x00020000u);   // this structure's rows, exactly
            // the lane's tile: two row pairs x two adjacent columns (four chains) -- or, where a row of FOUR columns is one aligned
            // 16-byte store (vec == 4: N % 4 == 0), two such halves that share the index arithmetic and the row-side points
            // (N = 64 .. 220: 58-65 / 60-66 / 43-50 us per 2^25 pairs against 62-73 / 68-79 / 48-59; without the wide store the
            // four-column tile loses: N = 99 76 against 70 us, N = 33 148 against 129; profiles/r05_k3_flat_tile_width.log)
            const bool wide = vec == 4;                      // (uniform)
            const unsigned TC = wide ? (unsigned)N >> 2 : (unsigned)(N + 1) >> 1, TR = (unsigned)(n_rp + 1) >> 1, FT = TR * TC;
            const unsigned ti = chunk * 64u + (unsigned)lane;
            const bool lt = ti < FT;
            const unsigned tcl = min(ti, FT - 1u);
            unsigned tr = __umulhi(tcl, rcpTC), tc = tcl - tr * TC;
            if (tc >= TC) ++tr, tc -= TC;
            const int c0 = (int)(wide ? 4u * tc : 2u * tc);
            const int rpA = (int)(2u * tr), rpB = min(rpA + 1, n_rp - 1);
            const bool lc1 = c0 + 1 < N, lrB = rpA + 1 < n_rp;
            f3v RA[NPIq], RB[NPIq];                   // the row-side points of the two row pairs ({row 2 rp, row 2 rp + 1} per component)
#pragma unroll
            for (int qi = 0; qi < NPI; ++qi) {
                const k3_f32x4 xa = rowp[(rpA * NPI + qi) * 2], xb = rowp[(rpB * NPI + qi) * 2];
                const f32x2 za = *reinterpret_cast<const f32x2*>(rowp + (rpA * NPI + qi) * 2 + 1);
                const f32x2 zb = *reinterpret_cast<const f32x2*>(rowp + (rpB * NPI + qi) * 2 + 1);
                RA[qi] = f3v{f32x2{xa.x, xa.y}, f32x2{xa.z, xa.w}, za};
                RB[qi] = f3v{f32x2{xb.x, xb.y}, f32x2{xb.z, xb.w}, zb};
            }
            // columns ca, ca + 1 (clamped: a dead column is never stored) against the two row pairs:
            // v[0], v[1] = row pair A x the two columns; v[2], v[3] = row pair B; .x / .y = the pair's rows
            auto half = [&](int ca, f32x2 (&v)[4]) {
                const int cb = min(ca + 1, N - 1);
                f3v P[NP][4];
                int qi = 0, qj = 0;
#pragma unroll
                for (int k = 0; k < NP; ++k) {
                    if ((SRC >> k) & 1) {
                        const k3_f32x4 p0 = slot[qj * N + ca], p1 = slot[qj * N + cb];
                        P[k][0] = P[k][2] = mk3v(f3{p0.x, p0.y, p0.z}, f3{p0.x, p0.y, p0.z});
                        P[k][1] = P[k][3] = mk3v(f3{p1.x, p1.y, p1.z}, f3{p1.x, p1.y, p1.z});
                        ++qj;
                    } else {
                        P[k][0] = P[k][1] = RA[qi];
                        P[k][2] = P[k][3] = RB[qi];
                        ++qi;
                    }
                }
                if constexpr (NP == 4 && FAITHFUL)
                    dihedral4v_ref_n<4>(P[0], P[1], P[2], P[3], v);
                else if constexpr (NP == 4)
                    dihedral4v_k3_n<4>(P[0], P[1], P[2], P[3], v);
                else if constexpr (FAITHFUL)
                    angle3v_ref_n<4>(P[0], P[1], P[2], v);
                else
                    angle3v_n<4>(P[0], P[1], P[2], v);
#pragma unroll
                for (int c = 0; c < 4; ++c) asm volatile("" : "+v"(v[c]));
            };
            f32x2 v[4];
            half(c0, v);
            const int offA = (int)__umul24((unsigned)(2 * rpA), (unsigned)N) * 4 + c0 * 4;
            const int offB = (int)__umul24((unsigned)(2 * rpB), (unsigned)N) * 4 + c0 * 4;
            const bool rA1 = 2 * rpA + 1 < rows, rB0 = lrB, rB1 = lrB && 2 * rpB + 1 < rows;   // rows 2 rpA + 1, 2 rpB, 2 rpB + 1 exist
            constexpr int DEAD = 0x7FFFFFF0;    // beyond num_records: dropped by the range check
            if (wide) {              // N % 4 == 0 and the rows 16-byte aligned: a row of the tile is one store
                f32x2 w[4];
                half(c0 + 2, w);     // (c0 + 3 < N)
                auto st4 = [&](float a, float b, float c, float d, int off) {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(k3_u32x4, k3_f32x4{a, b, c, d}), rsrc, off, 0, 0);
                };
                st4(v[0].x, v[1].x, w[0].x, w[1].x, lt ? offA : DEAD);
                st4(v[0].y, v[1].y, w[0].y, w[1].y, (lt && rA1) ? offA + N * 4 : DEAD);
                st4(v[2].x, v[3].x, w[2].x, w[3].x, (lt && rB0) ? offB : DEAD);
                st4(v[2].y, v[3].y, w[2].y, w[3].y, (lt && rB1) ? offB + N * 4 : DEAD);
            } else if (vec == 2) {   // (uniform) N even and the rows 8-byte aligned: the tile's two columns are one store
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(k3_u32x2, f32x2{v[0].x, v[1].x}), rsrc, lt ? offA : DEAD, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(k3_u32x2, f32x2{v[0].y, v[1].y}), rsrc, (lt && rA1) ? offA + N * 4 : DEAD, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(k3_u32x2, f32x2{v[2].x, v[3].x}), rsrc, (lt && rB0) ? offB : DEAD, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(k3_u32x2, f32x2{v[2].y, v[3].y}), rsrc, (lt && rB1) ? offB + N * 4 : DEAD, 0, 0);
            } else {
                const bool l1 = lt && lc1;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[0].x), rsrc, lt ? offA : DEAD, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[1].x), rsrc, l1 ? offA + 4 : DEAD, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[0].y), rsrc, (lt && rA1) ? offA + N * 4 : DEAD, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[1].y), rsrc, (l1 && rA1) ? offA + N * 4 + 4 : DEAD, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[2].x), rsrc, (lt && rB0) ? offB : DEAD, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[3].x), rsrc, (l1 && rB0) ? offB + 4 : DEAD, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[2].y), rsrc, (lt && rB1) ? offB + N * 4 : DEAD, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[3].y), rsrc, (l1 && rB1) ? offB + N * 4 + 4 : DEAD, 0, 0);
            }
            unsigned nx = 0;
            if (lane == 0) nx = atomicAdd(&next_task, 1u);
            t = (unsigned)__builtin_amdgcn_readfirstlane((int)nx);
        }
    }
}

// The sweep for even N (round 4): ONE 1024-thread workgroup per CU, NC = 2 or 4 consecutive column residues per lane
// (8- or 16-byte stores), two rows per trip in the halves of float2 registers.
//   * Work list: task t = (b * n_strips + strip) * n_chunks + chunk = CH rows x one strip of 64 * NC columns of one
//     structure.  A workgroup owns a contiguous range of it (1 / #CUs of the list); its LDS request keeps a second
//     workgroup off the CU, so every CU gets the same share whatever the dispatcher does.
//   * The row-side points of a (b, strip) segment are staged ONCE in LDS, pair-interleaved: one broadcast ds_read_b64 /
//     b128 delivers {row 2r, row 2r + 1} of a component already in the packed layout -- no scalar loads, no s_waitcnt on
//     SMEM and no SGPR -> VGPR moves inside the loop (the kernels before staged nothing: rows came by s_load per trip).
//   * The 16 waves PULL tasks from an LDS counter.  The SIMD arbiter serves its oldest wave first: with equal static
//     shares the four waves of a SIMD finish one after the other (18 / 28 / 38 / 47 us at config 3) and the last one runs
//     alone at half the issue rate; pulling lets the favoured waves take more tasks and all of them end together.
//   * Stores are write-through (sc1): the 134 MB of config 3 would otherwise leave up to 32 MB dirty in the L2s for the
//     end-of-kernel write-back, which is serial with everything (3-4 us of 57).
//   * The arithmetic of the NC columns is evaluated step by step across the columns (dihedral4v_k3_n, angle3v_n).
// Same operations per pair as the one-column kernel above: same bits.
// VEC = false (odd N, or an output that is not 4 * NC-byte aligned): the lane's NC columns are 64 apart instead of adjacent
// (column = strip * 64 * NC + 64 * c + lane), so every store instruction writes 64 consecutive floats of one row -- no
// alignment is needed at all -- at the price of NC dword stores per row instead of one vector store.
// One task's row pairs for L of the lane's NC columns (see the kernel; pinned results and interleaved chains as described there).
#define K3_SWEEP_ROWS(L)                                                                                                  \
    {                                                                                                                     \
            int i = i0;                                                                                                  \
            for (; i + 1 < i1; i += 2) {                                                                                 \
                f3v cur[NP];                                                                                             \
                rows(i >> 1, cur);                                                                                       \
                f3v P[NP][L];                                                                                            \
_Pragma("unroll")                                                                                                        \
                for (int k = 0; k < NP; ++k)                                                                             \
_Pragma("unroll")                                                                                                        \
                    for (int cc = 0; cc < L; ++cc) P[k][cc] = ((SRC >> k) & 1) ? mk3v(pj[cc][k], pj[cc][k]) : cur[k];    \
                f32x2 v[L];                                                                                              \
                if constexpr (NP == 4 && FAITHFUL)                                                                       \
                    dihedral4v_ref_n<L>(P[0], P[1], P[2], P[3], v);                                                      \
                else if constexpr (NP == 4)                                                                              \
                    dihedral4v_k3_n<L>(P[0], P[1], P[2], P[3], v);                                                       \
                else if constexpr (FAITHFUL)                                                                             \
                    angle3v_ref_n<L>(P[0], P[1], P[2], v);                                                               \
                else                                                                                                     \
                    angle3v_n<L>(P[0], P[1], P[2], v);                                                                   \
_Pragma("unroll")                                                                                                        \
                for (int cc = 0; cc < L; ++cc) asm volatile("" : "+v"(v[cc]));                                           \
                if constexpr (!VEC) {                                                                                    \
                    const int so = i * row_bytes;                                                                        \
_Pragma("unroll")                                                                                                        \
                    for (int cc = 0; cc < L; ++cc)                                                                       \
                        if (j0 + 64 * cc < N) {                                                                          \
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[cc].x), rsrc, lane_off + 256 * cc, so, POL); \
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[cc].y), rsrc, lane_off + 256 * cc, so + row_bytes, POL); \
                        }                                                                                                \
                } else if (live) {                                                                                       \
                    const int so = i * row_bytes;                                                                        \
                    if constexpr (NC == 4) {                                                                             \
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(k3_u32x4, k3_f32x4{v[0].x, v[1].x, v[2].x, v[3].x}), rsrc, lane_off, so, POL); \
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(k3_u32x4, k3_f32x4{v[0].y, v[1].y, v[2].y, v[3].y}), rsrc, lane_off, so + row_bytes, POL); \
                    } else {                                                                                             \
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(k3_u32x2, f32x2{v[0].x, v[1].x}), rsrc, lane_off, so, POL); \
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(k3_u32x2, f32x2{v[0].y, v[1].y}), rsrc, lane_off, so + row_bytes, POL); \
                    }                                                                                                    \
                }                                                                                                        \
            }                                                                                                            \
            if (i < i1) {                                                                                                \
                f3v cur[NP];                                                                                             \
                rows(i >> 1, cur);                                                                                       \
                float v[NC];                                                                                             \
_Pragma("unroll")                                                                                                        \
                for (int cc = 0; cc < L; ++cc) {                                                                         \
                    f3 p[NP];                                                                                            \
_Pragma("unroll")                                                                                                        \
                    for (int k = 0; k < NP; ++k) p[k] = ((SRC >> k) & 1) ? pj[cc][k] : mk3(cur[k].x.x, cur[k].y.x, cur[k].z.x); \
                    if constexpr (NP == 4)                                                                               \
                        v[cc] = FAITHFUL ? dihedral4_ref(p[0], p[1], p[2], p[3]) : dihedral4_k3(p[0], p[1], p[2], p[3]); \
                    else                                                                                                 \
                        v[cc] = FAITHFUL ? angle3_ref(p[0], p[1], p[2]) : angle3(p[0], p[1], p[2]);                      \
                }                                                                                                        \
                if constexpr (!VEC) {                                                                                    \
                    const int so = i * row_bytes;                                                                        \
_Pragma("unroll")                                                                                                        \
                    for (int cc = 0; cc < L; ++cc)                                                                       \
                        if (j0 + 64 * cc < N) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[cc]), rsrc, lane_off + 256 * cc, so, POL); \
                } else if (live) {                                                                                       \
                    const int so = i * row_bytes;                                                                        \
                    if constexpr (NC == 4)                                                                               \
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(k3_u32x4, k3_f32x4{v[0], v[1], v[2], v[3]}), rsrc, lane_off, so, POL); \
                    else                                                                                                 \
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(k3_u32x2, f32x2{v[0], v[1]}), rsrc, lane_off, so, POL); \
                }                                                                                                        \
            }                                                                                                            \
    }
#ifdef PS_K3_AB
// tools only (-DPS_K3_AB): per wave of the last k3_sweep launch, the 100 MHz wall clock at entry, after the first staging
// barrier, after the wave's first task, after its last task (tools/k3_stamps.py; DESIGN.md section 4, "where K3's time goes")
__device__ unsigned long long k3_stamps[512 * 16 * 4];
#define K3_STAMP(slot) if (lane == 0 && blockIdx.x < 512) k3_stamps[((size_t)blockIdx.x * 16 + wave) * 4 + (slot)] = wall_clock64()
// ... and of the last k3_featurise_tiles launch: entry, then per staging pass (the first five): pass begun (after the top
// barrier), rows staged (after the second barrier), this wave's last task of the pass done (tools/k3f_stamps.py)
__device__ unsigned long long k3f_stamps[512 * 8 * 16];
#define K3F_STAMP(slot) if (lane == 0 && blockIdx.x < 512 && (slot) < 16) k3f_stamps[((size_t)blockIdx.x * 8 + wave) * 16 + (slot)] = wall_clock64()
#else
#define K3_STAMP(slot)
#define K3F_STAMP(slot)
#endif

template <int NP, int SRC, int NC, bool VEC, bool FAITHFUL = false>
__global__ __launch_bounds__(k3_sweep_threads(NC, FAITHFUL)) void k3_sweep(const float* __restrict__ xyz, float* __restrict__ out, int N, int A,
                                                 AtomSel sel, int row_begin, int row_end, int out_rows,
                                                 int out_row_origin, int CH, int n_strips, int n_chunks,
                                                 unsigned n_tasks, unsigned tasks_per_wg) {
    static_assert(NC == 2 || NC == 4, "columns per lane");
    constexpr int NPI = NP - __builtin_popcount(SRC & ((1 << NP) - 1));   // points taken from the row residue
    constexpr int NPIq = NPI > 0 ? NPI : 1;
    constexpr int POL = 16;                       // sc1: write-through
    extern __shared__ __attribute__((aligned(16))) f32x2 k3_rowbuf[];   // [row pair][row point][xyz] of the current segment
    __shared__ unsigned next_task;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int n_waves = (int)(blockDim.x >> 6);
    const unsigned t0 = blockIdx.x * tasks_per_wg, t1 = min(t0 + tasks_per_wg, n_tasks);
    K3_STAMP(0);
    if (t0 >= t1) return;                         // whole workgroup
    int amap[NPIq];
    {
        int q = 0;
#pragma unroll
        for (int k = 0; k < NP; ++k)
            if (!((SRC >> k) & 1)) amap[q++] = sel.atom[k];
        if (NPI == 0) amap[0] = 0;
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
        // the column-side points are requested BEFORE the rows are staged: the two global-memory latencies of a segment's
        // set-up overlap instead of following each other (a 128-residue segment computes for ~6 us; its set-up was ~2)
        // VEC: NC adjacent columns (j0 .. j0 + NC - 1), N % NC == 0 so that they are all in or all out; otherwise NC columns 64 apart
        const int j0 = VEC ? (strip * 64 + lane) * NC : strip * 64 * NC + lane;
        constexpr int CSTEP = VEC ? 1 : 64;
        const bool live = j0 < N;                                 // the lane's FIRST column (VEC: all of them)
        f3 pj[NC][NP];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float* sj = xb + (size_t)min(j0 + c * CSTEP, N - 1) * (size_t)A * 3;   // clamped: dead columns are never stored
#pragma unroll
            for (int k = 0; k < NP; ++k) pj[c][k] = ((SRC >> k) & 1) ? load3(sj + sel.atom[k] * 3) : mk3(0.f, 0.f, 0.f);
        }
        if (NPI > 0 && (b != staged_b || r_lo != staged_lo || r_hi != staged_hi)) {
            // a row per thread, its 3 * NPI loads in flight together (an element per thread walked three dependent round
            // trips to L2 per 512-row segment)
            float* rb = reinterpret_cast<float*>(k3_rowbuf);
            for (int row = (int)threadIdx.x; row < r_hi - r_lo; row += (int)blockDim.x) {
                const float* pr = xb + (size_t)(r_lo + row) * (size_t)A * 3;
                float v[NPIq * 3];
#pragma unroll
                for (int q = 0; q < NPI; ++q) {
                    v[q * 3] = pr[amap[q] * 3]; v[q * 3 + 1] = pr[amap[q] * 3 + 1]; v[q * 3 + 2] = pr[amap[q] * 3 + 2];
                }
#pragma unroll
                for (int qc = 0; qc < NPI * 3; ++qc) rb[(((row >> 1) * (NPI * 3) + qc) << 1) + (row & 1)] = v[qc];
            }
            staged_b = b; staged_lo = r_lo; staged_hi = r_hi;
        }
        if (threadIdx.x == 0) next_task = (unsigned)(c_lo + n_waves);   // the first n_waves tasks are pre-assigned
        __syncthreads();
        K3_STAMP(1);
        // the segment's rows as one buffer: uniform base, the lane's constant byte offset, the row's byte offset as a scalar
        float* obase = out + ((size_t)b * out_rows + (size_t)(r_lo - out_row_origin)) * N;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(obase, 0, K3_PROBE_RECORDS(0xFFFFFFFFu), 0x00020000u);
        const int lane_off = j0 * 4;
        auto rows = [&](int r, f3v (&p)[NP]) {                    // r = row pair index inside the segment
            int q = 0;
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                if ((SRC >> k) & 1) {
                    p[k] = mk3v(mk3(0.f, 0.f, 0.f), mk3(0.f, 0.f, 0.f));
                } else {
                    p[k] = f3v{k3_rowbuf[(r * NPI + q) * 3 + 0], k3_rowbuf[(r * NPI + q) * 3 + 1], k3_rowbuf[(r * NPI + q) * 3 + 2]};
                    ++q;
                }
            }
        };
        int c = c_lo + wave;
        while (c < c_hi) {
            const int i0 = (c - c_lo) * CH;                       // rows relative to r_lo (CH is even: pairs stay aligned)
            const int i1 = min(i0 + CH, r_hi - r_lo);
            // the row pairs of the task (K3_SWEEP_ROWS, defined in front of the kernel; a macro rather than a generic lambda: inside
            // a lambda the register allocator spilled 1-30 registers in four instantiations).  !VEC: a strip may have fewer than NC
            // live column groups (64 columns each; the last strip of a row) -- the chains of the dead ones are not evaluated.
            if constexpr (VEC) {
                K3_SWEEP_ROWS(NC)
            } else {
                // (not three of four: that variant spilled 3-5 registers in some instantiations)
                constexpr bool SKIP = true;
                const int ncl = SKIP ? min(NC, (N - strip * 64 * NC + 63) >> 6) : NC;   // live column groups of this strip (uniform)
                if (!SKIP || ncl == NC || ncl == 3) K3_SWEEP_ROWS(NC)
                else if (NC == 4 && ncl == 2) K3_SWEEP_ROWS((NC == 4 ? 2 : 1))
                else K3_SWEEP_ROWS(1)
            }
#ifdef PS_K3_AB
            if (c == c_lo + wave) { K3_STAMP(2); }
#endif
            unsigned nx = 0;
            if (lane == 0) nx = atomicAdd(&next_task, 1u);
            c = __builtin_amdgcn_readfirstlane((int)nx);
        }
        K3_STAMP(3);
    }
}

#undef K3_SWEEP_ROWS

// Fused trRosetta featuriser (reference protstruc.py:790-817): the three atom-pair planes of K1 that
// inter_residue_geometry slices out (CA-CA, CB-CB, N-O), their masks, and the three K3 features, in
// one sweep -- 27 bytes written per residue pair instead of 1125.  Same lane layout as K3.
template <bool EXACT, bool FAITHFUL>
__global__ __launch_bounds__(256) void k3_inter_residue_geometry(
    const float* __restrict__ xyz, const uint8_t* __restrict__ amask, float* __restrict__ d_ca,
    float* __restrict__ d_cb, float* __restrict__ d_no, float* __restrict__ omega, float* __restrict__ theta,
    float* __restrict__ phi, uint8_t* __restrict__ m_ca, uint8_t* __restrict__ m_cb, uint8_t* __restrict__ m_no, int N,
    int A, int IR, int n_tiles, int n_chunks) {
    const unsigned w = blockIdx.x;   // 1-D grid as in k3_pairwise_angles
    const unsigned tile = w % (unsigned)n_tiles, rest = w / (unsigned)n_tiles;
    const int b = (int)(rest / (unsigned)n_chunks);
    const int j = (int)tile * (int)blockDim.x + threadIdx.x;   // a workgroup is 64, 128, 192 or 256 lanes wide (short chains)
    const int i0 = (int)(rest % (unsigned)n_chunks) * IR, i1 = min(i0 + IR, N);
    const bool live = j < N;
    const int jc = live ? j : N - 1;
    const float* sj = xyz + ((size_t)b * N + jc) * (size_t)A * 3;
    const f3 ca_j = load3(sj + 3), o_j = load3(sj + 9), cb_j = load3(sj + 12);
    uint8_t mj_ca = 1, mj_o = 1, mj_cb = 1;
    if (amask) {
        const uint8_t* mj = amask + ((size_t)b * N + jc) * A;
        mj_ca = mj[1] != 0; mj_o = mj[3] != 0; mj_cb = mj[4] != 0;
    }
    auto row_scalars = [&](int i, f3& n_i, f3& ca_i, f3& cb_i, uint8_t& mi_n, uint8_t& mi_ca, uint8_t& mi_cb) {
        const float* si = xyz + ((size_t)b * N + i) * (size_t)A * 3;  // wave-uniform
        n_i = load3(si); ca_i = load3(si + 3); cb_i = load3(si + 12);
        mi_n = mi_ca = mi_cb = 1;
        if (amask) {
            const uint8_t* mi = amask + ((size_t)b * N + i) * A;
            mi_n = mi[0] != 0; mi_ca = mi[1] != 0; mi_cb = mi[4] != 0;
        }
    };
    auto row_planes = [&](size_t o, f3 n_i, f3 ca_i, f3 cb_i, uint8_t mi_n, uint8_t mi_ca, uint8_t mi_cb) {
        d_ca[o] = dist3_t<EXACT>(ca_i, ca_j);
        d_cb[o] = dist3_t<EXACT>(cb_i, cb_j);
        d_no[o] = dist3_t<EXACT>(n_i, o_j);
        m_ca[o] = mi_ca & mj_ca;
        m_cb[o] = mi_cb & mj_cb;
        m_no[o] = mi_n & mj_o;
        phi[o] = FAITHFUL ? angle3_ref(ca_i, cb_i, cb_j) : angle3(ca_i, cb_i, cb_j);
    };
    int i = i0;
    for (; !FAITHFUL && i + 1 < i1; i += 2) {  // two rows per trip: distances, planar angle and both dihedrals as packed float2 math
        f3 n0, ca0, cb0, n1, ca1, cb1;
        uint8_t a0, b0, c0, a1, b1, c1;
        row_scalars(i, n0, ca0, cb0, a0, b0, c0);
        row_scalars(i + 1, n1, ca1, cb1, a1, b1, c1);
        if (!live) continue;
        const size_t o = ((size_t)b * N + i) * N + j;
        const f3v cav = mk3v(ca0, ca1), cbv = mk3v(cb0, cb1), nv = mk3v(n0, n1);
        const f3v cajv = mk3v(ca_j, ca_j), cbjv = mk3v(cb_j, cb_j), ojv = mk3v(o_j, o_j);
        const f32x2 dca = dist3v_t<EXACT>(cav, cajv), dcb = dist3v_t<EXACT>(cbv, cbjv), dno = dist3v_t<EXACT>(nv, ojv);
        const f32x2 ph = angle3v(cav, cbv, cbjv);
        const f32x2 om = dihedral4v_k3(cav, cbv, cajv, cbjv);   // as coded at protstruc.py:811
        const f32x2 th = dihedral4v_k3(nv, cav, cbv, cbjv);
        d_ca[o] = dca.x; d_ca[o + N] = dca.y;
        d_cb[o] = dcb.x; d_cb[o + N] = dcb.y;
        d_no[o] = dno.x; d_no[o + N] = dno.y;
        m_ca[o] = b0 & mj_ca; m_ca[o + N] = b1 & mj_ca;
        m_cb[o] = c0 & mj_cb; m_cb[o + N] = c1 & mj_cb;
        m_no[o] = a0 & mj_o;  m_no[o + N] = a1 & mj_o;
        phi[o] = ph.x; phi[o + N] = ph.y;
        omega[o] = om.x; omega[o + N] = om.y;
        theta[o] = th.x; theta[o + N] = th.y;
    }
    for (; i < i1; ++i) {
        f3 n_i, ca_i, cb_i;
        uint8_t mi_n, mi_ca, mi_cb;
        row_scalars(i, n_i, ca_i, cb_i, mi_n, mi_ca, mi_cb);
        if (!live) continue;
        const size_t o = ((size_t)b * N + i) * N + j;
        row_planes(o, n_i, ca_i, cb_i, mi_n, mi_ca, mi_cb);
        omega[o] = FAITHFUL ? dihedral4_ref(ca_i, cb_i, ca_j, cb_j) : dihedral4_k3(ca_i, cb_i, ca_j, cb_j);   // as coded at protstruc.py:811
        theta[o] = FAITHFUL ? dihedral4_ref(n_i, ca_i, cb_i, cb_j) : dihedral4_k3(n_i, ca_i, cb_i, cb_j);
    }
}

// The featuriser on the skeleton of k3_sweep (round 4): one workgroup per CU (1024 threads at NC = 2, 512 at NC = 4), the row
// residues' N / CA / CB points and their three mask bits staged in LDS (pair-interleaved), tasks pulled from an LDS
// counter, and the two dihedrals' and the planar angle's chains interleaved across the NC columns of a lane.  Same
// arithmetic per pair as k3_inter_residue_geometry above: same bits.
//   * Small tasks, in memory order.  The column points (CA, O, CB: staged once per structure, 36 bytes per residue) come from
//     LDS per task and strip.  The first cut gave a workgroup one (structure, strip) segment at a time, 8-row tasks and the
//     column points in registers for the whole segment.  That is only good when a workgroup's share divides into whole
//     segments (B = 128, N = 512: half a structure each): the waves of a workgroup wait for each other at every segment
//     boundary for up to one task, and a share cut into a short and a long segment (any other B, N) idled them for a fifth
//     of the time; and its staging loops walked ~13 dependent round trips to L2 per segment (an element per thread; now a
//     residue per thread with its nine loads in flight: N = 512 178 -> 168 us).  N = 500 284 us, N = 510 415, N = 200 344
//     against 211 at N = 512, 2^25 pairs each (profiles/r04_k3_featuriser_shapes.log).
//   * VEC: the lane's NC columns are consecutive and every float plane goes out in 4 * NC-byte stores (N % NC == 0, planes
//     4 * NC-byte aligned).  !VEC (any N, any 4-byte alignment): the columns are 64 apart and a store instruction writes 64
//     consecutive floats of a row, as in k3_sweep.
//   * M16 (VEC, N % 16 == 0, 16-byte aligned mask planes): the mask planes have their own lane map inside a strip:
//     a lane owns 16 consecutive column residues of one row, a store instruction carries four rows x 256 bytes (NC = 2:
//     up to eight rows x 128).
//     !M16: the mask planes are written FLAT.  The mask bytes of a task's rows are one contiguous run of every plane
//     (rows are contiguous in memory), whatever N is; the run is cut on the plane's absolute 16-byte grid -- a group that
//     starts in the task's rows belongs to the task, also where it runs into the next task's first row -- and a lane
//     builds 16 bytes at a time from bit sets: 16 bits of the structure's column mask (two LDS words, one v_alignbit) AND
//     the row's bit, the next row's where the group wraps, spread to bytes: a store instruction writes 1 KB.  The task of every
//     fourth chunk's first strip writes the masks of 16 rows (full lanes, one set-up).  Only at a structure's two ends are
//     the bytes outside whole groups written as bytes.
//   * WT: write-through stores (sc1) where every store covers whole 128-byte lines and strips are whole (N % 128 == 0);
//     write-back otherwise, so that lines shared by two stores merge in L2 instead of going out as two partial writes.
template <bool EXACT, int NC, bool VEC, bool M16, bool WT, bool FAITHFUL = false>
__global__ __launch_bounds__((NC == 4 || FAITHFUL) ? 512 : 1024) void k3_featurise(
    const float* __restrict__ xyz, const uint8_t* __restrict__ amask, float* __restrict__ d_ca,
    float* __restrict__ d_cb, float* __restrict__ d_no, float* __restrict__ omega, float* __restrict__ theta,
    float* __restrict__ phi, uint8_t* __restrict__ m_ca, uint8_t* __restrict__ m_cb, uint8_t* __restrict__ m_no, int N,
    int A, int CH, int n_strips, int n_chunks, unsigned n_tasks, unsigned tasks_per_wg, unsigned rcpN, int KS, int slot_bytes, int pull) {
    static_assert(NC == 2 || NC == 4, "columns per lane");
    static_assert(!M16 || VEC, "16-byte strip mask stores ride on the vector kernels");
    constexpr int POL = WT ? 16 : 0;
    extern __shared__ __attribute__((aligned(16))) f32x2 k3_rowbuf[];   // [row pair][N, CA, CB][xyz]; then the row mask words, the column points, the column masks
    __shared__ unsigned next_task;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int n_waves = (int)(blockDim.x >> 6);
    const unsigned t0 = blockIdx.x * tasks_per_wg, t1 = min(t0 + tasks_per_wg, n_tasks);
    if (t0 >= t1) return;                         // whole workgroup
    const int max_pairs = (N + 2) / 2, Np = (N + 3) & ~3;
    // one structure's slot: [row pair][N, CA, CB][xyz] as f32x2; the row mask words (per row pair: byte 0 = row 2r, byte 1 = row
    // 2r + 1; bits n = 1, ca = 2, cb = 4; 8 spare); the column points [CA, O, CB][xyz][Np]; the column masks -- M16:
    // colmask[plane][column] as bytes; !M16: colbits[plane][word] = bit sets (planes CA, CB, O; zero beyond N and two zero
    // words of padding), same region, 16-byte aligned.  KS slots of slot_bytes each (round 5: several structures per staging
    // pass -- short chains paid two barriers and a round trip to L2 per 4 096 pairs).
    const size_t colpts_off = ((size_t)max_pairs * (9 * 8 + 4) + 32 + 15) & ~(size_t)15;
    const int nw = (N + 31) / 32 + 2;
    const int row_bytes = N * 4;
    // tasks of one structure: (chunk of CH rows, strip), a chunk's strips adjacent.  (Tried: a chunk of two rows x ALL strips,
    // the wave walking the strips inside the row pair, so that every 64-byte segment two stores share is completed by one
    // wave within a trip: 3-7 % SLOWER than four-row tasks per strip, N = 500 267 against 259 us -- the column points then
    // come from LDS every trip.  Two-row tasks per strip are what helped: see the launcher.)
    const int tpc = n_strips;                                         // tasks per chunk
    const unsigned n_sub = (unsigned)n_chunks * (unsigned)tpc;
    const unsigned b_first = t0 / n_sub, b_last = (t1 - 1u) / n_sub;
    for (unsigned bs = b_first; bs <= b_last; bs += (unsigned)KS) {     // the structures of this workgroup's tasks, KS per staging pass
        const unsigned ks = min((unsigned)KS, b_last - bs + 1u);
        __syncthreads();                                          // the previous pass's readers are done
        // a team of tw waves per structure (all of them when the pass is one structure; one wave each from n_waves structures on)
        const int tw = max(1, n_waves / (int)ks), n_teams = n_waves / tw;
        const int team = wave / tw, tl = (wave - team * tw) * 64 + lane, tsz = tw * 64;   // (waves beyond n_teams * tw: no team)
        for (unsigned s = (unsigned)team; s < ks && team < n_teams; s += (unsigned)n_teams) {
            char* slot = reinterpret_cast<char*>(k3_rowbuf) + (size_t)s * slot_bytes;
            uint32_t* rowmask = reinterpret_cast<uint32_t*>(slot + (size_t)max_pairs * 72);
            float* colpts = reinterpret_cast<float*>(slot + colpts_off);
            uint8_t* colmask = reinterpret_cast<uint8_t*>(colpts + 9 * Np);
            uint32_t* colbits = reinterpret_cast<uint32_t*>(colmask);
            const float* xb = xyz + (size_t)(bs + s) * N * (size_t)A * 3;   // uniform
            const uint8_t* mb = amask ? amask + (size_t)(bs + s) * N * A : nullptr;
            // one residue per thread, its loads in flight together (an element per thread made every workgroup walk
            // 9 N / 512 dependent round trips to L2 per structure: ~10 us of a 170 us launch): rows N, CA, CB; columns CA, O, CB
            float* rb = reinterpret_cast<float*>(slot);
            for (int row = tl; row < Np; row += tsz) {
                const float* pr = xb + (size_t)min(row, N - 1) * (size_t)A * 3;   // (the padding repeats the last residue)
                const float v[15] = {pr[0], pr[1], pr[2], pr[3], pr[4], pr[5], pr[12], pr[13], pr[14], pr[9], pr[10], pr[11]};
                if (row < N) {
#pragma unroll
                    for (int qc = 0; qc < 9; ++qc) rb[(((row >> 1) * 9 + qc) << 1) + (row & 1)] = v[qc];
                }
                colpts[0 * Np + row] = v[3]; colpts[1 * Np + row] = v[4]; colpts[2 * Np + row] = v[5];       // CA
                colpts[3 * Np + row] = v[9]; colpts[4 * Np + row] = v[10]; colpts[5 * Np + row] = v[11];     // O
                colpts[6 * Np + row] = v[6]; colpts[7 * Np + row] = v[7]; colpts[8 * Np + row] = v[8];       // CB
            }
            const int n_pairs = N / 2 + 1 + (M16 ? 0 : (16 - CH) / 2);   // !M16: up to 16 - CH rows and one beyond the tasks' own (flat mask groups)
            for (int r = tl; r < n_pairs; r += tsz) {
                uint32_t w = 0x0707u;
                if (mb) {
                    const int ia = min(2 * r, N - 1), ib = min(ia + 1, N - 1);
                    const uint8_t* pa = mb + (size_t)ia * A;
                    const uint8_t* pb = mb + (size_t)ib * A;
                    w = (pa[0] != 0 ? 1u : 0u) | (pa[1] != 0 ? 2u : 0u) | (pa[4] != 0 ? 4u : 0u) |
                        (pb[0] != 0 ? 0x100u : 0u) | (pb[1] != 0 ? 0x200u : 0u) | (pb[4] != 0 ? 0x400u : 0u);
                }
                rowmask[r] = w;
            }
            if constexpr (M16) {
                for (int j = tl; j < N; j += tsz) {
                    uint8_t a = 1, c2 = 1, o = 1;
                    if (mb) {
                        const uint8_t* mj = mb + (size_t)j * A;
                        a = mj[1] != 0; c2 = mj[4] != 0; o = mj[3] != 0;
                    }
                    colmask[j] = a; colmask[N + j] = c2; colmask[2 * N + j] = o;
                }
            } else {
                for (int base = (wave - team * tw) * 64; base < nw * 32; base += tsz) {   // (uniform per wave)
                    const int j = base + lane;
                    bool a = false, c2 = false, o = false;
                    if (j < N) {
                        a = c2 = o = true;
                        if (mb) {
                            const uint8_t* mj = mb + (size_t)j * A;
                            a = mj[1] != 0; c2 = mj[4] != 0; o = mj[3] != 0;
                        }
                    }
                    const unsigned long long ba = __ballot(a), bc = __ballot(c2), bo = __ballot(o);
                    if (lane == 0) {
                        const int k = base >> 5;
                        colbits[k] = (uint32_t)ba; colbits[nw + k] = (uint32_t)bc; colbits[2 * nw + k] = (uint32_t)bo;
                        if (k + 1 < nw) {
                            colbits[k + 1] = (uint32_t)(ba >> 32); colbits[nw + k + 1] = (uint32_t)(bc >> 32);
                            colbits[2 * nw + k + 1] = (uint32_t)(bo >> 32);
                        }
                    }
                }
            }
        }
        const unsigned seg_t0 = max(t0, bs * n_sub), seg_t1 = min(t1, (bs + ks) * n_sub);
        if (threadIdx.x == 0) next_task = seg_t0 + (unsigned)(n_waves * pull);   // the first n_waves pulls are pre-assigned
        __syncthreads();
        constexpr int GS = 4 * NC, RS = 64 / GS;                  // M16: 16-column groups of a strip, rows of a store instruction (4 / 8)
        const int gq = lane % GS, rq = lane / GS;                 // ... this lane's group and row
        // a wave takes `pull` consecutive tasks at a time (1 in the product; round 5 tried 2 and 4 at rows that are not whole 64-byte
        // segments, so that the segments adjacent tasks share are completed by one wave: NOTES.md)
        unsigned t = seg_t0 + (unsigned)(wave * pull), t_end = min(t + (unsigned)pull, seg_t1);
        while (t < seg_t1) {
            // the task's structure (uniform): its slot in LDS, and per plane the structure's rows as one buffer (uniform base, the
            // lane's constant byte offset, the row's byte offset as a scalar)
            const unsigned b = t / n_sub;
            const int c = (int)(t - b * n_sub);
            const char* slot = reinterpret_cast<const char*>(k3_rowbuf) + (size_t)(b - bs) * slot_bytes;
            const f32x2* rowbuf = reinterpret_cast<const f32x2*>(slot);
            const uint32_t* rowmask = reinterpret_cast<const uint32_t*>(slot + (size_t)max_pairs * 72);
            const float* colpts = reinterpret_cast<const float*>(slot + colpts_off);
            const uint8_t* colmask = reinterpret_cast<const uint8_t*>(colpts + 9 * Np);
            const uint32_t* colbits = reinterpret_cast<const uint32_t*>(colmask);
            constexpr int r_lo = 0;
            const int r_hi = N;
            const size_t seg = (size_t)b * N * N, sbase = seg;
            auto rs = [&](void* base) { return __builtin_amdgcn_make_buffer_rsrc(base, 0, K3_PROBE_RECORDS(0xFFFFFFFFu), 0x00020000u); };
            const __amdgpu_buffer_rsrc_t r_dca = rs(d_ca + seg), r_dcb = rs(d_cb + seg), r_dno = rs(d_no + seg), r_om = rs(omega + seg),
                                         r_th = rs(theta + seg), r_ph = rs(phi + seg);
            const __amdgpu_buffer_rsrc_t r_mca = rs(m_ca + sbase), r_mcb = rs(m_cb + sbase), r_mno = rs(m_no + sbase);
            const int chunk = c / tpc, strip0 = c - chunk * tpc;
            const int i0 = chunk * CH - r_lo;                     // rows relative to r_lo (CH is even: pairs stay aligned)
            const int i1 = min(i0 + CH, r_hi - r_lo);
            // !M16: the mask bytes of 16 rows (several chunks: full lanes, and the set-up once per 16 rows), flat, plane by plane
            // (each plane has its own 16-byte grid), by the task of their first chunk's first strip.  What a task owns must not
            // depend on the workgroup that runs it: the rows' mask words are staged up to 16 rows beyond the workgroup's own rows.
            if (!M16 && strip0 == 0 && (chunk * CH) % 16 == 0) {
                const int R0 = chunk * CH, R1 = min(R0 + 16, N);
                const unsigned rel0 = (unsigned)R0 * (unsigned)N, rel1 = (unsigned)R1 * (unsigned)N;   // bytes from the structure's first
                const bool first = chunk == 0, more = R1 < N;     // the structure's first / not its last rows
                auto row_word = [&](unsigned i) {                 // the three mask bits of absolute row i (staged: r_lo <= i <= r_hi)
                    const unsigned ri = i - (unsigned)r_lo;
                    return rowmask[ri >> 1] >> (8u * (ri & 1u));
                };
                // one 16-byte group of one plane: 16 bits of the column mask from column j on (zero beyond N), AND the row's
                // bit; where the group runs into row i + 1 (N >= 64: at most once), that row's first columns
                auto group = [&](const uint32_t* W, uint32_t bit, uint32_t w0, uint32_t w1, unsigned j, bool wrap) {
                    const unsigned k = j >> 5, sh = j & 31u;
                    uint32_t x = (w0 & bit) ? (__builtin_amdgcn_alignbit(W[k + 1], W[k], sh) & 0xFFFFu) : 0u;
                    if (wrap && (w1 & bit)) x |= (W[0] << ((unsigned)N - j)) & 0xFFFFu;
                    return k3_u32x4{((x & 15u) * 0x00204081u) & 0x01010101u, (((x >> 4) & 15u) * 0x00204081u) & 0x01010101u,
                                    (((x >> 8) & 15u) * 0x00204081u) & 0x01010101u, (((x >> 12) & 15u) * 0x00204081u) & 0x01010101u};
                };
                auto span = [&](const uint8_t* first_byte, unsigned& bs, unsigned& be) {   // the whole groups that start in these rows
                    const unsigned sb16 = (unsigned)(reinterpret_cast<uintptr_t>(first_byte) & 15u);   // the structure's first byte on the grid
                    bs = rel0 + ((16u - ((sb16 + rel0) & 15u)) & 15u);
                    be = more ? rel1 + ((16u - ((sb16 + rel1) & 15u)) & 15u) : rel1 - ((sb16 + rel1) & 15u);
                };
                auto ends = [&](const __amdgpu_buffer_rsrc_t& rsrc, const uint32_t* W, uint32_t bit, unsigned bs, unsigned be) {
                    if ((first && bs > rel0) || (!more && be < rel1)) {   // (uniform) the structure's ends: bytes outside whole groups
                        const bool head = lane < 16;
                        const unsigned f = head ? rel0 + (unsigned)lane : be + (unsigned)(lane - 16);
                        if (lane < 32 && (head ? (first && f < bs) : (!more && f < rel1))) {
                            unsigned i = __umulhi(f, rcpN), j = f - i * (unsigned)N;
                            if (j >= (unsigned)N) ++i, j -= (unsigned)N;
                            const uint8_t v = (row_word(i) & bit) ? (uint8_t)((W[j >> 5] >> (j & 31u)) & 1u) : (uint8_t)0;
                            __builtin_amdgcn_raw_buffer_store_b8(v, rsrc, (int)f, 0, POL);
                        }
                    }
                };
                unsigned bs_ca, be_ca, bs_cb, be_cb, bs_no, be_no;
                span(m_ca + sbase, bs_ca, be_ca); span(m_cb + sbase, bs_cb, be_cb); span(m_no + sbase, bs_no, be_no);
                if (bs_ca == bs_cb && bs_ca == bs_no) {           // (uniform) one grid for the three planes: one decode per group
                    for (unsigned f = bs_ca + 16u * (unsigned)lane; f < be_ca; f += 16u * 64u) {   // a store instruction writes 1 KB
                        unsigned i = __umulhi(f, rcpN), j = f - i * (unsigned)N;
                        if (j >= (unsigned)N) ++i, j -= (unsigned)N;
                        const bool wrap = j + 16u > (unsigned)N;
                        const uint32_t w0 = row_word(i), w1 = wrap ? row_word(i + 1u) : 0u;
                        k3_u32x4 g1 = group(colbits, 2u, w0, w1, j, wrap), g2 = group(colbits + nw, 4u, w0, w1, j, wrap), g3 = group(colbits + 2 * nw, 1u, w0, w1, j, wrap);
                        __builtin_amdgcn_raw_buffer_store_b128(g1, r_mca, (int)f, 0, POL);
                        __builtin_amdgcn_raw_buffer_store_b128(g2, r_mcb, (int)f, 0, POL);
                        __builtin_amdgcn_raw_buffer_store_b128(g3, r_mno, (int)f, 0, POL);
                    }
                } else {
                    auto plane = [&](const __amdgpu_buffer_rsrc_t& rsrc, const uint32_t* W, uint32_t bit, unsigned bs, unsigned be) {
                        for (unsigned f = bs + 16u * (unsigned)lane; f < be; f += 16u * 64u) {
                            unsigned i = __umulhi(f, rcpN), j = f - i * (unsigned)N;
                            if (j >= (unsigned)N) ++i, j -= (unsigned)N;
                            const bool wrap = j + 16u > (unsigned)N;
                            __builtin_amdgcn_raw_buffer_store_b128(group(W, bit, row_word(i), wrap ? row_word(i + 1u) : 0u, j, wrap), rsrc, (int)f, 0, POL);
                        }
                    };
                    plane(r_mca, colbits, 2u, bs_ca, be_ca);
                    plane(r_mcb, colbits + nw, 4u, bs_cb, be_cb);
                    plane(r_mno, colbits + 2 * nw, 1u, bs_no, be_no);
                }
                ends(r_mca, colbits, 2u, bs_ca, be_ca);
                ends(r_mcb, colbits + nw, 4u, bs_cb, be_cb);
                ends(r_mno, colbits + 2 * nw, 1u, bs_no, be_no);
            }
            {
                const int strip = strip0;
                // this lane's NC columns of the strip
                const int jbase = VEC ? (strip * 64 + lane) * NC : strip * 64 * NC + lane;
                bool lv[NC];
                f3 ca_j[NC], o_j[NC], cb_j[NC];
                if constexpr (VEC) {                              // consecutive columns: one LDS read per coordinate
                    const int jr = min(jbase, Np - NC);
                    typedef typename std::conditional<NC == 4, k3_f32x4, f32x2>::type cvec;
                    cvec t[9];
#pragma unroll
                    for (int qc = 0; qc < 9; ++qc) t[qc] = *reinterpret_cast<const cvec*>(colpts + qc * Np + jr);
#pragma unroll
                    for (int cc = 0; cc < NC; ++cc) {
                        lv[cc] = jbase < N;                       // N % NC == 0: a lane's columns are all in or all out
                        ca_j[cc] = f3{t[0][cc], t[1][cc], t[2][cc]};
                        o_j[cc] = f3{t[3][cc], t[4][cc], t[5][cc]};
                        cb_j[cc] = f3{t[6][cc], t[7][cc], t[8][cc]};
                    }
                } else {
#pragma unroll
                    for (int cc = 0; cc < NC; ++cc) {
                        const int j = jbase + 64 * cc, jr = min(j, N - 1);
                        lv[cc] = j < N;
                        ca_j[cc] = f3{colpts[jr], colpts[Np + jr], colpts[2 * Np + jr]};
                        o_j[cc] = f3{colpts[3 * Np + jr], colpts[4 * Np + jr], colpts[5 * Np + jr]};
                        cb_j[cc] = f3{colpts[6 * Np + jr], colpts[7 * Np + jr], colpts[8 * Np + jr]};
                    }
                }
                const bool live = lv[0];
                const int lane_off = jbase * 4;
                if constexpr (M16) {                              // the strip's mask planes: four (NC = 2: eight) rows per store instruction
                    const int jg = strip * 64 * NC + 16 * gq;
                    if (jg < N) {                                 // N % 16 == 0: a group is in or out
                        const k3_u32x4 cm_ca = *reinterpret_cast<const k3_u32x4*>(colmask + jg);
                        const k3_u32x4 cm_cb = *reinterpret_cast<const k3_u32x4*>(colmask + N + jg);
                        const k3_u32x4 cm_o = *reinterpret_cast<const k3_u32x4*>(colmask + 2 * N + jg);
                        for (int k4 = i0; k4 < i1; k4 += RS) {
                            const int rr = k4 + rq;
                            if (rr < i1) {
                                const uint32_t w = rowmask[rr >> 1] >> (8 * (rr & 1));
                                const int vo = rr * N + jg;
                                const k3_u32x4 z = {0, 0, 0, 0};
                                __builtin_amdgcn_raw_buffer_store_b128((w & 2u) ? cm_ca : z, r_mca, vo, 0, POL);
                                __builtin_amdgcn_raw_buffer_store_b128((w & 4u) ? cm_cb : z, r_mcb, vo, 0, POL);
                                __builtin_amdgcn_raw_buffer_store_b128((w & 1u) ? cm_o : z, r_mno, vo, 0, POL);
                            }
                        }
                    }
                }
                // !VEC: the last strip of a row may have fewer than NC live column groups (64 columns each) -- the chains of
                // the dead ones are not evaluated (N = 129 at NC = 4: three of four; N = 300 at NC = 2: one of two)
                int i_from = i0;
                auto sweep_rows = [&](auto ncl_tag) {
                constexpr int L = decltype(ncl_tag)::value;
                for (int i = i_from; i < i1; i += 2) {
                    const int r = i >> 1;
                    const bool two = i + 1 < i1;                  // the last row of an odd N has no partner (uniform)
                    const f3v nv = {rowbuf[r * 9 + 0], rowbuf[r * 9 + 1], rowbuf[r * 9 + 2]};
                    const f3v cav = {rowbuf[r * 9 + 3], rowbuf[r * 9 + 4], rowbuf[r * 9 + 5]};
                    const f3v cbv = {rowbuf[r * 9 + 6], rowbuf[r * 9 + 7], rowbuf[r * 9 + 8]};
                    const int so = i * row_bytes;
                    // plane by plane, so that only one plane's results are live at a time; each plane's results are pinned
                    // before its store (see k3_sweep)
                    auto emit = [&](const __amdgpu_buffer_rsrc_t& rr, f32x2 (&v)[L]) {
#pragma unroll
                        for (int cc = 0; cc < L; ++cc) asm volatile("" : "+v"(v[cc]));
                        if constexpr (!VEC) {
#pragma unroll
                            for (int cc = 0; cc < L; ++cc)
                                if (lv[cc]) {
                                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[cc].x), rr, lane_off + 256 * cc, so, POL);
                                    if (two) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[cc].y), rr, lane_off + 256 * cc, so + row_bytes, POL);
                                }
                        } else if (live) {
                            if constexpr (NC == 4) {
                                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(k3_u32x4, k3_f32x4{v[0].x, v[1].x, v[2].x, v[3].x}), rr, lane_off, so, POL);
                                if (two) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(k3_u32x4, k3_f32x4{v[0].y, v[1].y, v[2].y, v[3].y}), rr, lane_off, so + row_bytes, POL);
                            } else {
                                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(k3_u32x2, f32x2{v[0].x, v[1].x}), rr, lane_off, so, POL);
                                if (two) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(k3_u32x2, f32x2{v[0].y, v[1].y}), rr, lane_off, so + row_bytes, POL);
                            }
                        }
                    };
                    f3v NV[L], CAV[L], CBV[L], CAJ[L], CBJ[L];
                    f32x2 v[L];
#pragma unroll
                    for (int cc = 0; cc < L; ++cc) {
                        NV[cc] = nv; CAV[cc] = cav; CBV[cc] = cbv;
                        CAJ[cc] = mk3v(ca_j[cc], ca_j[cc]); CBJ[cc] = mk3v(cb_j[cc], cb_j[cc]);
                    }
#pragma unroll
                    for (int cc = 0; cc < L; ++cc) v[cc] = dist3v_t<EXACT>(cav, CAJ[cc]);
                    emit(r_dca, v);
#pragma unroll
                    for (int cc = 0; cc < L; ++cc) v[cc] = dist3v_t<EXACT>(cbv, CBJ[cc]);
                    emit(r_dcb, v);
#pragma unroll
                    for (int cc = 0; cc < L; ++cc) v[cc] = dist3v_t<EXACT>(nv, mk3v(o_j[cc], o_j[cc]));
                    emit(r_dno, v);
                    if constexpr (FAITHFUL) angle3v_ref_n<L>(CAV, CBV, CBJ, v);
                    else angle3v_n<L>(CAV, CBV, CBJ, v);
                    emit(r_ph, v);
                    if constexpr (FAITHFUL) dihedral4v_ref_n<L>(CAV, CBV, CAJ, CBJ, v);      // as coded at protstruc.py:811
                    else dihedral4v_k3_n<L>(CAV, CBV, CAJ, CBJ, v);
                    emit(r_om, v);
                    if constexpr (FAITHFUL) dihedral4v_ref_n<L>(NV, CAV, CBV, CBJ, v);
                    else dihedral4v_k3_n<L>(NV, CAV, CBV, CBJ, v);
                    emit(r_th, v);
                }
                };
                // One live column group (chains of up to 64 residues in the 64-apart layout): a lane has a single column, so its four
                // interleaved chains are four consecutive ROW PAIRS instead of four columns (round 5; with one chain per lane the
                // dependent instructions of a chain followed each other with nothing in between)
                auto sweep_rows_by_row_pairs = [&]() {
                    constexpr int R = 4;
                    const f3v caj = mk3v(ca_j[0], ca_j[0]), cbj = mk3v(cb_j[0], cb_j[0]), oj = mk3v(o_j[0], o_j[0]);
                    int i = i0;
                    for (; i + 2 * R <= i1; i += 2 * R) {
                        f3v NV[R], CAV[R], CBV[R], CAJ[R], CBJ[R];
                        f32x2 v[R];
#pragma unroll
                        for (int q = 0; q < R; ++q) {
                            const int r = (i >> 1) + q;
                            NV[q] = f3v{rowbuf[r * 9 + 0], rowbuf[r * 9 + 1], rowbuf[r * 9 + 2]};
                            CAV[q] = f3v{rowbuf[r * 9 + 3], rowbuf[r * 9 + 4], rowbuf[r * 9 + 5]};
                            CBV[q] = f3v{rowbuf[r * 9 + 6], rowbuf[r * 9 + 7], rowbuf[r * 9 + 8]};
                            CAJ[q] = caj; CBJ[q] = cbj;
                        }
                        const int so = i * row_bytes;
                        auto emit = [&](const __amdgpu_buffer_rsrc_t& rr, f32x2 (&w)[R]) {
#pragma unroll
                            for (int q = 0; q < R; ++q) asm volatile("" : "+v"(w[q]));
                            if (lv[0]) {
#pragma unroll
                                for (int q = 0; q < R; ++q) {
                                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(w[q].x), rr, lane_off, so + 2 * q * row_bytes, POL);
                                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(w[q].y), rr, lane_off, so + (2 * q + 1) * row_bytes, POL);
                                }
                            }
                        };
#pragma unroll
                        for (int q = 0; q < R; ++q) v[q] = dist3v_t<EXACT>(CAV[q], caj);
                        emit(r_dca, v);
#pragma unroll
                        for (int q = 0; q < R; ++q) v[q] = dist3v_t<EXACT>(CBV[q], cbj);
                        emit(r_dcb, v);
#pragma unroll
                        for (int q = 0; q < R; ++q) v[q] = dist3v_t<EXACT>(NV[q], oj);
                        emit(r_dno, v);
                        if constexpr (FAITHFUL) angle3v_ref_n<R>(CAV, CBV, CBJ, v);
                        else angle3v_n<R>(CAV, CBV, CBJ, v);
                        emit(r_ph, v);
                        if constexpr (FAITHFUL) dihedral4v_ref_n<R>(CAV, CBV, CAJ, CBJ, v);      // as coded at protstruc.py:811
                        else dihedral4v_k3_n<R>(CAV, CBV, CAJ, CBJ, v);
                        emit(r_om, v);
                        if constexpr (FAITHFUL) dihedral4v_ref_n<R>(NV, CAV, CBV, CBJ, v);
                        else dihedral4v_k3_n<R>(NV, CAV, CBV, CBJ, v);
                        emit(r_th, v);
                    }
                    i_from = i;                                    // the task's remaining rows (fewer than eight): one pair at a time
                    if (i < i1) sweep_rows(std::integral_constant<int, 1>{});
                };
                if constexpr (VEC) {
                    sweep_rows(std::integral_constant<int, NC>{});
                } else {
                    const int ncl = min(NC, (N - strip * 64 * NC + 63) >> 6);   // live column groups of this strip (uniform)
                    if ((NC == 4 || FAITHFUL) && ncl == 1 && N <= 64) sweep_rows_by_row_pairs();   // (the instantiations with 256 VGPRs)
                    else if (ncl == NC) sweep_rows(std::integral_constant<int, NC>{});
                    else if (NC == 4 && ncl == 3) sweep_rows(std::integral_constant<int, NC == 4 ? 3 : 1>{});
                    else if (NC == 4 && ncl == 2) sweep_rows(std::integral_constant<int, NC == 4 ? 2 : 1>{});
                    else sweep_rows(std::integral_constant<int, 1>{});
                }
            }
            if (++t >= t_end) {
                unsigned nx = 0;
                if (lane == 0) nx = atomicAdd(&next_task, (unsigned)pull);
                t = (unsigned)__builtin_amdgcn_readfirstlane((int)nx);
                t_end = min(t + (unsigned)pull, seg_t1);
            }
        }
    }
}

// The featuriser on the TILE map of k3_flat (round 5): a lane's element is two row pairs x two adjacent columns of one structure
// (four interleaved chains per feature), the N / CA / CB row atoms, CA / O / CB column atoms and the six mask bits per residue of
// KS structures staged per pass, two workgroups per CU, tasks of 64 tiles pulled from an LDS counter.  Chains of up to a few hundred
// residues: every lane has work whatever N is, what depends on a row pair or a column alone is shared inside the tile, and
// the set-up of a structure is paid once per KS structures.  Float planes: 8-byte stores of the tile's two columns (N even, planes
// 8-byte aligned; else dword stores); mask planes: two bytes per row of the tile (N even, planes 2-byte aligned; else bytes).
// Same arithmetic per pair as k3_inter_residue_geometry: same bits.
template <bool EXACT, bool FAITHFUL>
__global__ __launch_bounds__(512) void k3_featurise_tiles(
    const float* __restrict__ xyz, const uint8_t* __restrict__ amask, float* __restrict__ d_ca,
    float* __restrict__ d_cb, float* __restrict__ d_no, float* __restrict__ omega, float* __restrict__ theta,
    float* __restrict__ phi, uint8_t* __restrict__ m_ca, uint8_t* __restrict__ m_cb, uint8_t* __restrict__ m_no, int N,
    int A, int KS, unsigned tps, unsigned n_tasks, unsigned tasks_per_wg, unsigned rcpN, unsigned rcpTC, int slot_vec4, int vecf,
    int vecm) {
    // [slot]: column atoms CA, O, CB as {x, y, z, -} per (atom, residue): 3 N vec4; row atoms N, CA, CB pair-interleaved
    // {x0, x1, y0, y1}, {z0, z1, -, -} per (row pair, atom): 6 RP vec4; then one byte per residue: bits 0..2 = row side
    // (N, CA, CB present), bits 4..6 = column side (CA, CB, O present)
    extern __shared__ __attribute__((aligned(16))) k3_f32x4 k3_tilebuf[];
    __shared__ unsigned next_task;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned n_waves = blockDim.x >> 6;
    const unsigned t0 = blockIdx.x * tasks_per_wg, t1 = min(t0 + tasks_per_wg, n_tasks);
    if (t0 >= t1) return;                         // whole workgroup
    const int n_rp = (N + 1) >> 1;
    const int col_vec4 = 3 * N, row_vec4 = 6 * n_rp;
    // tiles of two row pairs x two columns -- or four (vecf == 4: N % 4 == 0, 16-byte float rows, 4-byte mask rows)
    const bool wide = vecf == 4;                  // (uniform)
    const unsigned TC = wide ? (unsigned)N >> 2 : (unsigned)(N + 1) >> 1, TR = (unsigned)(n_rp + 1) >> 1, FT = TR * TC;
    const unsigned b_first = t0 / tps, b_last = (t1 - 1u) / tps;
    K3F_STAMP(0);
    [[maybe_unused]] int pass = 0;
    for (unsigned bs = b_first; bs <= b_last; bs += (unsigned)KS, ++pass) {
        const unsigned ks = min((unsigned)KS, b_last - bs + 1u);
        __syncthreads();                                          // the previous pass's readers are done
        K3F_STAMP(1 + 3 * pass);
        for (unsigned it = threadIdx.x; it < ks * (unsigned)N; it += blockDim.x) {   // one residue of one structure per thread
            unsigned sidx = __umulhi(it, rcpN), r = it - sidx * (unsigned)N;
            if (r >= (unsigned)N) ++sidx, r -= (unsigned)N;
            const float* pr = xyz + ((size_t)(bs + sidx) * N + r) * (size_t)A * 3;
            k3_f32x4* slot = k3_tilebuf + (size_t)sidx * slot_vec4;
            const float v[15] = {pr[0], pr[1], pr[2], pr[3], pr[4], pr[5], pr[9], pr[10], pr[11], pr[12], pr[13], pr[14]};   // N, CA, O, CB
            unsigned bits = 0x77u;
            if (amask) {
                const uint8_t* mr = amask + ((size_t)(bs + sidx) * N + r) * A;
                const unsigned mn = mr[0] != 0, mca = mr[1] != 0, mo = mr[3] != 0, mcb = mr[4] != 0;
                bits = mn | (mca << 1) | (mcb << 2) | (mca << 4) | (mcb << 5) | (mo << 6);
            }
            slot[0 * N + (int)r] = k3_f32x4{v[3], v[4], v[5], 0.0f};        // CA
            slot[1 * N + (int)r] = k3_f32x4{v[6], v[7], v[8], 0.0f};        // O
            slot[2 * N + (int)r] = k3_f32x4{v[9], v[10], v[11], 0.0f};      // CB
            float* rb = reinterpret_cast<float*>(slot + col_vec4) + (size_t)(r >> 1) * 24 + (r & 1);
            rb[0] = v[0]; rb[2] = v[1]; rb[4] = v[2];                       // N
            rb[8] = v[3]; rb[10] = v[4]; rb[12] = v[5];                     // CA
            rb[16] = v[9]; rb[18] = v[10]; rb[20] = v[11];                  // CB
            reinterpret_cast<uint8_t*>(slot + col_vec4 + row_vec4)[r] = (uint8_t)bits;
        }
        const unsigned seg_t0 = max(t0, bs * tps), seg_t1 = min(t1, (bs + ks) * tps);
        if (threadIdx.x == 0) next_task = seg_t0 + n_waves;       // the first n_waves tasks are pre-assigned
        __syncthreads();
        K3F_STAMP(2 + 3 * pass);
        unsigned t = seg_t0 + (unsigned)wave;
        while (t < seg_t1) {
            const unsigned b = t / tps, chunk = t - b * tps;      // (uniform)
            const k3_f32x4* slot = k3_tilebuf + (size_t)(b - bs) * slot_vec4;
            const k3_f32x4* rowp = slot + col_vec4;
            const uint8_t* mbits = reinterpret_cast<const uint8_t*>(slot + col_vec4 + row_vec4);
            const size_t sbase = (size_t)b * N * N;
            auto rs = [&](void* base, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)bytes, 0x00020000u); };
            const unsigned fbytes = (unsigned)N * (unsigned)N * 4u, mbytes = (unsigned)N * (unsigned)N;   // this structure's planes, exactly
            const __amdgpu_buffer_rsrc_t r_dca = rs(d_ca + sbase, fbytes), r_dcb = rs(d_cb + sbase, fbytes), r_dno = rs(d_no + sbase, fbytes),
                                         r_om = rs(omega + sbase, fbytes), r_th = rs(theta + sbase, fbytes), r_ph = rs(phi + sbase, fbytes);
            const __amdgpu_buffer_rsrc_t r_mca = rs(m_ca + sbase, mbytes), r_mcb = rs(m_cb + sbase, mbytes), r_mno = rs(m_no + sbase, mbytes);
            const unsigned ti = chunk * 64u + (unsigned)lane;
#ifdef PS_K3_AB
            const bool lt = ti < FT && k3f_probe != 2;
#else
            const bool lt = ti < FT;
#endif
            const unsigned tcl = min(ti, FT - 1u);
            unsigned tr = __umulhi(tcl, rcpTC), tc = tcl - tr * TC;
            if (tc >= TC) ++tr, tc -= TC;
            const int c0 = (int)(wide ? 4u * tc : 2u * tc);
            const int rpA = (int)(2u * tr), rpB = min(rpA + 1, n_rp - 1);
            const bool lc1 = c0 + 1 < N, lrB = rpA + 1 < n_rp;
            // rows 2 rpA (always there), 2 rpA + 1, 2 rpB, 2 rpB + 1
            const int iA0 = 2 * rpA, iA1 = min(iA0 + 1, N - 1), iB0 = 2 * rpB, iB1 = min(iB0 + 1, N - 1);
            const bool rA1 = iA0 + 1 < N, rB0 = lrB, rB1 = lrB && iB0 + 1 < N;
            // chain c of a half: row pair A (c = 0, 1) or B (2, 3) x the half's first / second column; .x / .y = the pair's rows
            f3v NV[4], CAV[4], CBV[4], CAJ[2][4], CBJ[2][4], OJ[2][4];
            {
                auto row_atom = [&](int rp, int q) {
                    const k3_f32x4 xy = rowp[(rp * 3 + q) * 2];
                    const f32x2 z = *reinterpret_cast<const f32x2*>(rowp + (rp * 3 + q) * 2 + 1);
                    return f3v{f32x2{xy.x, xy.y}, f32x2{xy.z, xy.w}, z};
                };
                NV[0] = NV[1] = row_atom(rpA, 0); CAV[0] = CAV[1] = row_atom(rpA, 1); CBV[0] = CBV[1] = row_atom(rpA, 2);
                NV[2] = NV[3] = row_atom(rpB, 0); CAV[2] = CAV[3] = row_atom(rpB, 1); CBV[2] = CBV[3] = row_atom(rpB, 2);
                auto cols = [&](int ca, f3v (&CA)[4], f3v (&CB)[4], f3v (&O)[4]) {
                    const int cb = min(ca + 1, N - 1);            // clamped: a dead column is never stored
                    const k3_f32x4 ca0 = slot[ca], ca1 = slot[cb], o0 = slot[N + ca], o1 = slot[N + cb], cb0 = slot[2 * N + ca], cb1 = slot[2 * N + cb];
                    CA[0] = CA[2] = mk3v(f3{ca0.x, ca0.y, ca0.z}, f3{ca0.x, ca0.y, ca0.z});
                    CA[1] = CA[3] = mk3v(f3{ca1.x, ca1.y, ca1.z}, f3{ca1.x, ca1.y, ca1.z});
                    O[0] = O[2] = mk3v(f3{o0.x, o0.y, o0.z}, f3{o0.x, o0.y, o0.z});
                    O[1] = O[3] = mk3v(f3{o1.x, o1.y, o1.z}, f3{o1.x, o1.y, o1.z});
                    CB[0] = CB[2] = mk3v(f3{cb0.x, cb0.y, cb0.z}, f3{cb0.x, cb0.y, cb0.z});
                    CB[1] = CB[3] = mk3v(f3{cb1.x, cb1.y, cb1.z}, f3{cb1.x, cb1.y, cb1.z});
                };
                cols(c0, CAJ[0], CBJ[0], OJ[0]);
                cols(wide ? c0 + 2 : c0, CAJ[1], CBJ[1], OJ[1]);   // (the second half exists only in a four-column tile)
            }
            constexpr int DEAD = 0x7FFFFFF0;    // beyond num_records: dropped by the range check
            const int offA = (int)__umul24((unsigned)iA0, (unsigned)N) + c0, offB = (int)__umul24((unsigned)iB0, (unsigned)N) + c0;   // in elements
            const bool l1 = lt && lc1;
            // one plane's chains: v = the tile's first two columns, w = its third and fourth (four-column tiles only)
            auto emit = [&](const __amdgpu_buffer_rsrc_t& rr, f32x2 (&v)[4], f32x2 (&w)[4]) {
#pragma unroll
                for (int c = 0; c < 4; ++c) asm volatile("" : "+v"(v[c]));
                if (wide) {      // (uniform) a row of the tile is one 16-byte store
#pragma unroll
                    for (int c = 0; c < 4; ++c) asm volatile("" : "+v"(w[c]));
                    auto st4 = [&](float a, float b, float c, float d, int off) {
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(k3_u32x4, k3_f32x4{a, b, c, d}), rr, off, 0, 0);
                    };
                    st4(v[0].x, v[1].x, w[0].x, w[1].x, lt ? offA * 4 : DEAD);
                    st4(v[0].y, v[1].y, w[0].y, w[1].y, (lt && rA1) ? (offA + N) * 4 : DEAD);
                    st4(v[2].x, v[3].x, w[2].x, w[3].x, (lt && rB0) ? offB * 4 : DEAD);
                    st4(v[2].y, v[3].y, w[2].y, w[3].y, (lt && rB1) ? (offB + N) * 4 : DEAD);
                } else if (vecf == 2) {
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(k3_u32x2, f32x2{v[0].x, v[1].x}), rr, lt ? offA * 4 : DEAD, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(k3_u32x2, f32x2{v[0].y, v[1].y}), rr, (lt && rA1) ? (offA + N) * 4 : DEAD, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(k3_u32x2, f32x2{v[2].x, v[3].x}), rr, (lt && rB0) ? offB * 4 : DEAD, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(k3_u32x2, f32x2{v[2].y, v[3].y}), rr, (lt && rB1) ? (offB + N) * 4 : DEAD, 0, 0);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[0].x), rr, lt ? offA * 4 : DEAD, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[1].x), rr, l1 ? offA * 4 + 4 : DEAD, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[0].y), rr, (lt && rA1) ? (offA + N) * 4 : DEAD, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[1].y), rr, (l1 && rA1) ? (offA + N) * 4 + 4 : DEAD, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[2].x), rr, (lt && rB0) ? offB * 4 : DEAD, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[3].x), rr, (l1 && rB0) ? offB * 4 + 4 : DEAD, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[2].y), rr, (lt && rB1) ? (offB + N) * 4 : DEAD, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[3].y), rr, (l1 && rB1) ? (offB + N) * 4 + 4 : DEAD, 0, 0);
                }
            };
            // the three mask planes first (their stores drain while the arithmetic runs): rows' bits 0..2 = N, CA, CB; columns' bits 4..6 = CA, CB, O
            {
                const unsigned mA0 = mbits[iA0], mA1 = mbits[iA1], mB0 = mbits[iB0], mB1 = mbits[iB1];
                const unsigned mc0 = mbits[c0] >> 4, mc1 = mbits[min(c0 + 1, N - 1)] >> 4;
                const unsigned mc2 = mbits[min(c0 + 2, N - 1)] >> 4, mc3 = mbits[min(c0 + 3, N - 1)] >> 4;   // (four-column tiles)
                auto plane = [&](const __amdgpu_buffer_rsrc_t& rr, unsigned rbit, unsigned cbit) {
                    const unsigned k0 = (mc0 >> cbit) & 1u, k1 = (mc1 >> cbit) & 1u;
                    const unsigned a0 = (mA0 >> rbit) & 1u, a1 = (mA1 >> rbit) & 1u, b0 = (mB0 >> rbit) & 1u, b1 = (mB1 >> rbit) & 1u;
                    if (wide) {      // (uniform) four bytes per row of the tile
                        const unsigned kk = k0 | (k1 << 8) | (((mc2 >> cbit) & 1u) << 16) | (((mc3 >> cbit) & 1u) << 24);
                        __builtin_amdgcn_raw_buffer_store_b32(a0 ? kk : 0u, rr, lt ? offA : DEAD, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(a1 ? kk : 0u, rr, (lt && rA1) ? offA + N : DEAD, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(b0 ? kk : 0u, rr, (lt && rB0) ? offB : DEAD, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(b1 ? kk : 0u, rr, (lt && rB1) ? offB + N : DEAD, 0, 0);
                    } else if (vecm == 2) {     // (uniform) two bytes per row of the tile
                        const unsigned kk = k0 | (k1 << 8);
                        __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(a0 ? kk : 0u), rr, lt ? offA : DEAD, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(a1 ? kk : 0u), rr, (lt && rA1) ? offA + N : DEAD, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(b0 ? kk : 0u), rr, (lt && rB0) ? offB : DEAD, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(b1 ? kk : 0u), rr, (lt && rB1) ? offB + N : DEAD, 0, 0);
                    } else {
                        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)(a0 & k0), rr, lt ? offA : DEAD, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)(a0 & k1), rr, l1 ? offA + 1 : DEAD, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)(a1 & k0), rr, (lt && rA1) ? offA + N : DEAD, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)(a1 & k1), rr, (l1 && rA1) ? offA + N + 1 : DEAD, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)(b0 & k0), rr, (lt && rB0) ? offB : DEAD, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)(b0 & k1), rr, (l1 && rB0) ? offB + 1 : DEAD, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)(b1 & k0), rr, (lt && rB1) ? offB + N : DEAD, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)(b1 & k1), rr, (l1 && rB1) ? offB + N + 1 : DEAD, 0, 0);
                    }
                };
                plane(r_mca, 1u, 0u);      // CA_i & CA_j
                plane(r_mcb, 2u, 1u);      // CB_i & CB_j
                plane(r_mno, 0u, 2u);      // N_i & O_j
            }
            f32x2 v[4], w[4];
#ifdef PS_K3_AB
            const int probe = k3f_probe;
            if (probe == 1) {
                for (int c = 0; c < 4; ++c) v[c] = w[c] = f32x2{(float)lane, (float)c};
                emit(r_dca, v, w); emit(r_dcb, v, w); emit(r_dno, v, w); emit(r_ph, v, w); emit(r_om, v, w); emit(r_th, v, w);
            } else {
#endif
            // each plane: the first half's four chains, the second half's (four-column tiles), the stores
#define K3F_TILE_PLANE(RSRC, EXPR0, EXPR1)      \
            { EXPR0; if (wide) { EXPR1; } emit(RSRC, v, w); }
#define K3F_DIST(R, C, OUT) _Pragma("unroll") for (int c = 0; c < 4; ++c) OUT[c] = dist3v_t<EXACT>(R[c], C[c])
            K3F_TILE_PLANE(r_dca, K3F_DIST(CAV, CAJ[0], v), K3F_DIST(CAV, CAJ[1], w))
            K3F_TILE_PLANE(r_dcb, K3F_DIST(CBV, CBJ[0], v), K3F_DIST(CBV, CBJ[1], w))
            K3F_TILE_PLANE(r_dno, K3F_DIST(NV, OJ[0], v), K3F_DIST(NV, OJ[1], w))
            if constexpr (FAITHFUL) {
                K3F_TILE_PLANE(r_ph, angle3v_ref_n<4>(CAV, CBV, CBJ[0], v), angle3v_ref_n<4>(CAV, CBV, CBJ[1], w))
                K3F_TILE_PLANE(r_om, dihedral4v_ref_n<4>(CAV, CBV, CAJ[0], CBJ[0], v), dihedral4v_ref_n<4>(CAV, CBV, CAJ[1], CBJ[1], w))   // as coded at protstruc.py:811
                K3F_TILE_PLANE(r_th, dihedral4v_ref_n<4>(NV, CAV, CBV, CBJ[0], v), dihedral4v_ref_n<4>(NV, CAV, CBV, CBJ[1], w))
            } else {
                K3F_TILE_PLANE(r_ph, angle3v_n<4>(CAV, CBV, CBJ[0], v), angle3v_n<4>(CAV, CBV, CBJ[1], w))
                K3F_TILE_PLANE(r_om, dihedral4v_k3_n<4>(CAV, CBV, CAJ[0], CBJ[0], v), dihedral4v_k3_n<4>(CAV, CBV, CAJ[1], CBJ[1], w))
                K3F_TILE_PLANE(r_th, dihedral4v_k3_n<4>(NV, CAV, CBV, CBJ[0], v), dihedral4v_k3_n<4>(NV, CAV, CBV, CBJ[1], w))
            }
#undef K3F_DIST
#undef K3F_TILE_PLANE
#ifdef PS_K3_AB
            }
#endif
            unsigned nx = 0;
            if (lane == 0) nx = atomicAdd(&next_task, 1u);
            t = (unsigned)__builtin_amdgcn_readfirstlane((int)nx);
        }
        K3F_STAMP(3 + 3 * pass);
    }
}

// Rows per task (even, 2..8).  The 16 waves of a workgroup pull the tasks of one (structure, strip) segment at a time, and
// the pulling only evens out the SIMD arbiter's oldest-first order if a segment has several tasks per wave: a segment of
// `rows` rows gets about 64 tasks (N = 512: 8 rows per task, 256: 4, 128: 2; with one task per wave a 128-residue segment
// ended when its slowest wave did: 61 -> 5x us at 2^25 pairs, profiles/r04_k3_shapes.log).  A task costs one LDS atomic.
inline int k3_rows_per_task(int rows, int min_rows) {
#ifdef PS_K3_AB
    static const int forced = getenv("PS_K3_CH") ? atoi(getenv("PS_K3_CH")) : 0;
    if (forced > 0) return forced;
#endif
    const int ch = (rows / 64) & ~1;
    return std::min(8, std::max(min_rows, ch));
}

// ---- launch context: the dispatchers either launch or only RECORD what they would launch (ps_k3_plan_f32, ----
// ps_featuriser_plan_f32).  Every K3 / featuriser launch goes through k3_go, so the plan a caller reads is by construction
// the dispatch a launch takes: same predicates, same grid, same LDS size (K1 has the same arrangement: k1_go).
struct K3Go {
    hipStream_t s;
    ps_k3_plan* plan;   // non-null: record only, launch nothing, make no HIP call
    int cus;            // compute units of the device the launch is for
};

struct K3Shape {        // what the plan reports besides the kernel's name and launch geometry
    int nc = 1, vec = 0, skips = 0, mask_mode = 0, wt = 0, faithful = 0, rows_per_task = 0, wgs_per_cu = 0, structs_per_segment = 0;
    unsigned n_tasks = 0, tasks_per_wg = 0;
};

template <typename... KArgs, typename... Args>
inline int k3_go(const K3Go& go, const char* family, const char* name, const K3Shape& sh, void (*kernel)(KArgs...),
                 unsigned long long (*prepared)[1], dim3 grid, dim3 block, size_t dyn, unsigned static_lds, Args&&... args) {
    if (!go.plan) {
        if (prepared)
            if (const int e = k3_allow_big_lds(kernel, *prepared, go.s)) return e;
        return ps_launch(kernel, grid, block, dyn, go.s, static_cast<Args&&>(args)...);
    }
    ps_k3_plan& pl = *go.plan;
    snprintf(pl.family, sizeof pl.family, "%s", family);
    snprintf(pl.kernel, sizeof pl.kernel, "%s", name);
    pl.n_launches = 1;
    pl.columns_per_lane = sh.nc; pl.vector_stores = sh.vec; pl.skips_dead_groups = sh.skips; pl.mask_store_mode = sh.mask_mode;
    pl.write_through = sh.wt; pl.faithful = sh.faithful; pl.rows_per_task = sh.rows_per_task; pl.workgroups_per_cu = sh.wgs_per_cu;
    pl.structures_per_segment = sh.structs_per_segment;
    pl.n_workgroups = grid.x; pl.threads_per_workgroup = (int)block.x; pl.lds_bytes = (unsigned)dyn + static_lds;
    pl.n_tasks = sh.n_tasks; pl.tasks_per_workgroup = sh.tasks_per_wg;
    return 0;
}

template <int NP, int SRC, int NC, bool VEC, bool FAITHFUL>
int launch_sweep(const float* xyz, float* out, int B, int N, int A, const AtomSel& sel, int row_begin, int row_end,
                 int out_rows, int out_row_origin, const K3Go& go) {
    constexpr int NPI = NP - __builtin_popcount(SRC & ((1 << NP) - 1));
    constexpr int THREADS = k3_sweep_threads(NC, FAITHFUL);
    const int rows = row_end - row_begin, cus = go.cus;
    const int n_strips = (N + 64 * NC - 1) / (64 * NC);
    const int CH = k3_rows_per_task(rows, 2);
    const int n_chunks = (rows + CH - 1) / CH;
    const unsigned long long n_tasks = (unsigned long long)n_strips * n_chunks * B;
    if (n_tasks > 0x7FFFFFFFull) return (int)hipErrorInvalidValue;
    const size_t need = (size_t)((rows + 2) / 2) * (NPI > 0 ? NPI : 1) * 3 * 8;   // one segment's rows, pair-interleaved
    // Every CU one full-width workgroup -- or, when a workgroup would walk four or more (structure, strip) segments (chains
    // of ~128 residues in large batches), TWO of half the width: a segment's set-up (two barriers, the global-load latency
    // of its rows and column points) idles all of a workgroup's waves for ~1.8 us, and a second workgroup on the CU computes
    // meanwhile.  Their LDS requests admit exactly two per CU.  Short lists: at least 4 tasks per workgroup.
    const unsigned long long segs = (unsigned long long)n_strips * B;
    const bool two = segs >= 4ull * cus * 2 && need <= K3_LDS_TWO_PER_CU;
    const unsigned slots = (unsigned)cus * (two ? 2u : 1u);
    const unsigned tasks_per_wg = (unsigned)std::max<unsigned long long>((n_tasks + slots - 1) / slots, 4ull);
    const unsigned grid = (unsigned)((n_tasks + tasks_per_wg - 1) / tasks_per_wg);
    const size_t dyn = two ? K3_LDS_TWO_PER_CU : std::max(need, K3_LDS_ONE_PER_CU);
    static unsigned long long prepared[1] = {0};   // bit d: device d allows this kernel its dynamic LDS
    char name[96];
    snprintf(name, sizeof name, "k3_sweep<NP=%d,SRC=%d,NC=%d,VEC=%d,FAITHFUL=%d>", NP, SRC, NC, (int)VEC, (int)FAITHFUL);
    K3Shape sh;
    sh.nc = NC; sh.vec = VEC; sh.skips = !VEC; sh.faithful = FAITHFUL; sh.rows_per_task = CH;
    sh.wgs_per_cu = two ? 2 : 1; sh.structs_per_segment = 1; sh.n_tasks = (unsigned)n_tasks; sh.tasks_per_wg = tasks_per_wg;
    return k3_go(go, "sweep", name, sh, k3_sweep<NP, SRC, NC, VEC, FAITHFUL>, &prepared, dim3(grid), dim3(THREADS >> (two ? 1 : 0)), dyn, 4u,
                 xyz, out, N, A, sel, row_begin, row_end, out_rows, out_row_origin, CH, n_strips, n_chunks, (unsigned)n_tasks, tasks_per_wg);
}

// LDS the flat kernel stages its structures in (one workgroup per CU: the request keeps a second one off the CU)
constexpr size_t K3_FLAT_LDS = 128 * 1024;

template <int NP, int SRC, bool FAITHFUL>
int launch_flat(const float* xyz, float* out, int B, int N, int A, const AtomSel& sel, int row_begin, int row_end, int out_rows,
                int out_row_origin, unsigned out_misalign_bytes, const K3Go& go) {
    constexpr int NPI = NP - __builtin_popcount(SRC & ((1 << NP) - 1)), NPJ = NP - NPI;
    const int rows = row_end - row_begin, rp = (rows + 1) / 2;
    // a row of the tile as one 16-byte store (then the tile is four columns wide), or as one 8-byte store, where the chain length
    // and the rows' alignment allow it
    const unsigned mis = out_misalign_bytes + (unsigned)(((long long)row_begin - out_row_origin) * N * 4);
    int vec = (N % 4 == 0 && (mis & 15u) == 0) ? 4 : (N % 2 == 0 && (mis & 7u) == 0) ? 2 : 0;
    const unsigned TR = (unsigned)(rp + 1) / 2;
    if (vec == 4) {
        // a structure's tiles fill whole tasks of 64: the wide tile's instruction saving (~10 %) must not go to idle lanes of the
        // last task (N = 36: 81 tiles in 2 tasks against 162 in 3: 113 against 105 us; N = 48: 144 in 3 against 288 in 5: level)
        const unsigned ft_w = TR * ((unsigned)N / 4), ft_n = TR * ((unsigned)N / 2);
        const unsigned long long lanes_w = 64ull * ((ft_w + 63) / 64) * 2, lanes_n = 64ull * ((ft_n + 63) / 64);   // in narrow-tile units
        if (lanes_w * 100 > lanes_n * 108) vec = 2;
    }
    const unsigned TC = vec == 4 ? (unsigned)N / 4 : (unsigned)(N + 1) / 2;       // tiles of two row pairs x four / two columns
    const unsigned tps = (TR * TC + 63u) / 64u;                                   // tasks (64 tiles) per structure
    const unsigned long long n_tasks = (unsigned long long)tps * B;
    if (n_tasks > 0x7FFFFFFFull) return (int)hipErrorInvalidValue;
    const int col_vec4 = NPJ * N, slot_vec4 = col_vec4 + rp * NPI * 2;   // 16-byte units: {x, y, z, -} per column atom; two per row-pair atom
#ifdef PS_K3_AB
    static const int wgs_env = getenv("PS_K3_FLAT_WGS") ? atoi(getenv("PS_K3_FLAT_WGS")) : 2;
    static const int passes_env = getenv("PS_K3_FLAT_PASSES") ? atoi(getenv("PS_K3_FLAT_PASSES")) : 4;
#else
    constexpr int wgs_env = 2, passes_env = 4;
#endif
    // two 512-thread workgroups per CU (their LDS requests admit exactly two): one computes while the other stages
    const int wgs = wgs_env == 1 ? 1 : 2;
    const unsigned slots = (unsigned)go.cus * (unsigned)wgs;
    const unsigned tasks_per_wg = (unsigned)std::max<unsigned long long>((n_tasks + slots - 1) / slots, 4ull);
    const unsigned grid = (unsigned)((n_tasks + tasks_per_wg - 1) / tasks_per_wg);
    // structures per staging pass: a quarter of a workgroup's share (so that the two workgroups of a CU interleave their
    // passes), at most what the LDS holds
    const size_t lds_cap = wgs == 2 ? K3_LDS_TWO_PER_CU : K3_FLAT_LDS;
    const unsigned share = (tasks_per_wg + tps - 1) / tps + 1;
    const size_t by_share = std::max<size_t>(1, (share + passes_env - 1) / passes_env);
    const int KS = (int)std::max<size_t>(1, std::min<size_t>(lds_cap / ((size_t)slot_vec4 * 16), by_share));
    const size_t dyn = wgs == 2 ? K3_LDS_TWO_PER_CU : std::max((size_t)KS * slot_vec4 * 16, K3_LDS_ONE_PER_CU);
    static unsigned long long prepared[1] = {0};
    char name[96];
    snprintf(name, sizeof name, "k3_flat<NP=%d,SRC=%d,FAITHFUL=%d>", NP, SRC, (int)FAITHFUL);
    K3Shape sh;
    sh.nc = 4; sh.skips = 1; sh.faithful = FAITHFUL; sh.rows_per_task = 0; sh.wgs_per_cu = wgs; sh.structs_per_segment = KS;
    sh.vec = vec / 2;                       // ps_k3_plan.vector_stores: 0 dword, 1 8-byte, 2 16-byte stores
    sh.n_tasks = (unsigned)n_tasks; sh.tasks_per_wg = tasks_per_wg;
    return k3_go(go, "flat_tiles", name, sh, k3_flat<NP, SRC, FAITHFUL>, &prepared, dim3(grid), dim3(1024 / wgs), dyn, 4u, xyz, out, N, A, sel, row_begin,
                 row_end, out_rows, out_row_origin, KS, tps, (unsigned)n_tasks, tasks_per_wg, (unsigned)((1ull << 32) / (unsigned)N), col_vec4,
                 slot_vec4, (unsigned)((1ull << 32) / std::max(1u, TC)), vec);
}

// whether one structure's selected atoms fit the flat kernel's LDS (two workgroups per CU)
inline bool k3_flat_fits(int N, int) { return (size_t)N * 4 * 16 + (size_t)((N + 1) / 2) * 4 * 32 <= K3_LDS_TWO_PER_CU; }

// One-column kernels: a workgroup as wide as the chain needs (whole waves, at most 256 lanes) -- with 256 lanes for a
// 64-residue chain three of four waves computed pairs that do not exist.
inline int k3_one_column_threads(int N) { return N >= 256 ? 256 : 64 * ((N + 63) / 64); }

// The per-CU sweep kernels pay two workgroup barriers and a staging pass per (structure, strip) segment and give a wave a
// strip of 64 * NC columns: below ~100 residues a segment has fewer tasks than the workgroup has waves and most lanes of a
// strip idle (N = 64: 143 us against 93 for the one-column kernel at 2^25 pairs; N = 16: 1427 against 282;
// profiles/r04_k3_shapes.log) -- shorter chains take k3_flat (round 5; the one-column kernel before).
constexpr int K3_SWEEP_MIN_N = 100;
// ... the featuriser's sweep (tasks of two rows, a structure's column points in LDS, two workgroups per CU, and since round 5
// several structures per staging pass) pays from 40 on: same-box, 2^25 pairs, min of 20 launches, sweep / one-column kernel:
// N = 63 276 / 344 us, 56 306 / 352, 48 333 / 371, 40 386 / 417, 33 480 / 449 (profiles/r05_featuriser_shapes.log)
constexpr int K3_FEATURISE_MIN_N = 40;
// the featuriser's tile kernel: chain lengths it takes (same-box A/B against the sweep and the one-column kernel: profiles/r05_featuriser_shapes.log)
// same box, 2^25 pairs, min of 20 launches, tiles / product before: N = 16 428 / 666 us, 24 372 / 605, 33 363 / 433, 40 271 / 355,
// 48 258 / 297, 64 224 / 234, 80 233 / 277, 99 265 / 267, 100 243-275 / 252, 129 262 / 339, 160 211 / 280, 200 221-245 / 232;
// the sweep wins where its lanes are full (128: 185 against 206, 256: 161 / 194) and at odd lengths above ~100 (101: 272 / 288:
// the tiles' stores are dwords and bytes there)
constexpr int K3F_TILES_MIN_N = 8, K3F_TILES_MAX_N = 96;     // every chain of 8 .. 96 residues ...
constexpr int K3F_TILES_MAX_N_EVEN = 200, K3F_TILES_UTIL_PERCENT = 85;   // ... and even lengths up to 200 where < 85 % of the sweep's lanes would have a column
constexpr unsigned K3F_TILES_WGS = 2;                        // (2 / 3 / 4 workgroups per CU: no difference beyond noise)
constexpr unsigned K3F_TILES_OVER = 1;                       // workgroups per resident slot
constexpr int K3_FLAT_MAX_N = 256;        // ... up to this length (above it the fast sweeps are level with the tiles: 57-64 / 59-66 / 38-42 us)
constexpr int K3_FLAT_MAX_N_WIDE = 448;   // ... this one where the tiles are four columns wide (N % 4 == 0, 16-byte rows): 56-59 / 54-61 / 38-44 us from 288 to 448
                                          // residues against the sweeps' 56-72 / 59-69 / 39-49 (level at 256 and 384; from 480 on the sweeps win)
constexpr int K3_FLAT_MAX_N_FAITHFUL = 480;   // (faithful: N = 300 94 / 84 / 60 us against 117 / 97 / 66; from 500 on the sweeps win)
constexpr int K3_FLAT_UTIL_PERCENT = 95; // ... and instead of a sweep of which fewer than this share of the lanes would have a column
constexpr int K3_FLAT_UTIL_PERCENT_FAITHFUL = 95;   // (the faithful sweeps, two waves per SIMD, lose even more to idle lanes: N = 160 149 us against 98)
constexpr int K3_SMALL_MAX_N = 32;    // k3_small: one wave per structure (33..64 measured: no better than the one-column kernel)

template <int NP, int SRC, bool FAITHFUL>
int launch_one_column(const float* xyz, float* out, int B, int N, int A, const AtomSel& sel, int row_begin, int row_end,
                      int out_rows, int out_row_origin, const K3Go& go) {
    const int rows = row_end - row_begin, thr1 = k3_one_column_threads(N), IR = 16;
    const int n_tiles = (N + thr1 - 1) / thr1, n_chunks = (rows + IR - 1) / IR;
    const unsigned long long n_wg = (unsigned long long)n_tiles * n_chunks * B;
    if (n_wg > 0x7FFFFFFFull) return (int)hipErrorInvalidValue;
    char name[96];
    snprintf(name, sizeof name, "k3_pairwise_angles<NP=%d,SRC=%d,FAITHFUL=%d>", NP, SRC, (int)FAITHFUL);
    K3Shape sh;
    sh.faithful = FAITHFUL; sh.rows_per_task = IR;
    return k3_go(go, "one_column", name, sh, k3_pairwise_angles<NP, SRC, FAITHFUL>, nullptr, dim3((unsigned)n_wg), dim3(thr1), 0, 0u, xyz, out,
                 N, A, sel, row_begin, row_end, out_rows, out_row_origin, IR, n_tiles, n_chunks);
}

// mode = exact_angles of the C ABI: bit 0 = the reference's order of operations (FAITHFUL), bit 1 = the one-column kernel
// whatever the shape (diagnostic: the layout-free twin the sweep kernels are held to, bit for bit, in either arithmetic)
template <int NP, int SRC, bool FAITHFUL>
int launch(const float* xyz, float* out, int B, int N, int A, const AtomSel& sel, int row_begin, int row_end,
           int out_rows, int out_row_origin, bool simple, unsigned out_misalign, const K3Go& go) {
    constexpr int NPI = NP - __builtin_popcount(SRC & ((1 << NP) - 1));
    const int rows = row_end - row_begin;
#ifdef PS_K3_AB
    static const int small_max = getenv("PS_K3_SMALL_MAX") ? atoi(getenv("PS_K3_SMALL_MAX")) : K3_SMALL_MAX_N;
#else
    constexpr int small_max = K3_SMALL_MAX_N;
#endif
    if (!simple && N <= small_max) {   // short chains: lanes = (row group, column), one wave per structure
        const unsigned long long n_wg = ((unsigned long long)B + 3) / 4;
        if (n_wg > 0x7FFFFFFFull) return (int)hipErrorInvalidValue;
        char name[96];
        snprintf(name, sizeof name, "k3_small<NP=%d,SRC=%d,FAITHFUL=%d>", NP, SRC, (int)FAITHFUL);
        K3Shape sh;
        sh.faithful = FAITHFUL; sh.rows_per_task = 4 * (64 >> (N <= 16 ? 4 : 5)); sh.nc = N <= 16 ? 16 : 32;
        return k3_go(go, "small", name, sh, k3_small<NP, SRC, FAITHFUL>, nullptr, dim3((unsigned)n_wg), dim3(256), 0, 0u, xyz, out, B, N, A, sel,
                     row_begin, row_end, out_rows, out_row_origin, N <= 16 ? 4 : N <= 32 ? 5 : 6);
    }
    // NC columns per lane need N % NC == 0 and 4 * NC-byte aligned rows; the segment's rows have to fit the LDS, and its
    // output has to be addressable by the kernel's 32-bit byte offsets (a row-sharded call on a very long chain)
    const bool fits = !simple && N >= K3_SWEEP_MIN_N && (size_t)((rows + 2) / 2) * (NPI > 0 ? NPI : 1) * 3 * 8 <= K3_LDS_MAX &&
                      (unsigned long long)rows * (unsigned long long)N * 4ull < (1ull << 31);
    // four columns per lane only where the instantiation keeps its registers (three or two column-side points of a
    // dihedral times four columns do not fit the 128 VGPRs of a 1024-thread workgroup: 4-95 spilled registers -- the (2,2)
    // split fitted with 124 until round 5's three-instruction dot products pushed it to 128 + 9 spilled; at two columns it
    // runs as fast (same-box A/B 54.8 / 52.5 against 54.0 / 52.0 us, profiles/r05_k3_modes_first.log) and skips dead groups;
    // the faithful kernels' 512-thread workgroups have 256 and take four columns for every split)
    constexpr bool NC4 = FAITHFUL ? K3_FAITHFUL_NC4(NP, SRC)
                                  : (NP == 3 || SRC == 0 || SRC == 1 || SRC == 2 || SRC == 4 || SRC == 8 || SRC == 15);
#ifdef PS_K3_AB
    static const int force_nc = getenv("PS_K3_NC") ? atoi(getenv("PS_K3_NC")) : 0;
    const bool allow4 = force_nc != 2;
    static const int flat_util = getenv("PS_K3_FLAT_UTIL") ? atoi(getenv("PS_K3_FLAT_UTIL")) : (FAITHFUL ? K3_FLAT_UTIL_PERCENT_FAITHFUL : K3_FLAT_UTIL_PERCENT);
    static const int flat_max_n = getenv("PS_K3_FLAT_MAX_N") ? atoi(getenv("PS_K3_FLAT_MAX_N")) : (FAITHFUL ? K3_FLAT_MAX_N_FAITHFUL : K3_FLAT_MAX_N);
#else
    constexpr int flat_max_n = FAITHFUL ? K3_FLAT_MAX_N_FAITHFUL : K3_FLAT_MAX_N;
    const bool allow4 = true;
    constexpr int flat_util = FAITHFUL ? K3_FLAT_UTIL_PERCENT_FAITHFUL : K3_FLAT_UTIL_PERCENT;
#endif
    const bool ok4 = NC4 && allow4 && fits && N % 4 == 0 && (out_misalign & 15u) == 0, ok2 = fits && N % 2 == 0 && (out_misalign & 7u) == 0;
    // lanes past the last column idle: take the width that wastes fewer of them (a tie goes to the wider stores)
    const long long w4 = (long long)((N + 255) / 256) * 256, w2 = (long long)((N + 127) / 128) * 128;
    // The 64-apart layout skips the dead column groups of a row's last strip: it computes ceil(N / 64) groups per row pair
    // where the vector layouts compute whole strips -- taken also for even N where that saves more than its dword stores cost
    const long long gn = (N + 63) / 64, gv = (NC4 && ok4 && (!ok2 || w4 <= w2) ? w4 : w2) / 64;
    constexpr bool SKIPS = true;   // (every instantiation skips dead groups since the fast (2,2) dihedral takes two columns per lane: round 5)
    // the sweep layout this launch would take: vector stores or columns 64 apart, columns per lane, and the 64-column groups
    // it evaluates per row pair (dead ones included unless skipped)
    const bool vec = (ok4 || ok2) && !(SKIPS && gn * 115 < gv * 100);
    const int nc = vec ? ((NC4 && ok4 && (!ok2 || w4 <= w2)) ? 4 : 2) : ((NC4 && allow4 && (SKIPS ? gn > 2 : w4 <= w2)) ? 4 : 2);
    const long long g_eval = vec ? gv : (SKIPS ? gn : (nc == 4 ? w4 : w2) / 64);
    // The flat kernel in its 2 x 2 tile map (every lane has an element whatever N is; 59-66 / 66-77 / 45-55 us per 2^25 pairs from
    // 48 to 140 residues, profiles/r05_k3_shapes.log): every chain shorter than the sweeps' minimum, and up to 256 residues wherever
    // fewer than K3_FLAT_UTIL_PERCENT of the sweep's lanes would have a column (N = 140: three groups of 64 for 140 columns,
    // 59 against 102 us; N = 180: 59 / 66 / 46 against 61 / 74 / 50).
    const bool wide_tiles = N % 4 == 0 && (out_misalign & 15u) == 0;
#ifdef PS_K3_AB
    const int flat_max = (getenv("PS_K3_FLAT_MAX_N") || FAITHFUL || !wide_tiles) ? flat_max_n : K3_FLAT_MAX_N_WIDE;
#else
    const int flat_max = (FAITHFUL || !wide_tiles) ? flat_max_n : K3_FLAT_MAX_N_WIDE;
#endif
    if (!simple && N > small_max && N >= 32 && k3_flat_fits(N, A) &&
        (N < K3_SWEEP_MIN_N || !fits ||
         (N <= flat_max && ((long long)N * 100 < (long long)flat_util * 64 * g_eval ||
                              // ... or where the sweep would write dwords (its columns 64 apart) and the tiles whole 16-byte rows: N = 192
                              (!vec && wide_tiles))))) {
        return launch_flat<NP, SRC, FAITHFUL>(xyz, out, B, N, A, sel, row_begin, row_end, out_rows, out_row_origin, out_misalign, go);
    }
    if (vec) {
        if constexpr (NC4) {
            if (nc == 4)
                return launch_sweep<NP, SRC, 4, true, FAITHFUL>(xyz, out, B, N, A, sel, row_begin, row_end, out_rows, out_row_origin, go);
        }
        return launch_sweep<NP, SRC, 2, true, FAITHFUL>(xyz, out, B, N, A, sel, row_begin, row_end, out_rows, out_row_origin, go);
    }
    if (fits) {   // odd N, a misaligned output, or fewer groups: the same sweep with the lane's columns 64 apart and dword stores
        if constexpr (NC4) {   // four columns from three groups on where dead groups are skipped; else by the lanes a strip wastes
            if (nc == 4)
                return launch_sweep<NP, SRC, 4, false, FAITHFUL>(xyz, out, B, N, A, sel, row_begin, row_end, out_rows, out_row_origin, go);
        }
        return launch_sweep<NP, SRC, 2, false, FAITHFUL>(xyz, out, B, N, A, sel, row_begin, row_end, out_rows, out_row_origin, go);
    }
    return launch_one_column<NP, SRC, FAITHFUL>(xyz, out, B, N, A, sel, row_begin, row_end, out_rows, out_row_origin, go);
}

template <int NP, int... SRCS>
int dispatch(int srcmask, const float* xyz, float* out, int B, int N, int A, const AtomSel& sel, int row_begin,
             int row_end, int out_rows, int out_row_origin, int mode, unsigned out_misalign, const K3Go& go) {
    int rc = (int)hipErrorInvalidValue;
    const bool simple = (mode & 2) != 0;
    if (mode & 1)
        (void)((srcmask == SRCS ? (rc = launch<NP, SRCS, true>(xyz, out, B, N, A, sel, row_begin, row_end, out_rows, out_row_origin, simple, out_misalign, go), true)
                                : false) || ...);
    else
        (void)((srcmask == SRCS ? (rc = launch<NP, SRCS, false>(xyz, out, B, N, A, sel, row_begin, row_end, out_rows, out_row_origin, simple, out_misalign, go), true)
                                : false) || ...);
    return rc;
}

// argument checks shared by the launcher and the plan query
inline int k3_check_args(int B, int N, int A, int n_points, const int* src, const int* atom, int row_begin, int row_end, int out_rows,
                         int out_row_origin, int exact_angles, AtomSel& sel, int& srcmask) {
    if (!src || !atom || B < 0 || N < 0 || A <= 0) return (int)hipErrorInvalidValue;
    if (exact_angles < 0 || exact_angles > 3) return (int)hipErrorInvalidValue;
    if (n_points != 3 && n_points != 4) return (int)hipErrorInvalidValue;
    if (row_begin < 0 || row_end > N || row_begin > row_end) return (int)hipErrorInvalidValue;
    if (out_row_origin > row_begin || row_end - out_row_origin > out_rows) return (int)hipErrorInvalidValue;
    sel = AtomSel{};
    srcmask = 0;
    for (int k = 0; k < n_points; ++k) {
        if (atom[k] < 0 || atom[k] >= A || (src[k] != 0 && src[k] != 1)) return (int)hipErrorInvalidValue;
        sel.atom[k] = atom[k];
        srcmask |= src[k] << k;
    }
    return 0;
}

inline int k3_run(const float* xyz, float* out, int B, int N, int A, int n_points, const AtomSel& sel, int srcmask, int row_begin,
                  int row_end, int out_rows, int out_row_origin, int exact_angles, unsigned out_misalign, const K3Go& go) {
    if (n_points == 4)
        return dispatch<4, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15>(srcmask, xyz, out, B, N, A, sel, row_begin, row_end, out_rows,
                                                                                 out_row_origin, exact_angles, out_misalign, go);
    return dispatch<3, 0, 1, 2, 3, 4, 5, 6, 7>(srcmask, xyz, out, B, N, A, sel, row_begin, row_end, out_rows, out_row_origin, exact_angles,
                                               out_misalign, go);
}

inline void k3_plan_reset(ps_k3_plan* plan) {
    const int sz = plan->struct_size;
    memset(plan, 0, sizeof *plan);
    plan->struct_size = sz;
    snprintf(plan->family, sizeof plan->family, "empty");
}

}  // namespace

extern "C" int ps_pairwise_angles_f32(const float* xyz, float* out, int B, int N, int A, int n_points, const int* src,
                                      const int* atom, int row_begin, int row_end, int out_rows, int out_row_origin,
                                      int exact_angles, void* stream) {
    if (!xyz || !out) return (int)hipErrorInvalidValue;
    AtomSel sel;
    int srcmask = 0;
    if (const int e = k3_check_args(B, N, A, n_points, src, atom, row_begin, row_end, out_rows, out_row_origin, exact_angles, sel, srcmask))
        return e;
    if (B == 0 || N == 0 || row_begin == row_end) return 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const K3Go go{s, nullptr, k3_cu_count(s)};
    return k3_run(xyz, out, B, N, A, n_points, sel, srcmask, row_begin, row_end, out_rows, out_row_origin, exact_angles,
                  (unsigned)(reinterpret_cast<uintptr_t>(out) & 15u), go);
}

extern "C" int ps_k3_plan_f32(int B, int N, int A, int n_points, const int* src, const int* atom, int row_begin, int row_end,
                              int out_rows, int out_row_origin, int out_misalign, int exact_angles, int cu_count,
                              ps_k3_plan* plan) {
    if (!plan || plan->struct_size != (int)sizeof(ps_k3_plan)) return (int)hipErrorInvalidValue;
    k3_plan_reset(plan);
    AtomSel sel;
    int srcmask = 0;
    if (const int e = k3_check_args(B, N, A, n_points, src, atom, row_begin, row_end, out_rows, out_row_origin, exact_angles, sel, srcmask))
        return e;
    if (out_misalign < 0 || out_misalign > 15 || (out_misalign & 3)) return (int)hipErrorInvalidValue;
    if (B == 0 || N == 0 || row_begin == row_end) return 0;
    const K3Go go{nullptr, plan, cu_count > 0 ? cu_count : 256};
    return k3_run(nullptr, nullptr, B, N, A, n_points, sel, srcmask, row_begin, row_end, out_rows, out_row_origin, exact_angles,
                  (unsigned)out_misalign, go);
}

namespace {

// argument checks shared by the featuriser's launcher and its plan query
inline int k3f_check_args(int B, int N, int A, int exact_sqrt, int exact_angles) {
    if (B < 0 || N < 0 || A < 5 || (exact_sqrt != 0 && exact_sqrt != 1) || exact_angles < 0 || exact_angles > 3)
        return (int)hipErrorInvalidValue;
    return 0;
}

// one featuriser instantiation by its run-time switches (EXACT: the distance planes' square root)
template <int NC, bool VEC, bool M16, bool WT, bool FAITHFUL, typename... Args>
inline int k3f_go(const K3Go& go, bool exact_sqrt, const K3Shape& sh, dim3 grid, dim3 block, size_t dyn, Args&&... args) {
    static unsigned long long prep[2][1] = {{0}, {0}};
    char name[96];
    snprintf(name, sizeof name, "k3_featurise<EXACT=%d,NC=%d,VEC=%d,M16=%d,WT=%d,FAITHFUL=%d>", (int)exact_sqrt, NC, (int)VEC, (int)M16, (int)WT,
             (int)FAITHFUL);
    if (exact_sqrt)
        return k3_go(go, "featurise", name, sh, k3_featurise<true, NC, VEC, M16, WT, FAITHFUL>, &prep[0], grid, block, dyn, 4u, static_cast<Args&&>(args)...);
    return k3_go(go, "featurise", name, sh, k3_featurise<false, NC, VEC, M16, WT, FAITHFUL>, &prep[1], grid, block, dyn, 4u, static_cast<Args&&>(args)...);
}

// The featuriser sweep's layout for a chain length and float-plane alignment: vector float stores where rows and planes allow
// (else 64 consecutive floats per store instruction: any N); columns per lane by the lanes a strip wastes
struct K3fSweepLayout {
    int nc;
    bool vec;
};
template <bool FAITHFUL>
inline K3fSweepLayout k3f_sweep_layout(int N, uintptr_t alf) {
    const bool v4 = N % 4 == 0 && (alf & 15u) == 0, v2 = N % 2 == 0 && (alf & 7u) == 0;
    const long long w4 = (long long)((N + 255) / 256) * 256, w2 = (long long)((N + 127) / 128) * 128;
    // ... or none at all: with the 64-floats-per-store layout the dead column groups of a row's last strip are skipped, so
    // it computes ceil(N / 64) groups per row pair where the vector layouts compute whole strips (N = 140: 3 against 4,
    // 342 against 376 us at 2^25 pairs) -- taken where that saves more than the ~15 % its dword stores cost
    constexpr bool CAN4 = K3_FEATURISE_NC4 && !FAITHFUL;   // (the faithful chains of four columns do not fit 256 VGPRs)
    const long long gn = (N + 63) / 64, gv = (CAN4 ? std::min(w4, w2) : w2) / 64;
    const bool prefer_scalar = gn * 115 < gv * 100;
    int NC = (CAN4 && w4 <= w2) ? 4 : 2;
    const bool vec = (NC == 4 ? v4 : v2) && !prefer_scalar;
    // ... in which four columns per lane beat two from three groups on (same-box A/B: N = 383 234 against 260 us, 301
    // 257 / 267; two groups, N = 101: 338 / 310)
    if (!vec) NC = (CAN4 && (gn > 2 || N <= 64)) ? 4 : 2;   // (N <= 64: one column group, four row pairs per lane -- the 256-VGPR instantiation)
    return K3fSweepLayout{NC, vec};
}

template <bool FAITHFUL>
int k3f_run(const float* xyz, const uint8_t* atom_mask, float* d_ca, float* d_cb, float* d_no, float* omega, float* theta, float* phi,
            uint8_t* d_ca_mask, uint8_t* d_cb_mask, uint8_t* d_no_mask, int B, int N, int A, int exact_sqrt, bool simple, uintptr_t alf,
            uintptr_t alm, const K3Go& go) {
    // one structure's rows (points + mask words), its column points, its column masks (bytes or bit sets)
    const size_t need = (size_t)((N + 2) / 2) * (9 * 8 + 4) + 32 + 16 + (size_t)((N + 3) & ~3) * 36 +
                        std::max<size_t>(3 * (size_t)N, 3 * ((size_t)(N + 31) / 32 + 2) * 4) + 16;
    // Chains of K3F_TILES_MIN_N .. K3F_TILES_MAX_N residues: the tile kernel (several structures per pass, every lane busy)
#ifdef PS_K3_AB
    static const int tiles_min = getenv("PS_K3F_TILES_MIN") ? atoi(getenv("PS_K3F_TILES_MIN")) : K3F_TILES_MIN_N;
    static const int tiles_max = getenv("PS_K3F_TILES_MAX") ? atoi(getenv("PS_K3F_TILES_MAX")) : K3F_TILES_MAX_N;
#else
    constexpr int tiles_min = K3F_TILES_MIN_N, tiles_max = K3F_TILES_MAX_N;
#endif
    {
        const int n_rp = (N + 1) / 2;
        const size_t slot_vec4 = (size_t)3 * N + (size_t)6 * n_rp + ((size_t)N + 15) / 16;   // column atoms, row atoms, one byte per residue
        // (odd lengths -- dword and byte stores in the tile kernel -- only below 70 %: N = 129 262 against 339 us, but 101 288 against 272)
        const bool tiles_even = N <= K3F_TILES_MAX_N_EVEN && (long long)N * 100 < (long long)(N % 2 == 0 ? K3F_TILES_UTIL_PERCENT : 70) * 64 * ((N + 63) / 64);
        // ... and even lengths whose sweep would fall back to its 64-floats-per-store layout (three or five column groups: N = 176,
        // 192, 272 .. 320): the tiles keep their 8-byte stores (192: 222 against 289 us, 288: 216-255 / 318, 320: 232 / 300,
        // profiles/r05_featuriser_shapes.log)
        const bool tiles_dword_sweep = N % 2 == 0 && (alf & 7u) == 0 && (alm & 1u) == 0 && !k3f_sweep_layout<FAITHFUL>(N, alf).vec;
        // ... and, in the faithful arithmetic (whose sweep has two columns per lane), every length with four-column tiles: level
        // with the sweep at 192 / 256 / 320 / 384, 5-12 % faster at 128, 224, 352, 448-500 (profiles/r05_featuriser_tile_width.log)
        const bool tiles_faithful = FAITHFUL && N % 4 == 0 && (alf & 15u) == 0 && (alm & 3u) == 0;
        if (!simple && N >= tiles_min && (N <= tiles_max || tiles_even || tiles_dword_sweep || tiles_faithful) && (alf & 3u) == 0 && slot_vec4 * 16 <= 48 * 1024 &&
            (unsigned long long)N * N < (1ull << 29)) {
            // four-column tiles where a row of the tile is one 16-byte float store and one 4-byte mask store -- unless a structure's
            // wide tiles would idle > 8 % more lanes of their last task of 64 (as in launch_flat)
            const unsigned TR = (unsigned)(n_rp + 1) / 2;
            int vecf = (N % 2 == 0 && (alf & 7u) == 0) ? 2 : 0, vecm = (N % 2 == 0 && (alm & 1u) == 0) ? 2 : 0;
#ifdef PS_K3_AB
            static const int tiles_wide = getenv("PS_K3F_TILES_WIDE") ? atoi(getenv("PS_K3F_TILES_WIDE")) : 1;
#else
            constexpr int tiles_wide = 1;
#endif
            if (tiles_wide && N % 4 == 0 && (alf & 15u) == 0 && (alm & 3u) == 0) {
                const unsigned ft_w = TR * ((unsigned)N / 4), ft_n = TR * ((unsigned)N / 2);
                const unsigned long long lanes_w = 64ull * ((ft_w + 63) / 64) * 2, lanes_n = 64ull * ((ft_n + 63) / 64);
                if (lanes_w * 100 <= lanes_n * 108) vecf = vecm = 4;
            }
            const unsigned TC = vecf == 4 ? (unsigned)N / 4 : (unsigned)(N + 1) / 2;
            const unsigned tps = (TR * TC + 63u) / 64u;
            const unsigned long long n_tasks = (unsigned long long)tps * B;
            if (n_tasks > 0x7FFFFFFFull) return (int)hipErrorInvalidValue;
#ifdef PS_K3_AB
            static const unsigned tiles_wgs = getenv("PS_K3F_TILES_WGS") ? (unsigned)atoi(getenv("PS_K3F_TILES_WGS")) : K3F_TILES_WGS;
            static const unsigned tiles_over = getenv("PS_K3F_TILES_OVER") ? (unsigned)atoi(getenv("PS_K3F_TILES_OVER")) : K3F_TILES_OVER;
#else
            constexpr unsigned tiles_wgs = K3F_TILES_WGS, tiles_over = K3F_TILES_OVER;
#endif
            // K3F_TILES_WGS 256-thread workgroups per CU (their LDS requests admit exactly that many; 132 / 167 VGPRs: three waves per SIMD)
            const size_t tiles_lds = tiles_wgs == 2 ? K3_LDS_TWO_PER_CU : ((size_t)160 * 1024 / tiles_wgs - 256) & ~(size_t)255;
            const unsigned slots = (unsigned)go.cus * tiles_wgs * tiles_over;
            const unsigned tasks_per_wg = (unsigned)std::max<unsigned long long>((n_tasks + slots - 1) / slots, 4ull);
            const unsigned grid = (unsigned)((n_tasks + tasks_per_wg - 1) / tasks_per_wg);
            const unsigned share = (tasks_per_wg + tps - 1) / tps + 1;
            const int KS = (int)std::max<size_t>(1, std::min<size_t>(tiles_lds / (slot_vec4 * 16), (share + 3) / 4));
            char name[96];
            snprintf(name, sizeof name, "k3_featurise_tiles<EXACT=%d,FAITHFUL=%d>", exact_sqrt, (int)FAITHFUL);
            K3Shape sh;
            sh.nc = 4; sh.vec = vecf / 2; sh.skips = 1; sh.mask_mode = vecm == 4 ? 4 : vecm == 2 ? 3 : 0; sh.faithful = FAITHFUL; sh.wgs_per_cu = (int)tiles_wgs; sh.structs_per_segment = KS;
            sh.n_tasks = (unsigned)n_tasks; sh.tasks_per_wg = tasks_per_wg;
            static unsigned long long prep[2][1] = {{0}, {0}};
            auto tiles = [&](auto kernel, unsigned long long (&prepared)[1]) {
                return k3_go(go, "featurise_tiles", name, sh, kernel, &prepared, dim3(grid), dim3(256), tiles_lds, 4u, xyz, atom_mask, d_ca, d_cb,
                             d_no, omega, theta, phi, d_ca_mask, d_cb_mask, d_no_mask, N, A, KS, tps, (unsigned)n_tasks, tasks_per_wg,
                             (unsigned)((1ull << 32) / (unsigned)N), (unsigned)((1ull << 32) / std::max(1u, TC)), (int)slot_vec4, vecf, vecm);
            };
            return exact_sqrt ? tiles(k3_featurise_tiles<true, FAITHFUL>, prep[0]) : tiles(k3_featurise_tiles<false, FAITHFUL>, prep[1]);
        }
    }
    // the per-CU sweep: any N >= K3_FEATURISE_MIN_N whose rows fit in LDS
#ifdef PS_K3_AB
    static const int feat_min_n = getenv("PS_K3F_MIN_N") ? atoi(getenv("PS_K3F_MIN_N")) : K3_FEATURISE_MIN_N;
#else
    constexpr int feat_min_n = K3_FEATURISE_MIN_N;
#endif
    if (!simple && N >= feat_min_n && need <= K3_LDS_MAX && (alf & 3u) == 0 && (unsigned long long)N * N < (1ull << 31)) {
        // vector float stores where rows and planes allow (else 64 consecutive floats per store instruction: any N);
        // columns per lane by the lanes a strip wastes
        constexpr bool CAN4 = K3_FEATURISE_NC4 && !FAITHFUL;
        const K3fSweepLayout lay = k3f_sweep_layout<FAITHFUL>(N, alf);
        const int NC = lay.nc;
        const bool vec = lay.vec;
        const bool m16 = vec && N % 16 == 0 && (alm & 15u) == 0;   // strip-local 16-byte mask stores: whole 16-column groups
        // write-through where strips are whole and every store covers whole lines; same-box A/B, trace: N = 512 174 against 178 us,
        // 256 181 / 187 -- but N = 480 (15 lines per row, a 224-column second strip) 239 against 212 and 160 308 / 297: there write-back
        const bool wt = m16 && N % 128 == 0 && (alf & 127u) == 0 && (alm & 127u) == 0;
        const int cus = go.cus;
        const int n_strips = (N + 64 * NC - 1) / (64 * NC);
        // a task is CH rows x one strip, the strips of a row chunk adjacent tasks: a workgroup's share has to be many tasks
        // per wave whatever B and N are -- its waves wait for each other at every structure boundary for up to one task.
        // Four rows with the M16 mask stores (four rows per instruction; rows are whole 64-byte segments there).  TWO
        // otherwise: where rows are not whole segments, the segment a row's strips share and the one a row's end shares with
        // the next row's start get their halves from adjacent tasks, i.e. from two waves up to a task apart -- with
        // four-row tasks (~16 us) longer than a line stays in L2 at this store rate, so that half of them went out as two
        // partial writes (2.2 % of the write requests at N = 500, none at N = 496: profiles/r04_featuriser_pmc.log);
        // same-box A/B, trace: N = 500 260 -> 228 us, 511 264 -> 241, 255 238 -> 218.
        const bool two_wg = (unsigned long long)B >= 4ull * cus;
        // (chains of up to 64 residues in the 64-apart layout: eight rows, so that a lane's four chains are four row pairs)
        const int CH = m16 ? 4 : (!vec && N <= 64) ? 8 : 2;
        const int n_chunks = (N + CH - 1) / CH;
        const unsigned long long n_tasks = (unsigned long long)n_chunks * n_strips * B;
        if (n_tasks > 0x7FFFFFFFull) return (int)hipErrorInvalidValue;
        // One workgroup per CU -- or, when a CU's share is four or more structures (chains of ~128 residues in large
        // batches), TWO of half the width: staging a structure (two barriers, a round trip to L2 for its rows and column
        // points) idles all of a workgroup's waves, and the other workgroup computes meanwhile.  Their LDS requests admit
        // exactly one / two per CU.
        const bool two = two_wg && need <= K3_LDS_TWO_PER_CU;
        const int wgs = two ? 2 * cus : cus;
        const unsigned tasks_per_wg = (unsigned)std::max<unsigned long long>((n_tasks + wgs - 1) / wgs, 4ull);
        const unsigned grid = (unsigned)((n_tasks + tasks_per_wg - 1) / tasks_per_wg);
        const size_t dyn = two ? K3_LDS_TWO_PER_CU : std::max(need, K3_LDS_ONE_PER_CU);
        // structures per staging pass (round 5): what the LDS request holds, at most a quarter of a workgroup's share (so that
        // the workgroups of a CU interleave their passes) -- short chains paid two barriers and a round trip to L2 per structure
        const size_t slot_bytes = (need + 15) & ~(size_t)15;
        const unsigned long long n_sub = (unsigned long long)n_chunks * n_strips;
        const unsigned long long share = (tasks_per_wg + n_sub - 1) / n_sub + 1;
#ifdef PS_K3_AB
        static const int ks_max = getenv("PS_K3F_KS_MAX") ? atoi(getenv("PS_K3F_KS_MAX")) : 1 << 20;
#else
        constexpr int ks_max = 1 << 20;
#endif
        const int KS = (int)std::max<unsigned long long>(1, std::min<unsigned long long>(std::min<unsigned long long>(dyn / slot_bytes, (share + 3) / 4), (unsigned long long)ks_max));
        const unsigned rcpN = (unsigned)((1ull << 32) / (unsigned)N);
#ifdef PS_K3_AB
        static const int pull = getenv("PS_K3F_PULL") ? std::max(1, atoi(getenv("PS_K3F_PULL"))) : 1;
#else
        constexpr int pull = 1;
#endif
        // four columns per lane need ~200 VGPRs (three column points x four columns + four interleaved chains): 8 waves; so do
        // the faithful chains of two columns
        const dim3 block((unsigned)(((NC == 4 || FAITHFUL) ? 512 : 1024) >> (two ? 1 : 0)));
        K3Shape sh;
        sh.nc = NC; sh.vec = vec; sh.skips = !vec; sh.mask_mode = m16 ? 2 : 1; sh.wt = wt; sh.faithful = FAITHFUL; sh.rows_per_task = CH;
        sh.wgs_per_cu = two ? 2 : 1; sh.structs_per_segment = KS; sh.n_tasks = (unsigned)n_tasks; sh.tasks_per_wg = tasks_per_wg;
#define K3F_GO(NC_, VEC_, M16_, WT_)                                                                                              \
    return k3f_go<NC_, VEC_, M16_, WT_, FAITHFUL>(go, exact_sqrt != 0, sh, dim3(grid), block, dyn, xyz, atom_mask, d_ca, d_cb, d_no, omega, theta, \
                                                  phi, d_ca_mask, d_cb_mask, d_no_mask, N, A, CH, n_strips, n_chunks, (unsigned)n_tasks,  \
                                                  tasks_per_wg, rcpN, KS, (int)slot_bytes, pull)
        if (wt && NC == 2) K3F_GO(2, true, true, true);
        if (m16 && NC == 2) K3F_GO(2, true, true, false);
        if constexpr (CAN4) {
            if (wt) K3F_GO(4, true, true, true);
            if (m16) K3F_GO(4, true, true, false);
            if (NC == 4 && vec) K3F_GO(4, true, false, false);
            if (NC == 4) K3F_GO(4, false, false, false);
        }
        if (vec) K3F_GO(2, true, false, false);
        K3F_GO(2, false, false, false);
#undef K3F_GO
    }
    const int IR = 16, thr1 = k3_one_column_threads(N);
    const int n_tiles = (N + thr1 - 1) / thr1, n_chunks = (N + IR - 1) / IR;
    const unsigned long long n_wg = (unsigned long long)n_tiles * n_chunks * B;
    if (n_wg > 0x7FFFFFFFull) return (int)hipErrorInvalidValue;
    char name[96];
    snprintf(name, sizeof name, "k3_inter_residue_geometry<EXACT=%d,FAITHFUL=%d>", exact_sqrt, (int)FAITHFUL);
    K3Shape sh;
    sh.faithful = FAITHFUL; sh.rows_per_task = IR;
    auto one = [&](auto kernel) {
        return k3_go(go, "one_column", name, sh, kernel, nullptr, dim3((unsigned)n_wg), dim3(thr1), 0, 0u, xyz, atom_mask, d_ca, d_cb, d_no, omega,
                     theta, phi, d_ca_mask, d_cb_mask, d_no_mask, N, A, IR, n_tiles, n_chunks);
    };
    return exact_sqrt ? one(k3_inter_residue_geometry<true, FAITHFUL>) : one(k3_inter_residue_geometry<false, FAITHFUL>);
}

}  // namespace

extern "C" int ps_inter_residue_geometry_f32(const float* xyz, const uint8_t* atom_mask, float* d_ca, float* d_cb,
                                             float* d_no, float* omega, float* theta, float* phi, uint8_t* d_ca_mask,
                                             uint8_t* d_cb_mask, uint8_t* d_no_mask, int B, int N, int A,
                                             int exact_sqrt, int exact_angles, void* stream) {
    if (!xyz || !d_ca || !d_cb || !d_no || !omega || !theta || !phi || !d_ca_mask || !d_cb_mask || !d_no_mask)
        return (int)hipErrorInvalidValue;
    if (const int e = k3f_check_args(B, N, A, exact_sqrt, exact_angles)) return e;
    if (B == 0 || N == 0) return 0;
    uintptr_t alf = 0, alm = 0;
    for (const void* p : {(const void*)d_ca, (const void*)d_cb, (const void*)d_no, (const void*)omega, (const void*)theta, (const void*)phi})
        alf |= reinterpret_cast<uintptr_t>(p);
    for (const void* p : {(const void*)d_ca_mask, (const void*)d_cb_mask, (const void*)d_no_mask}) alm |= reinterpret_cast<uintptr_t>(p);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const K3Go go{s, nullptr, k3_cu_count(s)};
    const bool simple = (exact_angles & 2) != 0;
    if (exact_angles & 1)
        return k3f_run<true>(xyz, atom_mask, d_ca, d_cb, d_no, omega, theta, phi, d_ca_mask, d_cb_mask, d_no_mask, B, N, A, exact_sqrt, simple, alf & 127u, alm & 127u, go);
    return k3f_run<false>(xyz, atom_mask, d_ca, d_cb, d_no, omega, theta, phi, d_ca_mask, d_cb_mask, d_no_mask, B, N, A, exact_sqrt, simple, alf & 127u, alm & 127u, go);
}

extern "C" int ps_featuriser_plan_f32(int B, int N, int A, int float_misalign, int mask_misalign, int exact_sqrt, int exact_angles,
                                      int cu_count, ps_k3_plan* plan) {
    if (!plan || plan->struct_size != (int)sizeof(ps_k3_plan)) return (int)hipErrorInvalidValue;
    k3_plan_reset(plan);
    if (const int e = k3f_check_args(B, N, A, exact_sqrt, exact_angles)) return e;
    if (float_misalign < 0 || float_misalign > 127 || mask_misalign < 0 || mask_misalign > 127) return (int)hipErrorInvalidValue;
    if (B == 0 || N == 0) return 0;
    const K3Go go{nullptr, plan, cu_count > 0 ? cu_count : 256};
    const bool simple = (exact_angles & 2) != 0;
    if (exact_angles & 1)
        return k3f_run<true>(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, B, N, A, exact_sqrt, simple,
                             (uintptr_t)float_misalign, (uintptr_t)mask_misalign, go);
    return k3f_run<false>(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, B, N, A, exact_sqrt, simple,
                          (uintptr_t)float_misalign, (uintptr_t)mask_misalign, go);
}

#ifdef PS_K3_AB
// tools only: the stamps of the last k3_sweep launch on the current device (after a synchronise)
extern "C" int ps_k3_debug_stamps(unsigned long long* host_dst, int n_words) {
    if (!host_dst || n_words <= 0 || n_words > 512 * 16 * 4) return (int)hipErrorInvalidValue;
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(k3_stamps), (size_t)n_words * 8, 0, hipMemcpyDeviceToHost);
}
extern "C" int ps_k3f_debug_probe(int mode) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(k3f_probe), &mode, sizeof mode, 0, hipMemcpyHostToDevice);
}
extern "C" int ps_k3f_debug_stamps(unsigned long long* host_dst, int n_words) {
    if (!host_dst || n_words <= 0 || n_words > 512 * 8 * 16) return (int)hipErrorInvalidValue;
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(k3f_stamps), (size_t)n_words * 8, 0, hipMemcpyDeviceToHost);
}
#endif
