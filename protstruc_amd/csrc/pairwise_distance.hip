// K1 -- all-atom pairwise distance matrix + pair mask.
// Replaces StructureBatch.pairwise_distance_matrix (reference protstruc.py:455-484).
//
// The output of one (structure b, residue row i) is one contiguous run of
// N*A*A floats (and N*A*A mask bytes): dist[b][i][0..N)[0..A)[0..A).  The kernel
// is a pure HBM *write* stream -- 1125 bytes written per residue pair at A = 15
// against ~1 byte read -- so everything is organised around emitting that
// stream as 16-byte-per-lane stores in address order:
//
//   * a workgroup owns a tile of JT column residues j and IR row residues i of
//     one structure; the 15 atoms of each staged residue sit in LDS padded to
//     float4 (one ds_read_b128 per atom) and the 15 atom-mask bytes of a
//     residue are packed into 15 bits;
//   * the (i, j-tile) run is cut into float4 slots in address order; a lane
//     decodes the slot's first element into (j, a, c) with constant divisions
//     (225 = 15*15, 15) and walks the next three elements with carry logic;
//   * the mask run is cut into 16-byte slots; a slot is a 16-bit window of the
//     concatenated 15-bit rows (m_i[a] ? bits(m_j) : 0), expanded bit->byte
//     with one multiply per four bytes;
//   * rows that are not 16-byte aligned (N % 4 != 0 for dist, N % 16 != 0 for
//     the mask) shift the slot grid so stores stay aligned and the ragged head
//     and tail are written element-wise.
//
// Three A = 15 kernels share that scheme.  Shapes with N % 16 == 0 take the "pattern" kernel (fixed per-lane
// decode, row atoms in registers); any other N >= 16 takes the "flat pattern" kernel (the same fixed decode laid
// over the flat pair axis, so rows need no alignment at all); the slot-decode kernel described above remains for
// unaligned planes and as the bit-identity cross-check in the tests (N < 16 takes the row-phase kernel).
//
// Other atom counts: A = 4, 8 the row-tile kernel; A = 14, 16, 24, 32 the fixed-A flat pattern kernel; every other
// A <= 64 the row-phase kernel (column atoms stationary in registers, any N; compile-time instantiations for the small
// counts, two run-time ones -- even / odd A -- for the rest); what is left (A > 64, unaligned planes) the
// element-per-lane kernel at the bottom.  ps_k1_plan_f32 reports which one a given launch takes (the dispatcher below,
// run in record-only mode).
#include "ps_common.hpp"
#include "../../include/protstruc_hip.h"

#include <cstdio>
#include <cstring>
#include <type_traits>

namespace {

constexpr int A15 = 15;
constexpr int AA15 = 225;

// Launch configuration: `ps_k1_config` of include/protstruc_hip.h, passed per call.  There is no process-global
// tuning state in this library (round 1 had one; a second device or thread could overwrite it mid-stream).
// Measured on MI355X at B=64, N=512 (tools/k1_probe.py): plain stores beat non-temporal ones for the pattern kernel
// (its 3600-byte groups are not 128-byte aligned, so lines are completed by a second wave and want to merge in
// L2); which granule per workgroup is fastest depends on the output allocation, so ops.py autotunes it per device.
using K1Cfg = ps_k1_config;

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ void store16(void* p, uint4 v) {
    u32x4_t w = {v.x, v.y, v.z, v.w};
    if (NT)
        __builtin_nontemporal_store(w, reinterpret_cast<u32x4_t*>(p));
    else
        *reinterpret_cast<u32x4_t*>(p) = w;
}

// A run of 16-byte slots addressed the buffer way: a uniform 64-bit base in a resource descriptor (scalar registers), this
// lane's constant 32-bit byte offset, and the slot group's byte offset as a scalar -- so stepping from group to group is
// scalar-unit work and the vector unit spends nothing on store addresses (it spent two 64-bit adds per store before).
// Raw buffer, stride 0, no range limit below 4 GB; 0x00020000 = 32-bit data format word of gfx90a / gfx94x / gfx950.
struct run16 {
    __amdgpu_buffer_rsrc_t rsrc;
    __device__ __forceinline__ explicit run16(void* base)
        : rsrc(__builtin_amdgcn_make_buffer_rsrc(base, 0, 0xFFFFFFFFu, 0x00020000u)) {}
    __device__ __forceinline__ void store(unsigned lane_off, unsigned group_off, uint4 v) const {
        u32x4_t w = {v.x, v.y, v.z, v.w};
        __builtin_amdgcn_raw_buffer_store_b128(w, rsrc, (int)lane_off, (int)group_off, 0);
    }
    __device__ __forceinline__ void store4(unsigned lane_off, unsigned group_off, uint32_t v) const {
        __builtin_amdgcn_raw_buffer_store_b32(v, rsrc, (int)lane_off, (int)group_off, 0);
    }
};

// EXACT = false: the hardware square root (v_sqrt_f32: exact for 84.95 % of all inputs, 1 ulp off for the rest,
// never more -- tools/microbench/sqrt_check.hip; the reference's own torch.norm is 1 ulp away from this formula on
// ~11 % of entries).  EXACT = true: correctly rounded (sqrt_rn_mk), 22 more VALU instructions per 16-byte slot, which
// on the devices that can store at 7 TB/s costs 10-16 % of K1's speed (profiles/r01_k1_ab_sqrt_mk.log).
// The subtractions and squares of x and y are one packed operation each (v_pk_add_f32 / v_pk_mul_f32 on the even-aligned
// (x, y) half of the float4 a ds_read_b128 delivers: two lanes of work per issue slot).  Every operation rounds exactly as
// its scalar twin and the sum keeps the reference's order (sx + sy) + sz, so the values are those of the scalar formula.
template <bool EXACT>
__device__ __forceinline__ float dist_pp(float4 p, float4 q) {
    const f32x2 dxy = f32x2{p.x, p.y} - f32x2{q.x, q.y};
    const f32x2 sxy = dxy * dxy;
    const float dz = p.z - q.z, sz = dz * dz;
    const float x = (sxy.x + sxy.y) + sz;
    return EXACT ? sqrt_rn_mk(x) : __builtin_amdgcn_sqrtf(x);
}

// four bits -> four bytes of 0/1 (bit k lands in byte k)
__device__ __forceinline__ uint32_t spread4(uint32_t nib) { return (nib * 0x00204081u) & 0x01010101u; }

// JT: column residues per tile (multiple of 16 so both planes split into whole 16-byte slots)
// DA / MA: every (i, j-tile) run of the distance / mask plane starts on a 16-byte
// boundary and is a whole number of 16-byte slots (N % 4 == 0 resp. N % 16 == 0
// and an aligned base pointer) -- the branch-free path.
template <int JT, bool NT, bool DA, bool MA, bool EXACT>
__global__ __launch_bounds__(256) void k1_pairdist_a15(const float* __restrict__ xyz,
                                                       const uint8_t* __restrict__ amask,
                                                       float* __restrict__ dist, uint8_t* __restrict__ dmask,
                                                       int N, int row_begin, int row_end, int out_rows,
                                                       int out_row_origin, int IR) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // LDS carve: xj[JT*15] float4 | xi[IR*15] float4 | mj[JT+1] u32 | mi[IR] u32
    float4* sxj = reinterpret_cast<float4*>(smem);
    float4* sxi = sxj + JT * A15;
    uint32_t* smj = reinterpret_cast<uint32_t*>(sxi + IR * A15);
    uint32_t* smi = smj + (JT + 4);

    const int tid = threadIdx.x;
    const int b = blockIdx.z;
    const int j0 = blockIdx.x * JT;
    const int jn = min(JT, N - j0);
    const int i0 = row_begin + blockIdx.y * IR;
    const int in = min(IR, row_end - i0);

    // ---- stage the coordinate tiles (coalesced dword loads of the flat rows) ----
    {
        const float* gj = xyz + ((size_t)b * N + j0) * (A15 * 3);
        float* lj = reinterpret_cast<float*>(sxj);
        for (int f = tid; f < jn * (A15 * 3); f += 256) {
            int atom = f / 3, comp = f - atom * 3;
            lj[atom * 4 + comp] = gj[f];
        }
        const float* gi = xyz + ((size_t)b * N + i0) * (A15 * 3);
        float* li = reinterpret_cast<float*>(sxi);
        for (int f = tid; f < in * (A15 * 3); f += 256) {
            int atom = f / 3, comp = f - atom * 3;
            li[atom * 4 + comp] = gi[f];
        }
        // 15 mask bytes of a residue -> 15 bits
        for (int r = tid; r < JT + 4 + IR; r += 256) {
            bool is_j = r < JT + 4;
            int rl = is_j ? r : r - (JT + 4);
            bool valid = is_j ? (rl < jn) : (rl < in);
            uint32_t bits = 0;
            if (valid) {
                if (amask) {
                    const uint8_t* m = amask + ((size_t)b * N + (is_j ? j0 : i0) + rl) * A15;
#pragma unroll
                    for (int c = 0; c < A15; ++c) bits |= (m[c] != 0 ? 1u : 0u) << c;
                } else {
                    bits = 0x7FFFu;
                }
            }
            (is_j ? smj : smi)[rl] = bits;
        }
    }
    __syncthreads();

    const unsigned nE = (unsigned)jn * AA15;  // elements of one (i, j-tile) run

    // =========================== distance plane ===========================
    if (dist) {
        for (int il = 0; il < in; ++il) {
            const size_t obase = (((size_t)b * out_rows + (size_t)(i0 + il - out_row_origin)) * N + j0) * AA15;
            float* orow = dist + obase;
            // misalignment of the run in floats; slot q covers elements [4q - sh, 4q - sh + 4)
            const unsigned sh = (unsigned)((reinterpret_cast<uintptr_t>(orow) >> 2) & 3u);
            const unsigned nQ = (nE + sh + 3u) >> 2;
            const unsigned ibase = (unsigned)il * A15;
            if (DA) {
                for (unsigned q = tid; q < (nE >> 2); q += 256) {
                    const unsigned e0 = 4u * q;
                    const unsigned jl = e0 / AA15;
                    const unsigned r = e0 - jl * AA15;
                    unsigned a = r / A15;
                    unsigned c = r - a * A15;
                    unsigned ja = jl * A15 + c;
                    unsigned ia = ibase + a;
                    float v[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        v[t] = dist_pp<EXACT>(sxi[ia], sxj[ja]);
                        // advance (a, c) -> next element; carries are selects, not branches
                        const bool wc = (c == A15 - 1);
                        const bool wa = wc && (a == A15 - 1);
                        c = wc ? 0u : c + 1u;
                        a = wa ? 0u : (wc ? a + 1u : a);
                        ja = wa ? ja + 1u : (wc ? ja - (A15 - 1) : ja + 1u);
                        ia = wa ? ia - (A15 - 1) : (wc ? ia + 1u : ia);
                    }
                    uint4 u = make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]),
                                         __float_as_uint(v[3]));
                    store16<NT>(orow + e0, u);
                }
                continue;
            }
            for (unsigned q = tid; q < nQ; q += 256) {
                const int e0 = (int)(4u * q) - (int)sh;
                const unsigned ef = e0 < 0 ? 0u : (unsigned)e0;  // first in-range element of the slot
                unsigned jl = ef / AA15;
                unsigned r = ef - jl * AA15;
                unsigned a = r / A15;
                unsigned c = r - a * A15;
                unsigned ja = jl * A15 + c;
                unsigned ia = ibase + a;
                float v[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int e = e0 + t;
                    if (e >= 0 && (unsigned)e < nE) {
                        v[t] = dist_pp<EXACT>(sxi[ia], sxj[ja]);
                        ++c;
                        ++ja;
                        if (c == A15) {
                            c = 0;
                            ja -= A15;
                            ++a;
                            ++ia;
                            if (a == A15) {
                                a = 0;
                                ia -= A15;
                                ja += A15;
                            }
                        }
                    } else {
                        v[t] = 0.f;
                    }
                }
                if (e0 >= 0 && (unsigned)(e0 + 4) <= nE) {
                    uint4 u = make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]),
                                         __float_as_uint(v[3]));
                    store16<NT>(orow + e0, u);
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int e = e0 + t;
                        if (e >= 0 && (unsigned)e < nE) orow[e] = v[t];
                    }
                }
            }
        }
    }

    // ============================= mask plane =============================
    if (dmask) {
        for (int il = 0; il < in; ++il) {
            const size_t obase = (((size_t)b * out_rows + (size_t)(i0 + il - out_row_origin)) * N + j0) * AA15;
            uint8_t* orow = dmask + obase;
            const unsigned sh = (unsigned)(reinterpret_cast<uintptr_t>(orow) & 15u);
            const unsigned nQ = (nE + sh + 15u) >> 4;
            const uint32_t mib = smi[il];
            if (MA) {
                for (unsigned q = tid; q < (nE >> 4); q += 256) {
                    const unsigned e0 = 16u * q;
                    const unsigned jl = e0 / AA15;
                    const unsigned r = e0 - jl * AA15;
                    const unsigned a = r / A15;
                    const unsigned c = r - a * A15;
                    const bool wa = (a == A15 - 1);
                    const unsigned a1 = wa ? 0u : a + 1u;
                    const unsigned jl1 = wa ? jl + 1u : jl;
                    const uint32_t row0 = ((mib >> a) & 1u) ? smj[jl] : 0u;
                    const uint32_t row1 = ((mib >> a1) & 1u) ? smj[jl1] : 0u;
                    const uint32_t win = (row0 | (row1 << 15)) >> c;
                    uint4 u = make_uint4(spread4(win & 15u), spread4((win >> 4) & 15u), spread4((win >> 8) & 15u),
                                         spread4((win >> 12) & 15u));
                    store16<NT>(orow + e0, u);
                }
                continue;
            }
            for (unsigned q = tid; q < nQ; q += 256) {
                const int e0 = (int)(16u * q) - (int)sh;
                const unsigned ef = e0 < 0 ? 0u : (unsigned)e0;
                const unsigned lead = (unsigned)((int)ef - e0);  // slot bytes before the run starts
                unsigned jl = ef / AA15;
                unsigned r = ef - jl * AA15;
                unsigned a = r / A15;
                unsigned c = r - a * A15;
                // 16-bit window starting at bit c of row (jl,a) | row(next) << 15 [| row(next2) << 30]
                unsigned a1 = a + 1, jl1 = jl;
                if (a1 == A15) {
                    a1 = 0;
                    ++jl1;
                }
                uint32_t row0 = ((mib >> a) & 1u) ? smj[jl] : 0u;
                uint32_t row1 = ((mib >> a1) & 1u) ? smj[jl1] : 0u;
                uint32_t win = ((row0 | (row1 << 15)) >> c) & 0xFFFFu;  // c + 16 <= 30
                // bits past the end of this run belong to nobody (rows >= jn hold 0 bits)
                if (e0 >= 0 && (unsigned)(e0 + 16) <= nE) {
                    uint4 u = make_uint4(spread4(win & 15u), spread4((win >> 4) & 15u), spread4((win >> 8) & 15u),
                                         spread4((win >> 12) & 15u));
                    store16<NT>(orow + e0, u);
                } else {
                    for (unsigned t = lead; t < 16u; ++t) {
                        const unsigned e = (unsigned)(e0 + (int)t);
                        if (e < nE) orow[e] = (uint8_t)((win >> (t - lead)) & 1u);
                    }
                }
            }
        }
    }
}

// ---- pattern kernel (aligned shapes: N % 16 == 0, 16-byte aligned planes) ----
// 4 column residues are 900 floats = exactly 225 float4 slots, and 16 column
// residues are 3600 mask bytes = exactly 225 16-byte slots.  So if lane t
// (t < 225) always takes slot t of a group, its four elements keep the SAME
// (j offset, a, c) in every group and every row: the index decode is a constant
// of the lane (a compile-time table, K1_PAT), the row atoms xi[a] it needs live
// in registers for a whole row, and the inner loop is LDS read (immediate
// offsets) -> 4 distances (x / y packed) -> one 16-byte buffer-addressed store
// with no vector integer arithmetic at all.  31 of 256 lanes idle in the sweep
// (they still help with staging).
//
// LDS image: a residue is 16 float4 slots (15 atoms + 1 pad) = 256 bytes = one
// full row of the 64 LDS banks, and atoms are fetched with ds_read_b128 (the
// loads are volatile so the compiler cannot narrow them to the 8-cycle b96
// form).  Lanes of one 16-lane service group that read the same residue then
// touch distinct banks (different atoms) or broadcast (same atom): conflict-free
// except where a group straddles two residues.
constexpr int RS = 16;  // float4 slots per staged residue

typedef float f32x4_t __attribute__((ext_vector_type(4)));

typedef const volatile f32x4_t __attribute__((address_space(3))) * lds_f32x4_ptr;

__device__ __forceinline__ float4 lds_atom(const float4* p) {
    const f32x4_t v = *(lds_f32x4_ptr)(p);  // generic -> LDS address space, volatile: stays one ds_read_b128
    return make_float4(v.x, v.y, v.z, v.w);
}

// MATH: 0 = product arithmetic with the hardware sqrt, 1 = with the correctly rounded sqrt (cfg.exact_sqrt).
// Builds made with -DPS_EXPERIMENTS (tools/ only, never the product library) add two timing experiments:
//       2 = stores without any arithmetic (WRONG values);  3 = the first correctly rounded routine (same values as 1).
template <int MATH>
__device__ __forceinline__ float dist_pp_m(float4 p, float4 q) {
#ifdef PS_EXPERIMENTS
    if (MATH == 3) {
        float dx = p.x - q.x, dy = p.y - q.y, dz = p.z - q.z;
        float sx = dx * dx, sy = dy * dy, sz = dz * dz;
        return sqrt_rn_pos((sx + sy) + sz);
    }
#else
    static_assert(MATH == 0 || MATH == 1, "experiment modes are compiled only with -DPS_EXPERIMENTS");
#endif
    return MATH == 0 ? dist_pp<false>(p, q) : dist_pp<true>(p, q);
}

// The pattern kernel's per-lane slot decode is the same for every workgroup, so it is a compile-time table
// (one 16-byte load per lane instead of ~90 VALU instructions of divisions by 225 and 15):
//   offj: byte k = column atom (jo * RS + c) of element 4 t + k inside a 4-residue group   (float4 slot index in LDS)
//   ai:   byte k = row atom a of that element
//   mask: the lane's 16-byte mask window inside a 16-residue group starts at byte 16 t = (jo, a, c); bits 0-3 jo,
//         4-7 jo1, 8-11 a, 12-15 a1, 16-19 c, 20 wa, where (jo1, a1) is the (column residue, row atom) the window runs on
//         into and wa says that jo1 = jo + 1 (a is the last row atom)
struct pat_lane_t { uint32_t offj, ai, mask, pad; };
struct pat_table_t { pat_lane_t lane[256]; };
constexpr pat_table_t make_pat_table() {
    pat_table_t t{};
    for (unsigned tid = 0; tid < 256; ++tid) {
        pat_lane_t l{0, 0, 0, 0};
        for (unsigned k = 0; k < 4; ++k) {
            const unsigned e = (4 * tid + k) % 900;   // lanes >= 225 are idle; keep their entries in range
            const unsigned jo = e / AA15, r = e % AA15, a = r / A15, c = r % A15;
            l.offj |= (jo * 16 + c) << (8 * k);
            l.ai |= a << (8 * k);
        }
        const unsigned e0 = (16 * tid) % 3600;
        const unsigned jo = e0 / AA15, r = e0 % AA15, a = r / A15, c = r % A15;
        const bool wa = (a == A15 - 1);
        const unsigned a1 = wa ? 0u : a + 1u, jo1 = wa ? jo + 1u : jo;   // jo1 <= 15: the last lane's window ends with its group
        l.mask = jo | ((jo1 & 15u) << 4) | (a << 8) | (a1 << 12) | (c << 16) | ((wa ? 1u : 0u) << 20);
        t.lane[tid] = l;
    }
    return t;
}
__device__ const pat_table_t K1_PAT = make_pat_table();

struct __attribute__((packed, aligned(4))) xyz12 { float x, y, z; };   // one atom of the (.., A, 3) input: 12 bytes, 4-byte aligned

// the 15 mask bits of staged residue r from the pattern kernel's bit stream (bit 15 r + c = atom c of residue r)
__device__ __forceinline__ uint32_t res_bits(const uint32_t* sbits, unsigned r) {
    const unsigned o = r * 15u;
    return __funnelshift_r(sbits[o >> 5], sbits[(o >> 5) + 1], o & 31u) & 0x7FFFu;
}

template <int JT, bool NT, int MATH, bool UNROLL>
__global__ __launch_bounds__(256) void k1_pairdist_a15_pat(const float* __restrict__ xyz,
                                                           const uint8_t* __restrict__ amask,
                                                           float* __restrict__ dist, uint8_t* __restrict__ dmask,
                                                           int N, int row_begin, int row_end, int out_rows,
                                                           int out_row_origin, int IR, int n_tiles, int n_ichunks,
                                                           int xcd_remap) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* sxj = reinterpret_cast<float4*>(smem);
    float4* sxi = sxj + JT * RS;
    constexpr int NPU = ((JT + 1) * A15 + 255) / 256;                // staging passes that cover JT column residues + one row residue
    uint32_t* sbits = reinterpret_cast<uint32_t*>(sxi + IR * RS);    // mask bit stream: 8 words per 256 staged atoms, + 1

    const int tid = threadIdx.x;
    // 1-D grid.  Workgroups are dealt round-robin to the 8 XCDs (ids equal mod 8 share an XCD and its L2), so with
    // xcd_remap each XCD sweeps its own contiguous eighth of the output: neighbouring runs -- which share 128-byte
    // lines where a run is not line-aligned -- then meet in ONE L2, and every L2 streams a sequential address range.
    // Store-only microbenchmark: 5.58 -> 5.90-5.99 TB/s at this granule (profiles/r01_store_microbench_10*.log).
    // Placement is only a speed matter: any map from workgroup id to run is correct.
    unsigned w = blockIdx.x;
    if (xcd_remap) w = (w & 7u) * (gridDim.x >> 3) + (w >> 3);   // host guarantees gridDim.x % 8 == 0 when set
    const unsigned tile = w % (unsigned)n_tiles, rest = w / (unsigned)n_tiles;
    const int b = (int)(rest / (unsigned)n_ichunks);
    const int j0 = (int)tile * JT;
    const int jn = min(JT, N - j0);  // multiple of 16
    const int i0 = row_begin + (int)(rest % (unsigned)n_ichunks) * IR;
    const int in = min(IR, row_end - i0);

    // Staging, one atom per lane and pass: a 12-byte global load and one 16-byte LDS write.  The staged residues (jn
    // column residues, then the `in` row residues) form one atom stream u = 0 .. 15 (jn + in) - 1; the loads of its first
    // NPU passes -- all of it when the workgroup has one row -- are issued before anything waits on them, so a
    // workgroup pays one memory latency.  Atom u's mask bit goes through a wave ballot into a bit stream (bit u), read
    // back 15 bits at a time by res_bits().  (Round 3: the float-per-lane loop this replaces spent ~17 VALU
    // instructions, four of them quarter-rate multiplies, per staged FLOAT -- a quarter of the kernel's vector time.)
    const pat_lane_t lane_pat = K1_PAT.lane[tid];   // this lane's fixed slot decode (a constant table; used after the barrier)
    {
        const int lane = tid & 63, wbase = tid - lane;
        const unsigned nj = (unsigned)jn * A15, total = nj + (unsigned)in * A15;
        const float* gb = xyz + (size_t)b * N * (A15 * 3);                       // this structure
        const uint8_t* mb = amask ? amask + (size_t)b * N * A15 : nullptr;
        const unsigned res_j = (unsigned)j0, res_i = (unsigned)i0 - (unsigned)jn;   // atom u lives in residue res_x + u / 15 ...
        const unsigned slot_i = 16u * (unsigned)(JT - jn);                       // ... and in LDS slot u + u / 15 (+ slot_i)
        auto src_atom = [&](unsigned u) { return (u < nj ? res_j : res_i) * A15 + u; };   // atom index in the structure
        auto dst_slot = [&](unsigned u) { return u + (__umul24(u, 34953u) >> 19) + (u < nj ? 0u : slot_i); };
        xyz12 pa[NPU];
        uint32_t ma[NPU];
#pragma unroll
        for (int p = 0; p < NPU; ++p) {
            const unsigned u = (unsigned)(p * 256 + tid);
            ma[p] = 0;
            if (u < total) {
                const unsigned at = src_atom(u);
                if (dist) pa[p] = *reinterpret_cast<const xyz12*>(gb + 3u * at);   // (a mask-only launch needs no coordinates)
                ma[p] = mb ? (uint32_t)mb[at] : 1u;
            }
        }
#pragma unroll
        for (int p = 0; p < NPU; ++p) {
            const unsigned u = (unsigned)(p * 256 + tid);
            if (dist && u < total) sxj[dst_slot(u)] = make_float4(pa[p].x, pa[p].y, pa[p].z, 0.f);
            const unsigned long long bal = __ballot(ma[p] != 0);
            if (lane < 2) sbits[((p * 256 + wbase) >> 5) + lane] = lane ? (uint32_t)(bal >> 32) : (uint32_t)bal;
        }
        for (unsigned base = NPU * 256; base < total; base += 256) {   // more than one row per workgroup (uniform trip count)
            const unsigned u = base + (unsigned)tid;
            uint32_t m = 0;
            if (u < total) {
                const unsigned at = src_atom(u);
                if (dist) {
                    const xyz12 q = *reinterpret_cast<const xyz12*>(gb + 3u * at);
                    sxj[dst_slot(u)] = make_float4(q.x, q.y, q.z, 0.f);
                }
                m = mb ? (uint32_t)mb[at] : 1u;
            }
            const unsigned long long bal = __ballot(m != 0);
            if (lane < 2) sbits[((base + wbase) >> 5) + lane] = lane ? (uint32_t)(bal >> 32) : (uint32_t)bal;
        }
    }
    __syncthreads();
    if (tid >= AA15) return;  // no barrier below

    const size_t row0 = ((size_t)b * out_rows + (size_t)(i0 - out_row_origin)) * N + j0;  // pair index of (i0, j0)

    if (dist) {
        unsigned offj[4], ai[4];   // element k of this lane's slot: column atom (jo * RS + c) of the group, row atom a
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            offj[k] = (lane_pat.offj >> (8 * k)) & 0xFFu;
            ai[k] = (lane_pat.ai >> (8 * k)) & 0xFFu;
        }
        const int ngroups = jn >> 2;
        for (int il = 0; il < in; ++il) {
            float4 pi[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) pi[k] = sxi[il * RS + ai[k]];
            float* ob = dist + (row0 + (size_t)il * N) * AA15;   // uniform: the row's run of this tile
            const run16 run(ob);
            const unsigned lo = 4u * (unsigned)tid;
            const float4* xj = sxj;
            auto group = [&](int g) {
                uint4 u;
#ifdef PS_EXPERIMENTS
                if (MATH == 2) {  // store-only timing run: WRONG values by design
                    u = make_uint4(__float_as_uint(pi[0].x), __float_as_uint(pi[1].x), __float_as_uint(pi[2].x),
                                   (unsigned)g);
                } else
#endif
                {
                    const float4* x = xj + g * (4 * RS);
                    const float4 q0 = lds_atom(x + offj[0]), q1 = lds_atom(x + offj[1]);
                    const float4 q2 = lds_atom(x + offj[2]), q3 = lds_atom(x + offj[3]);
                    u.x = __float_as_uint(dist_pp_m<MATH>(pi[0], q0));
                    u.y = __float_as_uint(dist_pp_m<MATH>(pi[1], q1));
                    u.z = __float_as_uint(dist_pp_m<MATH>(pi[2], q2));
                    u.w = __float_as_uint(dist_pp_m<MATH>(pi[3], q3));
                }
                if (NT) store16<true>(ob + (size_t)g * (4 * AA15) + lo, u);
                else run.store(4u * lo, (unsigned)g * (16u * AA15), u);
            };
            if (UNROLL && ngroups == JT / 4) {  // full tile: straight-line code, stores issued back to back
#pragma unroll
                for (int g = 0; g < JT / 4; ++g) group(g);
            } else {
#pragma unroll 4
                for (int g = 0; g < ngroups; ++g) group(g);
            }
        }
    }

    if (dmask) {
        // This lane's 16-byte window of a 16-residue group starts at byte (column residue jo, row atom a, column atom c)
        // and runs on into row atom a1 -- of the same column residue, or, when a is the last atom ("wa"), of the next one.
        // Column residues jo and jo + 1 are neighbours in the bit stream, so ONE 32-bit extraction per group gives
        // both: comp = [15 bits for row atom a | 15 bits for row atom a1], to be masked per row with the two row-atom bits.
        const unsigned jo = lane_pat.mask & 15u, a = (lane_pat.mask >> 8) & 15u, a1 = (lane_pat.mask >> 12) & 15u;
        const unsigned c = (lane_pat.mask >> 16) & 15u;
        const bool wa = (lane_pat.mask >> 20) & 1u;
        constexpr int NG = JT / 16;
        uint32_t comp[NG];
        const int ngroups = jn >> 4;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            comp[g] = 0u;
            if (g < ngroups) {
                const unsigned o = (unsigned)(g * 16 + (int)jo) * 15u;
                const uint32_t w = __funnelshift_r(sbits[o >> 5], sbits[(o >> 5) + 1], o & 31u);   // residues jo, jo + 1, ...
                const uint32_t m0 = w & 0x7FFFu;
                comp[g] = wa ? (w & 0x3FFFFFFFu) : (m0 | (m0 << 15));
            }
        }
        for (int il = 0; il < in; ++il) {
            const uint32_t mib = res_bits(sbits, (unsigned)(jn + il));
            const uint32_t km = (((mib >> a) & 1u) ? 0x7FFFu : 0u) | (((mib >> a1) & 1u) ? 0x3FFF8000u : 0u);
            uint8_t* ob = dmask + (row0 + (size_t)il * N) * AA15;
            const run16 run(ob);
            const unsigned lo = 16u * (unsigned)tid;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g < ngroups) {
                    const uint32_t win = (comp[g] & km) >> c;
                    uint4 u = make_uint4(spread4(win & 15u), spread4((win >> 4) & 15u), spread4((win >> 8) & 15u),
                                         spread4((win >> 12) & 15u));
                    if (NT) store16<true>(ob + (size_t)g * (16 * AA15) + lo, u);
                    else run.store(lo, (unsigned)g * (16u * AA15), u);
                }
            }
        }
    }
}

// ---- flat pattern kernel (any N >= 16, 16-byte aligned planes) ----
// The same fixed-lane pattern, laid over the FLAT pair axis P = (b*out_rows + il)*N + j instead of over one row:
// four consecutive pairs are 225 float4 slots and sixteen consecutive pairs are 225 16-byte mask slots whatever
// N is, because a pair is 225 elements and the planes start 16-byte aligned.  A workgroup takes chunks of FL = 128
// consecutive pairs.  Its LDS image is indexed by PAIR POSITION: slot p holds the column residue of pair P0 + p
// (j wraps at N and moves on to the next structure), so the inner loop is the pattern kernel's.  What changes:
//   * a chunk touches up to FR rows; the row atoms a lane needs are re-read from LDS at each row change, and the
//     (at most one per row) 4-pair group that straddles two rows takes an element-wise path;
//   * the mask uses a per-pair-position word (column bits | row bits << 16), so its 16-pair groups need no row
//     bookkeeping at all;
//   * [pbeg, pend) need not be 16-pair aligned (odd N; one structure of a row-sharded launch): partially active
//     groups take the element-wise path and inactive pairs are never written.
// FLOG2: log2(pairs per chunk).  The launcher's default is 6 (64 pairs = 72 KB + 18 KB of output per chunk); 7 (128
// pairs) and the small granules 4..5 of round 3's bounded A/B are cfg.flat_fl_log2 (profiles/r03_k1_ab_small_granule.log).
constexpr int FR = 16;   // row residues staged per chunk (N >= 16 -> a chunk of up to 128 pairs touches at most 9 rows)

// floor(x / d) for x, d < 2^23 with rcp = 1.0f / d
__device__ __forceinline__ unsigned udiv_rcp(unsigned x, unsigned d, float rcp) {
    unsigned q = (unsigned)((float)x * rcp);
    const int r = (int)x - (int)(q * d);
    if (r < 0) --q;
    else if (r >= (int)d) ++q;
    return q;
}

template <bool EXACT, bool HASMASK, int FLOG2>
__global__ __launch_bounds__(256, 4) void k1_pairdist_a15_flat(const float* __restrict__ xyz,
                                                            const uint8_t* __restrict__ amask,
                                                            float* __restrict__ dist, uint8_t* __restrict__ dmask,
                                                            int B, int N, int out_rows, int out_row_origin,
                                                            unsigned pbeg, unsigned pend, unsigned n_ranges,
                                                            unsigned range_stride, unsigned cpr, int cpw,
                                                            int xcd_remap, double rcpN_d, double rcpR_d) {
    constexpr int FL = 1 << FLOG2;
    __shared__ __attribute__((aligned(16))) float4 sxj[FL * RS];
    __shared__ __attribute__((aligned(16))) float4 sxi[FR * RS];
    __shared__ uint32_t smj[FL], smi[FR], smc[FL];

    const int tid = threadIdx.x;
    const pat_lane_t lane_pat = K1_PAT.lane[tid];   // this lane's fixed slot decode (constant table)
    unsigned w = blockIdx.x;
    if (xcd_remap) {  // XCD x (= w % 8) sweeps a contiguous share of the chunks; see the pattern kernel
        const unsigned n = gridDim.x, x = w & 7u;
        w = x * (n >> 3) + min(x, n & 7u) + (w >> 3);
    }
    const float rcpN = 1.0f / (float)N, rcpR = 1.0f / (float)out_rows;
    const int pl = tid >> 4, c16 = tid & 15;  // staging: 16 lanes per residue

    // The launch covers n_ranges pair ranges [pbeg, pend) + r * range_stride (one range, or the same rows of every
    // structure of a row-sharded launch), each cut into at most cpr chunks on the 128-pair grid.
    const unsigned n_chunks = n_ranges * cpr;
    for (int cc = 0; cc < cpw; ++cc) {
        const unsigned chunk = w * (unsigned)cpw + (unsigned)cc;
        if (chunk >= n_chunks) break;  // uniform
        if (cc) __syncthreads();       // the previous chunk's LDS image is still being read
        unsigned rg = 0, k = chunk;
        if (n_ranges > 1) {
            rg = chunk / cpr;
            k = chunk - rg * cpr;
        }
        const unsigned rbeg = pbeg + rg * range_stride, rend = pend + rg * range_stride;
        const unsigned P0 = ((rbeg >> FLOG2) + k) << FLOG2;
        if (P0 >= rend) continue;  // uniform: cpr is an upper bound when ranges start at different phases
        const int lo = rbeg > P0 ? (int)(rbeg - P0) : 0;
        const int hi = rend - P0 < (unsigned)FL ? (int)(rend - P0) : FL;
        // row / structure of the chunk's first pair: exact via fp64 (P0 < 2^32) with a +-1 fix-up
        unsigned R0 = (unsigned)((double)P0 * rcpN_d);
        if (R0 * (unsigned long long)N > P0) --R0;
        else if ((R0 + 1ull) * N <= P0) ++R0;
        const int j_start = (int)(P0 - R0 * (unsigned)N);
        unsigned b0 = (unsigned)((double)R0 * rcpR_d);
        if (b0 * (unsigned long long)out_rows > R0) --b0;
        else if ((b0 + 1ull) * out_rows <= R0) ++b0;
        const unsigned il0 = R0 - b0 * (unsigned)out_rows;
        const int nr = (j_start + FL - 1) / N + 1;  // rows the chunk touches

        // ---- stage: column residue of every pair position, the touched row residues, their mask bits ----
        // All global loads are issued before the first LDS write so the chunk pays one memory round trip, and the
        // (row, column, structure) of a lane's pair position is walked from pass to pass (p grows by 16 <= N, so at
        // most one row change per pass) instead of divided out again.
        float vx[FL / 16 + 1], vy[FL / 16 + 1], vz[FL / 16 + 1];
        unsigned vm[FL / 16 + 1];  // raw mask bytes: compared only after every load has been issued
        bool va[FL / 16 + 1];
        {
            unsigned rl = udiv_rcp((unsigned)(j_start + pl), (unsigned)N, rcpN);
            unsigned j = (unsigned)(j_start + pl) - rl * (unsigned)N;
            const unsigned db = udiv_rcp(il0 + rl, (unsigned)out_rows, rcpR);
            unsigned il = il0 + rl - db * (unsigned)out_rows;
            unsigned res0 = (b0 + db) * (unsigned)N;  // first residue of the pair's structure
#pragma unroll
            for (int pass = 0; pass < FL / 16; ++pass) {
                const int p = pass * 16 + pl;
                va[pass] = (p >= lo) && (p < hi) && (c16 < A15);
                const unsigned src = va[pass] ? (res0 + j) * A15 + c16 : 0u;  // inactive lanes re-read atom 0
                vx[pass] = xyz[src * 3u + 0];
                vy[pass] = xyz[src * 3u + 1];
                vz[pass] = xyz[src * 3u + 2];
                vm[pass] = HASMASK ? (unsigned)amask[src] : 1u;
                j += 16;
                if (j >= (unsigned)N) {
                    j -= (unsigned)N;
                    if (++il == (unsigned)out_rows) {
                        il = 0;
                        res0 += (unsigned)N;
                    }
                }
            }
        }
        {
            const unsigned ilr = il0 + (unsigned)pl;
            const unsigned db = udiv_rcp(ilr, (unsigned)out_rows, rcpR);
            const unsigned bb = b0 + db;
            const unsigned i = ilr - db * (unsigned)out_rows + (unsigned)out_row_origin;
            constexpr int L = FL / 16;
            va[L] = (pl < nr) && (bb < (unsigned)B) && (c16 < A15);
            const unsigned src = va[L] ? (bb * (unsigned)N + i) * A15 + c16 : 0u;
            vx[L] = xyz[src * 3u + 0];
            vy[L] = xyz[src * 3u + 1];
            vz[L] = xyz[src * 3u + 2];
            vm[L] = HASMASK ? (unsigned)amask[src] : 1u;
        }
#pragma unroll
        for (int pass = 0; pass < FL / 16; ++pass) {
            const int p = pass * 16 + pl;
            sxj[p * RS + c16] = va[pass] ? make_float4(vx[pass], vy[pass], vz[pass], 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
            const unsigned long long bal = __ballot(va[pass] && vm[pass] != 0u);
            if (c16 == 0) smj[p] = (uint32_t)(bal >> (16 * (pl & 3))) & 0x7FFFu;
        }
        {
            constexpr int L = FL / 16;
            sxi[pl * RS + c16] = va[L] ? make_float4(vx[L], vy[L], vz[L], 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
            const unsigned long long bal = __ballot(va[L] && vm[L] != 0u);
            if (c16 == 0) smi[pl] = (uint32_t)(bal >> (16 * (pl & 3))) & 0x7FFFu;
        }
        __syncthreads();
        if (dmask && tid < FL) {
            const unsigned rl = udiv_rcp((unsigned)(j_start + tid), (unsigned)N, rcpN);
            smc[tid] = (tid >= lo && tid < hi) ? (smj[tid] | (smi[rl] << 16)) : 0u;
        }
        __syncthreads();
        if (tid >= AA15) continue;  // idle in the sweep; rejoins at the next chunk's barrier

        if (dist) {
            unsigned offj[4], ai[4], jo[4];   // element k of this lane's slot (the pattern kernel's table, K1_PAT)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                offj[k] = (lane_pat.offj >> (8 * k)) & 0xFFu;   // jo * RS + c
                ai[k] = (lane_pat.ai >> (8 * k)) & 0xFFu;
                jo[k] = offj[k] >> 4;
            }
            float* o = dist + (size_t)P0 * AA15 + 4u * tid;
            const run16 run(dist + (size_t)P0 * AA15);   // the chunk's run: uniform base, scalar group offsets (see run16)
            int rl = 0;                // row of pair 4g
            int nb = N - j_start;      // pair position where row rl + 1 starts
            int g = 0;
            while (g < FL / 4) {
                const int p = 4 * g;
                if (p >= hi) break;
                while (nb <= p) {
                    ++rl;
                    nb += N;
                }
                const int pe = min(nb, hi);
                const int nfast = (p >= lo) ? (pe - p) >> 2 : 0;
                if (nfast > 0) {  // whole groups inside one row and inside the active range
                    float4 pi[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) pi[k] = sxi[rl * RS + ai[k]];
#pragma unroll 4
                    for (int q = 0; q < nfast; ++q) {
                        const float4* x = sxj + (g + q) * (4 * RS);
                        const float4 q0 = lds_atom(x + offj[0]), q1 = lds_atom(x + offj[1]);
                        const float4 q2 = lds_atom(x + offj[2]), q3 = lds_atom(x + offj[3]);
                        uint4 u;
                        u.x = __float_as_uint(dist_pp<EXACT>(pi[0], q0));
                        u.y = __float_as_uint(dist_pp<EXACT>(pi[1], q1));
                        u.z = __float_as_uint(dist_pp<EXACT>(pi[2], q2));
                        u.w = __float_as_uint(dist_pp<EXACT>(pi[3], q3));
                        run.store(16u * (unsigned)tid, (unsigned)(g + q) * (16u * AA15), u);
                    }
                    g += nfast;
                    continue;
                }
                if (p + 4 > lo) {  // straddles a row change or the edge of the active range: element-wise
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int pp = p + (int)jo[k];
                        if (pp >= lo && pp < hi) {
                            const int rk = rl + (pp >= nb ? 1 : 0);  // N >= 16: at most one row change per group
                            const float4 xi = sxi[rk * RS + ai[k]];
                            const float4 xq = sxj[g * (4 * RS) + offj[k]];
                            o[(size_t)g * (4 * AA15) + k] = dist_pp<EXACT>(xi, xq);
                        }
                    }
                }
                ++g;
            }
        }

        if (dmask) {
            // this lane's 16-byte window of a 16-pair group (K1_PAT: jo, jo1, a, a1, c)
            const unsigned jo = lane_pat.mask & 15u, jo1 = (lane_pat.mask >> 4) & 15u, a = (lane_pat.mask >> 8) & 15u;
            const unsigned a1 = (lane_pat.mask >> 12) & 15u, c = (lane_pat.mask >> 16) & 15u;
            uint8_t* o = dmask + (size_t)P0 * AA15 + 16u * tid;
            const run16 run(dmask + (size_t)P0 * AA15);
#pragma unroll
            for (int g = 0; g < FL / 16; ++g) {
                const int pg = 16 * g;
                if (pg < hi && pg + 16 > lo) {
                    const uint32_t w0 = smc[pg + jo], w1 = smc[pg + jo1];
                    const uint32_t row0 = ((w0 >> (16u + a)) & 1u) ? (w0 & 0x7FFFu) : 0u;
                    const uint32_t row1 = ((w1 >> (16u + a1)) & 1u) ? (w1 & 0x7FFFu) : 0u;
                    const uint32_t win = ((row0 | (row1 << 15)) >> c) & 0xFFFFu;
                    if (pg >= lo && pg + 16 <= hi) {
                        uint4 u = make_uint4(spread4(win & 15u), spread4((win >> 4) & 15u), spread4((win >> 8) & 15u),
                                             spread4((win >> 12) & 15u));
                        run.store(16u * (unsigned)tid, (unsigned)g * (16u * AA15), u);
                    } else {
                        for (unsigned t = 0; t < 16u; ++t) {
                            const int pp = pg + (int)((c + t < (unsigned)A15) ? jo : jo1);
                            if (pp >= lo && pp < hi) o[(size_t)g * (16 * AA15) + t] = (uint8_t)((win >> t) & 1u);
                        }
                    }
                }
            }
        }
    }
}

// ---- fixed-A flat pattern kernel (compile-time EVEN atom count, any N >= 16, 16-byte aligned planes) ----
// The A = 15 flat pattern kernel generalised to a compile-time atom count A with A*A >= 129: atom14 and the other even
// counts (`from_xyz` accepts any A, protstruc.py:94-128, tests/test_StructureBatch.py:11-21).  Four consecutive pairs
// are A*A float4 slots and sixteen consecutive pairs are A*A 16-byte mask slots for EVERY A (a pair is A*A elements and
// the planes start 16-byte aligned), so the fixed-lane pattern carries over: slot s of a group always decodes to the same
// (pair offset, a, c), and chunks of whole pairs are 128-byte-line aligned for any N.  What changes with A:
//   * A*A can exceed the 256 lanes: lane t owns slots t, t + 256, ... (SPL = ceil(A*A / 256) per group).  The sweep
//     takes one slot set at a time (u outer, the chunk's groups inner), so only ONE slot's pattern -- four column-atom
//     offsets, four row atoms -- is live in registers whatever SPL is;
//   * staging deals LPR = next power of two >= A lanes to a residue, and a residue's LDS image is RS float4 slots
//     (16 for A <= 16: the atoms of one residue on distinct banks; A for larger A);
//   * mask rows are A bits wide: a 16-byte mask slot spans up to 2 + 14/A rows; words are 64-bit from A = 32.
// Odd counts take the row-phase kernel (faster there); bit-identical to it and to the element-per-lane kernel
// (tests/test_gpu_k1_fuzz.py).
template <int A>
struct FlatA {
    static_assert(A >= 12 && A <= 64 && (A % 2 == 0 || A == 15), "fixed-A flat kernel: even atom counts (and 15 as the cross-check)");
    static constexpr int AA = A * A;
    static constexpr int LPR = A <= 16 ? 16 : (A <= 32 ? 32 : 64);       // lanes per staged residue
    static constexpr int RPP = 256 / LPR;                                 // residues staged per pass
    static constexpr int RS = A <= 16 ? 16 : A;                           // float4 slots per staged residue
    static constexpr int SPL = (AA + 255) / 256;                          // slots per lane per group
    // pairs per chunk: 128 / 64 / 32 / 16 by atom count (A/B-tested in round 2: profiles/r02_k1_ab_chunk_length.log)
    static constexpr int FL_LOG2 = A <= 16 ? 7 : (A <= 24 ? 6 : (A <= 40 ? 5 : 4));
    static constexpr int FLn = 1 << FL_LOG2;                              // pairs per chunk
    static constexpr int NR = (15 + FLn - 1) / 16 + 1;                    // rows a chunk can touch (N >= 16)
    static constexpr int FRn = ((NR + RPP - 1) / RPP) * RPP;              // row residues staged per chunk
    static constexpr int MROWS = 2 + 14 / A;                              // mask rows a 16-byte slot can span
    // (A = 32 needs the wide word too: the all-atoms mask is (1 << A) - 1)
    typedef typename std::conditional<(A >= 32), unsigned long long, uint32_t>::type mask_t;
};

template <int A, bool EXACT, bool HASMASK>
__global__ __launch_bounds__(256, 4) void k1_pairdist_flatA(const float* __restrict__ xyz,
                                                           const uint8_t* __restrict__ amask,
                                                           float* __restrict__ dist, uint8_t* __restrict__ dmask,
                                                           int B, int N, int out_rows, int out_row_origin,
                                                           unsigned pbeg, unsigned pend, unsigned n_ranges,
                                                           unsigned range_stride, unsigned cpr, int cpw,
                                                           int xcd_remap, double rcpN_d, double rcpR_d) {
    using G = FlatA<A>;
    constexpr int AA = G::AA, RSn = G::RS, FLn = G::FLn, FRn = G::FRn, LPR = G::LPR, RPP = G::RPP, SPL = G::SPL;
    constexpr int NPASS = FLn / RPP, RPASS = FRn / RPP;
    typedef typename G::mask_t mask_t;
    __shared__ __attribute__((aligned(16))) float4 sxj[FLn * RSn];
    __shared__ __attribute__((aligned(16))) float4 sxi[FRn * RSn];
    __shared__ mask_t smj[FLn], smi[FRn], smr[FLn];   // column bits / row bits per staged residue; row bits per pair position

    const int tid = threadIdx.x;
    unsigned w = blockIdx.x;
    if (xcd_remap) {
        const unsigned n = gridDim.x, x = w & 7u;
        w = x * (n >> 3) + min(x, n & 7u) + (w >> 3);
    }
    const float rcpN = 1.0f / (float)N, rcpR = 1.0f / (float)out_rows;
    const int pl = tid / LPR, cl = tid % LPR;          // staging: LPR lanes per residue
    const unsigned cs = (unsigned)cl;                  // LDS slot of this lane's atom
    const mask_t abits = (A >= 64) ? ~(mask_t)0 : (((mask_t)1 << (A & 63)) - 1);

    const unsigned n_chunks = n_ranges * cpr;
    for (int cc = 0; cc < cpw; ++cc) {
        const unsigned chunk = w * (unsigned)cpw + (unsigned)cc;
        if (chunk >= n_chunks) break;  // uniform
        if (cc) __syncthreads();
        unsigned rg = 0, k = chunk;
        if (n_ranges > 1) {
            rg = chunk / cpr;
            k = chunk - rg * cpr;
        }
        const unsigned rbeg = pbeg + rg * range_stride, rend = pend + rg * range_stride;
        const unsigned P0 = ((rbeg >> G::FL_LOG2) + k) << G::FL_LOG2;
        if (P0 >= rend) continue;  // uniform
        const int lo = rbeg > P0 ? (int)(rbeg - P0) : 0;
        const int hi = rend - P0 < (unsigned)FLn ? (int)(rend - P0) : FLn;
        unsigned R0 = (unsigned)((double)P0 * rcpN_d);
        if (R0 * (unsigned long long)N > P0) --R0;
        else if ((R0 + 1ull) * N <= P0) ++R0;
        const int j_start = (int)(P0 - R0 * (unsigned)N);
        unsigned b0 = (unsigned)((double)R0 * rcpR_d);
        if (b0 * (unsigned long long)out_rows > R0) --b0;
        else if ((b0 + 1ull) * out_rows <= R0) ++b0;
        const unsigned il0 = R0 - b0 * (unsigned)out_rows;
        const int nr = (j_start + FLn - 1) / N + 1;  // rows the chunk touches (<= G::NR)

        // ---- stage (all global loads before the first LDS write; (row, column, structure) walked, not divided) ----
        float vx[NPASS + RPASS], vy[NPASS + RPASS], vz[NPASS + RPASS];
        unsigned vm[NPASS + RPASS];
        bool va[NPASS + RPASS];
        {
            unsigned rl = udiv_rcp((unsigned)(j_start + pl), (unsigned)N, rcpN);
            unsigned j = (unsigned)(j_start + pl) - rl * (unsigned)N;
            const unsigned db = udiv_rcp(il0 + rl, (unsigned)out_rows, rcpR);
            unsigned il = il0 + rl - db * (unsigned)out_rows;
            unsigned res0 = (b0 + db) * (unsigned)N;
#pragma unroll
            for (int pass = 0; pass < NPASS; ++pass) {
                const int p = pass * RPP + pl;
                va[pass] = (p >= lo) && (p < hi) && (cl < A);
                const unsigned src = va[pass] ? (res0 + j) * A + cl : 0u;
                vx[pass] = xyz[src * 3u + 0];
                vy[pass] = xyz[src * 3u + 1];
                vz[pass] = xyz[src * 3u + 2];
                vm[pass] = HASMASK ? (unsigned)amask[src] : 1u;
                j += RPP;                      // RPP <= 16 <= N: at most one row change per pass
                if (j >= (unsigned)N) {
                    j -= (unsigned)N;
                    if (++il == (unsigned)out_rows) {
                        il = 0;
                        res0 += (unsigned)N;
                    }
                }
            }
        }
#pragma unroll
        for (int rp = 0; rp < RPASS; ++rp) {
            const unsigned rr = (unsigned)(rp * RPP + pl);
            const unsigned ilr = il0 + rr;
            const unsigned db = udiv_rcp(ilr, (unsigned)out_rows, rcpR);
            const unsigned bb = b0 + db;
            const unsigned i = ilr - db * (unsigned)out_rows + (unsigned)out_row_origin;
            const int L = NPASS + rp;
            va[L] = ((int)rr < nr) && (bb < (unsigned)B) && (cl < A);
            const unsigned src = va[L] ? (bb * (unsigned)N + i) * A + cl : 0u;
            vx[L] = xyz[src * 3u + 0];
            vy[L] = xyz[src * 3u + 1];
            vz[L] = xyz[src * 3u + 2];
            vm[L] = HASMASK ? (unsigned)amask[src] : 1u;
        }
        constexpr int FIELDS = 64 / LPR;   // residues per wave
#pragma unroll
        for (int pass = 0; pass < NPASS + RPASS; ++pass) {
            const bool is_row = pass >= NPASS;
            const int p = (is_row ? pass - NPASS : pass) * RPP + pl;
            if (cl < RSn)
                (is_row ? sxi : sxj)[p * RSn + cs] =
                    va[pass] ? make_float4(vx[pass], vy[pass], vz[pass], 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
            const unsigned long long bal = __ballot(va[pass] && vm[pass] != 0u);
            if (cl == 0) (is_row ? smi : smj)[p] = (mask_t)(bal >> (LPR * (pl % FIELDS))) & abits;
        }
        __syncthreads();
        if (dmask && tid < FLn) {
            const unsigned rl = udiv_rcp((unsigned)(j_start + tid), (unsigned)N, rcpN);
            smr[tid] = (tid >= lo && tid < hi) ? smi[rl] : (mask_t)0;   // smj is already 0 outside [lo, hi)
        }
        __syncthreads();

        if (dist) {
            // One slot set at a time (u outer, groups inner): only one slot's pattern -- four column-atom offsets and
            // four row atoms -- is live in registers, whatever SPL is.  A wave's store is still 1 KB contiguous.
#pragma unroll 1
            for (int u = 0; u < SPL; ++u) {
                const unsigned sl = (unsigned)tid + 256u * (unsigned)u;   // slot inside a 4-pair group
                if (sl >= (unsigned)AA) break;
                unsigned offj[4], ai[4], jo[4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const unsigned e = 4u * sl + kk;
                    jo[kk] = e / AA;
                    const unsigned r = e - jo[kk] * AA;
                    const unsigned a = r / A, c = r - a * A;
                    offj[kk] = jo[kk] * RSn + c;
                    ai[kk] = a;
                }
                float* o = dist + (size_t)P0 * AA + 4u * sl;
                const run16 run(dist + (size_t)P0 * AA);   // the chunk's run: uniform base, scalar group offsets (see run16)
                int rl = 0;                // row of pair 4g
                int nb = N - j_start;      // pair position where row rl + 1 starts
                int g = 0;
                while (g < FLn / 4) {
                    const int p = 4 * g;
                    if (p >= hi) break;
                    while (nb <= p) {
                        ++rl;
                        nb += N;
                    }
                    const int pe = min(nb, hi);
                    const int nfast = (p >= lo) ? (pe - p) >> 2 : 0;
                    if (nfast > 0) {  // whole groups inside one row and inside the active range
                        float4 pi[4];
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) pi[kk] = sxi[rl * RSn + ai[kk]];
#pragma unroll 4
                        for (int q = 0; q < nfast; ++q) {
                            const float4* x = sxj + (g + q) * (4 * RSn);
                            const float4 q0 = lds_atom(x + offj[0]), q1 = lds_atom(x + offj[1]);
                            const float4 q2 = lds_atom(x + offj[2]), q3 = lds_atom(x + offj[3]);
                            uint4 v;
                            v.x = __float_as_uint(dist_pp<EXACT>(pi[0], q0));
                            v.y = __float_as_uint(dist_pp<EXACT>(pi[1], q1));
                            v.z = __float_as_uint(dist_pp<EXACT>(pi[2], q2));
                            v.w = __float_as_uint(dist_pp<EXACT>(pi[3], q3));
                            run.store(16u * (unsigned)sl, (unsigned)(g + q) * (16u * AA), v);
                        }
                        g += nfast;
                        continue;
                    }
                    if (p + 4 > lo) {  // straddles a row change or the edge of the active range: element-wise
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) {
                            const int pp = p + (int)jo[kk];
                            if (pp >= lo && pp < hi) {
                                const int rk = rl + (pp >= nb ? 1 : 0);  // N >= 16: at most one row change per group
                                const float4 xi = sxi[rk * RSn + ai[kk]];
                                const float4 xq = sxj[g * (4 * RSn) + offj[kk]];
                                o[(size_t)g * (4 * AA) + kk] = dist_pp<EXACT>(xi, xq);
                            }
                        }
                    }
                    ++g;
                }
            }
        }

        if (dmask) {
            constexpr int MR = G::MROWS;
#pragma unroll 1
            for (int u = 0; u < SPL; ++u) {
                if ((unsigned)tid + 256u * (unsigned)u >= (unsigned)AA) break;
                const unsigned e0 = 16u * ((unsigned)tid + 256u * (unsigned)u);  // byte inside a 16-pair group
                const unsigned jo = e0 / AA, r = e0 - jo * AA;
                const unsigned a = r / A, c = r - a * A;
                unsigned prow[MR], arow[MR];
                int sh[MR];   // bit position of row m's first bit inside the 16-bit window (negative: starts before it)
#pragma unroll
                for (int m = 0; m < MR; ++m) {
                    const unsigned am = a + m;
                    const bool wp = am >= (unsigned)A;
                    prow[m] = min(jo + (wp ? 1u : 0u), 15u);   // rows with sh >= 16 contribute nothing (index clamped)
                    arow[m] = wp ? am - A : am;
                    sh[m] = m * A - (int)c;
                }
                uint8_t* o = dmask + (size_t)P0 * AA + e0;
                const run16 run(dmask + (size_t)P0 * AA);
#pragma unroll 1   // unrolled, the eight groups' row words stay live at once and the kernel spills
                for (int mg = 0; mg < FLn / 16; ++mg) {
                    const int pg = 16 * mg;
                    if (pg < hi && pg + 16 > lo) {
                        uint32_t win = 0;
#pragma unroll
                        for (int m = 0; m < MR; ++m) {
                            if (sh[m] < 16) {
                                const mask_t col = smj[pg + prow[m]], row = smr[pg + prow[m]];
                                const mask_t bits = ((row >> arow[m]) & 1) ? col : (mask_t)0;
                                win |= sh[m] <= 0 ? (uint32_t)(bits >> (-sh[m])) : ((uint32_t)bits << sh[m]);
                            }
                        }
                        win &= 0xFFFFu;
                        uint8_t* og = o + (size_t)mg * (16 * AA);
                        if (pg >= lo && pg + 16 <= hi) {
                            run.store(e0, (unsigned)mg * (16u * AA), make_uint4(spread4(win & 15u), spread4((win >> 4) & 15u),
                                                                                spread4((win >> 8) & 15u), spread4((win >> 12) & 15u)));
                        } else {
                            for (unsigned t = 0; t < 16u; ++t) {
                                const int pp = pg + (int)jo + ((r + t >= (unsigned)AA) ? 1 : 0);
                                if (pp >= lo && pp < hi) og[t] = (uint8_t)((win >> t) & 1u);
                            }
                        }
                    }
                }
            }
        }
    }
}

// ---- row-tile kernel for the small even atom counts whose pair block is a whole number of 16-byte slots (A = 4, 8) ----
// With A*A a multiple of 16 every pair's distance block (A*A floats) and mask block (A*A bytes) starts 16-byte aligned
// for ANY N, and a float4 slot never leaves one row atom a and one column residue j.  So the lane -> (j, a, c..c+3)
// pattern can be laid over a ROW instead of over pair positions: a workgroup owns IR rows x JT column residues of one
// structure, a lane owns the same slots of every row, and -- the point of this kernel -- its four COLUMN atoms stay in
// registers for all IR rows: the inner loop is one LDS read (the row atom) + four distances + one 16-byte store,
// against four LDS reads per slot and a per-chunk re-staging of every pair position in the flat kernels (whose
// staging is O(A) per pair against O(A^2) output: for A = 4 that is 64 bytes of LDS image per 80 bytes of output).
// A run of one row is JT*A*A*4 = 8 KB (mask: 2 KB); rows are N*A*A*4 bytes apart.  Row-sharded launches need nothing
// special (rows are the outer axis, as in the A = 15 pattern kernel).
template <int A>
struct RowTile {
    static_assert(A == 4 || A == 8, "row-tile kernel: A*A must be a multiple of 16 and a slot must stay inside one row atom");
    static constexpr int AA = A * A;
    static constexpr int JT = 2048 / AA;               // column residues per tile: 512 float4 slots = 8 KB per row
    static constexpr int SPL = 2;                      // slots per lane per row
    static constexpr int MS = JT * AA / 16;            // 16-byte mask slots per row = 128
    static constexpr int RPS = 16 / A;                 // row atoms (a) per mask slot
};

template <int A, bool EXACT>
__global__ __launch_bounds__(256) void k1_pairdist_rowtile(const float* __restrict__ xyz,
                                                           const uint8_t* __restrict__ amask,
                                                           float* __restrict__ dist, uint8_t* __restrict__ dmask,
                                                           int N, int row_begin, int row_end, int out_rows,
                                                           int out_row_origin, int IR, int n_tiles, int n_ichunks,
                                                           int xcd_remap) {
    using T = RowTile<A>;
    constexpr int AA = T::AA, JT = T::JT;
    extern __shared__ __attribute__((aligned(16))) char smem_rt[];
    float4* sxj = reinterpret_cast<float4*>(smem_rt);           // [JT * A]
    float4* sxi = sxj + JT * A;                                  // [IR * A]
    uint32_t* smj = reinterpret_cast<uint32_t*>(sxi + IR * A);  // [JT] column mask bits
    uint32_t* smi = smj + JT;                                    // [IR] row mask bits

    const int tid = threadIdx.x;
    unsigned w = blockIdx.x;
    if (xcd_remap) w = (w & 7u) * (gridDim.x >> 3) + (w >> 3);   // host guarantees gridDim.x % 8 == 0 when set
    const unsigned tile = w % (unsigned)n_tiles, rest = w / (unsigned)n_tiles;
    const int b = (int)(rest / (unsigned)n_ichunks);
    const int j0 = (int)tile * JT;
    const int jn = min(JT, N - j0);
    const int i0 = row_begin + (int)(rest % (unsigned)n_ichunks) * IR;
    const int in = min(IR, row_end - i0);

    {   // stage: coalesced dword loads of the flat coordinate rows, one float4 slot per atom
        const float* gj = xyz + ((size_t)b * N + j0) * (A * 3);
        float* lj = reinterpret_cast<float*>(sxj);
        for (int f = tid; f < jn * (A * 3); f += 256) {
            const int atom = f / 3, comp = f - atom * 3;
            lj[atom * 4 + comp] = gj[f];
        }
        const float* gi = xyz + ((size_t)b * N + i0) * (A * 3);
        float* li = reinterpret_cast<float*>(sxi);
        for (int f = tid; f < in * (A * 3); f += 256) {
            const int atom = f / 3, comp = f - atom * 3;
            li[atom * 4 + comp] = gi[f];
        }
        for (int r = tid; r < JT + IR; r += 256) {
            const bool is_j = r < JT;
            const int rl = is_j ? r : r - JT;
            const bool valid = is_j ? (rl < jn) : (rl < in);
            uint32_t bits = 0;
            if (valid) {
                if (amask) {
                    const uint8_t* m = amask + ((size_t)b * N + (is_j ? j0 : i0) + rl) * A;
#pragma unroll
                    for (int c = 0; c < A; ++c) bits |= (m[c] != 0 ? 1u : 0u) << c;
                } else {
                    bits = (1u << A) - 1u;
                }
            }
            (is_j ? smj : smi)[rl] = bits;
        }
    }
    __syncthreads();

    const size_t row0 = ((size_t)b * out_rows + (size_t)(i0 - out_row_origin)) * N + j0;  // pair index of (i0, j0)
    const size_t row_stride = (size_t)N * AA;

    if (dist) {
        // slot s = tid + 256 u of the row run: elements 4 s .. 4 s + 3 = (j, a, c0 .. c0 + 3), the same in every row
        float4 q[T::SPL][4];
        unsigned ai[T::SPL];
        bool act[T::SPL];
#pragma unroll
        for (int u = 0; u < T::SPL; ++u) {
            const unsigned e = 4u * ((unsigned)tid + 256u * u);
            const unsigned j = e / AA, r = e - j * AA;
            ai[u] = r / A;
            const unsigned c0 = r - ai[u] * A;
            act[u] = (int)j < jn;
            const unsigned jj = act[u] ? j : 0u;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) q[u][kk] = sxj[jj * A + c0 + kk];   // the column atoms: registers for all rows
        }
        float* orow = dist + row0 * AA;   // uniform: stores go the buffer way (run16), one descriptor per row
#pragma unroll 2
        for (int il = 0; il < in; ++il) {
            const run16 run(orow);
#pragma unroll
            for (int u = 0; u < T::SPL; ++u) {
                const float4 pi = sxi[il * A + ai[u]];
                uint4 v;
                v.x = __float_as_uint(dist_pp<EXACT>(pi, q[u][0]));
                v.y = __float_as_uint(dist_pp<EXACT>(pi, q[u][1]));
                v.z = __float_as_uint(dist_pp<EXACT>(pi, q[u][2]));
                v.w = __float_as_uint(dist_pp<EXACT>(pi, q[u][3]));
                if (act[u]) run.store(16u * (unsigned)tid, 4096u * (unsigned)u, v);
            }
            orow += row_stride;
        }
    }

    if (dmask) {
        // 128 mask slots per row: lane t takes slot t % 128 of the rows with parity t / 128
        const int ms = tid & (T::MS - 1), par = __builtin_amdgcn_readfirstlane(tid / T::MS);   // (uniform over a wave)
        const unsigned e0 = 16u * (unsigned)ms;
        const unsigned j = e0 / AA, a0 = (e0 - j * AA) / A;   // the slot covers row atoms a0 .. a0 + RPS - 1 of pair j
        if ((int)j < jn) {
            const uint32_t mj = smj[j];
            uint8_t* orow = dmask + (row0 + (size_t)par * N) * AA;
            for (int il = par; il < in; il += 2) {
                const run16 run(orow);
                const uint32_t mi = smi[il] >> a0;
                uint32_t win = 0;
#pragma unroll
                for (int t = 0; t < T::RPS; ++t) win |= ((mi >> t) & 1u) ? (mj << (A * t)) : 0u;
                run.store(e0, 0u, make_uint4(spread4(win & 15u), spread4((win >> 4) & 15u), spread4((win >> 8) & 15u),
                                             spread4((win >> 12) & 15u)));
                orow += 2 * row_stride;
            }
        }
    }
}

// ---- row-phase kernel: every atom count up to 64 that has no better kernel, ANY length ----
// The column-stationary idea of the row-tile kernels without their alignment conditions.  A row run of the distance plane
// starts (R * N * A*A) mod 4 floats past a 16-byte boundary (R = absolute row of the buffer); for even A that is always 0,
// for odd A it is one of four PHASES fixed by R mod 4.  Slots are cut on the ABSOLUTE 16-byte grid: slot s of a row with
// phase ph holds row elements 4 s - ph .. 4 s - ph + 3.  A lane owns slot s (two of them, 256 apart) of every row of its
// workgroup, keeps the column atoms of the SEVEN elements 4 s - 3 .. 4 s + 3 in registers (four when there is one phase
// only), and -- the phase being uniform over the workgroup -- takes one of four straight-line arms per row that uses the
// window 3 - ph .. 6 - ph of them.  So, unlike the phased variant of round 2's odd row-tile kernel that this replaced (rows
// of one residue class of R mod 4 per workgroup), a workgroup writes IR CONSECUTIVE rows.  It also replaced that round's
// small and odd fixed-A flat kernels, k1_mask_rows and the any-A flat kernel (same-process A/Bs in profiles/r03_*).
//   * tiles are cut in slot space, so a slot never belongs to two tiles; the only shared slots are the one that holds a
//     row's end and the next row's start, which both rows write element-wise (their own elements only);
//   * the mask plane rides in the same loop: the four mask bytes of a slot's elements are one aligned dword store (byte
//     offset = float offset of the slot), so it needs neither a 16-byte phase of its own nor a second launch;
//   * the four row atoms of a slot's elements are LDS reads (broadcasts: many lanes read the same few atoms) whose
//     fourth component carries the atom's mask bit; for A = 1 there is a single row atom, read once per row;
//   * everything per row (phase, row pointer, LDS row address) is wave-uniform and rides in SGPRs.
// ACT > 0: the atom count is the compile-time constant ACT (the small counts, 25 and atom37: index decode with constant divisors);
// ACT = 0 / -1: a RUN-TIME even / odd atom count up to 64 (same kernel; the decode -- once per workgroup -- divides at run
// time, the column mask words are 64 bits wide).  Only the parity of A decides the code shape (one phase or four).
template <int ACT>
struct RowPhase {
    static_assert(ACT >= -1 && ACT <= 64, "row-phase kernel: atom counts up to 64");
    static constexpr bool PHASED = ACT > 0 ? ((ACT * ACT) % 4 != 0) : (ACT == -1);   // odd A: A*A = 1 (mod 4)
    static constexpr int W = PHASED ? 7 : 4;           // elements per slot whose column atoms a lane keeps
    static constexpr int W0 = PHASED ? 3 : 0;          // window element of row element 4 s
    static constexpr int SPL = 2;                      // slots per lane per row
    static constexpr int TS = 256 * SPL;               // slots per tile-row at most (8 KB of distances)
    typedef typename std::conditional<(ACT > 0 && ACT < 32), uint32_t, unsigned long long>::type bits_t;   // A mask bits of a residue
};

// column residues a tile's elements can touch (host and device use the same formula for the LDS carve)
// -- never more than the chain has: a CA trace of 125 residues stages 2 KB, not the 33 KB of a full tile, and LDS stops
// being what caps the resident workgroups of short chains
__host__ __device__ inline int rowphase_maxres(int A, int N) { return min((4 * 512 + 2) / (A * A) + 2, N); }

// One slot of one row.  `xi_row`: LDS address of the row residue's atoms (uniform over the wave); od / om point at the slot
// (float / byte offset 4 s - ph of the row run), rund / runm + `so` address the same slot the buffer way (uniform row base
// in a descriptor + this lane's constant element offset: full slots spend no vector work on addresses); PH = the row's
// phase.  The row atom of every element arrives as one
// ds_read_b128 whose fourth component is that atom's mask bit (0 / 1 as an integer), so the mask costs no second lookup.
// HEAD (A = 1 only, the slots of u = 0): the window elements in front of the row's first element stand for the previous
// row's last elements (see the kernel), whose row atom is the previous row's: `xim_off` = -16 in the lane that owns slot 0
// (the previous row's atom sits right in front of this row's in LDS), 0 in every other lane.
template <int ACT, bool EXACT, int PH, bool HEAD>
__device__ __forceinline__ void rowphase_slot(const char* __restrict__ xi_row, const float (&col)[7][3],
                                              const uint32_t (&aoff)[7], uint32_t cm, uint32_t valid,
                                              float* __restrict__ od_, uint8_t* __restrict__ om_, bool wd, bool wm,
                                              const run16& rund, const run16& runm, unsigned so, int xim_off,
                                              bool skip_tail) {
    // wd / wm: whether the distance / mask plane is produced (uniform); the pointers are only meaningful when set
    float* od = wd ? od_ : nullptr;
    uint8_t* om = wm ? om_ : nullptr;
    using T = RowPhase<ACT>;
    constexpr int WO = T::W0 - PH;                     // window element of the slot's first element
    const uint32_t vm = (valid >> WO) & 15u;
    if (vm == 0u) return;
    float v[4];
    uint32_t rowbytes = 0;                             // byte k = mask bit of element k's row atom
    float4 xi0, xim;
    if (ACT == 1) {
        xi0 = *reinterpret_cast<const float4*>(xi_row);
        if (HEAD && PH > 0) xim = *reinterpret_cast<const float4*>(xi_row + xim_off);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float4 xi = (ACT == 1) ? ((HEAD && k < PH) ? xim : xi0) : *reinterpret_cast<const float4*>(xi_row + aoff[WO + k]);
        v[k] = dist_pp<EXACT>(xi, make_float4(col[WO + k][0], col[WO + k][1], col[WO + k][2], 0.f));
        rowbytes |= __float_as_uint(xi.w) << (8 * k);
    }
    const uint32_t mw = rowbytes & spread4((cm >> WO) & 15u);
    if (vm == 15u) {
        if (wd) rund.store(4u * so, 0u, make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]),
                                                   __float_as_uint(v[3])));
        if (wm) runm.store4(so, 0u, mw);
    } else {   // the slot that holds a row's start or end: this row's elements only.  skip_tail (uniform, A = 1): the next
               // row of this row chunk writes the slot whole (its head slot carries this row's last elements)
        if (ACT == 1 && skip_tail && !(vm & 8u)) return;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if ((vm >> k) & 1u) {
                if (wd) od[k] = v[k];
                if (wm) om[k] = (uint8_t)((mw >> (8 * k)) & 1u);
            }
    }
}

template <int ACT, bool EXACT>
__global__ __launch_bounds__(256) void k1_pairdist_rowphase(const float* __restrict__ xyz,
                                                            const uint8_t* __restrict__ amask,
                                                            float* __restrict__ dist, uint8_t* __restrict__ dmask,
                                                            int N, int A_rt, int row_begin, int row_end, int out_rows,
                                                            int out_row_origin, int IR, int n_tiles, int spt,
                                                            int n_ichunks, int lpg_log2, int xcd_remap, int flags,
                                                            unsigned rcpAA, unsigned rcpA) {
    using T = RowPhase<ACT>;
    typedef typename T::bits_t bits_t;
    constexpr int W = T::W;
    const int A = ACT > 0 ? ACT : A_rt;                                 // a constant wherever ACT > 0
    const int AA = A * A;
    const int maxres = rowphase_maxres(A, N);
    extern __shared__ __attribute__((aligned(16))) char smem_rp[];
    float4* sxi = reinterpret_cast<float4*>(smem_rp) + (ACT == 1 ? 1 : 0);   // [IR * A] row atoms: x, y, z, mask bit (A = 1: one
                                                                        // atom of padding in front, "row -1" of the seam logic)
    bits_t* smj = reinterpret_cast<bits_t*>(sxi + IR * A);              // [maxres] column mask bits
    float* sxj = reinterpret_cast<float*>(smj + maxres);                // [maxres * A * 3] column coordinates, as in HBM

    const int tid = threadIdx.x;
    unsigned w = blockIdx.x;
    if (xcd_remap) {
        const unsigned n = gridDim.x, x = w & 7u;
        w = x * (n >> 3) + min(x, n & 7u) + (w >> 3);
    }
    const unsigned tile = w % (unsigned)n_tiles, rest = w / (unsigned)n_tiles;
    const int b = (int)(rest / (unsigned)n_ichunks);
    const int i0 = row_begin + (int)(rest % (unsigned)n_ichunks) * IR;
    const int in = min(IR, row_end - i0);
    const int nel = N * AA;                                             // elements of one row run
    // Seams (A = 1).  Where a row run does not end on the 16-byte grid, one slot holds the end of row R and the start of
    // row R + 1.  Written from both sides it costs two visits of element-wise stores (up to 3 dword + 3 byte store
    // instructions each, one lane active): with the 2 KB rows of a CA trace that was 6 of the 10 store instructions of a row
    // (N = 501).  So the window elements in FRONT of a row's first element (t < 0: only the lane that owns slot 0 has them)
    // stand for the previous row's last elements -- column atoms of residues N + t, row atom = the previous row's, valid for
    // every row but the first of the chunk (`validp`) -- which makes the head slot a full slot like any other, and the
    // previous row skips its partial tail slot (skip_tail); only the chunk's first head and last tail are still written
    // element-wise.  (For the longer rows of A >= 3 the seams are 6 % of the store instructions and the same change
    // measured +-1 %, for 44 more registers: not taken.)  flags bit 0 [diagnostic]: off, as in round 3.
    const bool merge = ACT == 1 && (nel & 3) != 0 && nel >= 8 && !(flags & 1);
    const int nslots = (nel + 3 + ((nel & 3) == 0 ? 0 : (nel & 3) == 2 ? 2 : 3)) / 4;   // slots a row can touch, over the phases that occur
    const int s0 = (int)tile * spt, s1 = min(s0 + spt, nslots);
    // Short rows (a CA trace of 512 residues is 128 slots): the 256 lanes split into 256 >> lpg_log2 row groups of
    // LPG lanes (whole waves); group g takes rows g, g + G, ... so that no lane idles.  A lane's slots are LPG apart.
    const int LPG = 1 << lpg_log2, G = 256 >> lpg_log2;
    const int grp = __builtin_amdgcn_readfirstlane(tid >> lpg_log2);    // uniform over the wave: everything per row is scalar
    const int sl = tid & (LPG - 1);
    // column residues the tile's elements 4 s0 - 3 .. 4 s1 - 1 belong to
    const int t_lo = max(4 * s0 - T::W0, 0), t_hi = min(4 * s1 - 1, nel - 1);
    const int j_lo = t_lo / AA, j_hi = t_hi / AA, nres = j_hi - j_lo + 1;

    {   // stage: the tile's column residues are one contiguous float range of xyz; row atoms one float4 each
        const float* gj = xyz + ((size_t)b * N + j_lo) * (size_t)(A * 3);
        for (int f = tid; f < nres * (A * 3); f += 256) sxj[f] = gj[f];
        const float* gi = xyz + ((size_t)b * N + i0) * (size_t)(A * 3);
        float* li = reinterpret_cast<float*>(sxi);
        for (int f = tid; f < in * (A * 3); f += 256) {
            const int atom = f / 3, comp = f - atom * 3;
            li[atom * 4 + comp] = gi[f];
        }
        if (ACT == 1 && tid == 0) sxi[-1] = make_float4(0.f, 0.f, 0.f, 0.f);   // "row -1": read by the first row's head slot,
                                                     // its mask word is OR-ed in before the valid bytes are picked
        for (int f = tid; f < in * A; f += 256)      // fourth component: the row atom's mask bit
            reinterpret_cast<uint32_t*>(sxi)[f * 4 + 3] = amask ? (amask[((size_t)b * N + i0) * A + f] != 0 ? 1u : 0u) : 1u;
        for (int r = tid; r < nres; r += 256) {
            bits_t bits = A >= 64 ? ~(bits_t)0 : (((bits_t)1 << (A & 63)) - 1);
            if (amask) {
                const uint8_t* m = amask + ((size_t)b * N + j_lo + r) * A;
                bits = 0;
                if (ACT > 0) {
#pragma unroll
                    for (int c = 0; c < (ACT > 0 ? ACT : 1); ++c) bits |= (bits_t)(m[c] != 0 ? 1u : 0u) << c;
                } else {
                    for (int c = 0; c < A; ++c) bits |= (bits_t)(m[c] != 0 ? 1u : 0u) << c;
                }
            }
            smj[r] = bits;
        }
    }
    __syncthreads();

    // per-lane pattern: column atom, row-atom offset, column mask bit and validity of every window element
    float col[T::SPL][7][3];
    uint32_t aoff[T::SPL][7], cm[T::SPL], valid[T::SPL], so[T::SPL];
    uint32_t validp = 0;                               // A = 1: valid[0] with the previous row's last elements switched on
    const int xim_off = (ACT == 1 && merge && s0 + sl == 0) ? -(int)sizeof(float4) : 0;
#pragma unroll
    for (int u = 0; u < T::SPL; ++u) {
        const int s = s0 + LPG * u + sl;
        so[u] = 4u * (unsigned)s;
        cm[u] = valid[u] = 0;
#pragma unroll
        for (int wi = 0; wi < 7; ++wi) {
            col[u][wi][0] = col[u][wi][1] = col[u][wi][2] = 0.f;
            aoff[u][wi] = 0;
            if (wi < W) {
                const int t = 4 * s - T::W0 + wi;
                const bool ok = s < s1 && t >= 0 && t < nel;
                const unsigned e = ok ? (unsigned)t : (unsigned)t_lo;
                // constant divisors when ACT > 0; the run-time instantiations multiply by floor(2^32 / d) and correct once
                // (e < 2^28: the estimate is the quotient or one less) -- 28 run-time divisions per lane were ~15 % of a
                // workgroup's instructions
                unsigned j, r, a, c;
                if (ACT > 0) {
                    j = e / (unsigned)AA, r = e - j * (unsigned)AA;
                    a = r / (unsigned)A, c = r - a * (unsigned)A;
                } else {
                    j = __umulhi(e, rcpAA), r = e - j * (unsigned)AA;
                    if (r >= (unsigned)AA) ++j, r -= (unsigned)AA;
                    a = __umulhi(r, rcpA), c = r - a * (unsigned)A;
                    if (c >= (unsigned)A) ++a, c -= (unsigned)A;
                }
                const float* p = sxj + ((j - (unsigned)j_lo) * A + c) * 3;
                col[u][wi][0] = p[0];
                col[u][wi][1] = p[1];
                col[u][wi][2] = p[2];
                aoff[u][wi] = a * (unsigned)sizeof(float4);
                cm[u] |= (ok ? (uint32_t)((smj[j - (unsigned)j_lo] >> c) & 1u) : 0u) << wi;
                valid[u] |= (ok ? 1u : 0u) << wi;
                if (ACT == 1 && u == 0 && wi < T::W0 && merge && s == 0) {   // t = wi - 3 < 0: residue N + t of the previous row
                    const size_t ja = (size_t)b * N + (size_t)(N + t);      // (A = 1: element = residue; from L2: the last
                    col[u][wi][0] = xyz[ja * 3];                            //  residues are not among tile 0's staged ones
                    col[u][wi][1] = xyz[ja * 3 + 1];                        //  when a row has several tiles)
                    col[u][wi][2] = xyz[ja * 3 + 2];
                    cm[u] |= (amask ? (amask[ja] != 0 ? 1u : 0u) : 1u) << wi;
                    validp |= 1u << wi;
                }
            }
        }
    }
    validp |= valid[0];

    const long long Rb = (long long)b * out_rows - out_row_origin;    // absolute row of the buffer = Rb + i
    const unsigned nel4 = (unsigned)nel & 3u;
#pragma unroll 1
    for (int il = grp; il < in; il += G) {
        const long long R = Rb + i0 + il;
        const unsigned ph = T::PHASED ? (((unsigned)(R & 3) * nel4) & 3u) : 0u;    // (R * nel) mod 4
        const long long base = R * (long long)nel - (long long)ph;     // float index of row element -ph: 16-byte aligned, >= 0
        float* rd = dist + base;          // (only dereferenced when the plane is requested)
        uint8_t* rm = dmask + base;
        const bool wd = dist != nullptr, wm = dmask != nullptr;
        const char* xi_row = reinterpret_cast<const char*>(sxi + il * A);
        const run16 rund(rd), runm(rm);
        const bool skip_tail = merge && il + 1 < in;
#pragma unroll
        for (int u = 0; u < T::SPL; ++u) {
            float* od = rd + so[u];
            uint8_t* om = rm + so[u];
            constexpr bool HEAD = ACT == 1;
            const bool head = HEAD && u == 0;
            const uint32_t vrow = (head && il > 0) ? validp : valid[u];     // (uniform choice)
#define PS_RP_SLOT(PH_, H_) rowphase_slot<ACT, EXACT, PH_, H_>(xi_row, col[u], aoff[u], cm[u], vrow, od, om, wd, wm, rund, runm, (unsigned)so[u], xim_off, skip_tail)
            if constexpr (!T::PHASED) {
                PS_RP_SLOT(0, false);
            } else if (head) {
                switch (ph) {   // uniform over the wave
                    case 0: PS_RP_SLOT(0, HEAD); break;
                    case 1: PS_RP_SLOT(1, HEAD); break;
                    case 2: PS_RP_SLOT(2, HEAD); break;
                    default: PS_RP_SLOT(3, HEAD); break;
                }
            } else {
                switch (ph) {
                    case 0: PS_RP_SLOT(0, false); break;
                    case 1: PS_RP_SLOT(1, false); break;
                    case 2: PS_RP_SLOT(2, false); break;
                    default: PS_RP_SLOT(3, false); break;
                }
            }
#undef PS_RP_SLOT
        }
    }
}

// ---- A = 1, short chains (CA traces of peptides and small proteins, in large batches) ----
// The row-phase kernel keeps everything per row in scalar registers and gives a wave one row at a time: fine for 2 KB rows
// (N = 512), but a 64-residue trace is a 256-byte row -- 16 of a wave's 128 lane-slots -- and the per-row scalar work
// dominates (N = 64: 2.3 TB/s, N = 32: 0.9, N = 16: 0.3; profiles/r04_k1_n_sweep_a1.log).  Here the (B, N, N) output is
// ONE flat run of 16-byte slots; a workgroup owns CS consecutive slots (whatever structures they belong to), stages the
// atoms of those structures in LDS (x, y, z, mask bit per atom) and every lane decodes its slot's first element into
// (structure, i, j) with two reciprocal multiplications, steps through the other three by increment-and-wrap, and gathers
// both atoms of every element from LDS.  ~2x the vector work per slot of the row-phase kernel, but every lane of every
// store instruction is live whatever N is.  Same formula and operand order as every other K1 kernel: same bits.
// Full matrices only (row_begin = 0, row_end = out_rows = N): row-sharded launches keep the row-phase kernel.
constexpr int CA_K = 8;                  // slots per lane
constexpr int CA_CS = 256 * CA_K;        // slots per workgroup: 32 KB of distances + 8 KB of mask

__host__ __device__ inline int ca_max_structs(int N) { return (4 * CA_CS + N * N - 1) / (N * N) + 1; }

template <bool EXACT>
__global__ __launch_bounds__(256) void k1_pairdist_ca_flat(const float* __restrict__ xyz, const uint8_t* __restrict__ amask,
                                                           float* __restrict__ dist, uint8_t* __restrict__ dmask, int B,
                                                           int N, unsigned rcpNN, unsigned rcpN) {
    extern __shared__ __attribute__((aligned(16))) char smem_ca[];
    float4* sat = reinterpret_cast<float4*>(smem_ca);                   // [structures of this workgroup][N]
    const int tid = threadIdx.x;
    const unsigned NN = (unsigned)N * (unsigned)N;
    const unsigned long long total = (unsigned long long)B * NN;        // elements of the whole output
    const unsigned long long e_lo = (unsigned long long)blockIdx.x * (4ull * CA_CS);
    const unsigned long long e_hi = min(e_lo + 4ull * CA_CS, total);
    const unsigned b_lo = (unsigned)(e_lo / NN), b_hi = (unsigned)((e_hi - 1) / NN);
    const unsigned n_at = (b_hi - b_lo + 1u) * (unsigned)N;             // atoms to stage: one contiguous range of xyz
    {
        const float* g = xyz + (size_t)b_lo * N * 3;
        float* l = reinterpret_cast<float*>(sat);
        for (unsigned f = tid; f < n_at * 3u; f += 256u) {
            const unsigned at = f / 3u, comp = f - at * 3u;
            l[at * 4u + comp] = g[f];
        }
        const uint8_t* gm = amask ? amask + (size_t)b_lo * N : nullptr;
        for (unsigned at = tid; at < n_at; at += 256u)
            reinterpret_cast<uint32_t*>(sat)[at * 4u + 3u] = gm ? (gm[at] != 0 ? 1u : 0u) : 1u;
    }
    __syncthreads();
    const unsigned rel0 = (unsigned)(e_lo - (unsigned long long)b_lo * NN);   // < N * N
    const run16 rund(dist + e_lo), runm(dmask + e_lo);                  // (only used when the plane is requested)
    const bool wd = dist != nullptr, wm = dmask != nullptr;
    const char* base = reinterpret_cast<const char*>(sat);
#pragma unroll 2
    for (int k = 0; k < CA_K; ++k) {
        const unsigned sl = (unsigned)k * 256u + (unsigned)tid;         // slot within the workgroup's run
        const unsigned long long e0 = e_lo + 4ull * sl;
        if (e0 >= e_hi) break;
        // first element: rel = element index relative to structure b_lo (< 4 CS + N N: 32 bits)
        const unsigned rel = rel0 + 4u * sl;
        unsigned bl = __umulhi(rel, rcpNN), r = rel - bl * NN;
        if (r >= NN) ++bl, r -= NN;
        unsigned i = __umulhi(r, rcpN), j = r - i * (unsigned)N;
        if (j >= (unsigned)N) ++i, j -= (unsigned)N;
        unsigned row = (bl * (unsigned)N + i) * 16u, col = (bl * (unsigned)N + j) * 16u;   // LDS byte offsets of the two atoms
        const unsigned row_end = (bl * (unsigned)N + (unsigned)N) * 16u;                  // one past the structure's last atom
        float v[4];
        uint32_t mw = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 xi = *reinterpret_cast<const float4*>(base + row);
            const float4 xj = *reinterpret_cast<const float4*>(base + col);
            v[q] = dist_pp<EXACT>(xi, make_float4(xj.x, xj.y, xj.z, 0.f));
            mw |= (__float_as_uint(xi.w) & __float_as_uint(xj.w)) << (8 * q);
            // next element: j + 1; past the row's end -> next row, j = 0; past the structure's last row -> next structure
            col += 16u;
            if (col == row_end) {
                row += 16u;
                col = row_end - (unsigned)N * 16u;
                if (row == row_end) col = row_end;       // (row, col) = first atom of the next structure; its row_end is not
            }                                            // needed: a slot holds 4 elements and N * N >= 4 (N >= 8)
        }
        if (e0 + 4ull <= e_hi) {
            if (wd) rund.store(16u * sl, 0u, make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]),
                                                        __float_as_uint(v[3])));
            if (wm) runm.store4(4u * sl, 0u, mw);
        } else {   // the last slot of the whole output (B * N * N is not a multiple of 4)
            for (int q = 0; q < (int)(e_hi - e0); ++q) {
                if (wd) dist[e0 + q] = v[q];
                if (wm) dmask[e0 + q] = (uint8_t)((mw >> (8 * q)) & 1u);
            }
        }
    }
}

// ---- short chains of a few atoms per residue (backbone / CA+CB peptides in large batches) ----
// The flat idea of k1_pairdist_ca_flat for 2 <= A <= 16: where a row run (N * A * A elements) is a few hundred bytes the
// row-phase and row-tile kernels spend their time on per-row and per-workgroup set-up (A = 3, N = 16: 2.2 TB/s; A = 4, N = 16:
// 2.8; profiles/r04_k1_n_sweep_small_a_short.log).  The (B, N, N, A, A) output is one flat run of 16-byte slots, a workgroup owns
// CA_CS consecutive slots and stages the atoms of the structures they belong to; a lane decodes its slot's first element into
// (structure, i, j, a, c) with four reciprocal multiplications and steps through the other three by increment-and-carry.
template <bool EXACT>
__global__ __launch_bounds__(256) void k1_pairdist_small_flat(const float* __restrict__ xyz, const uint8_t* __restrict__ amask,
                                                              float* __restrict__ dist, uint8_t* __restrict__ dmask, int B,
                                                              int N, int A, unsigned rcpS, unsigned rcpR, unsigned rcpAA,
                                                              unsigned rcpA) {
    extern __shared__ __attribute__((aligned(16))) char smem_sf[];
    float4* sat = reinterpret_cast<float4*>(smem_sf);                   // [structures of this workgroup][N][A]
    const int tid = threadIdx.x;
    const unsigned AA = (unsigned)A * (unsigned)A, R = (unsigned)N * AA, S = (unsigned)N * R;   // elements of a pair / row / structure
    const unsigned long long total = (unsigned long long)B * S;
    const unsigned long long e_lo = (unsigned long long)blockIdx.x * (4ull * CA_CS);
    const unsigned long long e_hi = min(e_lo + 4ull * CA_CS, total);
    const unsigned b_lo = (unsigned)(e_lo / S), b_hi = (unsigned)((e_hi - 1) / S);
    const unsigned NA = (unsigned)N * (unsigned)A, n_at = (b_hi - b_lo + 1u) * NA;   // atoms to stage: one contiguous range of xyz
    {
        const float* g = xyz + (size_t)b_lo * NA * 3;
        float* l = reinterpret_cast<float*>(sat);
        for (unsigned f = tid; f < n_at * 3u; f += 256u) {
            const unsigned at = f / 3u, comp = f - at * 3u;
            l[at * 4u + comp] = g[f];
        }
        const uint8_t* gm = amask ? amask + (size_t)b_lo * NA : nullptr;
        for (unsigned at = tid; at < n_at; at += 256u)
            reinterpret_cast<uint32_t*>(sat)[at * 4u + 3u] = gm ? (gm[at] != 0 ? 1u : 0u) : 1u;
    }
    __syncthreads();
    const unsigned rel0 = (unsigned)(e_lo - (unsigned long long)b_lo * S);   // < S
    const run16 rund(dist + e_lo), runm(dmask + e_lo);                  // (only used when the plane is requested)
    const bool wd = dist != nullptr, wm = dmask != nullptr;
    const char* base = reinterpret_cast<const char*>(sat);
    const unsigned A16 = (unsigned)A * 16u, NA16 = NA * 16u;
#pragma unroll 2
    for (int k = 0; k < CA_K; ++k) {
        const unsigned sl = (unsigned)k * 256u + (unsigned)tid;         // slot within the workgroup's run
        const unsigned long long e0 = e_lo + 4ull * sl;
        if (e0 >= e_hi) break;
        // first element: rel = element index relative to structure b_lo (< 4 CS + S: 32 bits); quotients by floor(2^32 / d)
        // and one correction (rel < 2^28)
        const unsigned rel = rel0 + 4u * sl;
        unsigned bl = __umulhi(rel, rcpS), r = rel - bl * S;
        if (r >= S) ++bl, r -= S;
        unsigned i = __umulhi(r, rcpR), r2 = r - i * R;
        if (r2 >= R) ++i, r2 -= R;
        unsigned j = __umulhi(r2, rcpAA), r3 = r2 - j * AA;
        if (r3 >= AA) ++j, r3 -= AA;
        unsigned a = __umulhi(r3, rcpA), c = r3 - a * (unsigned)A;
        if (c >= (unsigned)A) ++a, c -= (unsigned)A;
        unsigned row = ((bl * (unsigned)N + i) * (unsigned)A + a) * 16u, col = ((bl * (unsigned)N + j) * (unsigned)A + c) * 16u;   // LDS byte offsets
        float v[4];
        uint32_t mw = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 xi = *reinterpret_cast<const float4*>(base + row);
            const float4 xj = *reinterpret_cast<const float4*>(base + col);
            v[q] = dist_pp<EXACT>(xi, make_float4(xj.x, xj.y, xj.z, 0.f));
            mw |= (__float_as_uint(xi.w) & __float_as_uint(xj.w)) << (8 * q);
            // next element: c + 1, carrying into a (next row atom, column atoms start over), j (next column residue), i (next
            // row residue, column residues start over) and the structure (both atoms move on to the next structure's first)
            ++c; col += 16u;
            if (c == (unsigned)A) {
                c = 0; col -= A16; ++a; row += 16u;
                if (a == (unsigned)A) {
                    a = 0; row -= A16; ++j; col += A16;
                    if (j == (unsigned)N) {
                        j = 0; col -= NA16; ++i; row += A16;
                        if (i == (unsigned)N) { i = 0; col += NA16; }
                    }
                }
            }
        }
        if (e0 + 4ull <= e_hi) {
            if (wd) rund.store(16u * sl, 0u, make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]),
                                                        __float_as_uint(v[3])));
            if (wm) runm.store4(4u * sl, 0u, mw);
        } else {   // the last slot of the whole output (B * N * N * A * A is not a multiple of 4)
            for (int q = 0; q < (int)(e_hi - e0); ++q) {
                if (wd) dist[e0 + q] = v[q];
                if (wm) dmask[e0 + q] = (uint8_t)((mw >> (8 * q)) & 1u);
            }
        }
    }
}

// ---- generic A: one output element per lane, runtime decode, scalar stores ----
template <bool EXACT>
__global__ __launch_bounds__(256) void k1_pairdist_generic(const float* __restrict__ xyz,
                                                           const uint8_t* __restrict__ amask,
                                                           float* __restrict__ dist, uint8_t* __restrict__ dmask,
                                                           int N, int A, int row_begin, int row_end, int out_rows,
                                                           int out_row_origin) {
    const int b = blockIdx.z;
    const int i = row_begin + blockIdx.y;
    const unsigned AA = (unsigned)A * A;
    const unsigned nE = (unsigned)N * AA;
    const size_t obase = (((size_t)b * out_rows + (size_t)(i - out_row_origin)) * N) * AA;
    const float* xi = xyz + ((size_t)b * N + i) * A * 3;
    const uint8_t* mi = amask ? amask + ((size_t)b * N + i) * A : nullptr;
    for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < nE; e += gridDim.x * 256u) {
        unsigned j = e / AA;
        unsigned r = e - j * AA;
        unsigned a = r / (unsigned)A;
        unsigned c = r - a * (unsigned)A;
        const float* xj = xyz + (((size_t)b * N + j) * A + c) * 3;
        const float* xa = xi + a * 3;
        if (dist) {
            float dx = xa[0] - xj[0], dy = xa[1] - xj[1], dz = xa[2] - xj[2];
            float sx = dx * dx, sy = dy * dy, sz = dz * dz;
            const float x = (sx + sy) + sz;
            dist[obase + e] = EXACT ? sqrt_rn_mk(x) : __builtin_amdgcn_sqrtf(x);
        }
        if (dmask) {
            uint8_t v = 1;
            if (mi) v = (mi[a] != 0) && (amask[((size_t)b * N + j) * A + c] != 0);
            dmask[obase + e] = v;
        }
    }
}

// ---- launch context: the dispatcher either launches or only RECORDS what it would launch (ps_k1_plan_f32) ----
// Every K1 launch goes through k1_go, so the plan a caller reads is by construction the dispatch a launch takes: same
// predicates, same grid, same LDS size -- there is no second copy of the selection logic to drift.
struct K1Go {
    hipStream_t s;
    ps_k1_plan* plan;   // non-null: record only, launch nothing
};

inline void plan_append(char* dst, size_t cap, const char* text) {
    size_t n = strlen(dst);
    if (n && n + 3 < cap) {
        memcpy(dst + n, " + ", 4);
        n += 3;
    }
    for (size_t i = 0; text[i] && n + 1 < cap; ++i) dst[n++] = text[i];
    dst[n] = 0;
}

// LDS of a launch: the dynamic request, and what the kernel declares statically (only the flat families declare any)
struct K1Lds {
    size_t dynamic;
    unsigned static_bytes;
    K1Lds(size_t d, unsigned st = 0) : dynamic(d), static_bytes(st) {}
    K1Lds(int d) : dynamic((size_t)d), static_bytes(0) {}
};

template <typename... KArgs, typename... Args>
inline int k1_go(const K1Go& go, const char* family, const char* name, int tparam, void (*kernel)(KArgs...), dim3 grid,
                 dim3 block, K1Lds lds_spec, Args&&... args) {
    const size_t lds = lds_spec.dynamic;
    if (!go.plan) return ps_launch(kernel, grid, block, lds, go.s, static_cast<Args&&>(args)...);
    ps_k1_plan& pl = *go.plan;
    char full[64];
    if (tparam >= 0) snprintf(full, sizeof full, "%s<%d>", name, tparam);
    else snprintf(full, sizeof full, "%s", name);
    plan_append(pl.kernel, sizeof pl.kernel, full);
    plan_append(pl.family, sizeof pl.family, family);
    // static LDS from the kernel's own declarations (compile-time constants at the call site): the plan query makes NO HIP
    // call -- it neither initialises the runtime in the calling process nor depends on a device being present
    const unsigned static_lds = lds_spec.static_bytes;
    const unsigned long long n_wg = (unsigned long long)grid.x * grid.y * grid.z;
    if (pl.n_launches == 0) {
        pl.n_workgroups = (unsigned)(n_wg > 0xFFFFFFFFull ? 0xFFFFFFFFull : n_wg);
        pl.lds_bytes = (unsigned)lds + static_lds;
        pl.threads_per_workgroup = (int)(block.x * block.y * block.z);
    } else {
        pl.n_workgroups_2 = (unsigned)(n_wg > 0xFFFFFFFFull ? 0xFFFFFFFFull : n_wg);
        pl.lds_bytes_2 = (unsigned)lds + static_lds;
    }
    ++pl.n_launches;
    return 0;
}

template <int JT>
int launch_a15(const K1Cfg& g, const float* xyz, const uint8_t* amask, float* dist, uint8_t* dmask, int B, int N,
               int row_begin, int row_end, int out_rows, int out_row_origin, const K1Go& go) {
    const int IR = g.rows_per_block;
    const int rows = row_end - row_begin;
    dim3 grid((N + JT - 1) / JT, (rows + IR - 1) / IR, B);
    size_t lds = (size_t)(JT + IR) * A15 * sizeof(float4) + (size_t)(JT + 4 + IR) * sizeof(uint32_t);
    const bool da = (N % 4 == 0) && ((reinterpret_cast<uintptr_t>(dist) & 15) == 0);
    const bool ma = (N % 16 == 0) && ((reinterpret_cast<uintptr_t>(dmask) & 15) == 0);
    const bool ex = g.exact_sqrt != 0;
#define PS_K1_LAUNCH1(NT_, DA_, MA_, EX_)                                                                         \
    k1_go(go, "slot_decode", "k1_pairdist_a15", JT, k1_pairdist_a15<JT, NT_, DA_, MA_, EX_>, grid, dim3(256), lds, xyz, \
          amask, dist, dmask, N, row_begin, row_end, out_rows, out_row_origin, IR)
#define PS_K1_LAUNCH(NT_, DA_, MA_) (ex ? PS_K1_LAUNCH1(NT_, DA_, MA_, true) : PS_K1_LAUNCH1(NT_, DA_, MA_, false))
    if (da && ma && g.variant == 0) {
        // LDS: padded float4 images of the JT column and IR row residues, then the two mask bit streams (at most
        // (JT * 15 / 256 + 1) * 8 + (IR * 15 / 256 + 1) * 8 + 1 words, which JT + 4 + IR words always cover for JT >= 32)
        const size_t lds_pat = (size_t)(JT + IR) * RS * sizeof(float4) + (size_t)(JT + 4 + IR) * sizeof(uint32_t) +
                               (size_t)(g.lds_pad_kb >= 0 ? g.lds_pad_kb : (N >= 256 ? 36 : 20)) * 1024;   // -1 = by chain length, see ps_k1_config
        const unsigned long long n_wg = (unsigned long long)grid.x * grid.y * grid.z;
        if (n_wg > 0x7FFFFFFFull) return (int)hipErrorInvalidValue;
        const int remap = (g.xcd_remap && n_wg % 8 == 0 && n_wg >= 64) ? 1 : 0;
#define PS_K1_PAT(NT_, M_, U_)                                                                                    \
    k1_go(go, "pattern", "k1_pairdist_a15_pat", JT, k1_pairdist_a15_pat<JT, NT_, M_, U_>, dim3((unsigned)n_wg),   \
          dim3(256), lds_pat, xyz, amask, dist, dmask, N, row_begin, row_end, out_rows, out_row_origin, IR,       \
          (int)grid.x, (int)grid.y, remap)
#ifdef PS_EXPERIMENTS
        const int math = g.experiment & 15;
        const bool unroll = (g.experiment & 16) != 0;
        if (math == 1) return PS_K1_PAT(false, 3, false);
        if (math == 2) return unroll ? PS_K1_PAT(false, 2, true) : PS_K1_PAT(false, 2, false);
        if (unroll) return ex ? PS_K1_PAT(false, 1, true) : PS_K1_PAT(false, 0, true);
#endif
        if (g.store_nt) return ex ? PS_K1_PAT(true, 1, false) : PS_K1_PAT(true, 0, false);
        return ex ? PS_K1_PAT(false, 1, false) : PS_K1_PAT(false, 0, false);
#undef PS_K1_PAT
    }
    if (B > 65535) return (int)hipErrorInvalidValue;   // slot-decode kernel: structure on grid.z
    if (da && ma) return g.store_nt ? PS_K1_LAUNCH(true, true, true) : PS_K1_LAUNCH(false, true, true);
    if (da) return PS_K1_LAUNCH(true, true, false);
    return PS_K1_LAUNCH(true, false, false);
#undef PS_K1_LAUNCH
#undef PS_K1_LAUNCH1
}

// Flat pattern kernel over the output pair range [pbeg, pend) (pair index P = (b*out_rows + il)*N + j).
bool flat_eligible(const K1Cfg& g, const float* dist, const uint8_t* dmask, int B, int N, int out_rows) {
    if (g.variant != 0 || g.flat == 0) return false;
    if (N < 16 || N >= (1 << 22) || out_rows < 1 || out_rows >= (1 << 22)) return false;
    if ((unsigned long long)B * out_rows * N > 0xFFFFFF00ull) return false;  // pair indices stay 32-bit
    if ((unsigned long long)B * N * (A15 * 3) >= 0x80000000ull) return false;  // and so do coordinate indices
    if ((reinterpret_cast<uintptr_t>(dist) & 15) || (reinterpret_cast<uintptr_t>(dmask) & 15)) return false;
    return g.flat == 2 || N % 16 != 0;
}

int launch_a15_flat(const K1Cfg& g, const float* xyz, const uint8_t* amask, float* dist, uint8_t* dmask, int B,
                    int N, int out_rows, int out_row_origin, unsigned pbeg, unsigned pend, unsigned n_ranges,
                    unsigned range_stride, const K1Go& go) {
    if (pbeg >= pend || n_ranges == 0) return 0;
    // chunks per range: exact for one range, an upper bound when the ranges start at different chunk phases
    // default chunk: 64 pairs (72 KB + 18 KB of output per workgroup): with the round-3 inner loop 4 % ahead of 128 pairs on
    // slow and medium buffers and equal on fast ones (profiles/r03_k1_flat_chunk_sweep_final.log); 32 pairs: +8 % / -3 to -10 %
    const int L2 = g.flat_fl_log2 ? g.flat_fl_log2 : 6;
    const unsigned FL = 1u << L2;
    const unsigned cpr = n_ranges == 1 ? ((pend + (FL - 1)) >> L2) - (pbeg >> L2) : ((pend - pbeg) >> L2) + 2;
    const unsigned long long n_chunks = (unsigned long long)n_ranges * cpr;
    if (n_chunks > 0x7FFFFFFFull) return (int)hipErrorInvalidValue;
    // like rows_per_block this depends on the output allocation: four chunks per workgroup are 1-2 % faster on some
    // and 8 % slower on others (profiles/r01_k1_ab_flat_cpw.log), so the default is 1 and ops.py autotunes it
    const unsigned cpw = (n_chunks >= 16384u) ? (unsigned)g.flat_cpw : 1u;
    const unsigned n_wg = (unsigned)((n_chunks + cpw - 1) / cpw);
    const int remap = (g.xcd_remap && n_wg >= 64) ? 1 : 0;
    const double rn = 1.0 / (double)N, rr = 1.0 / (double)out_rows;
    const size_t pad = (size_t)g.flat_lds_pad_kb * 1024;  // idle dynamic LDS: residency cap
#define PS_K1_FLAT2(EX_, HM_, L_)                                                                                 \
    k1_go(go, "flat", "k1_pairdist_a15_flat", 1 << L_, k1_pairdist_a15_flat<EX_, HM_, L_>, dim3(n_wg), dim3(256),     \
          K1Lds(pad, (unsigned)((((1 << L_) + FR) * RS) * sizeof(float4) + (2 * (1 << L_) + FR) * sizeof(uint32_t))),    \
          xyz, amask, dist, dmask, B, N, out_rows, out_row_origin, pbeg, pend, n_ranges, range_stride, cpr,       \
          (int)cpw, remap, rn, rr)
#define PS_K1_FLAT(EX_, HM_)                                                                                      \
    (L2 == 7 ? PS_K1_FLAT2(EX_, HM_, 7) : L2 == 6 ? PS_K1_FLAT2(EX_, HM_, 6) : L2 == 5 ? PS_K1_FLAT2(EX_, HM_, 5)  \
                                                                              : PS_K1_FLAT2(EX_, HM_, 4))
    if (g.exact_sqrt) return amask ? PS_K1_FLAT(true, true) : PS_K1_FLAT(true, false);
    return amask ? PS_K1_FLAT(false, true) : PS_K1_FLAT(false, false);
#undef PS_K1_FLAT
#undef PS_K1_FLAT2
}

// Row-tile kernel (A = 4, 8): any N, any row range; planes must be 16-byte aligned.
bool rowtile_eligible(const K1Cfg& g, const float* dist, const uint8_t* dmask, int A) {
    if (g.variant != 0 || g.flat != 1) return false;   // flat = 4 forces the flat kernels (cross-checks), 0 the simple one
    if (A != 4 && A != 8) return false;
    return !((reinterpret_cast<uintptr_t>(dist) & 15) || (reinterpret_cast<uintptr_t>(dmask) & 15));
}

template <int A>
int launch_rowtile(const K1Cfg& g, const float* xyz, const uint8_t* amask, float* dist, uint8_t* dmask, int B, int N,
                   int row_begin, int row_end, int out_rows, int out_row_origin, const K1Go& go) {
    constexpr int JT = RowTile<A>::JT;
    const int rows = row_end - row_begin;
    // Rows per workgroup: 6 (60 KB of output per workgroup).  Round 2 used 32; same-process sweeps of 2 .. 32 rows on two
    // boxes (profiles/r03_k1_rowtile_rows_per_workgroup.log) have 3-6 rows 3-12 % ahead of 32 at every shape (A = 4,
    // N = 500: 6.5-6.8 against 6.0 TB/s), 6 being the best or within 3 % of it also for short structures (N = 100 / 128),
    // where 2-4 rows lose.  cfg.rows_per_block > 1 overrides (A/B runs).
    const int irmax = g.rows_per_block > 1 ? g.rows_per_block : 6;
    const int IR = rows < irmax ? rows : irmax;
    const int n_tiles = (N + JT - 1) / JT, n_ichunks = (rows + IR - 1) / IR;
    const unsigned long long n_wg = (unsigned long long)n_tiles * n_ichunks * B;
    if (n_wg > 0x7FFFFFFFull) return (int)hipErrorInvalidValue;
    const int remap = (g.xcd_remap && n_wg % 8 == 0 && n_wg >= 64) ? 1 : 0;
    const size_t lds = (size_t)(JT + IR) * A * sizeof(float4) + (size_t)(JT + IR) * sizeof(uint32_t);
    if (g.exact_sqrt)
        return k1_go(go, "rowtile", "k1_pairdist_rowtile", A, k1_pairdist_rowtile<A, true>, dim3((unsigned)n_wg), dim3(256),
                     lds, xyz, amask, dist, dmask, N, row_begin, row_end, out_rows, out_row_origin, IR, n_tiles,
                     n_ichunks, remap);
    return k1_go(go, "rowtile", "k1_pairdist_rowtile", A, k1_pairdist_rowtile<A, false>, dim3((unsigned)n_wg), dim3(256),
                 lds, xyz, amask, dist, dmask, N, row_begin, row_end, out_rows, out_row_origin, IR, n_tiles, n_ichunks,
                 remap);
}

// Row-phase kernel: every atom count up to 64 other than 4, 8 (row-tile kernel) and 15; any N, any row range.  The small
// counts are compile-time instantiations; the rest share the two run-time instantiations (even / odd A).  cfg.rowphase:
// 0 = where it is the default (A <= 13, and the counts without a fixed-A flat kernel), 1 = every eligible count (A/B runs
// against the fixed-A flat and the A = 15 kernels), 2 = never (fixed-A flat / element kernels instead).
bool flatA_has(int A);

bool rowphase_eligible(const K1Cfg& g, const float* dist, const uint8_t* dmask, int N, int A) {
    const int mode = g.rowphase & 15;     // (bits 4..: A/B switches, see launch_rowphase)
    if (g.variant != 0 || g.flat != 1 || mode == 2) return false;
    if (A < 1 || A > 64 || A == 4 || A == 8) return false;
    if (A == 15 && mode != 1 && N >= 16) return false;                 // A = 15 has its own kernels from N = 16 on; batches of
                                                                       // shorter peptides take this kernel (1-D grid: any B)
    if (A != 15 && flatA_has(A) && mode != 1) return false;           // even counts with a fixed-A flat kernel
    if (N < 1 || (long long)N * A * A > (1ll << 28)) return false;   // slot and element indices of a row stay 32-bit
    return !((reinterpret_cast<uintptr_t>(dist) & 15) || (reinterpret_cast<uintptr_t>(dmask) & 15));
}

// CA traces (A = 1) of 8 .. CA_FLAT_MAX_N residues, full matrices: the flat kernel (see k1_pairdist_ca_flat).  cfg.rowphase = 1
// keeps the row-phase kernel (A/B runs).  The crossover (profiles/r04_k1_n_sweep_a1.log, same box, TB/s flat / row-phase):
// N = 64 5.74 / 2.25, 128 5.87 / 3.74, 200 5.54 / 4.84, 255 5.53 / 4.63; second box 256 5.72 / 6.12, 300 4.98 / 5.48,
// 512 4.75 / 6.75 -- the flat kernel stages whole structures per 8192 elements, which stops paying once a structure is
// longer than that.
constexpr int CA_FLAT_MAX_N = 255;
bool ca_flat_eligible(const K1Cfg& g, const float* dist, const uint8_t* dmask, int B, int N, int A, int row_begin, int row_end,
                      int out_rows, int out_row_origin) {
    if (A != 1 || g.variant != 0 || g.flat != 1 || (g.rowphase & 15) != 0) return false;
    if (N < 8 || N > CA_FLAT_MAX_N) return false;
    if (row_begin != 0 || row_end != N || out_rows != N || out_row_origin != 0) return false;
    return !((reinterpret_cast<uintptr_t>(dist) & 15) || (reinterpret_cast<uintptr_t>(dmask) & 15));
}

// Short chains of 2 .. 16 atoms per residue, full matrices: the flat kernel above up to the length where the row kernels catch up
// (same-box sweeps, TB/s flat / row kernel, profiles/r04_k1_n_sweep_small_a_short.log: A = 2 N = 8 4.7 / 0.7, 32 5.7 / 4.8, 64 5.8 /
// 5.8, 128 5.6 / 5.9; A = 3 N = 16 4.9 / 2.3, 64 5.1 / 4.7, 128 4.5 / 5.2; A = 4 N = 16 5.8 / 3.1, 64 5.7 / 5.4, 128 4.7 / 6.0;
// A = 5 N = 8 4.9 / 2.0, 16 5.1 / 4.1, 32 5.2 / 5.1, 64 4.7 / 6.1; A = 8 N = 8 5.8 / 3.7, 16 5.9 / 6.4; A = 13 N = 4 5.1 / 2.6,
// 16 5.2 / 4.4, 24 4.8 / 6.6).  cfg.rowphase = 1 keeps the row-phase / row-tile kernels (A/B runs).
// (A = 14 .. 16, TB/s flat / row-phase: N = 4 5.2-5.8 / 3.3-4.0, N = 8 .. 15 within 5 % of each other: up to seven residues)
inline int small_flat_max_n(int A) { return A <= 4 ? 64 : A == 5 ? 31 : A == 6 ? 24 : A == 7 ? 20 : A == 8 ? 8 : A <= 13 ? 16 : 7; }
bool small_flat_eligible(const K1Cfg& g, const float* dist, const uint8_t* dmask, int B, int N, int A, int row_begin, int row_end,
                         int out_rows, int out_row_origin) {
    if (A < 2 || A > 16 || g.variant != 0 || g.flat != 1 || (g.rowphase & 15) != 0) return false;
    if (N < 2 || N > small_flat_max_n(A)) return false;
    if (row_begin != 0 || row_end != N || out_rows != N || out_row_origin != 0) return false;
    return !((reinterpret_cast<uintptr_t>(dist) & 15) || (reinterpret_cast<uintptr_t>(dmask) & 15));
}

template <int ACT>
int launch_rowphase(const K1Cfg& g, const float* xyz, const uint8_t* amask, float* dist, uint8_t* dmask, int B, int N,
                    int A, int row_begin, int row_end, int out_rows, int out_row_origin, const K1Go& go) {
    using T = RowPhase<ACT>;
    const int rows = row_end - row_begin;
    const int nel = N * A * A;
    const int nslots = (nel + 3 + ((nel & 3) == 0 ? 0 : (nel & 3) == 2 ? 2 : 3)) / 4;   // over the phases that occur
    // short rows: 2 or 4 row groups of 128 / 64 lanes (every lane still takes ~32 rows), see the kernel
    const int lpg_log2 = nslots <= 128 ? 6 : (nslots <= 256 ? 7 : 8);
    const int G = 256 >> lpg_log2, tile_slots = T::SPL << lpg_log2;
    // Rows per lane (= rows per workgroup / G).  Short-lived workgroups suit the store stream (the fewer bytes a
    // workgroup writes, the better slow-class allocations absorb it), a workgroup's set-up -- staging, the per-lane index
    // decode -- wants to be amortised.  Same-process sweeps of 4 .. 32 rows on two boxes (profiles/
    // r03_k1_ab_rowphase_tiling.log, r03_k1_rowphase_rows_per_lane.log): the compile-time counts (cheap decode) peak at
    // 8-16 rows (A = 5, N = 501: 5.6-6.3 TB/s against 3.7-5.2 with 32), A = 1 / 2, whose rows are short, at 16-24; the
    // run-time even counts at 8-16; the run-time odd counts (seven-element windows) wanted 32 while their index decode
    // divided at run time, and 16-24 since it multiplies by reciprocals (round 4, profiles/r04_k1_rowphase_rows_runtime_odd.log:
    // A = 17 / 27 at N = 125 6.33 / 6.37 TB/s with 16 rows against 5.93 / 6.09 with 32; A = 35, 63: 24 rows).
    // cfg.rows_per_block > 1 overrides (A/B runs).
    const int rpl = g.rows_per_block > 1 ? g.rows_per_block : (ACT > 0 ? (ACT <= 2 ? 16 : 12) : (ACT == 0 ? 12 : (A < 32 ? 16 : 24)));
    const int n_ichunks = (rows + rpl * G - 1) / (rpl * G), IR = (rows + n_ichunks - 1) / n_ichunks;   // <= rpl * G rows, balanced
    // Tiles: FULL tiles of tile_slots and a short last one.  (Balanced tiles leave the last wave of EVERY workgroup idle
    // for a pass: A = 5, N = 500 ran 4.4 against 5.2 TB/s.)
    const int n_tiles = (nslots + tile_slots - 1) / tile_slots;
    const int spt = tile_slots;
    const unsigned long long n_wg = (unsigned long long)n_tiles * n_ichunks * B;
    if (n_wg > 0x7FFFFFFFull) return (int)hipErrorInvalidValue;
    const int remap = (g.xcd_remap && n_wg >= 64) ? 1 : 0;
    const int maxres = rowphase_maxres(A, N);
    const size_t lds = (size_t)(IR * A + (ACT == 1 ? 1 : 0)) * sizeof(float4) + (size_t)maxres * sizeof(typename T::bits_t) +
                       (size_t)maxres * A * 3 * sizeof(float);
    const int flags = g.rowphase >> 4;    // [diagnostic] A/B switches of the row-phase kernel (bit 0: A = 1 seam slots written
                                          // element-wise from both rows, as in round 3)
    const unsigned rcpAA = (unsigned)((1ull << 32) / (unsigned)(A * A)), rcpA = (unsigned)((1ull << 32) / (unsigned)A);   // (A = 1: unused)
    const int tp = ACT > 0 ? ACT : A;     // what the plan prints: the atom count
    const char* name = ACT > 0 ? "k1_pairdist_rowphase" : (ACT == 0 ? "k1_pairdist_rowphase_rt_even" : "k1_pairdist_rowphase_rt_odd");
    if (g.exact_sqrt)
        return k1_go(go, "rowphase", name, tp, k1_pairdist_rowphase<ACT, true>, dim3((unsigned)n_wg), dim3(256), lds, xyz,
                     amask, dist, dmask, N, A, row_begin, row_end, out_rows, out_row_origin, IR, n_tiles, spt, n_ichunks,
                     lpg_log2, remap, flags, rcpAA, rcpA);
    return k1_go(go, "rowphase", name, tp, k1_pairdist_rowphase<ACT, false>, dim3((unsigned)n_wg), dim3(256), lds, xyz,
                 amask, dist, dmask, N, A, row_begin, row_end, out_rows, out_row_origin, IR, n_tiles, spt, n_ichunks,
                 lpg_log2, remap, flags, rcpAA, rcpA);
}

// Fixed-A flat pattern kernels: the EVEN atom counts 14 (atom14), 16, 24, 32, where their line-aligned chunks make them
// as fast as the row-phase kernel on aligned lengths and faster on the others (profiles/
// r03_k1_a_sweep_rowphase_runtime_vs_flatA.log: A = 14, N = 250 6.45 against 5.49 TB/s).  For odd counts the row-phase
// kernel wins (25: 6.77 against 5.97; 27: 6.30 / 5.72; 37: 6.22 / 5.66), so their instantiations were removed.  A = 15
// is instantiated so that the template can be cross-checked against the hand-specialised A = 15 kernels (cfg.flat == 4).
bool flatA_has(int A) { return A == 14 || A == 15 || A == 16 || A == 24 || A == 32; }

bool flatA_eligible(const K1Cfg& g, const float* dist, const uint8_t* dmask, int B, int N, int A, int out_rows) {
    if (g.variant != 0 || g.flat == 0 || !flatA_has(A)) return false;
    if (A == 15 && g.flat != 4) return false;   // A = 15 has its own kernels
    if (N < 16 || N >= (1 << 22) || out_rows < 1 || out_rows >= (1 << 22)) return false;
    if ((unsigned long long)B * out_rows * N > 0xFFFFFF00ull) return false;      // pair indices stay 32-bit
    if ((unsigned long long)B * N * A * 3 >= 0x80000000ull) return false;        // and so do coordinate indices
    if ((reinterpret_cast<uintptr_t>(dist) & 15) || (reinterpret_cast<uintptr_t>(dmask) & 15)) return false;
    return true;
}

template <int A>
int launch_flatA(const K1Cfg& g, const float* xyz, const uint8_t* amask, float* dist, uint8_t* dmask, int B, int N,
                 int out_rows, int out_row_origin, unsigned pbeg, unsigned pend, unsigned n_ranges,
                 unsigned range_stride, const K1Go& go) {
    if (pbeg >= pend || n_ranges == 0) return 0;
    constexpr int L2 = FlatA<A>::FL_LOG2, FLn = FlatA<A>::FLn;
    const unsigned cpr = n_ranges == 1 ? ((pend + (FLn - 1)) >> L2) - (pbeg >> L2) : ((pend - pbeg) >> L2) + 2;
    const unsigned long long n_chunks = (unsigned long long)n_ranges * cpr;
    if (n_chunks > 0x7FFFFFFFull) return (int)hipErrorInvalidValue;
    const unsigned cpw = (n_chunks >= 16384u) ? (unsigned)g.flat_cpw : 1u;
    const unsigned n_wg = (unsigned)((n_chunks + cpw - 1) / cpw);
    const int remap = (g.xcd_remap && n_wg >= 64) ? 1 : 0;
    const double rn = 1.0 / (double)N, rr = 1.0 / (double)out_rows;
#define PS_K1_FLATA(EX_, HM_)                                                                                     \
    k1_go(go, "flatA", "k1_pairdist_flatA", A, k1_pairdist_flatA<A, EX_, HM_>, dim3(n_wg), dim3(256),             \
          K1Lds(0, (unsigned)((FlatA<A>::FLn + FlatA<A>::FRn) * FlatA<A>::RS * sizeof(float4) +                      \
                              (2 * FlatA<A>::FLn + FlatA<A>::FRn) * sizeof(typename FlatA<A>::mask_t))), xyz,       \
          amask, dist, dmask, B, N, out_rows, out_row_origin, pbeg, pend, n_ranges, range_stride, cpr, (int)cpw,  \
          remap, rn, rr)
    if (g.exact_sqrt) return amask ? PS_K1_FLATA(true, true) : PS_K1_FLATA(true, false);
    return amask ? PS_K1_FLATA(false, true) : PS_K1_FLATA(false, false);
#undef PS_K1_FLATA
}

// Range checks of a caller-supplied configuration; the defaults pass by construction.
bool cfg_valid(const K1Cfg& g) {
    if (g.struct_size != (int)sizeof(K1Cfg)) return false;
    if (g.variant < 0 || g.variant > 1 || g.flat < 0 || g.flat > 4 || g.flat == 3) return false;
    if (g.rows_per_block < 1 || g.rows_per_block > 32 || g.lds_pad_kb < -1 || g.lds_pad_kb > 120) return false;
    if (g.flat_cpw < 1 || g.flat_cpw > 64 || g.flat_lds_pad_kb < 0 || g.flat_lds_pad_kb > 100) return false;
    if (g.jt != 0 && g.jt != 16 && g.jt != 32 && g.jt != 64 && g.jt != 128) return false;
    if (g.flat_fl_log2 != 0 && (g.flat_fl_log2 < 4 || g.flat_fl_log2 > 7)) return false;
    if (g.rowphase < 0 || (g.rowphase & 15) > 2 || g.rowphase >= 256) return false;
#ifdef PS_EXPERIMENTS
    if (g.experiment < 0 || (g.experiment & 15) > 2 || g.experiment > 31) return false;
#else
    if (g.experiment != 0) return false;   // timing experiments do not exist in the product library
#endif
    return true;
}

}  // namespace

extern "C" void ps_k1_config_default(ps_k1_config* cfg) {
    if (!cfg) return;
    *cfg = ps_k1_config{};
    cfg->struct_size = (int)sizeof(ps_k1_config);
    cfg->flat = 1;
    cfg->rows_per_block = 1;
    // 20 KB of idle LDS per workgroup of the pattern kernel: with its 32-residue tiles (jt = 0 -> 32) five workgroups are
    // resident per CU; never more than 2 % behind the best configuration on the output buffers of ten boxes
    // (profiles/r03_k1_ab_lean_*.log; the 8 KB + 128-residue tiles of rounds 2-3 are a tuner candidate)
    cfg->lds_pad_kb = -1;
    cfg->flat_cpw = 1;
    cfg->xcd_remap = 1;
}

namespace {

// The one dispatcher: argument checks, then the first eligible kernel family in a fixed order.  `go` decides whether the
// chosen kernel is launched or only recorded (ps_k1_plan_f32); pointers are used for their NULL-ness and alignment only.
int k1_dispatch(const K1Go& go, const float* xyz, const uint8_t* atom_mask, float* dist, uint8_t* dist_mask, int B,
                int N, int A, int row_begin, int row_end, int out_rows, int out_row_origin, const ps_k1_config* cfg) {
    if (!xyz || (!dist && !dist_mask) || B < 0 || N < 0 || A <= 0) return (int)hipErrorInvalidValue;
    if (row_begin < 0 || row_end > N || row_begin > row_end) return (int)hipErrorInvalidValue;
    if (out_row_origin > row_begin || row_end - out_row_origin > out_rows) return (int)hipErrorInvalidValue;
    K1Cfg g;
    ps_k1_config_default(&g);
    if (cfg) {
        if (!cfg_valid(*cfg)) return (int)hipErrorInvalidValue;
        g = *cfg;   // by value: the caller may change or free its copy as soon as this call returns
    }
    if (B == 0 || N == 0 || row_begin == row_end) return 0;
    const int rows = row_end - row_begin;
    if (small_flat_eligible(g, dist, dist_mask, B, N, A, row_begin, row_end, out_rows, out_row_origin)) {
        const unsigned S = (unsigned)N * N * A * A;
        const unsigned long long total = (unsigned long long)B * S;
        const unsigned long long n_wg = (total + 4ull * CA_CS - 1) / (4ull * CA_CS);
        if (n_wg > 0x7FFFFFFFull) return (int)hipErrorInvalidValue;
        const size_t lds = (size_t)((4 * CA_CS + S - 1) / S + 1) * N * A * sizeof(float4);   // the structures a workgroup's slots can touch
        auto rcp = [](unsigned d) { return d == 1 ? 0xFFFFFFFFu : (unsigned)((1ull << 32) / d); };
        const unsigned rcpS = rcp(S), rcpR = rcp((unsigned)N * A * A), rcpAA = rcp((unsigned)A * A), rcpA = rcp((unsigned)A);
        if (g.exact_sqrt)
            return k1_go(go, "small_flat", "k1_pairdist_small_flat", A, k1_pairdist_small_flat<true>, dim3((unsigned)n_wg), dim3(256),
                         lds, xyz, atom_mask, dist, dist_mask, B, N, A, rcpS, rcpR, rcpAA, rcpA);
        return k1_go(go, "small_flat", "k1_pairdist_small_flat", A, k1_pairdist_small_flat<false>, dim3((unsigned)n_wg), dim3(256),
                     lds, xyz, atom_mask, dist, dist_mask, B, N, A, rcpS, rcpR, rcpAA, rcpA);
    }
    if (rowtile_eligible(g, dist, dist_mask, A)) {
        if (A == 4)
            return launch_rowtile<4>(g, xyz, atom_mask, dist, dist_mask, B, N, row_begin, row_end, out_rows, out_row_origin, go);
        return launch_rowtile<8>(g, xyz, atom_mask, dist, dist_mask, B, N, row_begin, row_end, out_rows, out_row_origin, go);
    }
    if (ca_flat_eligible(g, dist, dist_mask, B, N, A, row_begin, row_end, out_rows, out_row_origin)) {
        const unsigned long long total = (unsigned long long)B * N * N;
        const unsigned long long n_wg = (total + 4ull * CA_CS - 1) / (4ull * CA_CS);
        if (n_wg > 0x7FFFFFFFull) return (int)hipErrorInvalidValue;
        const size_t lds = (size_t)ca_max_structs(N) * N * sizeof(float4);
        const unsigned rcpNN = (unsigned)((1ull << 32) / (unsigned)(N * N)), rcpN = (unsigned)((1ull << 32) / (unsigned)N);
        if (g.exact_sqrt)
            return k1_go(go, "ca_flat", "k1_pairdist_ca_flat", -1, k1_pairdist_ca_flat<true>, dim3((unsigned)n_wg), dim3(256), lds,
                         xyz, atom_mask, dist, dist_mask, B, N, rcpNN, rcpN);
        return k1_go(go, "ca_flat", "k1_pairdist_ca_flat", -1, k1_pairdist_ca_flat<false>, dim3((unsigned)n_wg), dim3(256), lds,
                     xyz, atom_mask, dist, dist_mask, B, N, rcpNN, rcpN);
    }
    if (rowphase_eligible(g, dist, dist_mask, N, A)) {
#define PS_K1_RP(A_) case A_: return launch_rowphase<A_>(g, xyz, atom_mask, dist, dist_mask, B, N, A, row_begin, row_end, out_rows, out_row_origin, go);
        switch (A) {
            PS_K1_RP(1) PS_K1_RP(2) PS_K1_RP(3) PS_K1_RP(5) PS_K1_RP(6) PS_K1_RP(7) PS_K1_RP(9) PS_K1_RP(10) PS_K1_RP(11)
            PS_K1_RP(12) PS_K1_RP(13) PS_K1_RP(25) PS_K1_RP(37)   // (25: the reference's own test count; 37: atom37)
            PS_K1_RP(15)                                          // (N < 16; N >= 16 only with cfg.rowphase = 1: the A/B against the pattern kernels)
            default: break;
        }
#undef PS_K1_RP
        if (A & 1)
            return launch_rowphase<-1>(g, xyz, atom_mask, dist, dist_mask, B, N, A, row_begin, row_end, out_rows, out_row_origin, go);
        return launch_rowphase<0>(g, xyz, atom_mask, dist, dist_mask, B, N, A, row_begin, row_end, out_rows, out_row_origin, go);
    }
    if (flatA_eligible(g, dist, dist_mask, B, N, A, out_rows)) {
        // one contiguous pair range when every output row is computed, else the same rows of every structure
        const bool whole = rows == out_rows;
        const unsigned r0 = whole ? 0u : (unsigned)(row_begin - out_row_origin) * (unsigned)N;
        const unsigned r1 = whole ? (unsigned)((unsigned long long)B * out_rows * N) : r0 + (unsigned)rows * (unsigned)N;
        const unsigned nrg = whole ? 1u : (unsigned)B, stride = whole ? 0u : (unsigned)out_rows * (unsigned)N;
        switch (A) {
            case 14: return launch_flatA<14>(g, xyz, atom_mask, dist, dist_mask, B, N, out_rows, out_row_origin, r0, r1, nrg, stride, go);
            case 15: return launch_flatA<15>(g, xyz, atom_mask, dist, dist_mask, B, N, out_rows, out_row_origin, r0, r1, nrg, stride, go);
            case 16: return launch_flatA<16>(g, xyz, atom_mask, dist, dist_mask, B, N, out_rows, out_row_origin, r0, r1, nrg, stride, go);
            case 24: return launch_flatA<24>(g, xyz, atom_mask, dist, dist_mask, B, N, out_rows, out_row_origin, r0, r1, nrg, stride, go);
            case 32: return launch_flatA<32>(g, xyz, atom_mask, dist, dist_mask, B, N, out_rows, out_row_origin, r0, r1, nrg, stride, go);
            default: break;
        }
    }
    if (A == A15 && flat_eligible(g, dist, dist_mask, B, N, out_rows)) {
        // one contiguous pair range when every output row is computed, else the same rows of every structure
        if (rows == out_rows)
            return launch_a15_flat(g, xyz, atom_mask, dist, dist_mask, B, N, out_rows, out_row_origin, 0u,
                                   (unsigned)((unsigned long long)B * out_rows * N), 1u, 0u, go);
        const unsigned r0 = (unsigned)(row_begin - out_row_origin) * (unsigned)N;
        return launch_a15_flat(g, xyz, atom_mask, dist, dist_mask, B, N, out_rows, out_row_origin, r0,
                               r0 + (unsigned)rows * (unsigned)N, (unsigned)B, (unsigned)out_rows * (unsigned)N, go);
    }
    // the flat kernels above and the pattern kernel run on 1-D grids and take any batch size; the slot-decode and the
    // element-per-lane kernels put the structure on grid.z (checked where they are launched)
    if (A == A15) {
        // N < 16 (peptides): a row run is at most 13.5 KB, so the slot-decode kernel takes up to 16 rows per workgroup
        if (N < 16 && g.rows_per_block == 1) g.rows_per_block = rows < 16 ? rows : 16;
        // short chains (16 <= N < 64): a workgroup that writes one row of one tile is mostly set-up, so rows_per_block = 1
        // (the default) means "about 64 column residues' worth of rows": N = 16: 4 rows (6.7 against 4.3 TB/s), N = 32 and 48:
        // 2 rows (6.8 against 5.0, 6.5 against 5.9; profiles/r03_k1_short_chains_rows.log); consecutive rows are contiguous
        else if (N < 64 && g.rows_per_block == 1) g.rows_per_block = rows < (64 + N - 1) / N ? rows : (64 + N - 1) / N;
        if ((rows + g.rows_per_block - 1) / g.rows_per_block > 65535) return (int)hipErrorInvalidValue;
        // default tile: 32 column residues (36 KB + 9 KB of output per workgroup) at the default 20 KB of idle LDS = 5
        // workgroups per CU: with the round-3 kernel the best or within 2 % of it on the output buffers of ten boxes (7.0-7.3
        // TB/s fast class, 6.1-6.4 slow; 128-residue tiles + 8 KB, the default until then: 6.8-7.1 / 5.9)
        // (a mask-only launch stages no coordinates and writes a quarter of the bytes per tile: 128-residue tiles run it at the
        // fill rate of the plane, 6.8 TB/s, where 32-residue tiles are set-up bound at 5.4; profiles/r03_k1_plane_split_mask_only.log)
        const int jt = g.jt ? g.jt : (dist ? 32 : 128);
        if (jt == 128)
            return launch_a15<128>(g, xyz, atom_mask, dist, dist_mask, B, N, row_begin, row_end, out_rows,
                                   out_row_origin, go);
        if (jt == 16)
            return launch_a15<16>(g, xyz, atom_mask, dist, dist_mask, B, N, row_begin, row_end, out_rows, out_row_origin,
                                  go);
        if (jt == 32)
            return launch_a15<32>(g, xyz, atom_mask, dist, dist_mask, B, N, row_begin, row_end, out_rows, out_row_origin,
                                  go);
        return launch_a15<64>(g, xyz, atom_mask, dist, dist_mask, B, N, row_begin, row_end, out_rows, out_row_origin,
                              go);
    }
    if (rows > 65535 || B > 65535) return (int)hipErrorInvalidValue;   // element kernel: (row, structure) on grid.y / grid.z
    const unsigned long long nE = (unsigned long long)N * A * A;
    if (nE > 0xFFFFFFFFull) return (int)hipErrorInvalidValue;
    unsigned gx = (unsigned)((nE + 255) / 256);
    if (gx > 64) gx = 64;
    if (g.exact_sqrt)
        return k1_go(go, "element", "k1_pairdist_generic", -1, k1_pairdist_generic<true>, dim3(gx, rows, B), dim3(256), 0,
                     xyz, atom_mask, dist, dist_mask, N, A, row_begin, row_end, out_rows, out_row_origin);
    return k1_go(go, "element", "k1_pairdist_generic", -1, k1_pairdist_generic<false>, dim3(gx, rows, B), dim3(256), 0, xyz,
                 atom_mask, dist, dist_mask, N, A, row_begin, row_end, out_rows, out_row_origin);
}

}  // namespace

extern "C" int ps_pairwise_distance_cfg_f32(const float* xyz, const uint8_t* atom_mask, float* dist,
                                            uint8_t* dist_mask, int B, int N, int A, int row_begin, int row_end,
                                            int out_rows, int out_row_origin, const ps_k1_config* cfg,
                                            void* stream) {
    const K1Go go{reinterpret_cast<hipStream_t>(stream), nullptr};
    return k1_dispatch(go, xyz, atom_mask, dist, dist_mask, B, N, A, row_begin, row_end, out_rows, out_row_origin, cfg);
}

extern "C" int ps_k1_plan_f32(int B, int N, int A, int row_begin, int row_end, int out_rows, int out_row_origin,
                              int dist_misalign, int mask_misalign, int has_atom_mask, const ps_k1_config* cfg,
                              ps_k1_plan* plan) {
    if (!plan || plan->struct_size != (int)sizeof(ps_k1_plan)) return (int)hipErrorInvalidValue;
    if (dist_misalign > 15 || mask_misalign > 15 || (dist_misalign >= 0 && dist_misalign % 4 != 0))
        return (int)hipErrorInvalidValue;
    *plan = ps_k1_plan{};
    plan->struct_size = (int)sizeof(ps_k1_plan);
    // stand-in addresses: never dereferenced on the host, only tested for NULL and for their low four bits
    const uintptr_t base = 0x10000;
    const float* xyz = reinterpret_cast<const float*>(base);
    const uint8_t* am = has_atom_mask ? reinterpret_cast<const uint8_t*>(base) : nullptr;
    float* d = dist_misalign < 0 ? nullptr : reinterpret_cast<float*>(base + (uintptr_t)dist_misalign);
    uint8_t* m = mask_misalign < 0 ? nullptr : reinterpret_cast<uint8_t*>(base + (uintptr_t)mask_misalign);
    const K1Go go{nullptr, plan};
    const int rc = k1_dispatch(go, xyz, am, d, m, B, N, A, row_begin, row_end, out_rows, out_row_origin, cfg);
    if (rc == 0 && plan->n_launches == 0) {
        snprintf(plan->kernel, sizeof plan->kernel, "(nothing to launch)");
        snprintf(plan->family, sizeof plan->family, "empty");
    }
    return rc;
}

extern "C" int ps_pairwise_distance_f32(const float* xyz, const uint8_t* atom_mask, float* dist, uint8_t* dist_mask,
                                        int B, int N, int A, int row_begin, int row_end, int out_rows,
                                        int out_row_origin, void* stream) {
    return ps_pairwise_distance_cfg_f32(xyz, atom_mask, dist, dist_mask, B, N, A, row_begin, row_end, out_rows,
                                        out_row_origin, nullptr, stream);
}
