// K2 -- backbone phi/psi/omega + chain-terminus masks, one launch.
// Replaces StructureBatch.backbone_dihedrals, get_n_terminal_mask and
// get_c_terminal_mask (reference protstruc.py:435-453, :486-541) with
// geometry.dihedral (geometry.py:74-124) evaluated in registers.
//
// One lane per residue.  Each lane loads its own N, CA, C once; the C of the
// previous residue and the N, CA of the next one arrive by wavefront shuffles
// (__shfl_up / __shfl_down over 64 lanes), and only the two lanes at the edge of
// a wave fetch their neighbour from memory.  Everything a residue needs for its
// three torsions, its two terminus flags and its three mask bits is then in
// registers: 36 bytes read and 15 + 2 bytes written per residue.
#include "ps_common.hpp"

namespace {

__global__ __launch_bounds__(256) void k2_backbone_dihedrals(const float* __restrict__ xyz,
                                                             const float* __restrict__ chain_idx,
                                                             const uint8_t* __restrict__ residue_mask,
                                                             float* __restrict__ dihedrals,
                                                             uint8_t* __restrict__ dihedral_mask,
                                                             uint8_t* __restrict__ nterm_out,
                                                             uint8_t* __restrict__ cterm_out, int N, int A,
                                                             int n_tiles) {
    // 1-D grid (residue tile fastest, then structure): any batch size
    const int b = (int)(blockIdx.x / (unsigned)n_tiles);
    const int i = (int)(blockIdx.x % (unsigned)n_tiles) * 256 + threadIdx.x;
    const int lane = threadIdx.x & (PS_WAVE - 1);
    const bool live = i < N;
    const int ic = live ? i : N - 1;  // clamp so every lane of the wave can take part in the shuffles
    const size_t res = (size_t)b * N + ic;
    const float* p = xyz + res * (size_t)A * 3;

    const f3 n = load3(p), ca = load3(p + 3), c = load3(p + 6);
    const float ch = chain_idx[res];
    const bool rm = residue_mask[res] != 0;

    // neighbours through the wavefront
    f3 c_prev = mk3(__shfl_up(c.x, 1), __shfl_up(c.y, 1), __shfl_up(c.z, 1));
    float ch_prev = __shfl_up(ch, 1);
    f3 n_next = mk3(__shfl_down(n.x, 1), __shfl_down(n.y, 1), __shfl_down(n.z, 1));
    f3 ca_next = mk3(__shfl_down(ca.x, 1), __shfl_down(ca.y, 1), __shfl_down(ca.z, 1));
    float ch_next = __shfl_down(ch, 1);
    if (lane == 0 && ic > 0) {  // wave edge: predecessor lives in another wave
        c_prev = load3(p - (size_t)A * 3 + 6);
        ch_prev = chain_idx[res - 1];
    }
    if (lane == PS_WAVE - 1 && ic < N - 1) {
        n_next = load3(p + (size_t)A * 3);
        ca_next = load3(p + (size_t)A * 3 + 3);
        ch_next = chain_idx[res + 1];
    }
    if (!live) return;

    // protstruc.py:442-443 / :452-453 -- NaN padding on either side, and NaN != NaN
    const bool first = (i == 0), last = (i == N - 1);
    const bool nterm = (first || ch_prev != ch) && rm;
    const bool cterm = (last || ch != ch_next) && rm;

    if (dihedrals) {
        // protstruc.py:518-534: zero pad at the batch edge, then zero at the termini
        float phi = (first || nterm) ? 0.f : dihedral4(c_prev, n, ca, c);
        float psi = (last || cterm) ? 0.f : dihedral4(n, ca, c, n_next);
        float omega = (last || cterm) ? 0.f : dihedral4(ca, c, n_next, ca_next);
        float* o = dihedrals + res * 3;
        o[0] = phi;
        o[1] = psi;
        o[2] = omega;
    }
    if (dihedral_mask) {
        uint8_t* m = dihedral_mask + res * 3;
        m[0] = (uint8_t)(!nterm && rm);
        m[1] = (uint8_t)(!cterm && rm);
        m[2] = (uint8_t)(!cterm && rm);
    }
    if (nterm_out) nterm_out[res] = (uint8_t)nterm;
    if (cterm_out) cterm_out[res] = (uint8_t)cterm;
}

}  // namespace

extern "C" int ps_backbone_dihedrals_f32(const float* xyz, const float* chain_idx, const uint8_t* residue_mask,
                                         float* dihedrals, uint8_t* dihedral_mask, uint8_t* nterm, uint8_t* cterm,
                                         int B, int N, int A, void* stream) {
    if (!xyz || !chain_idx || !residue_mask || B < 0 || N < 0 || A < 3) return (int)hipErrorInvalidValue;
    if (B == 0 || N == 0) return 0;
    const int n_tiles = (N + 255) / 256;
    const unsigned long long n_wg = (unsigned long long)n_tiles * B;
    if (n_wg > 0x7FFFFFFFull) return (int)hipErrorInvalidValue;
    return ps_launch(k2_backbone_dihedrals, dim3((unsigned)n_wg), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     xyz, chain_idx, residue_mask, dihedrals, dihedral_mask, nterm, cterm, N, A, n_tiles);
}
