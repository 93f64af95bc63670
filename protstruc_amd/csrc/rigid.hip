// Rigid-body ops on the coordinate tensor -- the steps either side of the frames kernel.
//   ps_rigid_f32               translate / rotate / center_at / get_local_xyz  (reference protstruc.py:662-694,
//                              :759-788, :347-362): x' = R x + t  or  R^T x + t
//   ps_center_of_mass_f32      nanmean of one atom slot over the residues of a structure (protstruc.py:746-757)
//   ps_frames_to_backbone_f32  xyz = R_res * ideal_atom + t_res, zero padded to A slots (protstruc.py:264-319)
// All element-wise and tiny next to K1; one lane per atom (or per structure for the reduction).
#include "ps_common.hpp"

namespace {

// r_mode: 0 none, 1 one shared 3x3, 2 per structure (B,3,3), 3 per residue (B,N,3,3)
// t_mode: 0 none, 1 one shared (3), 2 per structure (B,3), 3 per residue (B,N,3), 4 per atom (B,N,A,3)
__global__ __launch_bounds__(256) void k_rigid(const float* __restrict__ in, float* __restrict__ out,
                                               const float* __restrict__ R, const float* __restrict__ t,
                                               size_t n_atoms_total, unsigned N, unsigned A, int r_mode, int t_mode,
                                               int transpose) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_atoms_total) return;
    const size_t res = i / A, b = res / N;
    f3 x = load3(in + i * 3);
    if (r_mode) {
        const float* m = R + (r_mode == 1 ? 0 : (r_mode == 2 ? b : res) * 9);
        f3 y;
        if (!transpose) {  // einsum "ij,j->i": sum over the column index
            y.x = dot3(mk3(m[0], m[1], m[2]), x);
            y.y = dot3(mk3(m[3], m[4], m[5]), x);
            y.z = dot3(mk3(m[6], m[7], m[8]), x);
        } else {           // einsum "ji,j->i"
            y.x = dot3(mk3(m[0], m[3], m[6]), x);
            y.y = dot3(mk3(m[1], m[4], m[7]), x);
            y.z = dot3(mk3(m[2], m[5], m[8]), x);
        }
        x = y;
    }
    if (t_mode) {
        const float* v = t + (t_mode == 1 ? 0 : (t_mode == 2 ? b : (t_mode == 3 ? res : i)) * 3);
        x.x += v[0];
        x.y += v[1];
        x.z += v[2];
    }
    float* o = out + i * 3;
    o[0] = x.x; o[1] = x.y; o[2] = x.z;
}

// one wave per structure: nanmean over residues of atom slot `atom`
__global__ __launch_bounds__(64) void k_center_of_mass(const float* __restrict__ xyz, float* __restrict__ com,
                                                       unsigned N, unsigned A, unsigned atom) {
    const unsigned b = blockIdx.x;
    double s[3] = {0, 0, 0}, cnt[3] = {0, 0, 0};
    for (unsigned n = threadIdx.x; n < N; n += 64) {
        const float* p = xyz + (((size_t)b * N + n) * A + atom) * 3;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float v = p[k];
            if (v == v) {
                s[k] += (double)v;
                cnt[k] += 1.0;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        for (int o = 32; o > 0; o >>= 1) {
            s[k] += __shfl_down(s[k], o);
            cnt[k] += __shfl_down(cnt[k], o);
        }
    }
    if (threadIdx.x == 0)
        for (int k = 0; k < 3; ++k) com[b * 3 + k] = (float)(s[k] / cnt[k]);  // 0/0 -> NaN like nanmean of all-NaN
}

__global__ __launch_bounds__(256) void k_frames_to_backbone(const float* __restrict__ rot, const float* __restrict__ trans,
                                                            const float* __restrict__ ideal, float* __restrict__ xyz,
                                                            size_t n_atoms_total, unsigned A, unsigned n_ideal) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_atoms_total) return;
    const size_t res = i / A;
    const unsigned a = (unsigned)(i - res * A);
    f3 x = mk3(0.f, 0.f, 0.f);
    if (a < n_ideal) {
        const float* m = rot + res * 9;
        const f3 p = load3(ideal + a * 3);
        x.x = dot3(mk3(m[0], m[1], m[2]), p) + trans[res * 3 + 0];
        x.y = dot3(mk3(m[3], m[4], m[5]), p) + trans[res * 3 + 1];
        x.z = dot3(mk3(m[6], m[7], m[8]), p) + trans[res * 3 + 2];
    }
    float* o = xyz + i * 3;
    o[0] = x.x; o[1] = x.y; o[2] = x.z;
}

}  // namespace

extern "C" int ps_rigid_f32(const float* xyz_in, float* xyz_out, const float* R, int r_mode, int transpose,
                            const float* t, int t_mode, int B, int N, int A, void* stream) {
    if (!xyz_in || !xyz_out || B < 0 || N < 0 || A <= 0) return (int)hipErrorInvalidValue;
    if (r_mode < 0 || r_mode > 3 || t_mode < 0 || t_mode > 4 || (r_mode && !R) || (t_mode && !t))
        return (int)hipErrorInvalidValue;
    const size_t n = (size_t)B * N * A;
    if (n == 0) return 0;
    return ps_launch(k_rigid, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       xyz_in, xyz_out, R, t, n, (unsigned)N, (unsigned)A, r_mode, t_mode, transpose);
}

extern "C" int ps_center_of_mass_f32(const float* xyz, float* com, int B, int N, int A, int atom, void* stream) {
    if (!xyz || !com || B < 0 || N < 0 || A <= 0 || atom < 0 || atom >= A) return (int)hipErrorInvalidValue;
    if (B == 0) return 0;
    return ps_launch(k_center_of_mass, dim3(B), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), xyz, com,
                       (unsigned)N, (unsigned)A, (unsigned)atom);
}

extern "C" int ps_frames_to_backbone_f32(const float* rot, const float* trans, const float* ideal, int n_ideal,
                                         float* xyz, int B, int N, int A, void* stream) {
    if (!rot || !trans || !ideal || !xyz || B < 0 || N < 0 || A <= 0 || n_ideal < 0 || n_ideal > A)
        return (int)hipErrorInvalidValue;
    const size_t n = (size_t)B * N * A;
    if (n == 0) return 0;
    return ps_launch(k_frames_to_backbone, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), rot, trans, ideal, xyz, n, (unsigned)A, (unsigned)n_ideal);
}
