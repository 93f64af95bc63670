// K4 -- per-residue backbone frames: Gram-Schmidt orientation + translation.
// Replaces StructureBatch.backbone_orientations / backbone_translations
// (reference protstruc.py:543-587) and geometry.gram_schmidt
// (geometry.py:413-439).  One lane per residue: 36 (+12) bytes read,
// 36 + 12 bytes written; the basis vectors are the COLUMNS of the 3x3.
#include "ps_common.hpp"

namespace {

__global__ __launch_bounds__(256) void k4_frames(const float* __restrict__ xyz, float* __restrict__ rot,
                                                 float* __restrict__ trans, size_t n_res, int A, int a1, int a2,
                                                 int a3, int t_atom) {
    const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_res) return;
    const float* p = xyz + r * (size_t)A * 3;
    if (rot) {
        f3 e1, e2, e3;
        gram_schmidt3(load3(p + a1 * 3), load3(p + a2 * 3), load3(p + a3 * 3), e1, e2, e3);
        float* o = rot + r * 9;
        o[0] = e1.x; o[1] = e2.x; o[2] = e3.x;
        o[3] = e1.y; o[4] = e2.y; o[5] = e3.y;
        o[6] = e1.z; o[7] = e2.z; o[8] = e3.z;
    }
    if (trans) {
        f3 t = load3(p + t_atom * 3);
        float* o = trans + r * 3;
        o[0] = t.x; o[1] = t.y; o[2] = t.z;
    }
}

}  // namespace

extern "C" int ps_frames_f32(const float* xyz, float* rot, float* trans, int B, int N, int A, int a1, int a2, int a3,
                             int t_atom, void* stream) {
    if (!xyz || (!rot && !trans) || B < 0 || N < 0 || A <= 0) return (int)hipErrorInvalidValue;
    if (rot && (a1 < 0 || a1 >= A || a2 < 0 || a2 >= A || a3 < 0 || a3 >= A)) return (int)hipErrorInvalidValue;
    if (trans && (t_atom < 0 || t_atom >= A)) return (int)hipErrorInvalidValue;
    const size_t n_res = (size_t)B * N;
    if (n_res == 0) return 0;
    hipLaunchKernelGGL(k4_frames, dim3((unsigned)((n_res + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), xyz, rot, trans, n_res, A, a1, a2, a3, t_atom);
    return ps_check_launch();
}
