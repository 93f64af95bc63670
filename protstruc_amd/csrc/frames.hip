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

// Point-wise forms of the geometry primitives for the free functions of protstruc.geometry
// (reference geometry.py:39-124, :413-439): a, b, c(, d) are (n,3) arrays.
//   mode 0: out[n]   = angle(a, b, c)        mode 1: out[n] = dihedral(a, b, c, d)
//   mode 2: out[n*9] = gram_schmidt(a, b, c) (3x3 row-major, basis vectors as columns)
__global__ __launch_bounds__(256) void k_pointwise(const float* __restrict__ a, const float* __restrict__ b,
                                                   const float* __restrict__ c, const float* __restrict__ d,
                                                   float* __restrict__ out, size_t n, int mode) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const f3 pa = load3(a + i * 3), pb = load3(b + i * 3), pc = load3(c + i * 3);
    if (mode == 0) {
        out[i] = angle3(pa, pb, pc);
    } else if (mode == 1) {
        out[i] = dihedral4(pa, pb, pc, load3(d + i * 3));
    } else {
        f3 e1, e2, e3;
        gram_schmidt3(pa, pb, pc, e1, e2, e3);
        float* o = out + i * 9;
        o[0] = e1.x; o[1] = e2.x; o[2] = e3.x;
        o[3] = e1.y; o[4] = e2.y; o[5] = e3.y;
        o[6] = e1.z; o[7] = e2.z; o[8] = e3.z;
    }
}

}  // namespace

extern "C" int ps_pointwise_f32(int mode, const float* a, const float* b, const float* c, const float* d, float* out,
                                long long n, void* stream) {
    if (mode < 0 || mode > 2 || !a || !b || !c || !out || n < 0 || (mode == 1 && !d)) return (int)hipErrorInvalidValue;
    if (n == 0) return 0;
    return ps_launch(k_pointwise, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), a, b, c, d, out, (size_t)n, mode);
}

extern "C" int ps_frames_f32(const float* xyz, float* rot, float* trans, int B, int N, int A, int a1, int a2, int a3,
                             int t_atom, void* stream) {
    if (!xyz || (!rot && !trans) || B < 0 || N < 0 || A <= 0) return (int)hipErrorInvalidValue;
    if (rot && (a1 < 0 || a1 >= A || a2 < 0 || a2 >= A || a3 < 0 || a3 >= A)) return (int)hipErrorInvalidValue;
    if (trans && (t_atom < 0 || t_atom >= A)) return (int)hipErrorInvalidValue;
    const size_t n_res = (size_t)B * N;
    if (n_res == 0) return 0;
    return ps_launch(k4_frames, dim3((unsigned)((n_res + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), xyz, rot, trans, n_res, A, a1, a2, a3, t_atom);
}
