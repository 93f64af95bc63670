// Batched Kabsch alignment -- replaces the per-structure Python loop of
// StructureBatch.align + geometry.kabsch (reference protstruc.py:880-918, geometry.py:442-480):
// for every structure, the rotation R and translation t that minimise the RMSD of R a + t against b over
// the masked atoms.  One workgroup per structure: masked centroids and the 3x3 covariance H = sum (a-ca)(b-cb)^T
// by fp64 block reductions, then (one lane) the optimal rotation in closed form from the eigenvectors of H^T H:
//   H = U S V^T  =>  R = V diag(1, 1, sign det(V U^T)) U^T = v0 u0^T + v1 u1^T + sign(det V) v2 (u0 x u1)^T
// with u_k = H v_k / s_k.  Also emits the minimum distance of every residue's atom to a point set
// (get_topk_nearest_residue_mask, protstruc.py:819-862).
#include "ps_common.hpp"

namespace {

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = PS_WAVE / 2; o > 0; o >>= 1) v += __shfl_down(v, o);
    return v;
}

// block-wide sums of NV doubles (result in every thread)
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* red /* NV*4 + NV */) {
    const int lane = threadIdx.x & (PS_WAVE - 1), wave = threadIdx.x / PS_WAVE;
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = wave_sum_d(v[k]);
    __syncthreads();
    if (lane == 0)
        for (int k = 0; k < NV; ++k) red[k * 4 + wave] = v[k];
    __syncthreads();
    if (threadIdx.x < NV) red[NV * 4 + threadIdx.x] = red[threadIdx.x * 4] + red[threadIdx.x * 4 + 1] + red[threadIdx.x * 4 + 2] + red[threadIdx.x * 4 + 3];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = red[NV * 4 + k];
}

// eigen-decomposition of a symmetric 3x3 (cyclic Jacobi); eigenvalues in w, eigenvectors in the columns of V
__device__ void jacobi3(double A[3][3], double V[3][3], double w[3]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) V[i][j] = (i == j);
    for (int sweep = 0; sweep < 12; ++sweep) {
        const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
        if (off < 1e-300) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (A[p][q] == 0.0) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) {  // A <- A J
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq;
                    A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {  // A <- J^T A
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk;
                    A[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < 3; ++i) w[i] = A[i][i];
}

__global__ __launch_bounds__(256) void k_kabsch(const float* __restrict__ src, const float* __restrict__ dst,
                                                const uint8_t* __restrict__ mask, float* __restrict__ R_out,
                                                float* __restrict__ t_out, unsigned n_atoms, size_t dst_stride,
                                                size_t mask_stride) {
    __shared__ double red[9 * 4 + 9];
    const unsigned b = blockIdx.x;
    const float* a = src + (size_t)b * n_atoms * 3;
    const float* bb = dst + (size_t)b * dst_stride;
    const uint8_t* m = mask + (size_t)b * mask_stride;

    double s7[7] = {0, 0, 0, 0, 0, 0, 0};  // sum a (3), sum b (3), count
    for (unsigned k = threadIdx.x; k < n_atoms; k += 256)
        if (m[k]) {
            s7[0] += a[k * 3]; s7[1] += a[k * 3 + 1]; s7[2] += a[k * 3 + 2];
            s7[3] += bb[k * 3]; s7[4] += bb[k * 3 + 1]; s7[5] += bb[k * 3 + 2];
            s7[6] += 1.0;
        }
    block_sum<7>(s7, red);
    const double ca[3] = {s7[0] / s7[6], s7[1] / s7[6], s7[2] / s7[6]};
    const double cb[3] = {s7[3] / s7[6], s7[4] / s7[6], s7[5] / s7[6]};

    double h[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // H[i][j] = sum (a_i - ca_i)(b_j - cb_j)
    for (unsigned k = threadIdx.x; k < n_atoms; k += 256)
        if (m[k]) {
            const double da[3] = {a[k * 3] - ca[0], a[k * 3 + 1] - ca[1], a[k * 3 + 2] - ca[2]};
            const double db[3] = {bb[k * 3] - cb[0], bb[k * 3 + 1] - cb[1], bb[k * 3 + 2] - cb[2]};
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) h[i * 3 + j] += da[i] * db[j];
        }
    block_sum<9>(h, red);
    if (threadIdx.x != 0) return;

    double K[3][3], V[3][3], w[3];   // K = H^T H
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) K[i][j] = h[0 * 3 + i] * h[0 * 3 + j] + h[1 * 3 + i] * h[1 * 3 + j] + h[2 * 3 + i] * h[2 * 3 + j];
    jacobi3(K, V, w);
    int o0 = 0, o1 = 1, o2 = 2;      // order eigenvalues descending
    if (w[o0] < w[o1]) { int t = o0; o0 = o1; o1 = t; }
    if (w[o0] < w[o2]) { int t = o0; o0 = o2; o2 = t; }
    if (w[o1] < w[o2]) { int t = o1; o1 = o2; o2 = t; }
    double v[3][3];                  // v[k] = k-th right singular vector
    const int ord[3] = {o0, o1, o2};
    for (int k = 0; k < 3; ++k)
        for (int i = 0; i < 3; ++i) v[k][i] = V[i][ord[k]];
    double u[2][3];
    for (int k = 0; k < 2; ++k) {
        double n2 = 0;
        for (int i = 0; i < 3; ++i) {
            u[k][i] = h[i * 3] * v[k][0] + h[i * 3 + 1] * v[k][1] + h[i * 3 + 2] * v[k][2];
            n2 += u[k][i] * u[k][i];
        }
        const double inv = 1.0 / sqrt(n2);
        for (int i = 0; i < 3; ++i) u[k][i] *= inv;
    }
    const double ux[3] = {u[0][1] * u[1][2] - u[0][2] * u[1][1], u[0][2] * u[1][0] - u[0][0] * u[1][2],
                          u[0][0] * u[1][1] - u[0][1] * u[1][0]};
    const double detV = v[0][0] * (v[1][1] * v[2][2] - v[1][2] * v[2][1]) - v[0][1] * (v[1][0] * v[2][2] - v[1][2] * v[2][0]) +
                        v[0][2] * (v[1][0] * v[2][1] - v[1][1] * v[2][0]);
    const double sgn = detV < 0 ? -1.0 : 1.0;
    double R[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R[i][j] = v[0][i] * u[0][j] + v[1][i] * u[1][j] + sgn * v[2][i] * ux[j];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) R_out[b * 9 + i * 3 + j] = (float)R[i][j];
        t_out[b * 3 + i] = (float)(cb[i] - (R[i][0] * ca[0] + R[i][1] * ca[1] + R[i][2] * ca[2]));
    }
}

// out[n] = min over query points q of | xyz[n][atom] - q |   (one structure)
__global__ __launch_bounds__(256) void k_min_dist(const float* __restrict__ xyz, const float* __restrict__ query,
                                                  float* __restrict__ out, unsigned N, unsigned A, unsigned atom,
                                                  unsigned n_query) {
    const unsigned n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const f3 p = load3(xyz + ((size_t)n * A + atom) * 3);
    float best = __builtin_huge_valf();
    bool any_nan = false;
    for (unsigned q = 0; q < n_query; ++q) {
        const f3 d = sub3(p, load3(query + (size_t)q * 3));
        const float dist = norm3(d);
        any_nan |= (dist != dist);
        best = fminf(best, dist);
    }
    out[n] = any_nan ? __builtin_nanf("") : best;  // torch.min propagates NaN
}

}  // namespace

extern "C" int ps_kabsch_f32(const float* src_xyz, const float* dst_xyz, const uint8_t* atom_mask, float* R, float* t,
                             int B, int n_atoms, int dst_is_shared, int mask_is_shared, void* stream) {
    if (!src_xyz || !dst_xyz || !atom_mask || !R || !t || B < 0 || n_atoms < 0) return (int)hipErrorInvalidValue;
    if (B == 0) return 0;
    return ps_launch(k_kabsch, dim3(B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src_xyz, dst_xyz,
                       atom_mask, R, t, (unsigned)n_atoms, dst_is_shared ? (size_t)0 : (size_t)n_atoms * 3,
                       mask_is_shared ? (size_t)0 : (size_t)n_atoms);
}

extern "C" int ps_min_dist_to_points_f32(const float* xyz, const float* query, float* out, int N, int A, int atom,
                                         int n_query, void* stream) {
    if (!xyz || !query || !out || N < 0 || A <= 0 || atom < 0 || atom >= A || n_query < 0) return (int)hipErrorInvalidValue;
    if (N == 0) return 0;
    return ps_launch(k_min_dist, dim3((N + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), xyz, query,
                       out, (unsigned)N, (unsigned)A, (unsigned)atom, (unsigned)n_query);
}
