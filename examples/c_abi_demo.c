/*
 * A plain-C consumer of the drop-in boundary: no Python, no PyTorch, no C++.
 *
 *   gcc -std=c99 -O1 -ffp-contract=off -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/c_abi_demo.c \
 *       -Lprotstruc_amd/lib -lprotstruc_hip -L/opt/rocm/lib -lamdhip64 -lm \
 *       -Wl,-rpath,$PWD/protstruc_amd/lib -Wl,-rpath,/opt/rocm/lib -o build/c_abi_demo
 *
 * Allocates with hipMalloc, calls ps_pairwise_distance_f32 / ps_pairwise_distance_cfg_f32 / ps_frames_f32 /
 * ps_diffuse_f32 on a stream of its own (the last two also captured into a hipGraph and replayed), and checks the
 * results on the host: distances against the float formula
 * sqrtf((dx*dx + dy*dy) + dz*dz) -- within 1 ulp with the default hardware square root, bit for bit with
 * ps_k1_config.exact_sqrt = 1 -- the pair mask exactly, frames for orthonormality.  Exit code 0 = all checks passed.
 * (Replaces, for a C host, what StructureBatch.pairwise_distance_matrix does in the reference: protstruc.py:455-484.)
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "protstruc_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)
#define PS(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "%s -> %d (%s)\n", #x, rc_, ps_error_string(rc_)); return 3; } } while (0)

static uint32_t lcg_state = 12345u;
static float frand(void) {   /* uniform in [-4, 4) */
    lcg_state = lcg_state * 1664525u + 1013904223u;
    return (float)(lcg_state >> 8) * (8.0f / 16777216.0f) - 4.0f;
}

static uint32_t bits_of(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

int main(void) {
    enum { B = 2, N = 21, A = 15 };
    const size_t n_xyz = (size_t)B * N * A * 3, n_mask = (size_t)B * N * A, n_out = (size_t)B * N * N * A * A;
    float* xyz = malloc(n_xyz * 4);
    uint8_t* mask = malloc(n_mask);
    float* dist = malloc(n_out * 4);
    uint8_t* dmask = malloc(n_out);
    if (!xyz || !mask || !dist || !dmask) return 1;
    for (size_t i = 0; i < n_xyz; ++i) xyz[i] = frand();
    for (size_t i = 0; i < n_mask; ++i) mask[i] = (uint8_t)((i % A) < 3 || (frand() > -3.2f));

    if (ps_abi_version() != PS_ABI_VERSION) { fprintf(stderr, "ABI %d != header %d\n", ps_abi_version(), PS_ABI_VERSION); return 4; }
    if (ps_has_experiments()) { fprintf(stderr, "experiments build\n"); return 4; }
    {   /* which kernel will the launch below take?  A pure host query: the library's own dispatcher in record-only mode */
        ps_k1_plan plan;
        memset(&plan, 0, sizeof plan);
        plan.struct_size = (int)sizeof plan;
        PS(ps_k1_plan_f32(B, N, A, 0, N, N, 0, 0, 0, 1, NULL, &plan));
        printf("K1 plan for B=%d N=%d A=%d: %s (%s), %u workgroups, %d launch(es)\n", B, N, A, plan.kernel, plan.family,
               plan.n_workgroups, plan.n_launches);
        if (plan.n_launches != 1 || plan.n_workgroups == 0 || !plan.kernel[0]) return 6;
        plan.struct_size = 8;   /* a caller built against another header is refused */
        if (ps_k1_plan_f32(B, N, A, 0, N, N, 0, 0, 0, 1, NULL, &plan) != 1) return 6;
    }

    float *d_xyz, *d_dist, *d_rot, *d_trans;
    uint8_t *d_mask, *d_dmask;
    hipStream_t stream;
    CK(hipStreamCreate(&stream));
    CK(hipMalloc((void**)&d_xyz, n_xyz * 4));
    CK(hipMalloc((void**)&d_mask, n_mask));
    CK(hipMalloc((void**)&d_dist, n_out * 4));
    CK(hipMalloc((void**)&d_dmask, n_out));
    CK(hipMalloc((void**)&d_rot, (size_t)B * N * 9 * 4));
    CK(hipMalloc((void**)&d_trans, (size_t)B * N * 3 * 4));
    CK(hipMemcpy(d_xyz, xyz, n_xyz * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_mask, mask, n_mask, hipMemcpyHostToDevice));

    for (int exact = 0; exact < 2; ++exact) {
        CK(hipMemset(d_dist, 0xFF, n_out * 4));
        CK(hipMemset(d_dmask, 7, n_out));
        if (exact) {
            ps_k1_config cfg;
            ps_k1_config_default(&cfg);
            cfg.exact_sqrt = 1;
            PS(ps_pairwise_distance_cfg_f32(d_xyz, d_mask, d_dist, d_dmask, B, N, A, 0, N, N, 0, &cfg, stream));
            cfg.struct_size = 4;   /* a caller built against another header is refused, nothing is launched */
            if (ps_pairwise_distance_cfg_f32(d_xyz, d_mask, d_dist, d_dmask, B, N, A, 0, N, N, 0, &cfg, stream) != 1) return 5;
        } else {
            PS(ps_pairwise_distance_f32(d_xyz, d_mask, d_dist, d_dmask, B, N, A, 0, N, N, 0, stream));
        }
        CK(hipStreamSynchronize(stream));
        CK(hipMemcpy(dist, d_dist, n_out * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(dmask, d_dmask, n_out, hipMemcpyDeviceToHost));
        long off_by_one = 0;
        for (int b = 0; b < B; ++b)
            for (int i = 0; i < N; ++i)
                for (int j = 0; j < N; ++j)
                    for (int a = 0; a < A; ++a)
                        for (int c = 0; c < A; ++c) {
                            const float* p = xyz + (((size_t)b * N + i) * A + a) * 3;
                            const float* q = xyz + (((size_t)b * N + j) * A + c) * 3;
                            const float dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
                            const float sx = dx * dx, sy = dy * dy, sz = dz * dz;
                            const float want = sqrtf((sx + sy) + sz);
                            const size_t o = ((((size_t)b * N + i) * N + j) * A + a) * A + c;
                            const uint32_t gb = bits_of(dist[o]), wb = bits_of(want);
                            const uint32_t ulps = gb > wb ? gb - wb : wb - gb;
                            if (ulps > (exact ? 0u : 1u)) {
                                fprintf(stderr, "dist[%d][%d][%d][%d][%d] = %.9g, want %.9g (exact=%d)\n", b, i, j, a, c, dist[o], want, exact);
                                return 6;
                            }
                            off_by_one += ulps;
                            const uint8_t wm = (uint8_t)(mask[((size_t)b * N + i) * A + a] && mask[((size_t)b * N + j) * A + c]);
                            if (dmask[o] != wm) { fprintf(stderr, "mask mismatch at %zu\n", o); return 7; }
                        }
        printf("pairwise_distance exact_sqrt=%d: %zu values ok (%ld one ulp off)\n", exact, n_out, off_by_one);
    }

    /* K4 through the same boundary: frames must be orthonormal, translations equal the CA atoms */
    PS(ps_frames_f32(d_xyz, d_rot, d_trans, B, N, A, 0, 1, 2, 1, stream));
    CK(hipStreamSynchronize(stream));
    float* rot = malloc((size_t)B * N * 9 * 4);
    float* tr = malloc((size_t)B * N * 3 * 4);
    CK(hipMemcpy(rot, d_rot, (size_t)B * N * 9 * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(tr, d_trans, (size_t)B * N * 3 * 4, hipMemcpyDeviceToHost));
    for (int r = 0; r < B * N; ++r) {
        const float* R = rot + (size_t)r * 9;
        for (int u = 0; u < 3; ++u)
            for (int v = 0; v < 3; ++v) {
                float dot = 0.f;
                for (int k = 0; k < 3; ++k) dot += R[k * 3 + u] * R[k * 3 + v];   /* columns are the basis vectors */
                if (fabsf(dot - (u == v ? 1.f : 0.f)) > 2e-5f) { fprintf(stderr, "frame %d not orthonormal\n", r); return 8; }
            }
        if (memcmp(tr + (size_t)r * 3, xyz + ((size_t)r * A + 1) * 3, 12) != 0) { fprintf(stderr, "translation %d\n", r); return 9; }
    }
    printf("frames: %d orthonormal, translations exact\n", B * N);

    /* The diffusion loop of BASELINE config 5 in miniature, captured into a hipGraph from C: T steps of
     * ps_diffuse_f32 + ps_frames_f32 (reference protstruc.py:864-878, :543-571).  The launchers allocate nothing and
     * never synchronise, so they are legal inside stream capture; the draw counter lives on the device and is advanced
     * by the kernels themselves, so every replay continues the noise stream. */
    {
        enum { T = 10 };
        uint64_t h_state[PS_RNG_STATE_WORDS];
        memset(h_state, 0, sizeof h_state);
        h_state[0] = 1234;   /* seed; word 1 = draw counter; the rest are the kernels' tickets (zero between launches) */
        uint64_t* d_state;
        float* d_beta;
        float h_beta[B] = {0.05f, 0.2f};
        CK(hipMalloc((void**)&d_state, sizeof h_state));
        CK(hipMalloc((void**)&d_beta, sizeof h_beta));
        CK(hipMemcpy(d_state, h_state, sizeof h_state, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_beta, h_beta, sizeof h_beta, hipMemcpyHostToDevice));
        hipGraph_t graph;
        hipGraphExec_t exec;
        CK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
        for (int t = 0; t < T; ++t) {
            PS(ps_diffuse_f32(d_xyz, d_beta, B, N * A * 3, d_state, NULL, stream));
            PS(ps_frames_f32(d_xyz, d_rot, d_trans, B, N, A, 0, 1, 2, 1, stream));
        }
        CK(hipStreamEndCapture(stream, &graph));
        CK(hipGraphInstantiate(&exec, graph, NULL, NULL, 0));
        float* x1 = malloc(n_xyz * 4);
        float* x2 = malloc(n_xyz * 4);
        CK(hipGraphLaunch(exec, stream));
        CK(hipStreamSynchronize(stream));
        CK(hipMemcpy(x1, d_xyz, n_xyz * 4, hipMemcpyDeviceToHost));
        CK(hipGraphLaunch(exec, stream));
        CK(hipStreamSynchronize(stream));
        CK(hipMemcpy(x2, d_xyz, n_xyz * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(h_state, d_state, sizeof h_state, hipMemcpyDeviceToHost));
        if (h_state[1] != 2 * T) { fprintf(stderr, "draw counter %llu, want %d\n", (unsigned long long)h_state[1], 2 * T); return 11; }
        for (int w = 2; w < PS_RNG_STATE_WORDS; ++w)
            if (h_state[w] != 0) { fprintf(stderr, "ticket word %d not reset\n", w); return 11; }
        if (memcmp(x1, x2, n_xyz * 4) == 0 || memcmp(x1, xyz, n_xyz * 4) == 0) { fprintf(stderr, "replay did not draw fresh noise\n"); return 11; }
        /* after 20 steps at beta 0.05 / 0.2 the coordinates are a mix of data and unit-variance noise: finite, moved */
        for (size_t i = 0; i < n_xyz; ++i)
            if (!(x2[i] == x2[i]) || fabsf(x2[i]) > 50.f) { fprintf(stderr, "coordinate %zu = %g\n", i, x2[i]); return 11; }
        CK(hipMemcpy(rot, d_rot, (size_t)B * N * 9 * 4, hipMemcpyDeviceToHost));
        for (int r = 0; r < B * N; ++r) {
            const float* R = rot + (size_t)r * 9;
            const float n0 = R[0] * R[0] + R[3] * R[3] + R[6] * R[6];
            if (fabsf(n0 - 1.f) > 2e-5f) { fprintf(stderr, "captured frame %d not unit\n", r); return 11; }
        }
        printf("hipGraph: %d captured steps replayed twice, draw counter %llu, fresh noise per replay\n", T, (unsigned long long)h_state[1]);
        hipGraphExecDestroy(exec); hipGraphDestroy(graph); hipFree(d_state); hipFree(d_beta); free(x1); free(x2);
    }

    /* argument errors come back as hipErrorInvalidValue before anything is launched */
    if (ps_pairwise_distance_f32(NULL, d_mask, d_dist, d_dmask, B, N, A, 0, N, N, 0, stream) != 1) return 10;
    if (ps_pairwise_distance_f32(d_xyz, d_mask, d_dist, d_dmask, B, N, A, 5, 3, N, 0, stream) != 1) return 10;
    hipFree(d_xyz); hipFree(d_mask); hipFree(d_dist); hipFree(d_dmask); hipFree(d_rot); hipFree(d_trans);
    hipStreamDestroy(stream);
    printf("c_abi_demo ok\n");
    return 0;
}
