/*
 * protstruc_hip.h -- C ABI of libprotstruc_hip.so (gfx950 / MI355X).
 *
 * The reference (dohlee/protstruc v0.0.7) has no FFI: its geometry hot path is
 * a set of Python methods on `StructureBatch` (protstruc/protstruc.py) and free
 * functions in protstruc/geometry.py.  Each entry point below replaces the
 * arithmetic of exactly one of those methods; the Python shell in
 * `protstruc_amd/structure_batch.py` binds them with ctypes and keeps the
 * reference's method signatures.  `file:line` citations are relative to the
 * reference checkout.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer owned by the caller (the only exception
 *     are the two small `src` / `atom` descriptor arrays of K3, which are HOST
 *     arrays read before the launch); the library never allocates, frees or
 *     retains memory;
 *   - coordinates are contiguous fp32 `xyz[B][N][A][3]`; masks are contiguous
 *     one-byte booleans (0 / 1; any non-zero input byte counts as true);
 *   - `stream` is a `hipStream_t` passed as `void*` (NULL = the default stream);
 *   - launches are asynchronous, perform no allocation, no synchronisation and
 *     no host read of device memory, so they may be captured into a hipGraph;
 *   - the library holds no mutable state (no tuning globals, no caches, no handles):
 *     every entry point is re-entrant and may be called concurrently from any
 *     number of threads, devices and streams;
 *   - the return value is a `hipError_t` as int (0 = success); argument errors
 *     return hipErrorInvalidValue (1) before anything is launched.
 */
#ifndef PROTSTRUC_HIP_H
#define PROTSTRUC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PS_RNG_STATE_WORDS 528

/* ABI version of this header (bumped on any signature change); ps_abi_version() returns the library's. */
#define PS_ABI_VERSION 5
int ps_abi_version(void);

/* 0 for the product library.  1 for builds made with -DPS_EXPERIMENTS (tools/ only), which contain timing
 * experiments that can write wrong values; those are unreachable -- not compiled -- in the product build. */
int ps_has_experiments(void);

/* Human-readable text for a code returned by any ps_* function. */
const char* ps_error_string(int code);

/*
 * Launch configuration of K1.  The library holds NO mutable state: every launcher is a pure function of its
 * arguments, so two threads / two devices / two streams of one process can never see each other's settings.
 * A NULL configuration (and ps_pairwise_distance_f32, which takes none) means ps_k1_config_default().
 * Every setting produces the same bits (the two square-root modes differ by at most 1 ulp); the knobs only move
 * work between kernels and change the granule a workgroup writes.  Not part of the drop-in surface; no reference
 * counterpart.
 *
 * Two kinds of fields (the layout is one flat struct for ABI stability):
 *   PRODUCTION -- what a caller or the explicit tuner (ops.autotune_pairwise_distance) sets:
 *       exact_sqrt; rows_per_block, lds_pad_kb, jt (pattern kernel granule and residency); flat_cpw, flat_lds_pad_kb,
 *       flat_fl_log2 = 0 / 6 / 7 (flat kernel granule); xcd_remap.
 *   DIAGNOSTIC -- kept only so that closed A/B comparisons and the cross-check tests stay reproducible; no caller needs
 *   them and the defaults (0) are the product:
 *       variant = 1 (the simple kernels everywhere: the cross-check of the parity tests), flat = 0 / 2 / 4 and
 *       rowphase = 1 / 2 (force or forbid a kernel family; + 16: an A/B switch of the row-phase kernel), store_nt = 1 (non-temporal stores: slower),
 *       flat_fl_log2 = 4 / 5 (the small granules of round 3's bounded A/B), experiment (must be 0: refused by the
 *       product library).
 */
typedef struct ps_k1_config {
    int struct_size;      /* = sizeof(ps_k1_config); a launcher refuses any other value (caller built against another header) */
    int exact_sqrt;       /* 0: hardware v_sqrt_f32 (exact for 85 % of inputs, 1 ulp off otherwise); 1: correctly rounded */
    int variant;          /* [diagnostic] 0: fast kernels (pattern / flat pattern / row-tile / row-phase / fixed-A flat); 1: the simple
                             kernels everywhere (slot-decode kernel for A = 15, element-per-lane kernel otherwise) */
    int flat;             /* 0: none of the fast kernels for A != 15 and no flat kernel for A = 15; 1 (default): every kernel
                             where it is the fast path; 2: force the A = 15 flat pattern kernel; 4: force the fixed-A flat
                             pattern kernel (A = 14, 15, 16, 24, 32) and no row-tile / row-phase kernel.  (3 was the any-A
                             flat kernel of rounds 1-2, removed: the row-phase kernel is faster at every atom count) */
    int rows_per_block;   /* pattern kernel: residue rows per workgroup, 1..32 (default 1, which for chains shorter than 64
                             residues means ceil(64 / N) rows); row-phase kernel: rows per lane when > 1 (default: 12 / 16 / 24
                             by atom count); row-tile kernel: rows per workgroup when > 1 (default 6) */
    int lds_pad_kb;       /* pattern kernel: KB of idle LDS per workgroup (caps resident workgroups per CU), 0..120; -1 (default): by chain
                             length -- 36 KB (3 workgroups per CU) from 256 residues on, 20 KB (5 per CU) below: eight boxes of round 4,
                             B=64, N=512: 36 KB 1.5-3.8 % ahead of 20 KB on seven; one box, N = 256 .. 2048 +2-3 %, N = 128 even, N <= 64 -7 % */
    int flat_cpw;         /* flat kernels: consecutive chunks per workgroup, 1..64 (default 1) */
    int flat_lds_pad_kb;  /* flat pattern kernel: idle LDS per workgroup, 0..100 */
    int jt;               /* pattern kernel: column residues per tile, 16 / 32 / 64 / 128, 0 = the default (32; 128 for a launch
                             that writes the mask plane only) */
    int xcd_remap;        /* 1 (default): each XCD sweeps one contiguous eighth of the output; 0: natural grid order */
    int store_nt;         /* [diagnostic] 1: non-temporal stores (slower on MI355X; kept for A/B runs) */
    int flat_fl_log2;     /* A = 15 flat pattern kernel: log2(pairs per chunk), 4..7; 0 = the default, 6 (64 pairs, 72 KB per chunk) */
    int rowphase;         /* [diagnostic] row-phase kernel (atom counts up to 64 other than 4, 8): 0 (default) where it is the default
                             dispatch -- every count up to 64 without a row-tile or fixed-A flat kernel, short chains excepted (A = 1, 8..255
                             residues; 2 <= A <= 16 up to 7..64 residues; full matrices): they take the two flat kernels; 1 also for A = 14, 15, 16, 24, 32 and
                             for those CA traces (A/B runs); 2 never (fixed-A flat / element kernels instead).  Bits 4..7 (value / 16) are A/B
                             switches of that kernel: 16 = the A = 1 seam slots written element-wise from both rows (round 3) */
    int experiment;       /* [diagnostic] must be 0 in the product library; timing experiments exist only in builds made with
                             -DPS_EXPERIMENTS (tools/), where 1 = first correctly rounded sqrt routine, 2 = store-only
                             run that writes WRONG values, +16 = fully unrolled group loop */
} ps_k1_config;

/* Fills *cfg with the defaults listed above (struct_size included). */
void ps_k1_config_default(ps_k1_config* cfg);

/*
 * K1 -- replaces StructureBatch.pairwise_distance_matrix (protstruc.py:455-484).
 *
 *   dist[b][i][j][a][c]      = | xyz[b][i][a] - xyz[b][j][c] |_2            (fp32)
 *   dist_mask[b][i][j][a][c] = atom_mask[b][i][a] && atom_mask[b][j][c]     (u8)
 *
 * The mask is NOT applied to dist (NaN / padded atoms propagate), as in the
 * reference.  Only residue rows i in [row_begin, row_end) are produced, which is
 * how the residue axis is sharded across GPUs.  The output buffers hold
 * `out_rows` residue rows per structure and row i is written at local row
 * `i - out_row_origin`:
 *     full-size buffers:   out_rows = N,                   out_row_origin = 0
 *     compact shard:       out_rows = row_end - row_begin, out_row_origin = row_begin
 * `atom_mask` may be NULL (all atoms present); `dist_mask` may be NULL (mask
 * plane not produced); `dist` may be NULL (only the mask plane produced).
 * Limits (hipErrorInvalidValue beyond them): B * out_rows * N < 2^32 pairs per launch for the flat kernels
 * (A = 15 at any N >= 16 that is not a multiple of 16; A = 14, 16, 24, 32), 2^31 workgroups for the pattern, row-tile
 * and row-phase kernels (the latter also N * A * A <= 2^28); B <= 65535 only for the two simple kernels that put the
 * structure on grid.z (A = 15 on unaligned planes or with variant = 1; A > 64; unaligned planes).  K2 / K3 run on 1-D grids:
 * any batch size up to 2^31 workgroups per launch.
 * Arithmetic: sqrt((dx*dx + dy*dy) + dz*dz) in fp32 without contraction; the square
 * root is the hardware instruction (exact for 85 % of inputs, 1 ulp off otherwise)
 * unless ps_k1_config.exact_sqrt selects the correctly rounded routine.
 */
int ps_pairwise_distance_f32(const float* xyz, const uint8_t* atom_mask,
                             float* dist, uint8_t* dist_mask,
                             int B, int N, int A,
                             int row_begin, int row_end,
                             int out_rows, int out_row_origin,
                             void* stream);

/* The same with an explicit launch configuration (NULL = defaults). */
int ps_pairwise_distance_cfg_f32(const float* xyz, const uint8_t* atom_mask,
                                 float* dist, uint8_t* dist_mask,
                                 int B, int N, int A,
                                 int row_begin, int row_end,
                                 int out_rows, int out_row_origin,
                                 const ps_k1_config* cfg, void* stream);

/*
 * Which K1 kernel a launch with these arguments takes -- a pure host query, nothing is launched and no device memory is
 * touched.  It runs the launcher's OWN dispatcher in a record-only mode (same eligibility predicates, same grid and LDS
 * arithmetic), so the answer cannot drift from what ps_pairwise_distance_cfg_f32 does.  `dist_misalign` /
 * `mask_misalign`: address of the plane modulo 16 (0 for anything torch.empty returns), or -1 when that plane is not
 * requested (NULL); `has_atom_mask`: whether atom_mask would be non-NULL.  Argument errors are the launcher's
 * (hipErrorInvalidValue).  No reference counterpart: it exists so that benchmarks and tests can name the kernel that ran
 * and assert that every kernel family is reached by some tested shape.
 */
typedef struct ps_k1_plan {
    int struct_size;            /* in: sizeof(ps_k1_plan) */
    int n_launches;             /* 0 (empty input) or 1 (every kernel writes both planes in one launch) */
    char family[48];            /* "pattern" | "flat" | "slot_decode" (A = 15); "rowtile" | "rowphase" | "flatA" |
                                   "ca_flat" | "small_flat" | "element" (other atom counts); "empty"; a second launch would be appended with " + " */
    char kernel[96];            /* kernel name with its leading template argument, e.g. "k1_pairdist_a15_pat<32>" */
    unsigned n_workgroups;      /* grid of the first launch */
    unsigned lds_bytes;         /* static + dynamic LDS per workgroup of the first launch */
    int threads_per_workgroup;
    unsigned n_workgroups_2;    /* second launch, if any */
    unsigned lds_bytes_2;
} ps_k1_plan;

int ps_k1_plan_f32(int B, int N, int A, int row_begin, int row_end, int out_rows, int out_row_origin,
                   int dist_misalign, int mask_misalign, int has_atom_mask,
                   const ps_k1_config* cfg, ps_k1_plan* plan);

/*
 * K2 -- replaces StructureBatch.backbone_dihedrals together with
 * get_n_terminal_mask / get_c_terminal_mask (protstruc.py:435-453, :486-541)
 * and geometry.dihedral (geometry.py:74-124).
 *
 *   dihedrals[b][i] = (phi, psi, omega), zero at the batch edge and at chain
 *   termini; dihedral_mask[b][i] = !(nterm, cterm, cterm) && residue_mask[b][i].
 * chain_idx is fp32 (NaN marks padding; NaN != NaN makes a terminus).
 * Atom slots 0,1,2 (N, CA, C) of each residue are read.  Any of the four
 * outputs may be NULL.
 */
int ps_backbone_dihedrals_f32(const float* xyz, const float* chain_idx,
                              const uint8_t* residue_mask,
                              float* dihedrals, uint8_t* dihedral_mask,
                              uint8_t* nterm, uint8_t* cterm,
                              int B, int N, int A, void* stream);

/*
 * K3 -- replaces StructureBatch.pairwise_dihedrals / pairwise_planar_angles and
 * the (B, N*N, n, 3) gather of _pairwise_xyz (protstruc.py:589-660), with
 * geometry.dihedral / geometry.angle (geometry.py:39-124) evaluated per pair.
 *
 *   n_points == 4: out[b][i][j] = dihedral(p0, p1, p2, p3)
 *   n_points == 3: out[b][i][j] = angle(p0, p1, p2)   (acos, no clamp -> NaN as in the reference)
 * where p_k = xyz[b][ src[k] ? j : i ][ atom[k] ].  Rows i in
 * [row_begin, row_end) are produced into an (B, out_rows, N) buffer with the same
 * row addressing as K1.
 * exact_angles (ABI 4; bit field since ABI 5), the angle counterpart of ps_k1_config.exact_sqrt:
 *   bit 0 -- arithmetic.
 *     0  the fast arithmetic: triple-product dihedral with one reciprocal square root, polynomial atan2 / acos, cosine
 *        through v_rsq_f32 -- exact where the reference is exact (+0 diagonal, NaN positions), otherwise within the
 *        conditioning gates of SURVEY hard part 3 (3.8e-6 of off-diagonal dihedrals more than 1e-5 from the reference
 *        at unit scale, max 6.5e-5; profiles/r04_k3_error_stats.log);
 *     1  the reference's order of operations (geometry.py:110-124, :64-66): three cross products, y / |b1| with a
 *        correctly rounded square root and an IEEE division, the device library's atan2f / acosf.  No entry more than
 *        1e-5 from the reference on well-conditioned inputs.  Since ABI 5 on the same per-CU sweep kernels as the fast
 *        arithmetic (the library routines restated instruction for instruction in packed form, bit-identical to the
 *        library calls of the one-column kernel); DESIGN.md section 4 has both modes' times side by side.
 *   bit 1 -- [diagnostic] the simple one-column kernel at every shape, in the arithmetic bit 0 selects: the layout-free
 *        cross-check kernel of the parity tests, like ps_k1_config.variant = 1.  Same bits as without it.
 *   (So 0 = fast, 1 = faithful, 2 = fast / one-column, 3 = faithful / one-column; anything else: hipErrorInvalidValue.)
 * The device a launch runs on is the stream's (hipStreamGetDevice); for the NULL stream, the calling thread's current device.
 */
int ps_pairwise_angles_f32(const float* xyz, float* out,
                           int B, int N, int A,
                           int n_points, const int* src, const int* atom,
                           int row_begin, int row_end,
                           int out_rows, int out_row_origin,
                           int exact_angles, void* stream);

/*
 * Which K3 / featuriser kernel a launch with these arguments takes -- pure host queries (ABI 5): nothing is launched,
 * no device memory is touched and NO HIP call is made.  They run the launchers' OWN dispatchers in a record-only mode
 * (same predicates, same grid and LDS arithmetic; K1 has the same arrangement: ps_k1_plan_f32), so the answer cannot
 * drift from what ps_pairwise_angles_f32 / ps_inter_residue_geometry_f32 do.  `out_misalign`: address of `out` modulo
 * 16 (a multiple of 4; 0 for anything torch.empty returns); `float_misalign` / `mask_misalign`: the OR of the six fp32 /
 * three mask plane addresses modulo 128; `cu_count`: compute units of the device (<= 0: 256, an MI355X).  Argument errors
 * are the launchers' (hipErrorInvalidValue).  No reference counterpart: they exist so that benchmarks and tests can name
 * the kernel that ran and assert that every arm of the dispatchers is reached by a shape that is held to the oracle.
 */
typedef struct ps_k3_plan {
    int struct_size;            /* in: sizeof(ps_k3_plan) */
    int n_launches;             /* 0 (empty input) or 1 */
    char family[32];            /* "sweep" | "flat_tiles" | "small" | "one_column" (K3); "featurise" | "featurise_tiles" | "one_column" (featuriser); "empty" */
    char kernel[96];            /* kernel name with its template arguments, e.g. "k3_sweep<NP=4,SRC=12,NC=4,VEC=1,FAITHFUL=0>" */
    int columns_per_lane;       /* sweep / featurise: 2 or 4 column residues per lane; flat_tiles: chains per lane (a 2 x 2 tile);
                                   small: the padded chain length (16 / 32); else 1 */
    int vector_stores;          /* 1: the lane's columns are adjacent (8- / 16-byte stores); 0: 64 apart / dword stores (any N);
                                   flat_tiles / featurise_tiles: 2 = four-column tiles, 16-byte stores (N % 4 == 0), 1 = two-column tiles, 8-byte stores */
    int skips_dead_groups;      /* 1: dead 64-column groups of a row's last strip are not evaluated */
    int mask_store_mode;        /* featuriser: 4 four bytes / 3 two bytes per tile row (tiles), 2 strip-local 16-byte stores, 1 flat 16-byte stores, 0 bytes */
    int write_through;          /* featuriser: sc1 stores */
    int faithful;               /* bit 0 of exact_angles */
    int rows_per_task;          /* rows of one pulled task (sweep / featurise); rows per workgroup otherwise */
    int workgroups_per_cu;      /* 1 or 2 for the per-CU kernels (their LDS request pins it), 0: not pinned */
    int structures_per_segment; /* sweep / featurise: 1; flat: structures staged in LDS together; else 0 */
    unsigned n_workgroups;
    int threads_per_workgroup;
    unsigned lds_bytes;         /* static + dynamic LDS per workgroup */
    unsigned n_tasks;           /* per-CU kernels: length of the task list, and a workgroup's share of it */
    unsigned tasks_per_workgroup;
} ps_k3_plan;

int ps_k3_plan_f32(int B, int N, int A, int n_points, const int* src, const int* atom,
                   int row_begin, int row_end, int out_rows, int out_row_origin,
                   int out_misalign, int exact_angles, int cu_count, ps_k3_plan* plan);

/*
 * K4 -- replaces StructureBatch.backbone_orientations + backbone_translations
 * (protstruc.py:543-587) and geometry.gram_schmidt (geometry.py:413-439).
 *
 *   rot[b][i] (3x3 row-major) has columns e1 = unit(a3-a2),
 *   e2 = unit((a1-a2) - (e1.(a1-a2)) e1), e3 = e1 x e2;  trans[b][i] = xyz[b][i][t_atom].
 * Either output may be NULL.
 */
int ps_frames_f32(const float* xyz, float* rot, float* trans,
                  int B, int N, int A, int a1, int a2, int a3, int t_atom,
                  void* stream);

/*
 * Point-wise geometry primitives -- replace the free functions geometry.angle,
 * geometry.dihedral and geometry.gram_schmidt (geometry.py:39-124, :413-439) on
 * (n,3) point arrays:  mode 0: out[n] = angle(a,b,c);  mode 1: out[n] =
 * dihedral(a,b,c,d);  mode 2: out[n][3][3] = gram_schmidt(a,b,c).  Radians.
 */
int ps_pointwise_f32(int mode, const float* a, const float* b, const float* c, const float* d,
                     float* out, long long n, void* stream);

/*
 * K5 -- replaces StructureBatch.diffuse_xyz (protstruc.py:864-878), in place:
 *   xyz[b] <- sqrt(1 - beta[b]) * xyz[b] + sqrt(beta[b]) * eps,  eps ~ N(0,1) iid.
 * `n_per_struct` = N*A*3.  If `noise` is non-NULL it supplies eps (parity with
 * the reference's deterministic part); otherwise eps comes from Philox4x32-10 +
 * Box-Muller keyed by rng_state[0] (seed) at counter offset rng_state[1], and
 * rng_state[1] is advanced on the device by the last workgroup to finish, so that
 * a captured graph replays with fresh noise without a second launch.  rng_state is
 * a device array of PS_RNG_STATE_WORDS (528) uint64: word 0 = seed, word 1 =
 * offset, the rest are completion tickets that must be zero when a call starts
 * (the kernels leave them zero).
 */
int ps_diffuse_f32(float* xyz, const float* beta, int B, int n_per_struct,
                   uint64_t* rng_state, const float* noise, void* stream);

/*
 * K6 -- replaces StructureBatch.standardize (protstruc.py:696-734), in place,
 * with per-structure statistics (what the reference computes at B == 1; its
 * B > 1 broadcast is a defect):
 *   cnt = sum(mask); mu = sum(nan_to_num(xyz*mask))/cnt;
 *   std = sqrt(sum((nan_to_num(xyz)-mu)^2 * mask)/cnt); xyz <- (xyz-mu)/std.
 * mu and std are (B,3) outputs.
 */
int ps_standardize_f32(float* xyz, const uint8_t* atom_mask, float* mu, float* std,
                       int B, int N, int A, void* stream);

/* The same with the kernel chosen explicitly: variant 0 = what ps_standardize_f32 does (the LDS-resident kernel --
 * one read and one write of the coordinates -- when a structure fits in 150 KB of LDS, i.e. N*A <= 12800 atoms);
 * variant 1 = always the three-sweep streaming kernel.  Both produce the same bits; this entry point exists for
 * that cross-check and for timing. */
int ps_standardize_variant_f32(float* xyz, const uint8_t* atom_mask, float* mu, float* std,
                               int B, int N, int A, int variant, void* stream);

/*
 * Replaces StructureBatch.unstandardize (protstruc.py:736-744), in place:
 *   xyz[b] <- xyz[b] * scale[b] + shift[b]   per coordinate axis; scale, shift are (B,3).
 */
int ps_affine_f32(float* xyz, const float* scale, const float* shift,
                  int B, int n_atoms_per_struct, void* stream);

/*
 * Fused step of the diffusion loop (BASELINE config 5): K5 followed by K4 on the
 * freshly diffused coordinates in one launch -- replaces the pair of calls
 * diffuse_xyz (protstruc.py:864-878) + backbone_orientations / backbone_translations
 * (protstruc.py:543-587).  Arguments as K5 and K4; the noise stream is identical
 * to the one ps_diffuse_f32 would draw from the same rng_state.
 */
int ps_diffuse_frames_f32(float* xyz, const float* beta, int B, int N, int A,
                          uint64_t* rng_state, const float* noise,
                          float* rot, float* trans, int a1, int a2, int a3, int t_atom,
                          void* stream);

/*
 * The whole diffusion loop (BASELINE config 5) in one launch: T steps of
 * ps_diffuse_frames_f32 with the coordinates resident in LDS between steps.
 * betas is (T, B); rot (T,B,N,3,3), trans (T,B,N,3) and xyz_traj (T,B,N,A,3) are
 * optional per-step outputs (NULL = not produced); xyz holds the final
 * coordinates.  Bit-identical to T successive calls of ps_diffuse_frames_f32
 * from the same rng_state (which ends up advanced by T).  rng_state is required.
 */
int ps_diffusion_trajectory_f32(float* xyz, const float* betas, int T, int B, int N, int A,
                                uint64_t* rng_state, float* rot, float* trans, float* xyz_traj,
                                int a1, int a2, int a3, int t_atom, void* stream);

/*
 * Fused trRosetta featuriser -- replaces StructureBatch.inter_residue_geometry
 * (protstruc.py:790-817) without materialising the (B,N,N,A,A) tensor: writes
 * six (B,N,N) fp32 planes d_ca, d_cb, d_no (slices [CA,CA], [CB,CB], [N,O] of
 * K1's dist), omega = dihedral(CA_i,CB_i,CA_j,CB_j), theta =
 * dihedral(N_i,CA_i,CB_i,CB_j), phi = angle(CA_i,CB_i,CB_j), and the three
 * (B,N,N) u8 mask planes.  atom_mask may be NULL (all present).  Needs A >= 5.
 * exact_sqrt: the square root of the three distance planes, as ps_k1_config.exact_sqrt -- 0: hardware v_sqrt_f32
 * (K1's default: the planes are then bit-identical to the slices of a default K1 launch), 1: correctly rounded
 * (bit-identical to K1 with exact_sqrt = 1).
 * exact_angles (ABI 4; bit field since ABI 5): omega, theta and phi in the arithmetic ps_pairwise_angles_f32 uses for the
 * same value (bit 0: 0 fast, 1 the reference's order of operations; bit 1: the one-column kernel at every shape): the three
 * planes equal the corresponding K3 launches bit for bit in every mode.
 * Placement of the nine planes: any 4-byte (fp32) / 1-byte (mask) boundary, each plane its own; results do not depend on
 * it.  Fastest where every plane starts on a 16-byte boundary (vector stores; the three mask planes then share one 16-byte
 * grid) -- protstruc_amd.ops pads the plane stride accordingly.
 */
int ps_inter_residue_geometry_f32(const float* xyz, const uint8_t* atom_mask,
                                  float* d_ca, float* d_cb, float* d_no,
                                  float* omega, float* theta, float* phi,
                                  uint8_t* d_ca_mask, uint8_t* d_cb_mask, uint8_t* d_no_mask,
                                  int B, int N, int A, int exact_sqrt, int exact_angles, void* stream);

/* The featuriser's twin of ps_k3_plan_f32 (ABI 5; see there). */
int ps_featuriser_plan_f32(int B, int N, int A, int float_misalign, int mask_misalign,
                           int exact_sqrt, int exact_angles, int cu_count, ps_k3_plan* plan);

/*
 * Rigid-body ops.  ps_rigid_f32 replaces StructureBatch.translate,
 * rotate, center_at and get_local_xyz (protstruc.py:662-694, :759-788, :347-362):
 *   out = R x + t   (transpose = 0)   or   out = R^T x + t   (transpose = 1)
 * r_mode: 0 none, 1 one shared 3x3, 2 per structure (B,3,3), 3 per residue (B,N,3,3);
 * t_mode: 0 none, 1 one shared (3), 2 per structure (B,3), 3 per residue (B,N,3),
 *         4 per atom (B,N,A,3).  xyz_out may alias xyz_in.
 */
int ps_rigid_f32(const float* xyz_in, float* xyz_out, const float* R, int r_mode, int transpose,
                 const float* t, int t_mode, int B, int N, int A, void* stream);

/* Replaces StructureBatch.center_of_mass (protstruc.py:746-757): com[b] = nanmean over residues of xyz[b][:][atom]. */
int ps_center_of_mass_f32(const float* xyz, float* com, int B, int N, int A, int atom, void* stream);

/*
 * Replaces the coordinate construction of StructureBatch.from_backbone_orientations_translations
 * (protstruc.py:289-312): xyz[b][n][a] = rot[b][n] * ideal[a] + trans[b][n] for a < n_ideal, 0 for
 * the remaining slots.  ideal is a device array (n_ideal, 3).
 */
int ps_frames_to_backbone_f32(const float* rot, const float* trans, const float* ideal, int n_ideal,
                              float* xyz, int B, int N, int A, void* stream);

/*
 * Batched Kabsch fit -- replaces the per-structure loop of StructureBatch.align and
 * geometry.kabsch (protstruc.py:880-918, geometry.py:442-480): R[b] (3x3), t[b] (3) minimising the RMSD of
 * R a + t against b over the atoms with atom_mask != 0.  src/dst are (B, n_atoms, 3); dst_is_shared /
 * mask_is_shared = 1 when one target / one mask serves every structure.  Apply with ps_rigid_f32.
 */
int ps_kabsch_f32(const float* src_xyz, const float* dst_xyz, const uint8_t* atom_mask, float* R, float* t,
                  int B, int n_atoms, int dst_is_shared, int mask_is_shared, void* stream);

/*
 * Distance of one atom slot of every residue of ONE structure to its nearest query point -- the distance
 * part of StructureBatch.get_topk_nearest_residue_mask (protstruc.py:844-849).  xyz (N,A,3), query (n_query,3).
 */
int ps_min_dist_to_points_f32(const float* xyz, const float* query, float* out, int N, int A, int atom,
                              int n_query, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PROTSTRUC_HIP_H */
