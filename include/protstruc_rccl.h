/*
 * protstruc_rccl.h -- C ABI of libprotstruc_rccl.so: the one exchange step of the geometry hot path.
 *
 * The residue-row sharded forms of StructureBatch.pairwise_distance_matrix (reference protstruc.py:455-484) and
 * pairwise_dihedrals / pairwise_planar_angles (protstruc.py:620-660) leave rank r of P with rows
 * [N*r/P, N*(r+1)/P) of every structure of a full-size (B, N, ...) output buffer (ps_pairwise_distance_f32 /
 * ps_pairwise_angles_f32 of protstruc_hip.h write straight into that slice).  ps_allgather_rows reassembles the full
 * tensor on every rank with RCCL over xGMI.  The reference has no counterpart (it is single-process); this is
 * what SURVEY 8(b) lists as `ps_allgather_rows`.
 *
 * Kept in its own shared library so that libprotstruc_hip.so has no RCCL dependency.  All pointers are device
 * pointers owned by the caller; `stream` is a hipStream_t passed as void*.  Return value: 0 on success,
 * a hipError_t (1..999) for HIP / argument errors, 1000 + ncclResult_t for RCCL errors; ps_comm_error_string
 * decodes both.  One communicator per process (= per GPU); create it once, outside any captured region.
 */
#ifndef PROTSTRUC_RCCL_H
#define PROTSTRUC_RCCL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PS_COMM_ID_BYTES 128

typedef struct ps_comm ps_comm; /* opaque: one RCCL communicator bound to the device that was current at creation */

/* ABI version of this header. */
int ps_rccl_abi_version(void);

/* Version of the RCCL library actually loaded (ncclGetVersion), e.g. 22606. */
int ps_rccl_version(int* version);

/* Rank 0: a fresh rendezvous id (ncclGetUniqueId), PS_COMM_ID_BYTES bytes, to be handed to every rank out of band
 * (the Python shell broadcasts it through torch.distributed's store). */
int ps_comm_unique_id(void* id_out);

/* Every rank, collectively: join the communicator of `world` ranks as `rank`, on the CURRENT HIP device. */
int ps_comm_create(ps_comm** comm_out, const void* id, int world, int rank);

int ps_comm_destroy(ps_comm* comm);

int ps_comm_rank(const ps_comm* comm, int* rank_out, int* world_out);

/* Rows [lo, hi) of `n_rows` that belong to `rank` of `world`: lo = n_rows*rank/world, hi = n_rows*(rank+1)/world
 * (the split every launcher-side caller must use for row_begin / row_end).  Precondition: n_rows >= 0, world >= 1,
 * 0 <= rank < world; anything else returns hipErrorInvalidValue with lo = hi = 0 (never a division by zero). */
int ps_shard_rows(int n_rows, int rank, int world, int* lo_out, int* hi_out);

/*
 * In-place all-gather of row slices.  `full` is this rank's (B, n_rows, row_bytes) buffer, contiguous, of which it
 * has filled rows ps_shard_rows(n_rows, rank, world) of every structure; on completion (in stream order) it holds
 * every rank's rows.  A rank's slice of one structure is one contiguous run, so the exchange is B collectives
 * issued as ONE group (ncclGroupStart / ncclGroupEnd): in-place ncclAllGather per structure when n_rows % world == 0
 * (rank r's rows sit at offset r * count of the receive buffer, the layout RCCL's in-place form expects), else one
 * in-place ncclBroadcast per (structure, owner).  A one-rank communicator is NOT short-cut: it issues the same group
 * (a self-gather moves nothing), so one GPU exercises every RCCL call a multi-GPU run makes.
 */
int ps_allgather_rows(ps_comm* comm, void* full, int B, int n_rows, long long row_bytes, void* stream);

/* The same with flags: PS_GATHER_FORCE_BROADCAST takes the per-(structure, owner) ncclBroadcast form even when
 * n_rows % world == 0 (same bits; lets a test or an A/B run reach the uneven-split branch at any shape). */
#define PS_GATHER_FORCE_BROADCAST 1
int ps_allgather_rows_ex(ps_comm* comm, void* full, int B, int n_rows, long long row_bytes, int flags, void* stream);

const char* ps_comm_error_string(int code);

#ifdef __cplusplus
}
#endif
#endif /* PROTSTRUC_RCCL_H */
