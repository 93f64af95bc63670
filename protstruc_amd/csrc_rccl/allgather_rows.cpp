// libprotstruc_rccl.so -- RCCL side of the row-sharded geometry path (see include/protstruc_rccl.h).
// Host code only: no kernels.  Every call is a thin, argument-checked sequence of RCCL calls on the caller's stream.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <new>

#include "../../include/protstruc_rccl.h"

struct ps_comm {
    ncclComm_t comm;
    int rank, world, device;
};

static_assert(sizeof(ncclUniqueId) == PS_COMM_ID_BYTES, "ncclUniqueId is not 128 bytes in this RCCL");

namespace {
inline int from_nccl(ncclResult_t r) { return r == ncclSuccess ? 0 : 1000 + (int)r; }
}  // namespace

extern "C" int ps_rccl_abi_version(void) { return 2; }

extern "C" int ps_rccl_version(int* version) {
    if (!version) return (int)hipErrorInvalidValue;
    return from_nccl(ncclGetVersion(version));
}

extern "C" int ps_comm_unique_id(void* id_out) {
    if (!id_out) return (int)hipErrorInvalidValue;
    return from_nccl(ncclGetUniqueId(static_cast<ncclUniqueId*>(id_out)));
}

extern "C" int ps_comm_create(ps_comm** comm_out, const void* id, int world, int rank) {
    if (!comm_out || !id || world < 1 || rank < 0 || rank >= world) return (int)hipErrorInvalidValue;
    *comm_out = nullptr;
    int dev = -1;
    hipError_t he = hipGetDevice(&dev);
    if (he != hipSuccess) return (int)he;
    ps_comm* c = new (std::nothrow) ps_comm{nullptr, rank, world, dev};
    if (!c) return (int)hipErrorOutOfMemory;
    ncclUniqueId uid = *static_cast<const ncclUniqueId*>(id);
    const ncclResult_t r = ncclCommInitRank(&c->comm, world, uid, rank);
    if (r != ncclSuccess) {
        delete c;
        return from_nccl(r);
    }
    *comm_out = c;
    return 0;
}

extern "C" int ps_comm_destroy(ps_comm* comm) {
    if (!comm) return 0;
    const ncclResult_t r = ncclCommDestroy(comm->comm);
    delete comm;
    return from_nccl(r);
}

extern "C" int ps_comm_rank(const ps_comm* comm, int* rank_out, int* world_out) {
    if (!comm) return (int)hipErrorInvalidValue;
    if (rank_out) *rank_out = comm->rank;
    if (world_out) *world_out = comm->world;
    return 0;
}

extern "C" int ps_shard_rows(int n_rows, int rank, int world, int* lo_out, int* hi_out) {
    if (lo_out) *lo_out = 0;
    if (hi_out) *hi_out = 0;
    if (n_rows < 0 || world < 1 || rank < 0 || rank >= world) return (int)hipErrorInvalidValue;  // (no division by 0)
    const long long n = n_rows;
    if (lo_out) *lo_out = (int)(n * rank / world);
    if (hi_out) *hi_out = (int)(n * (rank + 1) / world);
    return 0;
}

extern "C" int ps_allgather_rows_ex(ps_comm* comm, void* full, int B, int n_rows, long long row_bytes, int flags,
                                    void* stream) {
    if (!comm || !full || B < 0 || n_rows < 0 || row_bytes < 0 || (flags & ~PS_GATHER_FORCE_BROADCAST))
        return (int)hipErrorInvalidValue;
    if (B == 0 || n_rows == 0 || row_bytes == 0) return 0;
    // world == 1 is NOT short-cut: a one-rank communicator runs the same grouped collectives (a self-gather is legal
    // in RCCL and moves nothing), so a single GPU exercises every call a multi-GPU run makes.
    int dev = -1;
    hipError_t he = hipGetDevice(&dev);
    if (he != hipSuccess) return (int)he;
    if (dev != comm->device) return (int)hipErrorInvalidDevice;  // the communicator belongs to another GPU
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    char* base = static_cast<char*>(full);
    const size_t struct_bytes = (size_t)n_rows * (size_t)row_bytes;
    const int P = comm->world;

    ncclResult_t r = ncclGroupStart();
    if (r != ncclSuccess) return from_nccl(r);
    ncclResult_t first_err = ncclSuccess;
    if (n_rows % P == 0 && !(flags & PS_GATHER_FORCE_BROADCAST)) {
        const size_t count = (size_t)(n_rows / P) * (size_t)row_bytes;  // bytes per rank per structure
        for (int b = 0; b < B && first_err == ncclSuccess; ++b) {
            char* recv = base + (size_t)b * struct_bytes;
            first_err = ncclAllGather(recv + (size_t)comm->rank * count, recv, count, ncclUint8, comm->comm, s);
        }
    } else {
        for (int b = 0; b < B && first_err == ncclSuccess; ++b) {
            for (int owner = 0; owner < P && first_err == ncclSuccess; ++owner) {
                int lo, hi;
                ps_shard_rows(n_rows, owner, P, &lo, &hi);
                if (hi == lo) continue;  // more ranks than rows: this owner has nothing (same decision on every rank)
                char* p = base + (size_t)b * struct_bytes + (size_t)lo * (size_t)row_bytes;
                first_err = ncclBroadcast(p, p, (size_t)(hi - lo) * (size_t)row_bytes, ncclUint8, owner, comm->comm, s);
            }
        }
    }
    r = ncclGroupEnd();  // always closed, also after an error inside the group
    return from_nccl(first_err != ncclSuccess ? first_err : r);
}

extern "C" int ps_allgather_rows(ps_comm* comm, void* full, int B, int n_rows, long long row_bytes, void* stream) {
    return ps_allgather_rows_ex(comm, full, B, n_rows, row_bytes, 0, stream);
}

extern "C" const char* ps_comm_error_string(int code) {
    if (code >= 1000) return ncclGetErrorString(static_cast<ncclResult_t>(code - 1000));
    return hipGetErrorString(static_cast<hipError_t>(code));
}
