"""Multi-GPU form of the pairwise path: residue rows sharded over ranks.

One process per GPU (``torch.distributed``; backend "nccl" is RCCL on ROCm).  ``xyz`` / ``atom_mask`` are tiny
(180 B per residue) and replicated on every rank; only the O(N^2) outputs are sharded.  Rank r computes residue rows
[r*N/P, (r+1)*N/P) of every structure straight into its slice of a full-size (B, N, N, ...) buffer -- no staging
copy -- and, if ``gather`` is set, ONE grouped exchange per output plane reassembles the full tensor on every rank
over xGMI:

* backend "nccl": the native ``ps_allgather_rows`` of libprotstruc_rccl.so (include/protstruc_rccl.h) -- B in-place
  ``ncclAllGather`` calls (a rank's slice of one structure is one contiguous run) inside one
  ``ncclGroupStart`` / ``ncclGroupEnd``, on its own communicator created once per process group; uneven splits
  (N % P != 0) use one in-place ``ncclBroadcast`` per (structure, owner) in the same group;
* any other backend (gloo rehearsals on CPUs or on one GPU), or ``impl="torch"``: the same exchange through
  ``torch.distributed`` collectives.

The gather moves (P-1)/P of the whole result into every GPU and is bound by the xGMI links, not by the kernel
(SURVEY 8(e)); ``gather=False`` returns the row-sharded result, which is what a data-parallel consumer wants, and
``gather="recompute"`` produces the full tensor on every rank with no collective at all.

Sharded: ``pairwise_distance_matrix`` (reference protstruc.py:455-484) and ``pairwise_dihedrals`` /
``pairwise_planar_angles`` (protstruc.py:620-660).  The other kernels are per-residue / per-structure: they shard
over the batch with no exchange ("replicas only").
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from . import ops


def shard_rows(n_rows: int, rank: int, world: int) -> Tuple[int, int]:
    """Balanced contiguous split of ``n_rows`` residue rows: rows [lo, hi) belong to ``rank``
    (the same formula as ``ps_shard_rows`` of include/protstruc_rccl.h)."""
    return (n_rows * rank) // world, (n_rows * (rank + 1)) // world


# ---- native communicator: one per (process group), created outside any captured region -------------------------
_COMMS = {}


def _group_key(group):
    """Identity of a process group that does not survive destroy_process_group() + init_process_group(): the name
    torch gives every group it creates is unique within the process (a counter), the object id guards the rest."""
    pg = dist.group.WORLD if group is None else group
    return (getattr(pg, "group_name", None), id(pg))


def native_comm(group=None, device=None):
    """The libprotstruc_rccl communicator of (``group``, ``device``), created collectively on first use: rank 0 draws
    the RCCL unique id and ``torch.distributed`` carries it to the other ranks.  ``ps_comm_create`` binds the
    communicator to the HIP device that is current when it runs, and the object broadcast on the nccl backend uses
    the current device as well, so both happen with ``device`` (default: the current device) made current -- a caller
    whose tensors live on cuda:LOCAL_RANK but who never called ``torch.cuda.set_device`` still gets a communicator
    on the GPU its buffers are on."""
    from . import _rccl

    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if device.type != "cuda":
        raise ValueError(f"a native RCCL communicator needs a GPU, got {device}")
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    key = _group_key(group)
    if key in _COMMS:
        comm, _pg, dev_index = _COMMS[key]
        if dev_index != device.index:
            raise ValueError(f"the native communicator of this process group was created on cuda:{dev_index}; "
                             f"buffers on {device} cannot be gathered with it (one GPU per rank)")
        return comm
    lib = _rccl.load()
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    with torch.cuda.device(device):
        ident = [None]
        if rank == 0:
            buf = ctypes.create_string_buffer(_rccl.COMM_ID_BYTES)
            _rccl.check(lib.ps_comm_unique_id(buf), "ps_comm_unique_id")
            ident[0] = buf.raw
        src = dist.get_global_rank(group, 0) if group is not None and group is not dist.group.WORLD else 0
        dist.broadcast_object_list(ident, src=src, group=group)
        comm = ctypes.c_void_p()
        idbuf = ctypes.create_string_buffer(ident[0], _rccl.COMM_ID_BYTES)
        _rccl.check(lib.ps_comm_create(ctypes.byref(comm), idbuf, world, rank), "ps_comm_create")
    # the group object is kept alongside: while this entry exists its id cannot be reused by a later group
    _COMMS[key] = (comm, dist.group.WORLD if group is None else group, device.index)
    return comm


def destroy_native_comms() -> None:
    """Release every native communicator (call before ``dist.destroy_process_group()``)."""
    from . import _rccl

    while _COMMS:
        _, (comm, _pg, _dev) = _COMMS.popitem()
        _rccl.check(_rccl.load().ps_comm_destroy(comm), "ps_comm_destroy")


def _pick_impl(impl: Optional[str], tensor: torch.Tensor, group) -> str:
    if impl is None:
        impl = os.environ.get("PROTSTRUC_AMD_GATHER", "auto")
    if impl not in ("auto", "native", "torch"):
        raise ValueError(f"impl must be 'auto', 'native' or 'torch', got {impl!r}")
    if impl == "auto":
        impl = "native" if (dist.get_backend(group) == "nccl" and tensor.is_cuda) else "torch"
    if impl == "native" and not tensor.is_cuda:
        raise ValueError("the native RCCL gather needs device buffers")
    return impl


def allgather_rows(full: torch.Tensor, group=None, impl: Optional[str] = None, *, force_broadcast: bool = False) -> None:
    """In-place all-gather of row slices of a contiguous (B, N, ...) tensor: on entry rank r holds rows
    ``shard_rows(N, r, P)`` of every structure, on return every rank holds all rows.

    With one rank and no explicit ``impl`` there is nothing to exchange and nothing is called.  An EXPLICIT
    ``impl="native"`` / ``"torch"`` always issues the collectives, also at world 1 (a self-gather is legal and moves
    nothing): that is how a single GPU executes exactly the calls a multi-GPU run makes.  ``force_broadcast`` takes
    the per-(structure, owner) broadcast form that uneven splits use, whatever N is (same bits)."""
    if full.ndim < 2 or not full.is_contiguous():
        raise ValueError("allgather_rows needs a contiguous (B, N, ...) tensor")
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if world == 1 and impl in (None, "auto"):
        return
    B, N = full.shape[:2]
    if full.numel() == 0:
        return
    row_bytes = full[0, 0].numel() * full.element_size() if full.ndim > 2 else full.element_size()
    if _pick_impl(impl, full, group) == "native":
        from . import _rccl

        comm = native_comm(group, full.device)     # created (once) with the buffer's device current
        with torch.cuda.device(full.device):
            stream = ctypes.c_void_p(torch.cuda.current_stream(full.device).cuda_stream)
            rc = _rccl.load().ps_allgather_rows_ex(comm, ctypes.c_void_p(full.data_ptr()), B, N, row_bytes,
                                                   _rccl.GATHER_FORCE_BROADCAST if force_broadcast else 0, stream)
        _rccl.check(rc, "ps_allgather_rows")
        return
    _allgather_rows_torch(full.view(torch.uint8) if full.dtype == torch.bool else full, rank, world, group,
                          force_broadcast)


def _coalescing(group, device):
    """torch's grouped-collective context (ncclGroupStart/End underneath) where the backend has one."""
    import contextlib

    mgr = getattr(dist, "_coalescing_manager", None)
    if mgr is not None and dist.get_backend(group) == "nccl":
        return mgr(group=group, device=device, async_ops=False)
    return contextlib.nullcontext()


def _allgather_rows_torch(full: torch.Tensor, rank: int, world: int, group, force_broadcast: bool = False) -> None:
    B, N = full.shape[:2]
    lo, hi = shard_rows(N, rank, world)
    even = (N % world == 0) and not force_broadcast
    in_place = even and dist.get_backend(group) == "nccl"
    with _coalescing(group, full.device):
        for b in range(B):
            if even:
                # equal slices: rank r's rows sit at offset r * (N/world) of structure b -- the in-place
                # layout ncclAllGather expects (sendbuff == recvbuff + rank * count); gloo wants a separate input
                mine = full[b, lo:hi]
                dist.all_gather_into_tensor(full[b], mine if in_place else mine.clone(), group=group)
            else:
                for r in range(world):
                    rlo, rhi = shard_rows(N, r, world)
                    if rhi > rlo:
                        src = dist.get_global_rank(group, r) if group is not None and group is not dist.group.WORLD else r
                        dist.broadcast(full[b, rlo:rhi], src=src, group=group)


def pairwise_distance_matrix_sharded(xyz: torch.Tensor, atom_mask: Optional[torch.Tensor] = None, *,
                                     group=None, gather=True, impl: Optional[str] = None,
                                     out_dist: Optional[torch.Tensor] = None,
                                     out_mask: Optional[torch.Tensor] = None):
    """Row-sharded ``pairwise_distance_matrix``.

    Returns ``(dist, dist_mask, (row_lo, row_hi))``.  With ``gather=True`` both
    tensors are the full (B,N,N,A,A) result on every rank (bit-identical to the
    single-GPU kernel); with ``gather=False`` only rows [row_lo,row_hi) of them
    are defined on this rank.  ``gather="recompute"`` also leaves the full result
    on every rank but without any collective: the inputs are replicated, so each
    rank simply computes all rows itself -- one GPU writes the matrix at ~6 TB/s
    while an all-gather receives it at xGMI speed, so this is the faster way to
    the same bits whenever every rank really needs the whole matrix.
    ``impl``: "native" (RCCL through libprotstruc_rccl.so; default on the nccl backend), "torch"
    (torch.distributed collectives; default elsewhere)."""
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    B, N, A = xyz.shape[:3]
    if gather not in (True, False, "recompute"):
        raise ValueError(f"gather must be True, False or 'recompute', got {gather!r}")
    if gather == "recompute":
        d, m = ops.pairwise_distance(xyz, atom_mask, out_dist=out_dist, out_mask=out_mask)
        return d, m, (0, N)
    lo, hi = shard_rows(N, rank, world)
    shape = (B, N, N, A, A)
    if out_dist is None:
        out_dist = torch.empty(shape, dtype=torch.float32, device=xyz.device)
    if out_mask is None:
        out_mask = torch.empty(shape, dtype=torch.bool, device=xyz.device)
    ops.pairwise_distance(xyz, atom_mask, row_begin=lo, row_end=hi, out_dist=out_dist, out_mask=out_mask)
    if gather:
        allgather_rows(out_dist, group, impl)
        allgather_rows(out_mask, group, impl)
    return out_dist, out_mask, (lo, hi)


def pairwise_angles_sharded(xyz: torch.Tensor, slots_i: Sequence[int], slots_j: Sequence[int], n_points: int, *,
                            group=None, gather=True, impl: Optional[str] = None,
                            out: Optional[torch.Tensor] = None):
    """Row-sharded ``pairwise_dihedrals`` (n_points = 4) / ``pairwise_planar_angles`` (n_points = 3): the same row
    split and the same exchange as the distance matrix, on the (B, N, N) fp32 output.  Returns ``(out, (lo, hi))``."""
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    B, N = xyz.shape[:2]
    if gather not in (True, False, "recompute"):
        raise ValueError(f"gather must be True, False or 'recompute', got {gather!r}")
    if out is None:
        out = torch.empty(B, N, N, dtype=torch.float32, device=xyz.device)
    if gather == "recompute":
        return ops.pairwise_angles(xyz, slots_i, slots_j, n_points, out=out), (0, N)
    lo, hi = shard_rows(N, rank, world)
    ops.pairwise_angles(xyz, slots_i, slots_j, n_points, row_begin=lo, row_end=hi, out=out)
    if gather:
        allgather_rows(out, group, impl)
    return out, (lo, hi)
