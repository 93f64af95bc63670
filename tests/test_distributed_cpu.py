"""world_size-2 gloo test of the row-sharded pairwise-distance path (CPU).

The HIP kernel cannot run here, so ``ops.pairwise_distance`` is replaced by a
stand-in that fills the requested rows from the CPU oracle; what is under test
is the sharding arithmetic and the gather, which are device-independent.
"""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle_rows(xyz, atom_mask=None, *, row_begin=0, row_end=None, out_dist=None, out_mask=None, **kw):
    from oracle import protstruc_oracle as O
    d, m = O.pairwise_distance_matrix(xyz, atom_mask)
    out_dist[:, row_begin:row_end] = d[:, row_begin:row_end]
    out_mask[:, row_begin:row_end] = m[:, row_begin:row_end]
    return out_dist, out_mask


def _oracle_angle_rows(xyz, slots_i, slots_j, n_points, *, row_begin=0, row_end=None, compact=False, out=None):
    from oracle import protstruc_oracle as O
    fn = O.pairwise_dihedrals if n_points == 4 else O.pairwise_planar_angles
    full = fn(xyz, list(slots_i), list(slots_j))
    out[:, row_begin:row_end] = full[:, row_begin:row_end]
    return out


def _worker(rank, world, port, n_res, gather, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import protstruc_oracle as O
        from protstruc_amd import distributed as D
        from protstruc_amd import ops
        ops.pairwise_distance = _oracle_rows
        ops.pairwise_angles = _oracle_angle_rows
        g = torch.Generator().manual_seed(5)
        xyz = torch.randn(2, n_res, 15, 3, generator=g)
        mask = torch.rand(2, n_res, 15, generator=g) < 0.8
        nan = float("nan")
        out_d = torch.full((2, n_res, n_res, 15, 15), nan)
        out_m = torch.zeros(2, n_res, n_res, 15, 15, dtype=torch.bool)
        d, m, (lo, hi) = D.pairwise_distance_matrix_sharded(xyz, mask, gather=gather, out_dist=out_d, out_mask=out_m)
        rd, rm = O.pairwise_distance_matrix(xyz, mask)
        assert (lo, hi) == ((0, n_res) if gather == "recompute" else D.shard_rows(n_res, rank, world))
        if gather:
            ok = torch.equal(d, rd) and torch.equal(m, rm)
        else:
            ok = torch.equal(d[:, lo:hi], rd[:, lo:hi]) and torch.equal(m[:, lo:hi], rm[:, lo:hi])
            other = torch.ones(n_res, dtype=torch.bool)
            other[lo:hi] = False
            ok = ok and bool(torch.isnan(d[:, other]).all())  # nothing outside the shard was touched
        # the angle features shard the same way (reference protstruc.py:620-660)
        for npts, si, sj in ((4, [1, 4], [1, 4]), (3, [1, 4], [4])):
            buf = torch.full((2, n_res, n_res), 123.0)
            a, (alo, ahi) = D.pairwise_angles_sharded(xyz, si, sj, npts, gather=gather, out=buf)
            want = (O.pairwise_dihedrals if npts == 4 else O.pairwise_planar_angles)(xyz, si, sj)
            same = lambda x, y: torch.equal(x.isnan(), y.isnan()) and torch.equal(x.nan_to_num(9.0), y.nan_to_num(9.0))
            ok = ok and (alo, ahi) == (lo, hi)
            if gather:
                ok = ok and same(a, want)
            else:
                ok = ok and same(a[:, lo:hi], want[:, lo:hi]) and bool((a[:, other] == 123.0).all())
        q.put((rank, ok, lo, hi))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_res,gather", [(8, True), (7, True), (8, False), (9, "recompute")])
def test_row_sharded_distance_world2(n_res, gather):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + n_res + (100 if gather else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_res, gather, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(2))
    assert all(ok for _, ok, _, _ in got)
    if gather == "recompute":
        assert all((lo, hi) == (0, n_res) for _, _, lo, hi in got)  # every rank holds all rows, no collective
    else:
        assert got[0][2] == 0 and got[0][3] == got[1][2] and got[1][3] == n_res  # shards tile [0, N)


def test_native_shard_rows_matches_python():
    """ps_shard_rows (the split ps_allgather_rows assumes) is the split the launchers use."""
    import ctypes
    from protstruc_amd import _rccl
    from protstruc_amd.distributed import shard_rows
    lib = _rccl.load()   # loads RCCL itself; no GPU is touched
    assert _rccl.rccl_version() > 20000
    lo, hi = ctypes.c_int(), ctypes.c_int()
    for n in (1, 7, 8, 229, 512, 2047, 2048, 2 ** 30):
        for world in (1, 2, 3, 4, 7, 8):
            for r in range(world):
                assert lib.ps_shard_rows(n, r, world, ctypes.byref(lo), ctypes.byref(hi)) == 0
                assert (lo.value, hi.value) == shard_rows(n, r, world)
    # precondition violations are an error code with lo = hi = 0, never a division by zero in the caller's process
    for n, r, world in ((8, 0, 0), (8, 0, -1), (8, 2, 2), (8, -1, 2), (-1, 0, 1)):
        lo.value = hi.value = 99
        assert lib.ps_shard_rows(n, r, world, ctypes.byref(lo), ctypes.byref(hi)) == 1
        assert (lo.value, hi.value) == (0, 0)
    # argument errors come back before any RCCL call
    assert lib.ps_allgather_rows(None, None, 1, 8, 64, None) == 1
    assert lib.ps_allgather_rows_ex(None, None, 1, 8, 64, 0, None) == 1
    assert lib.ps_comm_create(None, None, 2, 0) == 1
    assert b"invalid" in lib.ps_comm_error_string(1).lower()


def test_rccl_header_matches_binding():
    import re
    from protstruc_amd import _rccl
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "protstruc_rccl.h")).read(), flags=re.S)
    declared = sorted(set(re.findall(r"\b(ps_[a-z0-9_]+)\s*\(", text)))
    assert declared == sorted(_rccl.SIGNATURES)
    lib = _rccl.load()
    for name in declared:
        assert hasattr(lib, name)


def test_shard_rows_partition():
    from protstruc_amd.distributed import shard_rows
    for n in (1, 7, 8, 512, 2048, 229):
        for world in (1, 2, 3, 4, 8):
            cuts = [shard_rows(n, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[r][1] == cuts[r + 1][0] for r in range(world - 1))
            assert max(hi - lo for lo, hi in cuts) - min(hi - lo for lo, hi in cuts) <= 1


def test_native_comm_needs_a_gpu_and_allgather_rows_validates():
    """The native communicator is bound to the device of the buffers it gathers: a CPU device is refused before any
    collective or RCCL call; allgather_rows refuses non-contiguous input and unknown implementations."""
    import pytest
    import torch
    from protstruc_amd import distributed as D
    with pytest.raises(ValueError, match="needs a GPU"):
        D.native_comm(device="cpu")
    with pytest.raises(ValueError, match="contiguous"):
        D.allgather_rows(torch.zeros(2, 4, 6)[:, :, ::2])
