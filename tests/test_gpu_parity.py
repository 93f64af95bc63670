"""GPU parity: HIP path (through the C ABI) vs golden fixtures from the reference
and vs the CPU oracle on seeded inputs.  Run on the MI355X box with -m gpu.

Tolerances (north_star): fp32 within 1e-5 abs at unit coordinate scale, masks
and NaN positions bit-exact.  For the ill-conditioned angle outputs (acos near
+-1, atan2 near the origin) the reference's own fp32 result is >1e-5 away from
an fp64 evaluation on ~1e-5 of entries (SURVEY hard part 3), so those gates
allow that fraction and bound the error against fp64 truth instead.
"""
import numpy as np
import pytest
import torch

from oracle import protstruc_oracle as O
from tests.conftest import load_golden

pytestmark = pytest.mark.gpu

TOL = 1e-5
SLOT = {"N": 0, "CA": 1, "C": 2, "O": 3, "CB": 4}
NAME = {v: k for k, v in SLOT.items()}


@pytest.fixture(scope="module")
def SB():
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    from protstruc_amd import StructureBatch
    from protstruc_amd import _lib
    _lib.load()
    return StructureBatch


def assert_close(got, want, tol=TOL, bad_frac=0.0):
    got = got.detach().cpu()
    assert got.shape == want.shape, (got.shape, want.shape)
    assert got.dtype == want.dtype, (got.dtype, want.dtype)
    assert torch.equal(torch.isnan(got), torch.isnan(want)), "NaN positions differ"
    err = (got - want).abs().nan_to_num(0.0)
    frac = (err > tol).float().mean().item()
    assert frac <= bad_frac, f"{frac:.3e} of entries differ by more than {tol} (max {err.max().item():.3e})"


def synth(seed, B, N, A=15, p=0.9, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    xyz = torch.randn(B, N, A, 3, generator=g) * scale
    mask = torch.rand(B, N, A, generator=g) < p
    mask[:, :, :3] = True
    return xyz, mask


# ----------------------------------------------------------------------------- K1
@pytest.mark.parametrize("name", ["g1_dist_b2_n8", "g1_dist_b1_n21", "g1_dist_b2_n6_a25", "g1_dist_floatmask",
                                  "g1_dist_nan"])
def test_k1_golden(SB, name):
    g = load_golden(name)
    d, m = SB.from_xyz(g["xyz"], g["atom_mask"]).pairwise_distance_matrix()
    assert_close(d, g["dist"])
    assert m.dtype == g["dist_mask"].dtype
    assert torch.equal(m.cpu(), g["dist_mask"])


@pytest.mark.parametrize("fixture,t", [("g13_dist_atom_counts", f"a{A}") for A in (14, 37, 25, 3, 4, 5, 8, 16)] +
                         [("g14_dist_small_atom_counts", t) for t in ("a1", "a2", "a6", "a7", "a10", "a13", "a1n7", "a9")])
def test_k1_golden_other_atom_counts(SB, fixture, t):
    """G13 / G14: the reference itself at the atom counts that take the fixed-A flat pattern, row-tile and row-phase
    kernels (G14: single atoms = CA traces, atom pairs, 6 / 7 / 9 / 10 / 13 atoms, lengths of every alignment phase and
    below 16): sampled whole (b,i,j) blocks within 1e-5 with NaN positions exact, mask blocks and every pair's mask
    count exact, per-pair distance sums.  (`from_xyz` accepts any atom count: reference protstruc.py:94-128,
    tests/test_StructureBatch.py:11-21.)"""
    g = load_golden(fixture)
    sb = SB.from_xyz(g[f"{t}_xyz"], g[f"{t}_atom_mask"])
    d, m = sb.pairwise_distance_matrix()
    assert m.dtype == torch.bool and d.shape[-1] == g[f"{t}_xyz"].shape[2]
    b, i, j = g[f"{t}_b"].long().cuda(), g[f"{t}_i"].long().cuda(), g[f"{t}_j"].long().cuda()
    assert_close(d[b, i, j], g[f"{t}_dist_blocks"])
    assert torch.equal(m[b, i, j].cpu(), g[f"{t}_mask_blocks"])
    assert torch.equal(m.sum((3, 4)).to(torch.int32).cpu(), g[f"{t}_mask_row_sums"])
    sums = torch.nan_to_num(d, nan=0.0).double().sum((3, 4)).float().cpu()
    assert torch.allclose(sums, g[f"{t}_dist_row_nansum"], rtol=2e-6, atol=1e-4)


def test_k1_golden_protein_scale(SB):
    g = load_golden("g1_dist_b1_n12_protein_scale")
    d, m = SB.from_xyz(g["xyz"], g["atom_mask"]).pairwise_distance_matrix()
    ulp = torch.finfo(torch.float32).eps * g["dist"].abs().clamp_min(1.0)
    assert ((d.cpu() - g["dist"]).abs() <= 2 * ulp).all()
    assert torch.equal(m.cpu(), g["dist_mask"])


@pytest.mark.parametrize("B,N", [(1, 1), (1, 2), (2, 5), (1, 16), (2, 64), (1, 65), (1, 100), (3, 128), (1, 229),
                                 (2, 256)])
def test_k1_vs_oracle(SB, B, N):
    """Ragged N (not multiples of 4 / 16 / the 64-residue tile) exercise the unaligned head/tail paths."""
    xyz, mask = synth(100 + N, B, N)
    d, m = SB.from_xyz(xyz, mask).pairwise_distance_matrix()
    rd, rm = O.pairwise_distance_matrix_chunked(xyz, mask)
    assert_close(d, rd)
    assert torch.equal(m.cpu(), rm)


def test_k1_every_kernel_family_vs_oracle(SB):
    """tests/k1_families.py: one launch per kernel family behind the K1 entry point.  Each entry is first confirmed --
    through the library's own dispatcher, for the very buffers used -- to select the family it names, then run on the
    GPU and held to the oracle (distances 1e-5 with NaN positions exact, mask exact), with sentinels around the output
    and, for row ranges written into a full-size buffer, in the rows that must stay untouched."""
    from protstruc_amd import _lib, ops
    from tests.k1_families import ALL_FAMILIES, FAMILY_SHAPES
    keys = sorted({k for e in FAMILY_SHAPES for k in e[5]})
    saved = {k: _lib.get_tuning(k) for k in keys}
    SENT, ran = 777.0, set()
    try:
        for B, N, A, rows, compact, overrides, family in FAMILY_SHAPES:
            for k in keys:
                _lib.set_tuning(k, overrides.get(k, saved[k]))
            xyz, mask = synth(900 + 7 * N + A, B, N, A=A)
            xyz[B - 1, N // 2, A - 1] = float("nan")
            r0, r1 = rows if rows else (0, N)
            out_rows = (r1 - r0) if compact else N
            numel, pad = B * out_rows * N * A * A, 64
            bd = torch.full((numel + 2 * pad,), SENT, device="cuda")
            bm = torch.full((numel + 2 * pad,), 7, dtype=torch.uint8, device="cuda")
            d = bd[pad:pad + numel].view(B, out_rows, N, A, A)
            m = bm[pad:pad + numel].view(torch.bool).view(B, out_rows, N, A, A)
            plan = _lib.k1_plan(B, N, A, r0, r1, compact=compact, dist_misalign=d.data_ptr() % 16,
                                mask_misalign=m.data_ptr() % 16)
            assert plan["family"] == family, (B, N, A, rows, overrides, plan)
            ops.pairwise_distance(xyz.cuda(), mask.cuda(), row_begin=r0, row_end=r1, compact=compact, out_dist=d, out_mask=m)
            rd, rm = O.pairwise_distance_matrix(xyz, mask)
            got_d = d if compact else d[:, r0:r1]
            got_m = m if compact else m[:, r0:r1]
            assert_close(got_d, rd[:, r0:r1])
            assert torch.equal(got_m.cpu(), rm[:, r0:r1]), (family, B, N, A)
            assert (bd[:pad] == SENT).all() and (bd[pad + numel:] == SENT).all(), (family, B, N, A)
            assert (bm[:pad] == 7).all() and (bm[pad + numel:] == 7).all(), (family, B, N, A)
            if not compact:
                assert (d[:, :r0] == SENT).all() and (d[:, r1:] == SENT).all(), (family, B, N, A)
                raw = bm[pad:pad + numel].view(B, N, N, A, A)
                assert (raw[:, :r0] == 7).all() and (raw[:, r1:] == 7).all(), (family, B, N, A)
            ran.add(family)
    finally:
        for k, v in saved.items():
            _lib.set_tuning(k, v)
    assert ran == ALL_FAMILIES


def test_k1_config2_exact_shape(SB):
    """BASELINE config 2's K1 half at its exact shape (B=64, N=256; 4.7 GB of output): sampled blocks against the oracle's
    formula, exact mask checksum of every structure, bitwise symmetry of two structures, no element left unwritten."""
    from protstruc_amd import _lib, ops
    B, N = 64, 256
    xyz, mask = synth(2, B, N)
    xg, mg = xyz.cuda(), mask.cuda()
    d = torch.full((B, N, N, 15, 15), float("nan"), device="cuda")
    m = torch.zeros(B, N, N, 15, 15, dtype=torch.bool, device="cuda")
    assert _lib.k1_plan(B, N, 15)["kernel"] == "k1_pairdist_a15_pat<32>"
    ops.pairwise_distance(xg, mg, out_dist=d, out_mask=m)
    g = torch.Generator().manual_seed(22)
    bs = torch.randint(0, B, (256,), generator=g)
    is_ = torch.cat([torch.randint(0, N, (252,), generator=g), torch.tensor([0, N - 1, N - 1, 0])])
    js = torch.cat([torch.randint(0, N, (252,), generator=g), torch.tensor([0, N - 1, 0, N - 1])])
    want = torch.norm(xyz[bs, is_][:, :, None, :] - xyz[bs, js][:, None, :, :], dim=-1)
    assert_close(d[bs.cuda(), is_.cuda(), js.cuda()], want)
    assert torch.equal(m[bs.cuda(), is_.cuda(), js.cuda()].cpu(), mask[bs, is_][:, :, None] & mask[bs, js][:, None, :])
    per_struct = mask.reshape(B, -1).sum(1).to(torch.int64)
    assert torch.equal(torch.stack([torch.count_nonzero(m[b]) for b in range(B)]).cpu(), per_struct * per_struct)
    assert not torch.isnan(d).any()
    for b in (0, B - 1):
        assert torch.equal(d[b], d[b].permute(1, 0, 3, 2))
    # one structure in full against the oracle
    rd, rm = O.pairwise_distance_matrix(xyz[17:18], mask[17:18])
    assert_close(d[17:18], rd)
    assert torch.equal(m[17:18].cpu(), rm)


@pytest.mark.parametrize("A,N", [(15, 16), (15, 21), (14, 18), (5, 20), (37, 16), (7, 12), (4, 9), (4, 131), (8, 33), (8, 7), (5, 32), (3, 16), (5, 20), (3, 28), (5, 21), (3, 18), (5, 7)])
def test_k1_special_values(SB, A, N):
    """Infinite, huge, tiny, NaN and signed-zero coordinates propagate exactly as in the reference's arithmetic
    (protstruc.py:477-479: difference, square, sum, square root; nothing is masked or clamped): NaN and inf positions
    equal the oracle's, finite values within 1e-5 relative, in every K1 kernel family and both square-root modes."""
    from protstruc_amd import ops
    g = torch.Generator().manual_seed(90 + A)
    xyz = torch.randn(2, N, A, 3, generator=g)
    xyz[0, 1, 0] = float("inf")
    xyz[0, 2, 1, 0] = float("-inf")
    xyz[0, 3, 2] = float("nan")
    xyz[0, 4, 0] = 3e19            # squares overflow to inf
    xyz[0, 5, 0] = torch.tensor([1e-25, -1e-25, 0.0])
    xyz[0, 6, 0] = torch.tensor([-0.0, 0.0, -0.0])
    xyz[1, 0, :] = xyz[1, 1, :]    # two identical residues: exact zeros off the diagonal
    mask = torch.rand(2, N, A, generator=g) < 0.8
    want, wmask = O.pairwise_distance_matrix(xyz, mask)
    for exact in (False, True):
        ops.set_exact_sqrt(exact)
        try:
            d, m = SB.from_xyz(xyz, mask).pairwise_distance_matrix()
        finally:
            ops.set_exact_sqrt(False)
        d = d.cpu()
        assert torch.equal(m.cpu(), wmask)
        assert torch.equal(d.isnan(), want.isnan()), "NaN positions"
        assert torch.equal(d.isinf(), want.isinf()), "inf positions"
        fin = torch.isfinite(want)
        rel = ((d[fin] - want[fin]).abs() / want[fin].clamp_min(1e-30))
        assert (rel[want[fin] > 0] <= 1e-5).all()
        assert (d[fin][want[fin] == 0] == 0).all(), "exact zeros stay exact zeros"


def test_k1_no_mask_and_symmetry(SB):
    xyz, _ = synth(7, 2, 48)
    d, m = SB.from_xyz(xyz).pairwise_distance_matrix()
    assert m.dtype == torch.bool and bool(m.all())
    # size-independent properties: d[b,i,j,a,c] == d[b,j,i,c,a] bit-for-bit, zero self-distance
    assert torch.equal(d, d.permute(0, 2, 1, 4, 3))
    idx = torch.arange(48)
    self_d = d[:, idx, idx][:, :, torch.arange(15), torch.arange(15)]
    assert (self_d == 0).all()


def test_k1_row_shards_reassemble(SB):
    from protstruc_amd import ops
    xyz, mask = synth(8, 2, 96)
    xyz, mask = xyz.cuda(), mask.cuda()
    full_d, full_m = ops.pairwise_distance(xyz, mask)
    out_d = torch.full_like(full_d, float("nan"))
    out_m = torch.zeros_like(full_m)
    for r0, r1 in [(0, 24), (24, 50), (50, 51), (51, 96)]:
        ops.pairwise_distance(xyz, mask, row_begin=r0, row_end=r1, out_dist=out_d, out_mask=out_m)
        cd, cm = ops.pairwise_distance(xyz, mask, row_begin=r0, row_end=r1, compact=True)
        assert torch.equal(cd, full_d[:, r0:r1]) and torch.equal(cm, full_m[:, r0:r1])
    assert torch.equal(out_d, full_d) and torch.equal(out_m, full_m)


def test_k1_store_policy_variants_agree(SB):
    from protstruc_amd import _lib, ops
    xyz, mask = synth(9, 2, 208)   # 208 = 3 full 64-tiles + a 16-residue tail; 1 full 128-tile + 80
    xyz, mask = xyz.cuda(), mask.cuda()
    base = ops.pairwise_distance(xyz, mask)
    rd, rm = O.pairwise_distance_matrix_chunked(xyz.cpu(), mask.cpu())
    assert_close(base[0], rd)
    nt0, rows0, var0, jt0 = (_lib.get_tuning(k) for k in ("k1_store_nt", "k1_rows_per_block", "k1_variant", "k1_jt"))
    try:
        for var in (0, 1):          # pattern kernel / slot-decode kernel
            for jt in (0, 16, 32, 64, 128):
                for nt in (0, 1):
                    for rows in (1, 3, 8, 16, 32):
                        _lib.set_tuning("k1_variant", var)
                        _lib.set_tuning("k1_jt", jt)
                        _lib.set_tuning("k1_store_nt", nt)
                        _lib.set_tuning("k1_rows_per_block", rows)
                        for remap in (0, 1):
                            _lib.set_tuning("k1_xcd_remap", remap)
                            d = torch.full_like(base[0], float("nan"))
                            m = torch.zeros_like(base[1])
                            ops.pairwise_distance(xyz, mask, out_dist=d, out_mask=m)
                            assert torch.equal(d, base[0]) and torch.equal(m, base[1]), (var, jt, nt, rows, remap)
    finally:
        _lib.set_tuning("k1_store_nt", nt0)
        _lib.set_tuning("k1_rows_per_block", rows0)
        _lib.set_tuning("k1_variant", var0)
        _lib.set_tuning("k1_jt", jt0)


def test_k1_pattern_kernel_single_plane_launches(SB):
    """The pattern kernel with only one plane requested: a distance-only launch and a mask-only launch (which stages no
    coordinates and takes 128-residue tiles by default) reproduce the planes of the fused launch bit for bit, at every tile
    length, with and without an atom mask, for row ranges and several rows per workgroup."""
    from protstruc_amd import _lib, ops
    keys = ("k1_jt", "k1_rows_per_block", "k1_lds_pad_kb")
    saved = {k: _lib.get_tuning(k) for k in keys}
    try:
        for (B, N) in [(3, 16), (2, 48), (2, 208), (1, 256)]:
            xyz, mask = synth(300 + N, B, N)
            xyz[0, N // 3] = float("nan")
            xg, mg = xyz.cuda(), mask.cuda()
            for am in (mg, None):
                for k, v in saved.items():
                    _lib.set_tuning(k, v)
                assert _lib.k1_plan(B, N, 15)["family"] == "pattern"
                ref_d, ref_m = ops.pairwise_distance(xg, am)
                for jt, rows in [(0, 1), (16, 1), (32, 3), (64, 1), (128, 2), (16, 32)]:
                    _lib.set_tuning("k1_jt", jt)
                    _lib.set_tuning("k1_rows_per_block", rows)
                    d0, none_m = ops.pairwise_distance(xg, am, want_mask=False)
                    none_d, m1 = ops.pairwise_distance(xg, am, want_dist=False)
                    assert none_m is None and none_d is None
                    assert torch.equal(d0.view(torch.int32), ref_d.view(torch.int32)), (B, N, jt, rows)
                    assert torch.equal(m1, ref_m), (B, N, jt, rows)
                    r0, r1 = N // 4, N - 1
                    cd, _ = ops.pairwise_distance(xg, am, row_begin=r0, row_end=r1, compact=True, want_mask=False)
                    _, cm = ops.pairwise_distance(xg, am, row_begin=r0, row_end=r1, compact=True, want_dist=False)
                    assert torch.equal(cd.view(torch.int32), ref_d[:, r0:r1].contiguous().view(torch.int32))
                    assert torch.equal(cm, ref_m[:, r0:r1])
    finally:
        for k, v in saved.items():
            _lib.set_tuning(k, v)


@pytest.mark.parametrize("exact", [0, 1])
def test_k1_flat_kernel_bit_identical_to_slot_decode(SB, exact):
    """The flat pattern kernel (any N >= 16) against the slot-decode kernel: same bits, nothing written outside the
    requested rows, for full / compact / in-place row ranges, chunks that span rows and structures, NaN atoms."""
    from protstruc_amd import _lib, ops
    keys = ("k1_variant", "k1_flat", "k1_flat_cpw", "k1_store_nt", "k1_exact_sqrt", "k1_flat_fl_log2")
    saved = {k: _lib.get_tuning(k) for k in keys}
    SENT = 12345.0
    try:
        _lib.set_tuning("k1_exact_sqrt", exact)
        for (B, N) in [(1, 16), (3, 17), (2, 18), (5, 19), (2, 37), (3, 100), (2, 127), (2, 128), (1, 437), (2, 250)]:
            xyz, mask = synth(100 + N, B, N)
            xyz[0, N // 3] = float("nan")
            mask[0, N // 3] = False
            xg, mg = xyz.cuda(), mask.cuda()
            _lib.set_tuning("k1_variant", 1)
            ref_d, ref_m = ops.pairwise_distance(xg, mg)
            ref_d0, _ = ops.pairwise_distance(xg, None)
            _lib.set_tuning("k1_variant", 0)
            _lib.set_tuning("k1_flat", 2)
            numel = ref_d.numel()
            # (chunks per workgroup, non-temporal stores, log2 of pairs per chunk: 0 = the default 128-pair chunks;
            # 4 / 5 / 6 = the 16 / 32 / 64-pair chunks of round 3's small-granule A/B)
            for cpw, nt, fl in [(1, 0, 0), (2, 0, 0), (5, 1, 0), (1, 0, 4), (3, 0, 5), (2, 0, 6)]:
                _lib.set_tuning("k1_flat_cpw", cpw)
                _lib.set_tuning("k1_store_nt", nt)
                _lib.set_tuning("k1_flat_fl_log2", fl)
                # full matrix, written into the middle of a larger sentinel buffer (16-byte aligned offset)
                pad = 64
                bd = torch.full((numel + 2 * pad,), SENT, device="cuda")
                bm = torch.full((numel + 2 * pad,), 7, dtype=torch.uint8, device="cuda")
                d = bd[pad:pad + numel].view(ref_d.shape)
                m = bm[pad:pad + numel].view(torch.bool).view(ref_m.shape)
                ops.pairwise_distance(xg, mg, out_dist=d, out_mask=m)
                assert torch.equal(d.view(torch.int32), ref_d.view(torch.int32)), (B, N, cpw)
                assert torch.equal(m, ref_m), (B, N, cpw)
                assert (bd[:pad] == SENT).all() and (bd[pad + numel:] == SENT).all()
                assert (bm[:pad] == 7).all() and (bm[pad + numel:] == 7).all()
                d0, _ = ops.pairwise_distance(xg, None, want_mask=False)
                assert torch.equal(d0.view(torch.int32), ref_d0.view(torch.int32))
                _, m1 = ops.pairwise_distance(xg, mg, want_dist=False)
                assert torch.equal(m1, ref_m)
                # row ranges: compact buffer, and in place inside a full-size buffer whose other rows stay untouched
                for r0, r1 in [(0, 1), (1, N - 1), (N // 2, N), (N - 1, N)]:
                    if r0 >= r1:
                        continue
                    cd, cm = ops.pairwise_distance(xg, mg, row_begin=r0, row_end=r1, compact=True)
                    assert torch.equal(cd.view(torch.int32), ref_d[:, r0:r1].contiguous().view(torch.int32))
                    assert torch.equal(cm, ref_m[:, r0:r1])
                    fd = torch.full_like(ref_d, SENT)
                    fm = torch.full(ref_m.shape, 7, dtype=torch.uint8, device="cuda")
                    ops.pairwise_distance(xg, mg, row_begin=r0, row_end=r1, out_dist=fd, out_mask=fm.view(torch.bool))
                    assert torch.equal(fd[:, r0:r1].contiguous().view(torch.int32),
                                       ref_d[:, r0:r1].contiguous().view(torch.int32)), (B, N, r0, r1)
                    assert torch.equal(fm[:, r0:r1].view(torch.bool), ref_m[:, r0:r1])
                    assert (fd[:, :r0] == SENT).all() and (fd[:, r1:] == SENT).all()
                    assert (fm[:, :r0] == 7).all() and (fm[:, r1:] == 7).all()
    finally:
        for k, v in saved.items():
            _lib.set_tuning(k, v)


def _same_floats(a, b):
    """Bit-identical where finite/inf, NaN in the same places (NaN payloads may differ between sqrt routines)."""
    return torch.equal(a.isnan(), b.isnan()) and torch.equal(a.nan_to_num(7.0).view(torch.int32),
                                                             b.nan_to_num(7.0).view(torch.int32))


@pytest.mark.parametrize("exact", [0, 1])
def test_k1_any_atom_count_kernel_matches_element_kernel(SB, exact):
    """The fixed-A flat pattern kernels (k1_flat=4), the row-tile kernels of A = 4 / 8 and the row-phase kernel of every
    other atom count up to 64 (k1_flat=1, the default dispatch; k1_rowphase=1 where another kernel is the default)
    against the element-per-lane kernel (k1_flat=0) for atom counts other than 15 -- and against the pattern kernels
    at A = 15 -- over full, compact and in-place row ranges, with sentinel guards around every output."""
    from protstruc_amd import _lib, ops
    keys = ("k1_variant", "k1_flat", "k1_flat_cpw", "k1_store_nt", "k1_exact_sqrt", "k1_rowphase")
    saved = {k: _lib.get_tuning(k) for k in keys}
    _lib.set_tuning("k1_exact_sqrt", exact)
    SENT = 12345.0
    cases = [(2, 16, 1), (3, 17, 2), (2, 33, 3), (3, 50, 4), (2, 100, 5), (2, 37, 14), (2, 64, 16), (2, 21, 25),
             (1, 40, 37), (1, 19, 64), (2, 250, 4), (2, 37, 15), (1, 128, 15),
             # the fixed-A flat pattern kernel (atom14 / atom37, and A = 15 as the template's cross-check): shapes that
             # cross rows inside a 4-pair group, chunks that span several rows and structures, N = 16 (shortest)
             (2, 16, 14), (3, 300, 14), (2, 129, 14), (1, 16, 37), (2, 17, 37), (2, 130, 37), (2, 200, 15), (3, 19, 15),
             (2, 90, 16), (2, 70, 25), (1, 16, 25),
             # backbone-only layouts at short and long lengths
             (3, 16, 3), (2, 100, 3), (4, 16, 4), (2, 17, 4), (3, 19, 5), (2, 300, 5), (2, 23, 8), (2, 200, 8),
             # row-tile kernel (A = 4, 8; the default dispatch): partial tiles, more rows than one workgroup takes
             (2, 129, 4), (1, 300, 4), (2, 33, 8), (1, 70, 8),
             # A = 3, 5 with N % 16 == 0 (one alignment phase): one tile, partial last tile, several tiles
             (2, 32, 5), (1, 160, 5), (2, 240, 5), (2, 16, 3), (1, 224, 3), (1, 448, 3),
             # N % 4 == 0 but not % 16
             (2, 20, 5), (3, 100, 5), (2, 500, 5), (2, 36, 3), (1, 228, 3), (3, 44, 3),
             # any other N: all four alignment phases (odd N), two (N % 4 == 2), tiny N
             (2, 17, 5), (3, 101, 5), (2, 499, 5), (2, 30, 5), (2, 19, 3), (1, 229, 3), (2, 6, 3), (2, 3, 5), (4, 2, 5),
             # row-phase kernel (round 3; A = 1, 2, 3, 5, 6, 7, 9..13): single atoms and pairs of atoms, every phase of
             # odd A, several tiles per row (A = 1: N > 2042; A = 5: N > 81; A = 13: N > 12), more rows than one
             # workgroup takes (N > 32), tiny N
             (2, 500, 1), (2, 501, 1), (2, 502, 1), (2, 503, 1), (1, 2100, 1), (3, 5, 1), (2, 1, 1), (1, 1030, 2),
             (2, 2, 2), (2, 99, 2), (2, 100, 6), (2, 37, 7), (2, 3, 7), (1, 64, 9), (2, 50, 10), (2, 33, 11), (1, 40, 12),
             (2, 29, 13), (1, 90, 5), (1, 91, 5), (1, 93, 3), (1, 94, 3),
             # CA traces of 8 .. 255 residues (round 4: the flat A = 1 kernel): many structures per workgroup, a workgroup
             # boundary inside a structure, a last slot that is not full (B * N * N % 4 != 0), both ends of the range
             (3, 8, 1), (3, 9, 1), (700, 11, 1), (5, 16, 1), (37, 37, 1), (3, 64, 1), (2, 100, 1), (1, 129, 1), (2, 255, 1),
             (1, 256, 1),
             # the fixed-A flat pattern kernels added in round 3
             (2, 40, 24), (1, 33, 27), (1, 30, 32),
             # atom counts served by the run-time instantiations of the row-phase kernel (even / odd), short and long rows
             (2, 40, 20), (1, 37, 33), (1, 9, 64), (2, 10, 21), (1, 70, 18), (1, 45, 40), (1, 31, 63), (2, 5, 17)]

    def paths(A):
        """(k1_flat, k1_rowphase) settings that reach a fast kernel for this atom count.  k1_flat 1: default dispatch
        (row-tile / row-phase kernels, fixed-A flat kernels for 14, 16, 24, 32); 4: fixed-A flat pattern kernel.
        k1_rowphase 1: the row-phase kernel also where a fixed-A flat kernel or the A = 15 kernels are the default."""
        if A in (4, 8):
            return [(1, 0)]
        if A in (14, 15, 16, 24, 32):
            return [(4, 0), (1, 1)]
        return [(1, 0)]

    try:
        for (B, N, A), (flat, rowphase) in [(c, f) for c in cases for f in paths(c[2])]:
            _lib.set_tuning("k1_rowphase", rowphase)
            xyz, mask = synth(300 + N + A, B, N, A=A)
            xyz[0, N // 3] = float("nan")
            mask[0, N // 3] = False
            xg, mg = xyz.cuda(), mask.cuda()
            _lib.set_tuning("k1_flat", 0 if A != 15 else 1)       # reference path: element kernel / pattern kernels
            ref_d, ref_m = ops.pairwise_distance(xg, mg)
            ref_d0, _ = ops.pairwise_distance(xg, None)
            rd, rm = O.pairwise_distance_matrix(xyz, mask)
            assert_close(ref_d, rd)
            _lib.set_tuning("k1_flat", flat)
            numel = ref_d.numel()
            for cpw, nt in [(1, 0), (3, 1)]:
                _lib.set_tuning("k1_flat_cpw", cpw)
                _lib.set_tuning("k1_store_nt", nt)
                pad = 64
                bd = torch.full((numel + 2 * pad,), SENT, device="cuda")
                bm = torch.full((numel + 2 * pad,), 7, dtype=torch.uint8, device="cuda")
                d = bd[pad:pad + numel].view(ref_d.shape)
                m = bm[pad:pad + numel].view(torch.bool).view(ref_m.shape)
                ops.pairwise_distance(xg, mg, out_dist=d, out_mask=m)
                assert _same_floats(d, ref_d), (B, N, A, cpw, flat, rowphase)
                assert torch.equal(m, ref_m) and torch.equal(m.cpu(), rm), (B, N, A, cpw, flat, rowphase)
                assert (bd[:pad] == SENT).all() and (bd[pad + numel:] == SENT).all()
                assert (bm[:pad] == 7).all() and (bm[pad + numel:] == 7).all()
                d0, _ = ops.pairwise_distance(xg, None, want_mask=False)
                assert _same_floats(d0, ref_d0)
                _, m1 = ops.pairwise_distance(xg, mg, want_dist=False)
                assert torch.equal(m1, ref_m)
                for r0, r1 in [(0, 1), (1, N - 1), (N // 2, N)]:
                    if r0 >= r1:
                        continue
                    cd, cm = ops.pairwise_distance(xg, mg, row_begin=r0, row_end=r1, compact=True)
                    assert _same_floats(cd, ref_d[:, r0:r1].contiguous()) and torch.equal(cm, ref_m[:, r0:r1])
                    fd = torch.full_like(ref_d, SENT)
                    fm = torch.full(ref_m.shape, 7, dtype=torch.uint8, device="cuda")
                    ops.pairwise_distance(xg, mg, row_begin=r0, row_end=r1, out_dist=fd, out_mask=fm.view(torch.bool))
                    assert _same_floats(fd[:, r0:r1].contiguous(), ref_d[:, r0:r1].contiguous()), (B, N, A, r0, r1)
                    assert torch.equal(fm[:, r0:r1].view(torch.bool), ref_m[:, r0:r1])
                    assert (fd[:, :r0] == SENT).all() and (fd[:, r1:] == SENT).all()
                    assert (fm[:, :r0] == 7).all() and (fm[:, r1:] == 7).all()
    finally:
        for k, v in saved.items():
            _lib.set_tuning(k, v)


def test_k1_square_root_modes(SB):
    """K1's two arithmetic modes.  Exact mode is the correctly rounded sqrt of the fp32 sum ((dx^2 + dy^2) + dz^2):
    bit-identical to numpy evaluating that formula in float32.  The default mode uses the hardware square root: never
    more than 1 ulp away from exact mode, identical on most entries.  Every kernel behind the entry point is covered
    (pattern, flat pattern, slot-decode, row-tile, row-phase incl. its run-time atom counts, fixed-A flat, element-per-lane)."""
    from protstruc_amd import _lib, ops
    keys = ("k1_variant", "k1_flat", "k1_exact_sqrt")
    saved = {k: _lib.get_tuning(k) for k in keys}

    def numpy_formula(xyz):
        x = xyz.numpy().astype(np.float32)
        d = x[:, :, None, :, None, :] - x[:, None, :, None, :, :]
        sq = d * d
        return torch.from_numpy(np.sqrt((sq[..., 0] + sq[..., 1]) + sq[..., 2]))

    try:
        for (B, N, A, variant, flat) in [(2, 64, 15, 0, 1), (2, 37, 15, 0, 1), (2, 37, 15, 1, 1), (2, 40, 5, 0, 1),
                                         (2, 40, 5, 0, 0), (1, 24, 37, 0, 1), (2, 33, 4, 0, 1), (2, 30, 14, 0, 1),
                                         (1, 20, 40, 0, 1)]:
            xyz, mask = synth(900 + N + A, B, N, A=A, scale=float(N % 7 + 1))
            xg, mg = xyz.cuda(), mask.cuda()
            _lib.set_tuning("k1_variant", variant)
            _lib.set_tuning("k1_flat", flat)
            want = numpy_formula(xyz)
            _lib.set_tuning("k1_exact_sqrt", 1)
            de, _ = ops.pairwise_distance(xg, mg)
            assert torch.equal(de.cpu().view(torch.int32), want.view(torch.int32)), (B, N, A, variant, flat)
            _lib.set_tuning("k1_exact_sqrt", 0)
            df, _ = ops.pairwise_distance(xg, mg)
            ulps = (df.cpu().view(torch.int32) - want.view(torch.int32)).abs()
            assert int(ulps.max()) <= 1, (B, N, A, variant, flat)
            assert float((ulps == 0).float().mean()) > 0.75
    finally:
        for k, v in saved.items():
            _lib.set_tuning(k, v)


def test_k1_misaligned_output_buffers(SB):
    """Outputs that are only 4-byte (dist) / 1-byte (mask) aligned -- contiguous views into larger buffers -- must
    still be written correctly and completely (they take the slot-decode / element kernels)."""
    from protstruc_amd import ops
    for (B, N, A) in [(2, 37, 15), (2, 32, 15), (1, 40, 5), (2, 7, 15)]:
        xyz, mask = synth(500 + N + A, B, N, A=A)
        xg, mg = xyz.cuda(), mask.cuda()
        ref_d, ref_m = ops.pairwise_distance(xg, mg)
        numel = ref_d.numel()
        for off in (1, 2, 3, 5):
            bd = torch.full((numel + 16,), 777.0, device="cuda")
            bm = torch.full((numel + 16,), 7, dtype=torch.uint8, device="cuda")
            d = bd[off:off + numel].view(ref_d.shape)
            m = bm[off:off + numel].view(torch.bool).view(ref_m.shape)
            assert d.data_ptr() % 16 != 0 and m.data_ptr() % 16 != 0
            ops.pairwise_distance(xg, mg, out_dist=d, out_mask=m)
            assert _same_floats(d, ref_d), (B, N, A, off)
            assert torch.equal(m, ref_m), (B, N, A, off)
            assert (bd[:off] == 777.0).all() and (bd[off + numel:] == 777.0).all()
            assert (bm[:off] == 7).all() and (bm[off + numel:] == 7).all()


def test_k1_capi_argument_validation_streams_and_capture(SB):
    """The C entry point itself: invalid arguments are rejected before anything is launched (hipErrorInvalidValue = 1),
    launches on different streams are independent, and the flat / row-phase kernels are hipGraph-capturable."""
    import ctypes
    from protstruc_amd import _lib, ops
    lib = _lib.load()
    xyz, mask = synth(31, 2, 20)
    xg, mg = xyz.cuda(), mask.cuda().view(torch.uint8)
    d = torch.full((2, 20, 20, 15, 15), 5.0, device="cuda")
    m = torch.zeros(2, 20, 20, 15, 15, dtype=torch.uint8, device="cuda")
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    call = lambda *a: lib.ps_pairwise_distance_f32(*a, st)
    ok = (P(xg), P(mg), P(d), P(m), 2, 20, 15, 0, 20, 20, 0)
    bad = [
        (None,) + ok[1:],                                   # no coordinates
        ok[:2] + (None, None) + ok[4:],                     # neither output plane requested
        ok[:4] + (-1,) + ok[5:],                            # negative batch
        ok[:6] + (0,) + ok[7:],                             # A = 0
        ok[:7] + (-1, 20, 20, 0),                           # row_begin < 0
        ok[:7] + (5, 21, 20, 0),                            # row_end > N
        ok[:7] + (7, 5, 20, 0),                             # row_begin > row_end
        ok[:7] + (5, 10, 4, 5),                             # compact buffer too small for the rows
        ok[:7] + (5, 10, 20, 6),                            # origin after the first computed row
    ]
    for args in bad:
        assert call(*args) == 1, args
    torch.cuda.synchronize()
    assert (d == 5.0).all() and (m == 0).all()             # nothing was written by the rejected calls
    assert call(*ok) == 0
    assert call(*(ok[:4] + (0,) + ok[5:])) == 0             # B = 0 is a no-op, not an error
    ref_d, ref_m = d.clone(), m.clone()

    # two streams, two shapes (flat pattern kernel and row-phase kernel), interleaved launches
    xyz5, mask5 = synth(32, 3, 33, A=5)
    x5, m5 = xyz5.cuda(), mask5.cuda()
    want5 = ops.pairwise_distance(x5, m5)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    outs = []
    for _ in range(4):
        with torch.cuda.stream(s1):
            outs.append(("a", ops.pairwise_distance(xg, mask.cuda())))
        with torch.cuda.stream(s2):
            outs.append(("b", ops.pairwise_distance(x5, m5)))
    torch.cuda.synchronize()
    for tag, (od, om) in outs:
        if tag == "a":
            assert torch.equal(od, ref_d) and torch.equal(om.view(torch.uint8), ref_m)
        else:
            assert torch.equal(od, want5[0]) and torch.equal(om, want5[1])

    # graph capture of both kernels; replay after the inputs changed in place
    g = torch.cuda.CUDAGraph()
    xa, xb, mgb = xg.clone(), x5.clone(), mask.cuda()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        ga = ops.pairwise_distance(xa, mgb)
        gb = ops.pairwise_distance(xb, m5)
    xa.mul_(2.0); xb.add_(1.0)
    g.replay(); torch.cuda.synchronize()
    ea = ops.pairwise_distance(xa, mgb); eb = ops.pairwise_distance(xb, m5)
    assert torch.equal(ga[0], ea[0]) and torch.equal(ga[1], ea[1])
    assert torch.equal(gb[0], eb[0]) and torch.equal(gb[1], eb[1])


def test_k1_caller_buffers_validated_before_launch(SB):
    """Caller-supplied outputs (the path ``pairwise_distance_matrix_sharded(out_dist=, out_mask=)`` takes) and the atom
    mask are dereferenced by a kernel on xyz's GPU: a CPU tensor, a wrong dtype / shape or a non-contiguous view must
    raise before anything is launched, never become a wild device write."""
    from protstruc_amd import ops
    xyz, mask = synth(33, 2, 20)
    xg, mg = xyz.cuda(), mask.cuda()
    shape = (2, 20, 20, 15, 15)
    good_d = torch.full(shape, 5.0, device="cuda")
    good_m = torch.zeros(shape, dtype=torch.bool, device="cuda")
    bad_outs = [
        dict(out_dist=torch.empty(shape)),                                          # CPU distance buffer
        dict(out_mask=torch.empty(shape, dtype=torch.bool)),                        # CPU mask buffer
        dict(out_dist=torch.empty(shape, dtype=torch.float64, device="cuda")),      # wrong dtype
        dict(out_mask=torch.empty(shape, dtype=torch.uint8, device="cuda")),
        dict(out_dist=torch.empty((2, 20, 20, 15, 14), device="cuda")),             # wrong shape
        dict(out_dist=torch.empty((2, 20, 20, 15, 30), device="cuda")[..., ::2]),   # right shape, strided
    ]
    for kw in bad_outs:
        with pytest.raises(ValueError, match="must be a contiguous"):
            ops.pairwise_distance(xg, mg, **{"out_dist": good_d, "out_mask": good_m, **kw})
    with pytest.raises(RuntimeError, match="HIP-only"):
        ops.pairwise_distance(xg, mask)                                             # CPU atom mask
    torch.cuda.synchronize()
    assert (good_d == 5.0).all() and not good_m.any()                               # nothing was launched
    d, m = ops.pairwise_distance(xg, mg, out_dist=good_d, out_mask=good_m)
    assert d is good_d and m is good_m and not torch.isnan(d).any()


def test_k1_knobs_flipped_on_another_thread(SB):
    """The library holds no tuning state: every launch carries a snapshot of its device's host-side table.  One
    thread rewrites that table as fast as it can while two others launch K1 (pattern and flat-pattern shapes) on
    their own streams; every result must be the reference bits (all settings compute the same values)."""
    import threading
    from protstruc_amd import _lib, ops
    xa, ma = synth(61, 3, 64)            # N % 16 == 0: pattern kernel
    xb, mb = synth(62, 2, 50)            # flat pattern kernel
    xa, ma, xb, mb = xa.cuda(), ma.cuda(), xb.cuda(), mb.cuda()
    want_a = ops.pairwise_distance(xa, ma)
    want_b = ops.pairwise_distance(xb, mb)
    torch.cuda.synchronize()
    keys = ("k1_rows_per_block", "k1_jt", "k1_flat_cpw", "k1_xcd_remap", "k1_store_nt", "k1_lds_pad_kb", "k1_flat")
    saved = {k: _lib.get_tuning(k) for k in keys}
    stop = threading.Event()
    errors = []

    def flipper():
        rng = np.random.default_rng(5)
        while not stop.is_set():
            _lib.set_tuning("k1_rows_per_block", int(rng.choice([1, 2, 3, 4, 8])))
            _lib.set_tuning("k1_jt", int(rng.choice([0, 16, 32, 64, 128])))
            _lib.set_tuning("k1_flat_cpw", int(rng.choice([1, 2, 5])))
            _lib.set_tuning("k1_xcd_remap", int(rng.integers(0, 2)))
            _lib.set_tuning("k1_store_nt", int(rng.integers(0, 2)))
            _lib.set_tuning("k1_lds_pad_kb", int(rng.choice([0, 8])))
            _lib.set_tuning("k1_flat", int(rng.choice([1, 2])))

    def launcher(x, m, want):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for _ in range(150):
                    d, k = ops.pairwise_distance(x, m)
                    st.synchronize()
                    if not (torch.equal(d, want[0]) and torch.equal(k, want[1])):
                        errors.append("bits differ")
                        return
        except Exception as exc:  # noqa: BLE001
            errors.append(repr(exc))

    threads = [threading.Thread(target=flipper), threading.Thread(target=launcher, args=(xa, ma, want_a)),
               threading.Thread(target=launcher, args=(xb, mb, want_b))]
    try:
        for t in threads:
            t.start()
        for t in threads[1:]:
            t.join()
    finally:
        stop.set()
        threads[0].join()
        for k, v in saved.items():
            _lib.set_tuning(k, v)
    assert not errors, errors
    # another device's entry was never touched by any of this
    assert _lib.get_tuning("k1_rows_per_block", device=7) == 1


def test_k1_explicit_config_through_the_c_abi(SB):
    """ps_pairwise_distance_cfg_f32 called directly with caller-built structs: every valid configuration gives the
    bits of the default call, malformed ones return hipErrorInvalidValue and write nothing."""
    import ctypes
    from protstruc_amd import _lib
    lib = _lib.load()
    xyz, mask = synth(63, 2, 48)
    xg, mg = xyz.cuda(), mask.cuda().view(torch.uint8)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run(cfg):
        d = torch.full((2, 48, 48, 15, 15), -3.0, device="cuda")
        m = torch.full((2, 48, 48, 15, 15), 9, dtype=torch.uint8, device="cuda")
        rc = lib.ps_pairwise_distance_cfg_f32(P(xg), P(mg), P(d), P(m), 2, 48, 15, 0, 48, 48, 0,
                                              None if cfg is None else ctypes.byref(cfg), st)
        torch.cuda.synchronize()
        return rc, d, m

    rc, ref_d, ref_m = run(None)
    assert rc == 0 and (ref_m <= 1).all() and not (ref_d == -3.0).any()
    default = _lib.K1Config()
    lib.ps_k1_config_default(ctypes.byref(default))
    for over in [{}, {"rows_per_block": 4}, {"jt": 64, "rows_per_block": 3}, {"variant": 1}, {"flat": 2, "flat_cpw": 3},
                 {"flat": 4}, {"flat": 0}, {"rowphase": 1}, {"rowphase": 2}, {"flat": 2, "flat_fl_log2": 5},
                 {"xcd_remap": 0, "store_nt": 1}, {"lds_pad_kb": 8}]:
        cfg = _lib.K1Config(**{**{f: getattr(default, f) for f, _ in _lib.K1Config._fields_}, **over})
        rc, d, m = run(cfg)
        assert rc == 0 and torch.equal(d, ref_d) and torch.equal(m, ref_m), over
    for over in [{"struct_size": 4}, {"experiment": 2}, {"rows_per_block": 0}, {"flat": 9}, {"flat": 3}, {"rowphase": 3},
                 {"flat_fl_log2": 2}]:
        cfg = _lib.K1Config(**{**{f: getattr(default, f) for f, _ in _lib.K1Config._fields_}, **over})
        rc, d, m = run(cfg)
        assert rc == 1 and (d == -3.0).all() and (m == 9).all(), over


def test_terminal_masks_request_only_their_output(SB):
    from protstruc_amd import ops
    g = load_golden("g2_bbdih_chains")
    sb = SB.from_xyz(g["xyz"], g["atom_mask"], g["chain_idx"], chain_ids=[["A", "B", "C"]] * g["xyz"].shape[0])
    assert torch.equal(sb.get_n_terminal_mask().cpu(), g["nterm"]) and torch.equal(sb.get_c_terminal_mask().cpu(), g["cterm"])
    only = ops.backbone_dihedrals(sb.xyz, sb.chain_idx, sb.residue_mask, want_dihedrals=False, want_mask=False,
                                  want_nterm=False)
    assert only[0] is None and only[1] is None and only[2] is None and torch.equal(only[3].cpu(), g["cterm"])
    with pytest.raises(ValueError):
        ops.backbone_dihedrals(sb.xyz, sb.chain_idx, sb.residue_mask, want_dihedrals=False, want_mask=False,
                               want_nterm=False, want_cterm=False)


def test_k1_allocate_fast_outputs(SB):
    """The allocation-shopping helper returns usable buffers and a report; results do not depend on the choice."""
    from protstruc_amd import ops
    xyz, mask = synth(51, 4, 128)
    xg, mg = xyz.cuda(), mask.cuda()
    d, m, rep = ops.allocate_fast_outputs(xg, mg, candidates=3)
    assert d.shape == (4, 128, 128, 15, 15) and m.shape == d.shape and m.dtype == torch.bool
    assert len(rep["ms_per_candidate"]) == 3 and 0 <= rep["chosen"] < 3
    assert rep["ms_per_candidate"][rep["chosen"]] == min(rep["ms_per_candidate"])
    ops.pairwise_distance(xg, mg, out_dist=d, out_mask=m)
    rd, rm = ops.pairwise_distance(xg, mg)
    assert torch.equal(d, rd) and torch.equal(m, rm)


def test_k1_autotune_is_explicit_and_transparent(SB):
    """Nothing is timed behind the caller's back: a large call leaves the device's configuration alone.  The explicit
    tuner (ops.autotune_pairwise_distance) and the PROTSTRUC_AMD_AUTOTUNE=1 / set_implicit_autotune opt-in change speed only -- results are
    bit-identical before and after -- and never run during stream capture."""
    import os
    from protstruc_amd import _lib, ops
    xyz, mask = synth(12, 16, 512)   # 4.2 M pairs: a shape the tuner accepts
    xg, mg = xyz.cuda(), mask.cuda()
    saved_tuned = ops._K1_TUNED.pop(xg.device, None)
    rows0, pad0 = _lib.get_tuning("k1_rows_per_block"), _lib.get_tuning("k1_lds_pad_kb")
    try:
        assert (rows0, pad0) == (1, -1) or saved_tuned is not None     # the measured-best default (idle LDS by chain length)
        d0, m0 = ops.pairwise_distance(xg, mg)
        assert ops.k1_autotune_result(xg.device) is None, "an ordinary call must not tune"
        out_d, out_m = torch.empty_like(d0), torch.empty_like(m0)
        res = ops.autotune_pairwise_distance(xg, mg, out_d, out_m)
        assert res is not None and res["rows_per_block"] in (1, 2)
        # every candidate but the one that merely NAMES the default launch at this length (36 KB of idle LDS + 32-residue tiles from
        # N = 256: timing it against candidate 0 was noise, and a "pick" of it made bench.py re-time the default for nothing)
        alias = {"rows_per_block": 1, "lds_pad_kb": 36, "jt": 32}
        assert set(res["ms"]) == {ops._cand_label(c) for c in ops._K1_CANDIDATE_PATTERN if c != alias} and len(res["ms"]) == 7
        assert _lib.get_tuning("k1_rows_per_block") == res["rows_per_block"]
        assert _lib.get_tuning("k1_lds_pad_kb") == res["lds_pad_kb"] and _lib.get_tuning("k1_jt") == res["jt"]
        assert torch.equal(out_d, d0) and torch.equal(out_m, m0)
        d1, m1 = ops.pairwise_distance(xg, mg)
        assert torch.equal(d0, d1) and torch.equal(m0, m1)
        # opt-in through the environment; a captured call never tunes and still works
        ops._K1_TUNED.pop(xg.device, None)
        ops.set_implicit_autotune(True)     # what PROTSTRUC_AMD_AUTOTUNE=1 at import time selects
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            d2, m2 = ops.pairwise_distance(xg, mg)
        g.replay(); torch.cuda.synchronize()
        assert ops.k1_autotune_result(xg.device) is None and torch.equal(d2, d0)
        ops.pairwise_distance(xg, mg)
        assert ops.k1_autotune_result(xg.device) is not None
        # the flat kernel's chunks per workgroup (lengths that are not a multiple of 16) tune the same way
        xyz2, mask2 = synth(13, 23, 437)   # 4.39 M pairs
        x2, m2g = xyz2.cuda(), mask2.cuda()
        ops.set_implicit_autotune(False)
        e0, f0 = ops.pairwise_distance(x2, m2g)
        assert "flat_cpw" not in ops.k1_autotune_result(xg.device)
        e1, f1 = torch.empty_like(e0), torch.empty_like(f0)
        res2 = ops.autotune_pairwise_distance(x2, m2g, e1, f1)
        assert res2["flat_cpw"] in (1, 2, 4) and _lib.get_tuning("k1_flat_cpw") == res2["flat_cpw"]
        assert _lib.get_tuning("k1_flat_lds_pad_kb") == res2["flat_lds_pad_kb"]
        assert res2["flat_fl_log2"] in (0, 5, 7) and _lib.get_tuning("k1_flat_fl_log2") == res2["flat_fl_log2"]
        assert len(res2["flat_ms"]) == len(ops._K1_CANDIDATE_FLAT)
        assert torch.equal(e0, e1) and torch.equal(f0, f1)
    finally:
        ops.set_implicit_autotune(False)
        for k, v in (("k1_rows_per_block", rows0), ("k1_lds_pad_kb", pad0), ("k1_jt", 0), ("k1_flat_cpw", 1),
                     ("k1_flat_lds_pad_kb", 0), ("k1_flat_fl_log2", 0)):
            _lib.set_tuning(k, v)
        ops._K1_TUNED.pop(xg.device, None)
        if saved_tuned is not None:
            ops._K1_TUNED[xg.device] = saved_tuned


def test_k1_headline_shape_properties(SB):
    """BASELINE headline shape B=64, N=512: too big for the CPU oracle, so check
    sampled blocks against it plus whole-tensor properties."""
    B, N = 64, 512
    xyz, mask = synth(0, B, N)
    sb = SB.from_xyz(xyz, mask)
    d, m = sb.pairwise_distance_matrix()
    assert d.shape == (B, N, N, 15, 15) and m.shape == d.shape
    g = torch.Generator().manual_seed(1)
    bs = torch.randint(0, B, (64,), generator=g)
    is_ = torch.randint(0, N, (64,), generator=g)
    js = torch.randint(0, N, (64,), generator=g)
    got = d[bs.cuda(), is_.cuda(), js.cuda()].cpu()
    want = torch.norm(xyz[bs, is_][:, :, None, :] - xyz[bs, js][:, None, :, :], dim=-1)
    assert_close(got, want)
    gm = m[bs.cuda(), is_.cuda(), js.cuda()].cpu()
    assert torch.equal(gm, mask[bs, is_][:, :, None] & mask[bs, js][:, None, :])
    # checksum of the mask plane: sum over (j,c) of mask = count_i * total, exact integer identity
    per_struct = mask.reshape(B, -1).sum(1).to(torch.int64)
    assert torch.equal(m.reshape(B, -1).sum(1, dtype=torch.int64).cpu(), per_struct * per_struct)
    # symmetry on one structure (full tensor transposes of 19 GB are avoided)
    assert torch.equal(d[3], d[3].permute(1, 0, 3, 2))
    # no element left unwritten: a NaN-prefilled buffer comes back NaN-free
    from protstruc_amd import ops
    buf = torch.full((4, N, N, 15, 15), float("nan"), device="cuda")
    ops.pairwise_distance(sb.xyz[:4], sb.atom_mask[:4], out_dist=buf, want_mask=False)
    assert not torch.isnan(buf).any()


def test_k1_large_n_2048(SB):
    """BASELINE config 4's residue count (N=2048): 64-bit addressing and the row-range path at scale.
    One structure (4.7 GB); sampled blocks vs the oracle's formula, mask checksum, one row shard."""
    from protstruc_amd import ops
    N = 2048
    xyz, mask = synth(4, 1, N)
    xg, mg = xyz.cuda(), mask.cuda()
    d, m = ops.pairwise_distance(xg, mg)
    g = torch.Generator().manual_seed(2)
    is_ = torch.cat([torch.randint(0, N, (200,), generator=g), torch.tensor([0, N - 1, N - 1, 0])])
    js = torch.cat([torch.randint(0, N, (200,), generator=g), torch.tensor([0, N - 1, 0, N - 1])])
    want = torch.norm(xyz[0, is_][:, :, None, :] - xyz[0, js][:, None, :, :], dim=-1)
    assert_close(d[0, is_.cuda(), js.cuda()], want)
    assert torch.equal(m[0, is_.cuda(), js.cuda()].cpu(), mask[0, is_][:, :, None] & mask[0, js][:, None, :])
    cnt = int(mask.sum())
    assert int(m.sum(dtype=torch.int64)) == cnt * cnt
    # rank 5 of 8 (rows 1280..1536) written into a NaN-prefilled compact buffer equals the full result
    cd, cm = ops.pairwise_distance(xg, mg, row_begin=1280, row_end=1536, compact=True)
    assert torch.equal(cd, d[:, 1280:1536]) and torch.equal(cm, m[:, 1280:1536])


@pytest.mark.parametrize("B,N,A", [(2, 2047, 15), (3, 1001, 15), (4, 511, 37), (8, 1023, 4)])
def test_k1_large_ragged_shapes_properties(SB, B, N, A):
    """Multi-GB launches of the flat kernels at lengths that are not multiples of anything: sampled blocks against
    the formula, exact mask checksum per structure, symmetry of a block, no element left unwritten."""
    from protstruc_amd import ops
    xyz, mask = synth(40 + N, B, N, A=A)
    xg, mg = xyz.cuda(), mask.cuda()
    d = torch.full((B, N, N, A, A), float("nan"), device="cuda")
    m = torch.zeros(B, N, N, A, A, dtype=torch.bool, device="cuda")
    ops.pairwise_distance(xg, mg, out_dist=d, out_mask=m)
    g = torch.Generator().manual_seed(3)
    bs = torch.randint(0, B, (96,), generator=g)
    is_ = torch.cat([torch.randint(0, N, (92,), generator=g), torch.tensor([0, N - 1, N - 1, 0])])
    js = torch.cat([torch.randint(0, N, (92,), generator=g), torch.tensor([0, N - 1, 0, N - 1])])
    want = torch.norm(xyz[bs, is_][:, :, None, :] - xyz[bs, js][:, None, :, :], dim=-1)
    assert_close(d[bs.cuda(), is_.cuda(), js.cuda()], want)
    assert torch.equal(m[bs.cuda(), is_.cuda(), js.cuda()].cpu(), mask[bs, is_][:, :, None] & mask[bs, js][:, None, :])
    per_struct = mask.reshape(B, -1).sum(1).to(torch.int64)
    assert torch.equal(torch.stack([torch.count_nonzero(m[b]) for b in range(B)]).cpu(), per_struct * per_struct)
    assert not torch.isnan(d[B - 1]).any() and not torch.isnan(d[0, N - 1]).any()
    k = min(N, 200)
    assert torch.equal(d[B - 1, :k, N - k:], d[B - 1, N - k:, :k].permute(1, 0, 3, 2))
    del d, m
    torch.cuda.empty_cache()


def test_k1_config4_full_size_properties(SB):
    """BASELINE config 4 at full size on one GPU (B=32, N=2048: 134 M pairs, 151 GB of output -- the largest shape
    BASELINE names): sampled blocks against the oracle's formula, exact mask checksum per structure, symmetry."""
    from protstruc_amd import ops
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    if free < 170 * 2**30:
        pytest.skip("needs 170 GB of free HBM")
    B, N = 32, 2048
    xyz, mask = synth(0, B, N)
    d, m = ops.pairwise_distance(xyz.cuda(), mask.cuda())
    g = torch.Generator().manual_seed(1)
    bs = torch.randint(0, B, (128,), generator=g)
    is_ = torch.cat([torch.randint(0, N, (124,), generator=g), torch.tensor([0, N - 1, N - 1, 0])])
    js = torch.cat([torch.randint(0, N, (124,), generator=g), torch.tensor([0, N - 1, 0, N - 1])])
    want = torch.norm(xyz[bs, is_][:, :, None, :] - xyz[bs, js][:, None, :, :], dim=-1)
    assert_close(d[bs.cuda(), is_.cuda(), js.cuda()], want)
    assert torch.equal(m[bs.cuda(), is_.cuda(), js.cuda()].cpu(), mask[bs, is_][:, :, None] & mask[bs, js][:, None, :])
    per_struct = mask.reshape(B, -1).sum(1).to(torch.int64)
    got = torch.stack([torch.count_nonzero(m[b]) for b in range(B)]).cpu()
    assert torch.equal(got, per_struct * per_struct)
    assert torch.equal(d[B - 1, :256, 1024:1280], d[B - 1, 1024:1280, :256].permute(1, 0, 3, 2))
    del d, m
    torch.cuda.empty_cache()


# ----------------------------------------------------------------------------- K2
@pytest.mark.parametrize("name", ["g2_bbdih_chains", "g2_bbdih_padded_nan", "g2_bbdih_default_a25"])
def test_k2_golden(SB, name):
    g = load_golden(name)
    if "atom_mask" in g:
        sb = SB.from_xyz(g["xyz"], g["atom_mask"], chain_idx=g["chain_idx"], chain_ids=[["A", "B", "C"]] * len(g["xyz"]))
    else:
        sb = SB.from_xyz(g["xyz"])
    dih, dmask = sb.backbone_dihedrals()
    assert_close(dih, g["dihedrals"])
    assert dmask.dtype == torch.bool and torch.equal(dmask.cpu(), g["dihedral_mask"])
    assert torch.equal(sb.get_n_terminal_mask().cpu(), g["nterm"])
    assert torch.equal(sb.get_c_terminal_mask().cpu(), g["cterm"])


@pytest.mark.parametrize("B,N", [(1, 1), (1, 2), (2, 63), (2, 64), (2, 65), (4, 300), (64, 256)])
def test_k2_vs_oracle(SB, B, N):
    xyz, mask = synth(200 + N, B, N)
    chain_idx = torch.zeros(B, N)
    if N >= 2:
        chain_idx[:, N // 2:] = 1.0
    sb = SB.from_xyz(xyz, mask, chain_idx=chain_idx, chain_ids=[["A", "B"]] * B)
    dih, dmask = sb.backbone_dihedrals()
    rdih, rmask = O.backbone_dihedrals(xyz, chain_idx, mask.any(-1))
    assert_close(dih, rdih)
    assert torch.equal(dmask.cpu(), rmask)
    # properties asserted by the reference's own test (tests/test_StructureBatch.py:68-96)
    assert ((dih >= -np.pi) & (dih <= np.pi)).all()
    nterm, cterm = sb.get_n_terminal_mask(), sb.get_c_terminal_mask()
    assert (dih[:, :, 0][nterm] == 0).all() and (dih[:, :, 1][cterm] == 0).all() and (dih[:, :, 2][cterm] == 0).all()


# ----------------------------------------------------------------------------- K3
def _f64_angles(xyz, si, sj, n_points):
    x = xyz.double()
    p = O.pairwise_points(x, si, sj)
    n = xyz.shape[1]
    if n_points == 4:
        return O.dihedral(p[:, :, 0], p[:, :, 1], p[:, :, 2], p[:, :, 3]).reshape(-1, n, n)
    return O.angle(p[:, :, 0], p[:, :, 1], p[:, :, 2]).reshape(-1, n, n)


def test_k3_golden(SB):
    g = load_golden("g3_pairwise_angles")
    sb = SB.from_xyz(g["xyz"], g["atom_mask"])
    checked = 0
    for key, want in g.items():
        if key[:4] not in ("dih_", "ang_"):
            continue
        left, right = key[4:].split("__")
        ai = [t for t in left.split("_") if t]
        aj = [t for t in right.split("_") if t]
        got = (sb.pairwise_dihedrals if key.startswith("dih_") else sb.pairwise_planar_angles)(ai, aj)
        # 2 * 33 * 33 entries: allow at most one ill-conditioned entry beyond 1e-5
        assert_close(got, want, bad_frac=1.0 / want.numel())
        checked += 1
    assert checked == 8
    omega = sb.pairwise_dihedrals(["CA", "CB"], ["CA", "CB"])
    diag = torch.diagonal(omega, dim1=1, dim2=2)
    assert (diag == 0).all() and not torch.signbit(diag).any(), "diagonal must be exactly +0.0 (no FMA contraction)"
    assert torch.diagonal(sb.pairwise_planar_angles(["CA", "CB"], ["CB"]), dim1=1, dim2=2).isnan().all()


@pytest.mark.parametrize("ai,aj,npts", [
    (["CA", "CB"], ["CA", "CB"], 4), (["N", "CA", "CB"], ["CB"], 4), (["C"], ["N", "CA", "C"], 4),
    (["N", "CA", "C", "O"], [], 4), ([], ["N", "CA", "C", "O"], 4),
    (["CA", "CB"], ["CB"], 3), (["CA"], ["CA", "CB"], 3), ([], ["N", "CA", "C"], 3),
])
def test_k3_vs_oracle_conditioning_gate(SB, ai, aj, npts):
    """Gate of SURVEY hard part 3: >= (1 - 1e-4) of entries within 1e-5 of the
    oracle, and the build no worse than the oracle against an fp64 evaluation."""
    B, N = 4, 192
    xyz, mask = synth(300, B, N)
    sb = SB.from_xyz(xyz, mask)
    si, sj = [SLOT[a] for a in ai], [SLOT[a] for a in aj]
    if npts == 4:
        got, ref = sb.pairwise_dihedrals(ai, aj).cpu(), O.pairwise_dihedrals(xyz, si, sj)
    else:
        got, ref = sb.pairwise_planar_angles(ai, aj).cpu(), O.pairwise_planar_angles(xyz, si, sj)
    truth = _f64_angles(xyz, si, sj, npts)
    off = ~torch.eye(N, dtype=torch.bool).expand(B, N, N)
    if not aj or not ai:
        off = torch.ones(B, N, N, dtype=torch.bool)  # no (i == j) degeneracy when one side supplies every point
    both = ~(torch.isnan(ref) | torch.isnan(got))
    sel = off & both

    def wrap(d):  # angular distance (atan2 branch cut at +-pi)
        return torch.minimum(d.abs(), (2 * np.pi - d.abs()).abs()) if npts == 4 else d.abs()

    err_vs_ref = wrap(got - ref)[sel]
    assert (err_vs_ref > 1e-5).float().mean().item() <= 1e-4
    e_got = wrap(got.double() - truth)[sel]
    e_ref = wrap(ref.double() - truth)[sel]
    assert e_got.max().item() <= max(2 * e_ref.max().item(), 2e-5)   # (SURVEY hard part 3: "no worse than"; measured <= 1.6x, profiles/r05_k3_error_stats.log)
    assert (e_got > 1e-5).float().mean().item() <= (e_ref > 1e-5).float().mean().item() + 1e-4
    # NaNs only where the reference has them, up to |cos| rounding just past 1 (acos without clamp)
    nan_mismatch = (torch.isnan(got) != torch.isnan(ref))[off].float().mean().item()
    assert nan_mismatch <= 1e-5


def test_k3_config3_shape(SB):
    """BASELINE config 3 (B=128, N=512): full launch, four structures checked against the oracle and fp64."""
    B, N = 128, 512
    xyz, mask = synth(3, B, N)
    sb = SB.from_xyz(xyz, mask)
    pick = [0, 37, 64, 127]
    for (ai, aj, si, sj, npts) in [(["CA", "CB"], ["CA", "CB"], [1, 4], [1, 4], 4), (["N", "CA", "CB"], ["CB"], [0, 1, 4], [4], 4),
                                   (["CA", "CB"], ["CB"], [1, 4], [4], 3)]:
        got = (sb.pairwise_dihedrals if npts == 4 else sb.pairwise_planar_angles)(ai, aj)
        assert got.shape == (B, N, N)
        g = got[pick].cpu()
        ref = (O.pairwise_dihedrals if npts == 4 else O.pairwise_planar_angles)(xyz[pick], si, sj)
        truth = _f64_angles(xyz[pick], si, sj, npts)
        off = ~torch.eye(N, dtype=torch.bool).expand(len(pick), N, N)
        ok = off & ~(ref.isnan() | g.isnan())
        wrap = (lambda d: torch.minimum(d.abs(), (2 * np.pi - d.abs()).abs())) if npts == 4 else (lambda d: d.abs())
        assert (wrap(g - ref)[ok] > 1e-5).float().mean().item() <= 1e-4
        # against fp64: the MAXIMUM over 10^6 entries is one tail event of each arithmetic's own ill-conditioning pattern (the
        # fast form's worst entry is not the oracle's worst entry), so the maxima are held to 4x (measured 3.4x here, 1.6x at
        # N = 192 where test_k3_vs_oracle_conditioning_gate holds them to 2x) and the 1 - 1e-5 quantiles -- ten entries
        # from the top, a statistic rather than an event -- to 2x
        e_got, e_ref = wrap(g.double() - truth)[ok], wrap(ref.double() - truth)[ok]
        assert e_got.max().item() <= max(4 * e_ref.max().item(), 2e-5)
        kth = max(1, int(e_got.numel() * (1 - 1e-5)))
        assert e_got.kthvalue(kth).values.item() <= max(2 * e_ref.kthvalue(kth).values.item(), 1e-5)
        diag = torch.diagonal(g, dim1=1, dim2=2)
        assert ((diag == 0) & ~torch.signbit(diag)).all() if npts == 4 else diag.isnan().all()
    geo = sb.inter_residue_geometry()
    assert torch.equal(geo["omega"].isnan(), sb.pairwise_dihedrals(["CA", "CB"], ["CA", "CB"]).isnan())
    assert torch.equal(geo["omega"].nan_to_num(0), sb.pairwise_dihedrals(["CA", "CB"], ["CA", "CB"]).nan_to_num(0))


def test_k3_config3_shape_faithful_mode(SB):
    """BASELINE config 3 (B=128, N=512) in the reference's order of operations (`set_exact_angles(True)`; north_star: "fp32
    within 1e-5 abs, NaN positions identical"): full launches on the per-CU sweep kernels; on four structures NO dihedral --
    diagonal included -- is more than 1e-5 from the oracle (angular distance; max <= 2e-6), the planar angle is within 1e-5
    wherever the angle is more than 0.05 rad from 0 and pi (nearer, acos amplifies the last bits of the cosine -- a 3-ulp
    difference between this division / square root chain and numpy's is 1.2e-5 at 0.015 rad) and within its conditioning gate everywhere; NaN positions are EQUAL to the
    oracle's in all three; the featuriser's angle planes equal the K3 launches bit for bit."""
    from protstruc_amd import ops
    B, N = 128, 512
    xyz, mask = synth(3, B, N)
    sb = SB.from_xyz(xyz, mask)
    pick = [0, 37, 64, 127]
    same = lambda a, b: torch.equal(a.isnan(), b.isnan()) and torch.equal(a.nan_to_num(5.0), b.nan_to_num(5.0))
    try:
        ops.set_exact_angles(True)
        outs = {}
        for (key, ai, aj, si, sj, npts) in [("omega", ["CA", "CB"], ["CA", "CB"], [1, 4], [1, 4], 4), ("theta", ["N", "CA", "CB"], ["CB"], [0, 1, 4], [4], 4),
                                            ("phi", ["CA", "CB"], ["CB"], [1, 4], [4], 3)]:
            plan = __import__("protstruc_amd._lib", fromlist=["k3_plan"]).k3_plan(B, N, 15, si, sj, npts, exact_angles=1)
            assert plan["family"] == "sweep" and plan["faithful"] == 1, plan
            got = (sb.pairwise_dihedrals if npts == 4 else sb.pairwise_planar_angles)(ai, aj)
            outs[key] = got
            assert got.shape == (B, N, N)
            g = got[pick].cpu()
            ref = (O.pairwise_dihedrals if npts == 4 else O.pairwise_planar_angles)(xyz[pick], si, sj)
            assert torch.equal(g.isnan(), ref.isnan()), key                      # NaN positions identical
            ok = ~ref.isnan()
            if npts == 4:
                err = torch.minimum((g - ref).abs(), (2 * np.pi - (g - ref).abs()).abs())[ok]
                assert (err > 1e-5).sum().item() == 0 and err.max().item() <= 2e-6, (key, err.max().item())
                diag = torch.diagonal(g, dim1=1, dim2=2)
                assert ((diag == 0) & ~torch.signbit(diag)).all()
            else:
                err = (g - ref).abs()
                well = ok & ((ref - np.pi).abs() > 5e-2) & (ref.abs() > 5e-2)   # |cos| <= 1 - 1.25e-3: an ulp of the cosine moves the angle by <= 1.2e-6
                assert (err[well] > 1e-5).sum().item() == 0, err[well].max().item()
                assert (err[ok] > 1e-5).float().mean().item() <= 1e-4
                assert torch.diagonal(g, dim1=1, dim2=2).isnan().all()
        geo = sb.inter_residue_geometry()
        for key in ("omega", "theta", "phi"):
            assert same(geo[key], outs[key]), key
    finally:
        ops.set_exact_angles(False)


def test_config5_shape_diffusion_loop(SB):
    """BASELINE config 5 shape (B=256, N=384): standardize once, then a short loop three ways --
    step-by-step, fused steps, and the LDS-resident trajectory kernel -- must agree bit for bit."""
    B, N, T = 256, 384, 6
    xyz, mask = synth(5, B, N, scale=8.0)
    betas = torch.linspace(1e-4, 0.05, T)[:, None].expand(T, B).contiguous()
    a = SB.from_xyz(xyz.clone(), mask).manual_seed(77)
    b = SB.from_xyz(xyz.clone(), mask).manual_seed(77)
    c = SB.from_xyz(xyz.clone(), mask).manual_seed(77)
    for sb in (a, b, c):
        sb.standardize()
    out, mu, std = O.standardize(xyz[:8], mask[:8])
    assert_close(a.mu[:8], mu, tol=3e-5)
    assert_close(a.get_xyz()[:8], out, tol=3e-5)
    ra = []
    for t in range(T):
        a.diffuse_xyz(betas[t]); ra.append(a.backbone_orientations())
        rb, _ = b.diffuse_xyz_and_frames(betas[t])
        assert torch.equal(ra[-1], rb)
    rc, tc, _ = c.diffuse_trajectory(betas)
    assert torch.equal(rc, torch.stack(ra)) and torch.equal(a.get_xyz(), b.get_xyz()) and torch.equal(a.get_xyz(), c.get_xyz())
    assert_close(rc[-1][:8], O.backbone_orientations(c.get_xyz()[:8].cpu()), bad_frac=1e-3)
    # after T small-beta steps the masked statistics are still ~ (0, 1): mean drifts by O(sqrt(beta/n)), var stays ~1
    z = c.get_xyz()[:8].cpu(); w = mask[:8].unsqueeze(-1).float()
    mean = (z * w).sum((1, 2)) / w.sum((1, 2))
    assert mean.abs().max() < 0.05


def cosine_betas(T, s=8e-3, beta_max=0.999):
    """Variance schedule of the reference's diffusion tutorial (docs/tutorials/diffusing_xyz_coordinates.ipynb,
    cell 2; loop shape README.md:120-150): alpha_bar_t = cos^2((t/T + s)/(1 + s) * pi/2) / alpha_bar_0,
    beta_t = clip(1 - alpha_bar_t / alpha_bar_{t-1}, 1e-5, beta_max), beta_0 = 0; the loop uses beta[0..T-1]."""
    t = torch.arange(T + 1)
    f = torch.cos((t / T + s) / (1 + s) * torch.pi / 2.0).square()
    abar = f / f[0]
    beta = torch.cat([torch.zeros(1), (1 - abar[1:] / abar[:-1]).clamp(min=1e-5, max=beta_max)])
    return beta[:T].float()


def test_config5_full_T300_graph(SB):
    """BASELINE config 5 at its stated size: B=256, N_res=384, cosine schedule T=300, standardize once, then
    (a) the loop diffuse_xyz + backbone_orientations captured in ONE hipGraph and replayed, against
    (b) the LDS-resident trajectory kernel without and with the per-step coordinates (5.3 GB, > 2^32 bytes).
    Reference: protstruc.py:696-734 (standardize), :864-878 (diffuse_xyz), :543-571 (frames); README.md:120-150."""
    B, N, A, T = 256, 384, 15, 300
    xyz, mask = synth(55, B, N, scale=8.0)
    betas = cosine_betas(T)
    assert betas[0] == 0 and betas[-1] > 0.5 and betas.shape == (T,)
    betas_TB = betas[:, None].expand(T, B).contiguous().cuda()
    checks = (0, 149, 299)  # steps 1, 150, 300

    a = SB.from_xyz(xyz.clone(), mask).manual_seed(1234)
    a.standardize()
    ref_std, mu, sd = O.standardize(xyz[:8], mask[:8])
    assert_close(a.mu[:8], mu, tol=3e-5)
    assert_close(a.std[:8], sd, tol=3e-5)
    assert_close(a.get_xyz()[:8], ref_std, tol=3e-5)
    xyz0 = a.get_xyz().clone()          # standardized start, shared by all three runs
    seed_state = a._rng_state.clone()

    # ---- (a) the whole loop in one captured graph --------------------------------------------------------------
    beta_dev = [betas_TB[t] for t in range(T)]   # (B,) device views: nothing is copied at launch time
    a.diffuse_xyz(beta_dev[1]); a.backbone_orientations()    # warm-up outside capture (library + allocator)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    rots_a = []
    with torch.cuda.graph(graph):
        for t in range(T):
            a.diffuse_xyz(beta_dev[t])
            rots_a.append(a.backbone_orientations())
    a.get_xyz().copy_(xyz0)              # undo the warm-up: same start and same draw counter as (b)
    a._rng_state.copy_(seed_state)
    graph.replay()
    torch.cuda.synchronize()
    assert int(a._rng_state[1]) == T and not a._rng_state[2:].any(), "draw counter must advance by T, tickets back to 0"
    final_a = a.get_xyz().clone()
    keep_a = {t: rots_a[t].clone() for t in checks}
    # a second replay continues the noise stream (fresh noise, not a repeat)
    graph.replay()
    torch.cuda.synchronize()
    assert int(a._rng_state[1]) == 2 * T and not torch.equal(a.get_xyz(), final_a)
    del graph, rots_a

    # ---- (b1) trajectory kernel, frames only ---------------------------------------------------------------------
    b = SB.from_xyz(xyz0.clone(), mask).manual_seed(1234)
    rot_b, trans_b, none_xyz = b.diffuse_trajectory(betas_TB)
    torch.cuda.synchronize()
    assert none_xyz is None and rot_b.shape == (T, B, N, 3, 3) and trans_b.shape == (T, B, N, 3)
    assert int(b._rng_state[1]) == T
    for t in checks:
        assert torch.equal(rot_b[t], keep_a[t]), f"step {t + 1}: trajectory kernel != captured loop"
    assert torch.equal(b.get_xyz(), final_a)
    assert torch.equal(trans_b[T - 1], final_a[:, :, 1])

    # ---- (b2) the same with the per-step coordinates into a NaN-prefilled 5.3 GB buffer -----------------------
    c = SB.from_xyz(xyz0.clone(), mask).manual_seed(1234)
    traj = torch.full((T, B, N, A, 3), float("nan"), device="cuda")
    assert traj.numel() * 4 > 2 ** 32
    rot_c = torch.full((T, B, N, 3, 3), float("nan"), device="cuda")
    rot_c2, trans_c, traj2 = c.diffuse_trajectory(betas_TB, want_translations=False, out_orientations=rot_c, out_xyz=traj)
    torch.cuda.synchronize()
    assert rot_c2 is rot_c and traj2 is traj and trans_c is None
    assert not torch.isnan(traj).any(), "unwritten part of xyz_traj"
    assert not torch.isnan(rot_c[:, :8]).any()
    assert torch.equal(traj[T - 1], c.get_xyz()) and torch.equal(c.get_xyz(), final_a)
    assert torch.equal(traj[0], xyz0), "beta_0 = 0 leaves the coordinates untouched (sqrt(0) * eps = 0)"
    for t in checks:
        assert torch.equal(rot_c[t], keep_a[t])
        assert torch.equal(trans_b[t], traj[t][:, :, 1])
    # consecutive steps obey the update rule with SOME unit-variance noise: (x_t - sqrt(1-b) x_{t-1}) / sqrt(b)
    t = 150
    eps = (traj[t] - (1 - betas[t]).sqrt().item() * traj[t - 1]) / betas[t].sqrt().item()
    assert abs(eps.mean().item()) < 5e-3 and abs(eps.var().item() - 1.0) < 5e-3

    # ---- frames of 8 sampled structures at step 300 against the oracle on the final coordinates -----------------
    idx = torch.tensor([0, 31, 64, 100, 127, 200, 254, 255])
    fin = traj[T - 1][idx.cuda()].cpu()
    want = O.backbone_orientations(fin)
    truth = O.backbone_orientations(fin.double())
    got = rot_c[T - 1][idx.cuda()]
    assert_close(got, want, bad_frac=1e-3)
    e_got = (got.cpu().double() - truth).abs().max().item()
    e_ref = (want.double() - truth).abs().max().item()
    assert e_got <= max(4 * e_ref, 2e-5)
    # ---- after 300 steps alpha_bar ~ 0: every coordinate is N(0,1); masked per-structure statistics ~ (0, 1) ---
    z = traj[T - 1]
    w = mask.cuda().unsqueeze(-1).float()
    cnt = w.sum((1, 2))
    mean = (z * w).sum((1, 2)) / cnt
    var = (((z - mean[:, None, None]) ** 2) * w).sum((1, 2)) / cnt
    assert mean.abs().max().item() < 0.08 and (var - 1).abs().max().item() < 0.12


@pytest.mark.parametrize("N", [5, 64, 257, 300])
def test_k3_row_ranges_compact_and_in_place(SB, N):
    """K3's row_begin / row_end addressing (what the multi-GPU row sharding uses): compact and in-place results of
    arbitrary row ranges equal the corresponding rows of the full result bit for bit, nothing outside the range is
    written, for dihedrals and planar angles and several point splits."""
    from protstruc_amd import ops
    xyz, _ = synth(70 + N, 3, N)
    xg = xyz.cuda()
    same = lambda a, b: torch.equal(a.isnan(), b.isnan()) and torch.equal(a.nan_to_num(5.0), b.nan_to_num(5.0))
    rng = np.random.default_rng(N)
    for npts, si, sj in [(4, [1, 4], [1, 4]), (4, [0, 1, 4], [4]), (4, [2], [0, 1, 2]), (3, [1, 4], [4]), (3, [1], [1, 4])]:
        full = ops.pairwise_angles(xg, si, sj, npts)
        for _ in range(4):
            r0 = int(rng.integers(0, N)); r1 = int(rng.integers(r0, N + 1))
            c = ops.pairwise_angles(xg, si, sj, npts, row_begin=r0, row_end=r1, compact=True)
            assert c.shape == (3, r1 - r0, N) and same(c, full[:, r0:r1])
            buf = torch.full((3, N, N), 321.0, device="cuda")
            out = ops.pairwise_angles(xg, si, sj, npts, row_begin=r0, row_end=r1, out=buf)
            assert out is buf and same(buf[:, r0:r1], full[:, r0:r1])
            assert (buf[:, :r0] == 321.0).all() and (buf[:, r1:] == 321.0).all()
    with pytest.raises(ValueError):
        ops.pairwise_angles(xg, [1, 4], [1, 4], 4, row_begin=3, row_end=2)
    with pytest.raises(ValueError):
        ops.pairwise_angles(xg, [1, 4], [1, 4], 4, out=torch.empty(3, N, N + 1, device="cuda"))


# 140, 300, 330, 450: a last strip with one / one / two / three live column groups of 64 (the dead ones are skipped)
@pytest.mark.parametrize("faithful", [False, True], ids=["fast", "faithful"])
@pytest.mark.parametrize("N", [6, 64, 101, 130, 140, 255, 256, 300, 330, 384, 450, 511, 512, 516])
def test_k3_sweep_kernels_bit_identical_to_the_one_column_kernel(SB, N, faithful):
    """The per-CU sweep kernels (two / four column residues per lane, LDS-staged rows, pulled tasks, arithmetic
    interleaved across the columns) evaluate the same operations per pair as the one-column kernel: same bits.  The
    one-column kernel is asked for explicitly (`exact_angles = 2`, diagnostic); the dispatcher's pick for an aligned output
    (vector stores for even N, the 64-apart column layout with dword stores for odd N) and for a 4-byte-misaligned output
    (always the dword layout; short chains: the one-column kernel) are both held to it; full launches and an odd row range.
    `faithful`: the same in the reference's order of operations (`set_exact_angles(True)`, round 5): there the one-column
    kernel CALLS the device library (atan2f, acosf, the compiler's IEEE division) per element while the sweep kernels run
    the library's instruction sequences restated in packed form (ps_common.hpp: atan2_lib_vn, acos_lib_vn, div_ieee_vn) --
    this identity is what pins the restatement to the library."""
    from protstruc_amd import ops
    B = 3
    xyz, _ = synth(900 + N, B, N)
    xyz[1, N // 2] = float("nan")                      # NaN rows and columns propagate identically
    xyz[2, 1] = xyz[2, 0]                              # coincident residues: exact zeros / NaNs as the reference has them
    xg = xyz.cuda()
    same = lambda a, b: torch.equal(a.isnan(), b.isnan()) and torch.equal(a.nan_to_num(5.0), b.nan_to_num(5.0))
    splits = [(4, [1, 4], [1, 4]), (4, [0, 1, 4], [4]), (4, [2], [0, 1, 2]), (4, [1], [4, 1, 0]), (4, [0, 1], [2, 3]),
              (4, [0, 1, 2, 3], []), (4, [], [0, 1, 2, 3]), (3, [1, 4], [4]), (3, [1], [1, 4]), (3, [], [0, 1, 2]), (3, [4, 1, 0], [])]
    try:
        ops.set_exact_angles(faithful)
        for npts, si, sj in splits:
            one = ops.pairwise_angles(xg, si, sj, npts, _one_column=True)
            big = torch.full((B * N * N + 1,), 7.0, device="cuda")
            mis = ops.pairwise_angles(xg, si, sj, npts, out=big[1:].view(B, N, N))      # misaligned: columns 64 apart, dword stores
            fast = ops.pairwise_angles(xg, si, sj, npts)
            assert same(fast, one) and same(mis, one), (npts, si, sj)
            assert big[0] == 7.0
            if N > 8:
                r0, r1 = 3, N - 2                                                      # odd number of rows, odd first row
                c = ops.pairwise_angles(xg, si, sj, npts, row_begin=r0, row_end=r1, compact=True)
                assert same(c, one[:, r0:r1]), (npts, si, sj)
    finally:
        ops.set_exact_angles(False)


def test_k3_exact_angles_mode(SB):
    """`ops.set_exact_angles(True)` (ABI 4: `exact_angles` of ps_pairwise_angles_f32 / ps_inter_residue_geometry_f32) selects
    geometry.dihedral / geometry.angle in the reference's order of operations (geometry.py:110-124, :64-66), as
    `exact_sqrt` does for K1.  At B=8, N=256, unit scale: NO off-diagonal dihedral more than 1e-5 from the oracle, max
    <= 1e-6 (the fast default: 3.8e-6 of entries beyond 1e-5, max 6.5e-5); the planar angle within its conditioning
    gate and closer to the oracle than the fast form; exact where the reference is exact; the featuriser's three angle
    planes equal the K3 launches bit for bit in this mode too; the default comes back unchanged."""
    from protstruc_amd import ops
    B, N = 8, 256
    g = torch.Generator().manual_seed(0)
    xyz = torch.randn(B, N, 15, 3, generator=g)
    sb = SB.from_xyz(xyz)
    off = ~torch.eye(N, dtype=torch.bool).expand(B, N, N)
    wrap = lambda d: torch.minimum(d.abs(), (2 * np.pi - d.abs()).abs())
    same = lambda a, b: torch.equal(a.isnan(), b.isnan()) and torch.equal(a.nan_to_num(5.0), b.nan_to_num(5.0))
    assert ops.get_exact_angles() is False
    fast = {}
    for key, (ai, aj) in {"omega": (["CA", "CB"], ["CA", "CB"]), "theta": (["N", "CA", "CB"], ["CB"])}.items():
        fast[key] = sb.pairwise_dihedrals(ai, aj)
    fast["phi"] = sb.pairwise_planar_angles(["CA", "CB"], ["CB"])
    try:
        ops.set_exact_angles(True)
        assert ops.get_exact_angles() is True
        ex = {}
        for key, (ai, aj, si, sj) in {"omega": (["CA", "CB"], ["CA", "CB"], [1, 4], [1, 4]),
                                      "theta": (["N", "CA", "CB"], ["CB"], [0, 1, 4], [4])}.items():
            ex[key] = sb.pairwise_dihedrals(ai, aj)
            ref = O.pairwise_dihedrals(xyz, si, sj)
            err = wrap(ex[key].cpu() - ref)[off]
            assert err.max().item() <= 1e-6 and (err > 1e-5).sum().item() == 0, (key, err.max().item())
            diag = torch.diagonal(ex[key], dim1=1, dim2=2)
            assert ((diag == 0) & ~torch.signbit(diag)).all()             # exactly +0.0 where the reference is
        ex["phi"] = sb.pairwise_planar_angles(["CA", "CB"], ["CB"])
        ref = O.pairwise_planar_angles(xyz, [1, 4], [4])
        both = off & ~(ref.isnan() | ex["phi"].cpu().isnan())
        e_exact, e_fast = (ex["phi"].cpu() - ref).abs()[both], (fast["phi"].cpu() - ref).abs()[both & ~fast["phi"].cpu().isnan()]
        assert (e_exact > 1e-5).float().mean().item() <= 1e-4
        assert e_exact.median().item() <= e_fast.median().item() + 1e-9
        assert torch.diagonal(ex["phi"], dim1=1, dim2=2).isnan().all()
        geo = sb.inter_residue_geometry()
        for key in ("omega", "theta", "phi"):
            assert same(geo[key], ex[key]), key
            assert not torch.equal(ex[key].nan_to_num(5.0), fast[key].nan_to_num(5.0))   # it IS another arithmetic
        # odd N and a row range take the same path
        x5 = xyz[:2, :37].contiguous().cuda()
        full = ops.pairwise_angles(x5, [1, 4], [1, 4], 4)
        part = ops.pairwise_angles(x5, [1, 4], [1, 4], 4, row_begin=5, row_end=20, compact=True)
        assert same(part, full[:, 5:20])
        assert wrap(full.cpu() - O.pairwise_dihedrals(xyz[:2, :37], [1, 4], [1, 4]))[~torch.eye(37, dtype=torch.bool).expand(2, 37, 37)].max() <= 1e-6
    finally:
        ops.set_exact_angles(False)
    assert same(sb.pairwise_dihedrals(["CA", "CB"], ["CA", "CB"]), fast["omega"])
    geo = sb.inter_residue_geometry()
    assert same(geo["omega"], fast["omega"]) and same(geo["phi"], fast["phi"])
    lib = __import__("protstruc_amd._lib", fromlist=["load"]).load()
    import ctypes
    arr = (ctypes.c_int * 4)
    xg = xyz[:1].contiguous().cuda(); out = torch.empty(1, N, N, device="cuda")
    rc = lib.ps_pairwise_angles_f32(xg.data_ptr(), out.data_ptr(), 1, N, 15, 4, arr(0, 0, 1, 1), arr(1, 4, 1, 4), 0, N, N, 0, 4, None)
    assert rc == 1                                                        # exact_angles outside {0, 1, 2, 3}: refused before any launch


def test_k3_planar_angle_collinear_and_extreme_arms(SB):
    """Where the fast planar angle (cos = dot * rsq(|ba|^2) * rsq(|bc|^2), polynomial acos) may leave the reference, pinned:
    exactly collinear points (the reference's correctly rounded division gives cos = -1 or 1 exactly -> pi or 0; an ulp of
    v_rsq_f32 puts the fast cosine at 1 -+ 6e-8 -> within 5e-4 of it or NaN).  Documented in INTEGRATION.md.  Extreme arm
    lengths are NOT such a case any more (round 5: each arm's reciprocal length is applied before the next factor comes in;
    rounds 3-4 multiplied the squared lengths first, which overflowed at 1e10 -> pi / 2 and underflowed at 1e-12 -> NaN):
    arms of 1e10 and of 1e-12 give the angles of the unit-scale configuration.  The faithful mode
    (`set_exact_angles(True)`) equals the oracle to 1e-6 everywhere, collinear points included."""
    from protstruc_amd import ops
    A = 5
    xyz = torch.zeros(1, 4, A, 3)
    # residue r: CA = slot 1, CB = slot 4.  planar(CA_i, CB_i, CB_j) with i = 0
    xyz[0, 0, 1] = torch.tensor([0.0, 0.0, 0.0]); xyz[0, 0, 4] = torch.tensor([3.0, 0.0, 0.0])     # arm (-3, 0, 0)
    xyz[0, 1, 4] = torch.tensor([6.0, 0.0, 0.0])          # collinear, opposite side: angle pi
    xyz[0, 2, 4] = torch.tensor([1.0, 0.0, 0.0])          # collinear, same side: angle 0
    xyz[0, 3, 4] = torch.tensor([3.0, 7.0, 0.0])          # right angle
    sb = SB.from_xyz(xyz)
    ref = O.pairwise_planar_angles(xyz, [1, 4], [4])[0, 0]
    assert ref[1].item() == pytest.approx(np.pi, abs=1e-6) and ref[2].item() == 0.0 and ref[3].item() == pytest.approx(np.pi / 2, abs=1e-6)
    fast = sb.pairwise_planar_angles(["CA", "CB"], ["CB"])[0, 0].cpu()
    for j in (1, 2):                                       # ill-conditioned: sqrt(2 ulp) ~ 3.5e-4, or NaN just past 1
        assert fast[j].isnan() or abs(fast[j].item() - ref[j].item()) <= 5e-4
    assert abs(fast[3].item() - np.pi / 2) <= 1e-6
    big = xyz.clone(); big[0, :, :, :] *= 1e10
    tiny = xyz.clone(); tiny[0, :, :, :] *= 1e-12
    fb = SB.from_xyz(big).pairwise_planar_angles(["CA", "CB"], ["CB"])[0, 0].cpu()
    ft = SB.from_xyz(tiny).pairwise_planar_angles(["CA", "CB"], ["CB"])[0, 0].cpu()
    for f in (fb, ft):                                     # scale-free: the unit-scale answers at 1e10 and at 1e-12
        assert abs(f[3].item() - np.pi / 2) <= 1e-6
        assert f[1].isnan() or abs(f[1].item() - np.pi) <= 5e-4
        assert f[2].isnan() or abs(f[2].item()) <= 5e-4
    try:
        ops.set_exact_angles(True)
        for x in (xyz, big, tiny):
            got = SB.from_xyz(x).pairwise_planar_angles(["CA", "CB"], ["CB"])[0, 0].cpu()
            want = O.pairwise_planar_angles(x, [1, 4], [4])[0, 0]
            assert torch.equal(got.isnan(), want.isnan())
            assert (got - want).nan_to_num(0).abs().max().item() <= 1e-6
    finally:
        ops.set_exact_angles(False)


@pytest.mark.parametrize("N", [2048, 4608])
def test_k3_sweep_kernels_long_chains(SB, N):
    """Long chains: one structure's rows fill 49-110 KB of the workgroup's LDS (N = 2048, 4608 with two row-side points), a
    split with three row-side points no longer fits at N = 4608 and falls back to the one-column kernel, and a workgroup's
    share of the task list is a fraction of one (structure, strip) segment.  Same bits as the one-column kernel."""
    from protstruc_amd import ops
    B = 1
    xyz, _ = synth(1000 + N, B, N)
    xg = xyz.cuda()
    same = lambda a, b: torch.equal(a.isnan(), b.isnan()) and torch.equal(a.nan_to_num(5.0), b.nan_to_num(5.0))
    for npts, si, sj in [(4, [1, 4], [1, 4]), (4, [0, 1, 4], [4]), (3, [1, 4], [4])]:
        one = ops.pairwise_angles(xg, si, sj, npts, _one_column=True)
        big = torch.full((B * N * N + 1,), 7.0, device="cuda")
        mis = ops.pairwise_angles(xg, si, sj, npts, out=big[1:].view(B, N, N))      # misaligned: dword-store layout
        fast = ops.pairwise_angles(xg, si, sj, npts)
        assert same(fast, one) and same(mis, one), (npts, si, sj)
        part = ops.pairwise_angles(xg, si, sj, npts, row_begin=N // 3, row_end=N // 3 + 777, compact=True)   # a shard
        assert same(part, one[:, N // 3:N // 3 + 777])
    geo = SB.from_xyz(xyz).inter_residue_geometry()
    assert same(geo["omega"], ops.pairwise_angles(xg, [1, 4], [1, 4], 4)) and same(geo["phi"], ops.pairwise_angles(xg, [1, 4], [4], 3))


def test_k3_differential_fuzz(SB):
    """Random shapes, point splits, row ranges and output forms: the sweep kernels (and whatever the dispatcher picks for
    the shape -- four or two columns per lane with vector stores, the 64-apart layout with dword stores for odd N, the
    one-column kernel for short chains) against the one-column kernel asked for explicitly, bit for bit; nothing outside
    the requested rows is written."""
    from protstruc_amd import ops
    import os
    # PS_K3_FUZZ_SEED / PS_K3_FUZZ_TRIALS: one-off longer runs (the committed defaults are what CI runs)
    rng = np.random.default_rng(int(os.environ.get("PS_K3_FUZZ_SEED", "20260404")))
    same = lambda a, b: torch.equal(a.isnan(), b.isnan()) and torch.equal(a.nan_to_num(5.0), b.nan_to_num(5.0))
    for trial in range(int(os.environ.get("PS_K3_FUZZ_TRIALS", "80"))):
        B = int(rng.integers(1, 5))
        N = int(rng.choice([int(rng.integers(1, 40)), int(rng.integers(40, 700)), 4 * int(rng.integers(1, 160)), 256, 384]))
        A = int(rng.choice([5, 15, 25]))
        npts = int(rng.choice([3, 4]))
        n_i = int(rng.integers(0, npts + 1))
        slots = [int(x) for x in rng.integers(0, A, size=npts)]
        si, sj = slots[:n_i], slots[n_i:]
        g = torch.Generator().manual_seed(7000 + trial)
        xyz = torch.randn(B, N, A, 3, generator=g)
        if trial % 5 == 0 and N > 2:
            xyz[0, int(rng.integers(0, N))] = float("nan")
        if trial % 7 == 0 and N > 2:
            xyz[-1, 1] = xyz[-1, 0]
        xg = xyz.cuda()
        ops.set_exact_angles(trial % 3 == 2)        # every third trial in the reference's order of operations
        one = ops.pairwise_angles(xg, si, sj, npts, _one_column=True)
        if trial % 4 == 1:          # the misaligned-output path as well
            big = torch.full((B * N * N + 1,), 7.0, device="cuda")
            assert same(ops.pairwise_angles(xg, si, sj, npts, out=big[1:].view(B, N, N)), one) and big[0] == 7.0
        r0 = int(rng.integers(0, N)); r1 = int(rng.integers(r0, N + 1))
        if trial % 3 == 0:
            r0, r1 = 0, N
        if trial % 2 == 0:
            got = ops.pairwise_angles(xg, si, sj, npts, row_begin=r0, row_end=r1, compact=True)
            assert same(got, one[:, r0:r1]), (trial, B, N, A, npts, si, sj, r0, r1)
        else:
            buf = torch.full((B, N, N), 321.0, device="cuda")
            ops.pairwise_angles(xg, si, sj, npts, row_begin=r0, row_end=r1, out=buf)
            assert same(buf[:, r0:r1], one[:, r0:r1]), (trial, B, N, A, npts, si, sj, r0, r1)
            assert (buf[:, :r0] == 321.0).all() and (buf[:, r1:] == 321.0).all()
    ops.set_exact_angles(False)


@pytest.mark.parametrize("N", [1, 2, 3, 5, 8, 15, 16, 17, 31, 32, 33, 48, 63, 64])
def test_k3_short_chain_kernel_bit_identical_to_the_one_column_kernel(SB, N):
    """Short chains (N <= 64) take a kernel of their own -- one wave per structure, lanes = (row group, column), four rows
    per trip -- with the one-column kernel's arithmetic: same bits, for every point split, batch sizes that fill a
    workgroup's four waves partly, row ranges, compact and in-place outputs."""
    from protstruc_amd import ops
    same = lambda a, b: torch.equal(a.isnan(), b.isnan()) and torch.equal(a.nan_to_num(5.0), b.nan_to_num(5.0))
    splits = [(4, [1, 4], [1, 4]), (4, [0, 1, 4], [4]), (4, [2], [0, 1, 2]), (4, [0, 1, 2, 3], []), (4, [], [0, 1, 2, 3]),
              (3, [1, 4], [4]), (3, [1], [1, 4]), (3, [], [0, 1, 2]), (3, [4, 1, 0], [])]
    for B in (1, 5, 9):
        xyz, _ = synth(1300 + N + B, B, N)
        if N > 1:
            xyz[0, N // 2] = float("nan")
            xyz[B - 1, 1] = xyz[B - 1, 0]
        xg = xyz.cuda()
        for faithful in (False, True):               # the fast arithmetic, and the reference's order of operations
            try:
                ops.set_exact_angles(faithful)
                for npts, si, sj in splits:
                    one = ops.pairwise_angles(xg, si, sj, npts, _one_column=True)
                    assert same(ops.pairwise_angles(xg, si, sj, npts), one), (N, B, npts, si, sj, faithful)
                    r0, r1 = N // 3, N - N // 4
                    assert same(ops.pairwise_angles(xg, si, sj, npts, row_begin=r0, row_end=r1, compact=True), one[:, r0:r1])
                    buf = torch.full((B, N, N), 321.0, device="cuda")
                    ops.pairwise_angles(xg, si, sj, npts, row_begin=r0, row_end=r1, out=buf)
                    assert same(buf[:, r0:r1], one[:, r0:r1]) and (buf[:, :r0] == 321.0).all() and (buf[:, r1:] == 321.0).all()
            finally:
                ops.set_exact_angles(False)


def _angle_gates(got, ref, npts, faithful, where):
    """The parity gates of a K3 plane against the oracle (SURVEY hard part 3).  fast: >= 1 - 1e-4 of the off-diagonal entries
    within 1e-5 (angular distance), NaN positions equal up to 1e-5 of entries (|cos| rounding just past 1); faithful: NO
    dihedral beyond 1e-5 anywhere and NaN positions EQUAL; the faithful planar angle within 1e-5 wherever the angle is
    more than 0.05 rad from 0 and pi, and inside the 1e-4 gate everywhere."""
    n = ref.shape[-1]
    off = ~torch.eye(n, dtype=torch.bool).expand_as(ref) if ref.shape[-2] == n else torch.ones_like(ref, dtype=torch.bool)
    d = (got - ref).abs()
    err = torch.minimum(d, (2 * np.pi - d).abs()) if npts == 4 else d
    both = ~(got.isnan() | ref.isnan())
    slack = 2.0 / max(1, int((both & off).sum()))                 # short chains: one ill-conditioned entry is allowed
    if faithful:
        assert torch.equal(got.isnan(), ref.isnan()), where
        if npts == 4:
            assert (err[both] > 1e-5).sum().item() == 0, (where, err[both].max().item())
        else:
            well = both & ((ref - np.pi).abs() > 5e-2) & (ref.abs() > 5e-2)
            assert (err[well] > 1e-5).sum().item() == 0, (where, err[well].max().item())
            assert (err[both] > 1e-5).float().mean().item() <= 1e-4 + slack, where
    else:
        assert (err[both & off] > 1e-5).float().mean().item() <= 1e-4 + slack, (where, (err[both & off] > 1e-5).float().mean().item())
        assert (got.isnan() != ref.isnan())[off].float().mean().item() <= 1e-5 + slack, where


def _picks(B, N):
    return sorted({0, B // 3, (2 * B) // 3, B - 1}) if N < 1000 else [0]


def _k3_table_ids():
    from tests.k3_families import K3_SHAPES
    return [e for e in K3_SHAPES if e[8]]


@pytest.mark.parametrize("entry", _k3_table_ids(), ids=lambda e: f"B{e[0]}-N{e[1]}-np{e[2][0]}-i{len(e[2][1])}-mis{e[5]}-mode{e[6]}")
def test_k3_every_dispatch_arm_vs_oracle(SB, entry):
    """tests/k3_families.py: one launch per arm of K3's dispatcher (k3_small at 16 / 32 padded columns; the sweep at four / two
    columns per lane, adjacent or 64 apart, dead column groups skipped or not, one or two workgroups per CU; the one-column
    kernel; each in both arithmetic modes).  tests/test_k3_plan.py (CPU) holds the table to the dispatcher; here each entry is
    launched through the C ABI into a sentinel-framed buffer at the alignment the entry names, the plan of THAT launch
    (actual address, this device's CU count) must be the arm the table names, and the result is held to the ORACLE."""
    import ctypes
    from protstruc_amd import _lib, ops
    from tests.k3_families import K3_ARM_KEYS, arm_key
    B, N, (npts, si, sj), rows, compact, mis, mode, want, _ = entry
    r0, r1 = rows if rows else (0, N)
    xyz, _m = synth(8800 + N + B, B, N)
    xg = xyz.cuda()
    out_rows = (r1 - r0) if compact else N
    pad = 64
    big = torch.full((B * out_rows * N + 2 * pad + 4,), 777.0, device="cuda")
    base = pad + ((mis - big.data_ptr() % 16) % 16) // 4
    out = big[base:base + B * out_rows * N].view(B, out_rows, N)
    assert out.data_ptr() % 16 == mis
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    plan = _lib.k3_plan(B, N, 15, si, sj, npts, r0, r1, compact=compact, out_misalign=mis, exact_angles=mode, cu_count=cus)
    if cus == 256:
        assert arm_key(plan, K3_ARM_KEYS) == tuple(want[k] for k in K3_ARM_KEYS), plan
    try:
        ops.set_exact_angles(bool(mode & 1))
        got = ops.pairwise_angles(xg, si, sj, npts, row_begin=r0, row_end=r1, compact=compact, out=out, _one_column=bool(mode & 2))
    finally:
        ops.set_exact_angles(False)
    torch.cuda.synchronize()
    assert got is out
    assert (big[:base] == 777.0).all() and (big[base + B * out_rows * N:] == 777.0).all(), "a sentinel around the output changed"
    lo = 0 if compact else r0
    if not compact:
        assert (out[:, :r0] == 777.0).all() and (out[:, r1:] == 777.0).all(), "rows outside the requested range were written"
    assert not (out[:, lo:lo + (r1 - r0)] == 777.0).any(), "part of the requested rows was not written"
    pick = _picks(B, N)
    ref = (O.pairwise_dihedrals if npts == 4 else O.pairwise_planar_angles)(xyz[pick], si, sj)[:, r0:r1]
    g = out[pick][:, lo:lo + (r1 - r0)].cpu()
    _angle_gates(g, ref, npts, bool(mode & 1), (B, N, npts, si, sj, mode, plan["kernel"]))


def _featuriser_table_ids():
    from tests.k3_families import FEATURISER_SHAPES
    return [e for e in FEATURISER_SHAPES if e[6]]


@pytest.mark.parametrize("entry", _featuriser_table_ids(), ids=lambda e: f"B{e[0]}-N{e[1]}-f{e[2]}-m{e[3]}-mode{e[4]}")
def test_featuriser_every_dispatch_arm_vs_oracle(SB, entry):
    """tests/k3_families.py: one launch per arm of the featuriser's dispatcher (four / two columns per lane, vector / dword
    float stores, strip-local / flat mask stores, write-through or not, one or two workgroups per CU, the one-column kernel;
    both arithmetic modes), through the C ABI into sentinel-framed planes at the alignment the entry names, each plane held
    to the ORACLE: distances <= 1e-5, masks exact, angles by the gates of `_angle_gates`."""
    from protstruc_amd import _lib
    from protstruc_amd.ops import _ptr, _stream
    from tests.k3_families import FEATURISER_ARM_KEYS, arm_key
    B, N, fmis, mmis, mode, want, _ = entry
    xyz, mask = synth(9900 + N + B, B, N)
    mask[0, N // 2] = False
    xg, mg = xyz.cuda(), mask.cuda().to(torch.uint8)
    plane, pad = B * N * N, 256
    fbuf = torch.full((6 * (plane + pad) + pad,), 777.0, device="cuda")
    mbuf = torch.full((3 * (plane + pad) + pad,), 7, dtype=torch.uint8, device="cuda")
    # plane k of the six / three starts at the entry's misalignment (modulo 128) -- the first plane exactly, the others too
    foff = [pad + i * (plane + pad) for i in range(6)]
    foff = [o + ((fmis - (fbuf.data_ptr() + 4 * o) % 128) % 128) // 4 for o in foff]
    moff = [pad + i * (plane + pad) for i in range(3)]
    moff = [o + (mmis - (mbuf.data_ptr() + o) % 128) % 128 for o in moff]
    alf = 0
    for o in foff:
        alf |= (fbuf.data_ptr() + 4 * o) % 128
    alm = 0
    for o in moff:
        alm |= (mbuf.data_ptr() + o) % 128
    assert alf == fmis and alm == mmis
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    plan = _lib.featuriser_plan(B, N, 15, float_misalign=alf, mask_misalign=alm, exact_angles=mode, cu_count=cus)
    if cus == 256:
        assert arm_key(plan, FEATURISER_ARM_KEYS) == tuple(want[k] for k in FEATURISER_ARM_KEYS), plan
    rc = _lib.load().ps_inter_residue_geometry_f32(_ptr(xg), _ptr(mg), *[fbuf.data_ptr() + 4 * o for o in foff],
                                                   *[mbuf.data_ptr() + o for o in moff], B, N, 15, 0, mode, _stream(xg))
    assert rc == 0
    torch.cuda.synchronize()
    keepf = torch.ones_like(fbuf, dtype=torch.bool)
    keepm = torch.ones_like(mbuf, dtype=torch.bool)
    for o in foff:
        keepf[o:o + plane] = False
    for o in moff:
        keepm[o:o + plane] = False
    assert (fbuf[keepf] == 777.0).all() and (mbuf[keepm] == 7).all(), "a sentinel between the planes changed"
    pick = _picks(B, N)
    refd = O.inter_residue_geometry(xyz[pick][:, :, :5].contiguous(), mask[pick][:, :, :5].contiguous())   # slots N, CA, C, O, CB only
    names = ["d_ca", "d_cb", "d_no", "omega", "theta", "phi"]
    for k, name in enumerate(names):
        g = fbuf[foff[k]:foff[k] + plane].view(B, N, N)[pick].cpu()
        assert not (g == 777.0).any(), name
        where = (B, N, mode, name, plan["kernel"])
        if name.startswith("d_"):
            assert torch.equal(g.isnan(), refd[name].isnan()) and (g - refd[name]).abs().nan_to_num(0).max().item() <= 1e-5, where
        else:
            _angle_gates(g, refd[name], 3 if name == "phi" else 4, bool(mode & 1), where)
    for k, name in enumerate(["d_ca_mask", "d_cb_mask", "d_no_mask"]):
        g = mbuf[moff[k]:moff[k] + plane].view(B, N, N)[pick].cpu()
        assert torch.equal(g.bool(), refd[name].bool()) and int(g.max()) <= 1, (B, N, mode, name)


def test_k3_inside_a_captured_graph(SB):
    """The sweep kernels ask for more than 64 KB of dynamic LDS, which has to be allowed once per kernel
    (hipFuncSetAttribute at an instantiation's first launch, possibly a captured one: legal during capture --
    tools/microbench/capture_attr_test.hip, profiles/r04_capture_attr_test.log).  Captured launches of K3 must replay to
    the eager result bit for bit."""
    from protstruc_amd import ops
    N = 134                                      # a per-CU sweep length (>= 100) that no other test launches these splits at
    xyz, mask = synth(4242, 2, N)
    xg, mg = xyz.cuda(), mask.cuda()
    si, sj = [3, 0], [2, 1]                      # O_i, N_i | C_j, CA_j
    si2, sj2 = [3], [0, 2, 1]                    # planar O_i | N_j, C_j
    out = torch.empty(2, N, N, device="cuda")
    out2 = torch.empty(2, N, N, device="cuda")
    xo, mo = synth(4243, 3, 131)                 # odd: the featuriser's 64-floats-per-store layout with flat mask stores
    xog, mog = xo.cuda(), mo.cuda()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            ops.pairwise_angles(xg, si, sj, 4, out=out)
            ops.pairwise_angles(xg, si2, sj2, 3, out=out2)
            geo = ops.inter_residue_geometry(xog, mog)
    out.fill_(7.0); out2.fill_(7.0)
    for v in geo.values():
        v.fill_(1)
    g.replay()
    torch.cuda.synchronize()
    same = lambda a, b: torch.equal(a.isnan(), b.isnan()) and torch.equal(a.nan_to_num(5.0), b.nan_to_num(5.0))
    assert same(out, ops.pairwise_angles(xg, si, sj, 4)) and same(out2, ops.pairwise_angles(xg, si2, sj2, 3))
    eager = ops.inter_residue_geometry(xog, mog)
    for k, v in geo.items():
        assert (torch.equal(v, eager[k]) if v.dtype == torch.bool else same(v, eager[k])), k


def test_k3_errors(SB):
    xyz, mask = synth(5, 1, 8)
    sb = SB.from_xyz(xyz, mask)
    with pytest.raises(ValueError, match="Atom XX is not valid."):
        sb.pairwise_dihedrals(["CA", "XX"], ["CA", "CB"])
    with pytest.raises(ValueError):
        sb.pairwise_planar_angles(["CA"], ["CG"])
    assert sb.pairwise_dihedrals(["ca", "cb"], ["Ca", "Cb"]).shape == (1, 8, 8)


# ----------------------------------------------------------------------------- K4
def test_k4_golden(SB):
    g = load_golden("g5_frames")
    sb = SB.from_xyz(g["xyz"], g["atom_mask"])
    assert_close(sb.backbone_orientations(), g["rot_default"])
    assert_close(sb.backbone_orientations("C", "CA", "N"), g["rot_C_CA_N"])
    assert_close(sb.backbone_orientations("CB", "CA", "O"), g["rot_CB_CA_O"])
    assert torch.equal(sb.backbone_translations().cpu(), g["trans_CA"])
    assert torch.equal(sb.backbone_translations("N").cpu(), g["trans_N"])
    rot, trans = sb.backbone_orientations_and_translations()
    assert_close(rot, g["rot_default"])
    assert torch.equal(trans.cpu(), g["trans_CA"]) and trans.is_contiguous()
    with pytest.raises(KeyError):
        sb.backbone_orientations("N", "CA", "CX")
    # reference tests/test_geometry.py:246-262: ideal backbone -> identity frame, exactly
    ideal = SB.from_xyz(g["ideal_xyz"]).backbone_orientations().cpu()
    assert (ideal == torch.eye(3).expand(2, 10, -1, -1)).all()


def test_k4_vs_oracle_orthonormal(SB):
    """Config-5 shape.  Random-normal N/CA/C are occasionally almost collinear, which makes the
    normalisation of the second axis ill-conditioned (the oracle itself is then >1e-5 from an fp64
    evaluation), so gate like K3: almost all entries within 1e-5 and no worse than the oracle vs fp64."""
    xyz, mask = synth(400, 256, 384)
    rot = SB.from_xyz(xyz, mask).backbone_orientations()
    ref = O.backbone_orientations(xyz)
    truth = O.backbone_orientations(xyz.double())
    assert_close(rot, ref, bad_frac=1e-4)
    e_got = (rot.cpu().double() - truth).abs()
    e_ref = (ref.double() - truth).abs()
    assert e_got.max().item() <= max(4 * e_ref.max().item(), 2e-5)
    assert (e_got > 1e-5).float().mean().item() <= (e_ref > 1e-5).float().mean().item() + 1e-5
    eye = torch.eye(3, device=rot.device).expand_as(rot)
    dev_got = (rot.transpose(-1, -2) @ rot - eye).abs().max().item()
    dev_ref = (ref.transpose(-1, -2) @ ref - eye.cpu()).abs().max().item()
    assert dev_got <= max(2 * dev_ref, 1e-5), "frames must be as orthonormal as the reference's"


# ----------------------------------------------------------------------------- K6
def test_k6_golden_and_roundtrip(SB):
    g = load_golden("g6_standardize")
    for k in range(4):
        sb = SB.from_xyz(g[f"xyz_{k}"].clone(), g[f"atom_mask_{k}"])
        sb.standardize()
        assert_close(sb.mu, g[f"mu_{k}"], tol=2e-5)      # coordinates here are at 3..25 A scale
        assert_close(sb.std, g[f"std_{k}"], tol=2e-5)
        assert_close(sb.get_xyz(), g[f"std_xyz_{k}"], tol=2e-5)
        with pytest.raises(ValueError, match="already standardized"):
            sb.standardize()
        sb.unstandardize()
        want = g[f"unstd_xyz_{k}"]
        got = sb.get_xyz().cpu()
        assert torch.equal(got.isnan(), want.isnan())
        # reference's own round-trip tolerance (tests/test_StructureBatch.py:255)
        assert torch.allclose(got.nan_to_num(0), want.nan_to_num(0), rtol=1e-4, atol=1e-5)
        with pytest.raises(ValueError, match="not standardized"):
            sb.unstandardize()


@pytest.mark.parametrize("B,N,A", [(256, 384, 15), (4, 100, 15), (3, 33, 5), (2, 853, 15), (2, 1000, 15), (5, 1, 15)])
def test_k6_lds_resident_equals_streaming(SB, B, N, A):
    """The LDS-resident standardize kernel (one read + one write of the coordinates; taken whenever a structure fits in
    150 KB of LDS) against the three-sweep streaming kernel: mu, std and coordinates bit for bit, incl. structures that
    do not start 16-byte aligned (N*A*3 odd), NaN atoms under the mask, the largest structure that still fits
    (N=853) and one that does not (N=1000: both variants stream)."""
    import ctypes
    from protstruc_amd import _lib
    lib = _lib.load()
    xyz, mask = synth(600 + N, B, N, A=A, scale=9.0)
    xyz[0, N // 2, A - 1] = float("nan")
    mask[0, N // 2, A - 1] = False
    outs = []
    for variant in (0, 1):
        x = xyz.clone().cuda()
        m = mask.cuda().view(torch.uint8)
        mu = torch.empty(B, 3, device="cuda")
        sd = torch.empty(B, 3, device="cuda")
        rc = lib.ps_standardize_variant_f32(x.data_ptr(), m.data_ptr(), mu.data_ptr(), sd.data_ptr(), B, N, A, variant,
                                            torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert rc == 0
        outs.append((x, mu, sd))
    same = lambda a, b: torch.equal(a.isnan(), b.isnan()) and torch.equal(a.nan_to_num(3.0), b.nan_to_num(3.0))
    assert same(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
    ref, rmu, rsd = O.standardize(xyz[:2], mask[:2])
    assert_close(outs[0][1][:2], rmu, tol=3e-5)
    assert_close(outs[0][0][:2], ref, tol=3e-5)
    assert lib.ps_standardize_variant_f32(0, 0, 0, 0, B, N, A, 2, None) == 1


def test_k6_batched_vs_oracle(SB):
    xyz, mask = synth(500, 16, 100, scale=12.0)
    xyz = xyz + torch.randn(16, 1, 1, 3) * 20
    sb = SB.from_xyz(xyz.clone(), mask)
    sb.standardize()
    out, mu, std = O.standardize(xyz, mask)
    assert_close(sb.mu, mu, tol=2e-5)
    assert_close(sb.std, std, tol=2e-5)
    assert_close(sb.get_xyz(), out, tol=2e-5)
    # masked statistics of the result are (0, 1) per structure and axis
    z = sb.get_xyz().cpu()
    w = mask.unsqueeze(-1).float()
    cnt = w.sum((1, 2))
    mean = (z * w).sum((1, 2)) / cnt
    var = ((z - mean[:, None, None]) ** 2 * w).sum((1, 2)) / cnt
    assert mean.abs().max() < 1e-5 and (var - 1).abs().max() < 1e-4
    with pytest.raises(ValueError, match="Only one of"):
        SB.from_xyz(xyz, mask).standardize(atom_mask=mask, residue_mask=mask.any(-1))
    # mask arguments restrict the atoms used (reference intent, Q3/Q4)
    sel = mask.any(-1)
    sel[:, 50:] = False
    sb2 = SB.from_xyz(xyz.clone(), mask)
    sb2.standardize(residue_mask=sel)
    out2, mu2, std2 = O.standardize(xyz, mask & sel.unsqueeze(-1))
    assert_close(sb2.mu, mu2, tol=2e-5)
    assert_close(sb2.get_xyz(), out2, tol=5e-5)


# ----------------------------------------------------------------------------- K5
def test_k5_golden_deterministic_part(SB):
    g = load_golden("g7_diffuse")
    sb = SB.from_xyz(g["xyz"].clone(), g["atom_mask"])
    sb.diffuse_xyz(g["beta"], noise=g["noise"])
    assert torch.equal(sb.get_xyz().cpu(), g["out"]), "sqrt(1-b)*x + eps*sqrt(b) must be bit-exact given eps"


@pytest.mark.parametrize("N", [5, 7, 16])
def test_k5_ragged_struct_boundaries(SB, N):
    xyz, mask = synth(600 + N, 3, N)
    beta = torch.tensor([0.1, 0.5, 0.9])
    noise = torch.randn(3, N, 15, 3, generator=torch.Generator().manual_seed(1))
    sb = SB.from_xyz(xyz.clone(), mask)
    sb.diffuse_xyz(beta, noise=noise)
    # Bit-exactness is asserted against the committed golden fixture above; against the live oracle
    # allow 1 ulp, because torch's CPU sqrt kernel is host-dependent (on the MI355X box's host it is
    # 1 ulp off the correctly rounded sqrt(0.1), sqrt(0.9) that the GPU and numpy produce).
    assert torch.allclose(sb.get_xyz().cpu(), O.diffuse_xyz(xyz, beta, noise), rtol=2e-7, atol=1e-7)
    x, e, bb = xyz.numpy(), noise.numpy(), beta.numpy().reshape(3, 1, 1, 1)
    strict = (np.sqrt(np.float32(1) - bb) * x).astype(np.float32) + (e * np.sqrt(bb)).astype(np.float32)
    assert np.array_equal(sb.get_xyz().cpu().numpy(), strict), "must equal unfused IEEE fp32 arithmetic bit-for-bit"


def test_k5_sampler_statistics(SB):
    """The device sampler cannot match CPU mt19937 bit-for-bit; check it statistically."""
    B, N = 8, 512
    xyz = torch.zeros(B, N, 15, 3)
    sb = SB.from_xyz(xyz.clone()).manual_seed(1234)
    beta = torch.full((B,), 1.0)           # xyz <- eps exactly
    sb.diffuse_xyz(beta)
    e1 = sb.get_xyz().clone()
    sb.diffuse_xyz(beta)                    # sqrt(0)*eps1 + eps2
    e2 = sb.get_xyz().clone()
    n = e1.numel()
    for e in (e1, e2):
        assert abs(e.mean().item()) < 5 / np.sqrt(n)
        assert abs(e.var().item() - 1) < 5 * np.sqrt(2 / n)
        assert abs((e ** 3).mean().item()) < 5 * np.sqrt(15 / n)
        assert abs((e ** 4).mean().item() - 3) < 5 * np.sqrt(96 / n)
    # successive draws and neighbouring coordinates are uncorrelated
    assert abs((e1 * e2).mean().item()) < 5 / np.sqrt(n)
    f = e1.flatten()
    assert abs((f[1:] * f[:-1]).mean().item()) < 5 / np.sqrt(n)
    assert not torch.equal(e1, e2)
    # Kolmogorov-Smirnov against N(0,1)
    from scipy import stats
    ks = stats.kstest(f[:200000].cpu().numpy(), "norm")
    assert ks.pvalue > 1e-3
    # same seed -> same stream
    sb2 = SB.from_xyz(xyz.clone()).manual_seed(1234)
    sb2.diffuse_xyz(beta)
    assert torch.equal(sb2.get_xyz(), e1)
    # variance schedule honoured per structure
    sb3 = SB.from_xyz(xyz.clone()).manual_seed(7)
    betas = torch.linspace(0.1, 0.9, B)
    sb3.diffuse_xyz(betas)
    v = sb3.get_xyz().reshape(B, -1).var(dim=1).cpu()
    assert torch.allclose(v, betas, rtol=0.05)


def test_k5_argument_validation_before_launch(SB):
    """Everything the sampler kernels dereference is validated on the host: a short / CPU / mistyped rng_state, a
    mis-shaped noise tensor or output buffer raises ValueError before anything is launched."""
    from protstruc_amd import ops
    B, N, A = 2, 8, 15
    xyz = torch.randn(B, N, A, 3, device="cuda")
    beta = torch.full((B,), 0.1, device="cuda")
    betas = beta[None].expand(3, B).contiguous()
    good = torch.zeros(ops.RNG_STATE_WORDS, dtype=torch.int64, device="cuda")
    before = xyz.clone()
    bad_states = [
        torch.zeros(2, dtype=torch.int64, device="cuda"),                      # the [seed, offset] pair alone
        torch.zeros(ops.RNG_STATE_WORDS, dtype=torch.int64),                   # host memory
        torch.zeros(ops.RNG_STATE_WORDS, dtype=torch.int32, device="cuda"),    # wrong word size
        torch.zeros(2 * ops.RNG_STATE_WORDS, dtype=torch.int64, device="cuda")[::2],   # strided
    ]
    for st in bad_states:
        with pytest.raises(ValueError):
            ops.diffuse_(xyz, beta, st)
        with pytest.raises(ValueError):
            ops.diffuse_frames_(xyz, beta, 0, 1, 2, 1, st)
        with pytest.raises(ValueError):
            ops.diffusion_trajectory_(xyz, betas, 0, 1, 2, 1, st)
    with pytest.raises(ValueError):
        ops.diffuse_frames_(xyz, beta, 0, 1, 2, 1, None, noise=torch.zeros(B, N, A - 1, 3, device="cuda"))
    with pytest.raises(ValueError):
        ops.diffuse_frames_(xyz, beta[:1], 0, 1, 2, 1, good)
    with pytest.raises(ValueError):
        ops.diffuse_frames_(xyz, beta, 0, 1, A, 1, good)                        # atom slot out of range
    with pytest.raises(ValueError):
        ops.diffuse_frames_(xyz, beta, 0, 1, 2, 1, good, out_rot=torch.empty(B, N, 9, device="cuda"))
    with pytest.raises(ValueError):
        ops.diffuse_frames_(xyz, beta, 0, 1, 2, 1, good, out_trans=torch.empty(B, N, 3, dtype=torch.float64, device="cuda"))
    with pytest.raises(ValueError):
        ops.diffuse_frames_(xyz, beta, 0, 1, 2, 1, good, out_rot=torch.empty(B, N, 3, 3))   # host buffer
    with pytest.raises(ValueError):
        ops.diffusion_trajectory_(xyz, betas, 0, 1, 2, 1, good, out_xyz=torch.empty(2, B, N, A, 3, device="cuda"))
    with pytest.raises(ValueError):
        ops.diffuse_(xyz, beta.cpu().to("cuda")[:1], good)
    torch.cuda.synchronize()
    assert torch.equal(xyz, before) and not good.any(), "a refused call must not have launched anything"
    ops.diffuse_(xyz, beta, good)                                             # and the good state still works
    torch.cuda.synchronize()
    assert int(good[1]) == 1 and not good[2:].any() and not torch.equal(xyz, before)


def test_k5_back_to_back_launches_draw_consecutive_offsets(SB):
    """K5's draw counter lives on the device and is advanced by the LAST workgroup of a launch to take its ticket
    (diffusion.hip: rng_take_ticket / rng_finish), which is only correct if every wave of every workgroup has read
    (seed, offset) before its workgroup takes a ticket.  Pinned here at a grid of ~23 000 workgroups (90 rounds of the chip):
    the noise of draw k is extracted by one isolated launch per k (xyz = 0, beta = 1: the update returns eps itself, offset set
    by hand), then eight back-to-back sampler launches on real coordinates must equal, bit for bit, eight launches with that
    noise injected -- i.e. launch k drew with offset k in every workgroup, none with k - 1 or k + 1 -- and leave the counter at
    8 with every ticket word back at zero.  The fused K5 + K4 step draws the same stream."""
    from protstruc_amd import ops
    B, N, A, K = 2048, 512, 15, 8
    g = torch.Generator().manual_seed(5)
    xyz0 = torch.randn(B, N, A, 3, generator=g).cuda()
    beta = (torch.rand(B, generator=g) * 0.5 + 0.01).cuda()
    one = torch.ones(B, device="cuda")
    sb = SB.from_xyz(torch.zeros(1, 2, A, 3)).manual_seed(2024)
    state0 = sb._rng_state.clone()
    eps = []
    for k in range(K):                                   # draw k in isolation
        st = state0.clone()
        st[1] = k
        z = torch.zeros(B, N, A, 3, device="cuda")
        ops.diffuse_(z, one, st)
        torch.cuda.synchronize()
        assert int(st[1]) == k + 1 and not st[2:].any()
        eps.append(z)
    assert not torch.equal(eps[0], eps[1]) and abs(eps[0].mean().item()) < 1e-3 and abs(eps[0].var().item() - 1) < 1e-3
    a, st = xyz0.clone(), state0.clone()
    for k in range(K):                                   # back to back: launch k + 1 is queued while launch k runs
        ops.diffuse_(a, beta, st)
    b = xyz0.clone()
    for k in range(K):
        ops.diffuse_(b, beta, None, eps[k])
    torch.cuda.synchronize()
    assert int(st[1]) == K and not st[2:].any(), "draw counter must advance by one per launch, tickets back to 0"
    assert torch.equal(a, b), "a launch drew with another offset than its own"
    # the fused step (another kernel, the same ticket protocol read through LDS) draws the same stream
    Bf = 64
    c, st = xyz0[:Bf].clone(), state0.clone()
    for k in range(K):
        ops.diffuse_frames_(c, beta[:Bf], 0, 1, 2, 1, st)
    d = xyz0[:Bf].clone()
    for k in range(K):
        ops.diffuse_(d, beta[:Bf], None, eps[k][:Bf].contiguous())
    torch.cuda.synchronize()
    assert int(st[1]) == K and not st[2:].any() and torch.equal(c, d)


def test_k5_graph_capture_replays_fresh_noise(SB):
    B, N = 4, 64
    sb = SB.from_xyz(torch.zeros(B, N, 15, 3)).manual_seed(99)
    beta = torch.full((B,), 0.5, device="cuda")
    sb.diffuse_xyz(beta)  # warm-up outside capture
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        sb.diffuse_xyz(beta)
        rot = sb.backbone_orientations()
    snaps = []
    for _ in range(3):
        graph.replay()
        torch.cuda.synchronize()
        snaps.append(sb.get_xyz().clone())
    assert not torch.equal(snaps[0], snaps[1]) and not torch.equal(snaps[1], snaps[2])
    assert_close(rot, O.backbone_orientations(snaps[2].cpu()))


# ----------------------------------------------------------------------------- callers
def test_inter_residue_geometry_golden(SB):
    g = load_golden("g8_inter_residue_geometry")
    geo = SB.from_xyz(g["xyz"], g["atom_mask"]).inter_residue_geometry()
    assert set(geo) == {"d_ca", "d_ca_mask", "d_cb", "d_cb_mask", "d_no", "d_no_mask", "omega", "theta", "phi"}
    for k, v in geo.items():
        if k.endswith("_mask"):
            assert torch.equal(v.cpu(), g[k])
        else:
            assert_close(v, g[k], bad_frac=1.0 / g[k].numel())


# 34, 63: below the per-CU sweep (one-column kernel); 64, 77: its shortest chains (one live column group of two); 256, 512: four columns per lane, vector stores, strip-local 16-byte mask
# stores; 208: the same with a partial last strip (13 of 16 column groups live); 200, 500: four columns, vector float stores,
# flat mask stores; 258, 100: two columns per lane (vector); 101, 129: odd, two columns, 64-floats-per-store layout; 301: the
# same with three strips; 511, 257: odd, four columns
@pytest.mark.parametrize("N", [100, 101, 258, 34, 256, 200, 208, 512, 129, 301, 257, 511, 500, 64, 77, 63])
def test_inter_residue_geometry_matches_unfused_kernels(SB, N):
    """The fused featuriser must equal the K1 slices -- in BOTH square-root modes of the device (it takes the mode K1
    takes: the hardware square root by default, the correctly rounded one after set_exact_sqrt(True)), the two modes
    within 1 ulp of each other -- and the K3 calls it replaces, bit for bit."""
    from protstruc_amd import ops
    xyz, mask = synth(77, 3, N)
    sb = SB.from_xyz(xyz, mask)
    was = ops.get_exact_sqrt()
    geos = {}
    try:
        for exact in (False, True):
            ops.set_exact_sqrt(exact)
            geos[exact] = sb.inter_residue_geometry()
            d, m = sb.pairwise_distance_matrix()
            for key, (a, c) in {"d_ca": (1, 1), "d_cb": (4, 4), "d_no": (0, 3)}.items():
                assert torch.equal(geos[exact][key], d[:, :, :, a, c]), (key, exact)
                assert torch.equal(geos[exact][key + "_mask"], m[:, :, :, a, c]), (key, exact)
    finally:
        ops.set_exact_sqrt(was)
    for key in ("d_ca", "d_cb", "d_no"):
        ulps = (geos[True][key].view(torch.int32) - geos[False][key].view(torch.int32)).abs()
        assert int(ulps.max()) <= 1
    geo = geos[bool(was)]

    def same(x, y):
        return torch.equal(x.isnan(), y.isnan()) and torch.equal(x.nan_to_num(0), y.nan_to_num(0))

    assert same(geo["omega"], sb.pairwise_dihedrals(["CA", "CB"], ["CA", "CB"]))
    assert same(geo["theta"], sb.pairwise_dihedrals(["N", "CA", "CB"], ["CB"]))
    assert same(geo["phi"], sb.pairwise_planar_angles(["CA", "CB"], ["CB"]))
    fm = SB.from_xyz(xyz, mask.float()).inter_residue_geometry()
    assert fm["d_ca_mask"].dtype == torch.float32 and torch.equal(fm["d_ca_mask"].bool(), geo["d_ca_mask"])


def _featuriser_in_sentinels(xyz, mask, shifts, exact_sqrt=0, faithful=0):
    """ps_inter_residue_geometry_f32 into planes placed `shifts` = [(6 float shifts in floats, 3 mask shifts in bytes), ...]
    away from their slots inside sentinel-filled buffers, each compared bit for bit with the one-column kernel
    (exact_angles = 2 | faithful: same arithmetic, the plain layout); no sentinel may change."""
    from protstruc_amd import _lib
    from protstruc_amd.ops import _ptr, _stream
    B, N = xyz.shape[:2]
    xg, mg = xyz.cuda(), mask.cuda().to(torch.uint8)
    plane, pad = B * N * N, 40
    lib = _lib.load()

    def run(mode, f_shift, m_shift):
        fbuf = torch.full((6 * (plane + pad) + pad,), 777.0, device="cuda")
        mbuf = torch.full((3 * (plane + pad) + pad,), 7, dtype=torch.uint8, device="cuda")
        foff = [pad + i * (plane + pad) + f_shift[i] for i in range(6)]
        moff = [pad + i * (plane + pad) + m_shift[i] for i in range(3)]
        rc = lib.ps_inter_residue_geometry_f32(_ptr(xg), _ptr(mg), *[fbuf.data_ptr() + 4 * o for o in foff],
                                               *[mbuf.data_ptr() + o for o in moff], B, N, 15, exact_sqrt, mode, _stream(xg))
        assert rc == 0
        torch.cuda.synchronize()
        fs = [fbuf[o:o + plane] for o in foff]
        ms = [mbuf[o:o + plane] for o in moff]
        keepf = torch.ones_like(fbuf, dtype=torch.bool)
        keepm = torch.ones_like(mbuf, dtype=torch.bool)
        for o in foff:
            keepf[o:o + plane] = False
        for o in moff:
            keepm[o:o + plane] = False
        assert (fbuf[keepf] == 777.0).all() and (mbuf[keepm] == 7).all(), (N, B, mode, f_shift, m_shift)
        return fs, ms

    ref_f, ref_m = run(2 | faithful, [0] * 6, [0] * 3)
    for f_shift, m_shift in shifts:
        fs, ms = run(faithful, f_shift, m_shift)
        for k in range(6):
            assert torch.equal(fs[k].isnan(), ref_f[k].isnan()) and torch.equal(fs[k].nan_to_num(0), ref_f[k].nan_to_num(0)), (N, B, k, f_shift)
        for k in range(3):
            assert torch.equal(ms[k], ref_m[k]), (N, B, k, m_shift)
            assert int(ms[k].max()) <= 1


def test_inter_residue_geometry_differential_fuzz(SB):
    """Random lengths (40 .. 700: one to three strips, every residue of the length modulo 16), batch sizes, masks and plane
    placements: the per-CU featuriser (vector / 64-floats-per-store float planes, strip-local / flat mask stores, one or
    two workgroups per CU) against the one-column kernel, bit for bit, inside sentinels."""
    import os
    # PS_FEAT_FUZZ_SEED / PS_FEAT_FUZZ_TRIALS: one-off longer runs (the committed defaults are what CI runs)
    rng = torch.Generator().manual_seed(int(os.environ.get("PS_FEAT_FUZZ_SEED", "20241004")))
    for trial in range(int(os.environ.get("PS_FEAT_FUZZ_TRIALS", "36"))):
        N = int(torch.randint(40, 701, (1,), generator=rng))
        if trial % 6 == 0:
            N = (N // 16) * 16                   # the strip-local mask form
        B = int(torch.randint(1, 5, (1,), generator=rng)) if N > 200 else int(torch.randint(1, 40, (1,), generator=rng))
        xyz, mask = synth(7000 + trial, B, N)
        keep = float(torch.rand(1, generator=rng))
        mask = torch.rand(B, N, 15, generator=rng) < keep
        if trial % 5 == 0:
            mask = None if trial % 10 == 0 else torch.ones(B, N, 15, dtype=torch.bool)
        fsh = [int(v) for v in torch.randint(0, 4, (6,), generator=rng)] if trial % 2 else [0] * 6
        msh = [int(v) for v in torch.randint(0, 16, (3,), generator=rng)] if trial % 3 else [0] * 3
        if mask is None:
            mask_t = torch.ones(B, N, 15, dtype=torch.bool)
        else:
            mask_t = mask
        _featuriser_in_sentinels(xyz, mask_t, [(fsh, msh)], exact_sqrt=trial % 2, faithful=int(trial % 3 == 1))


# ... 2048: the longest chain whose rows, column points and masks fit in LDS (158 KB); 2100: beyond it (one-column kernel)
@pytest.mark.parametrize("N,B", [(40, 9), (48, 70), (63, 5), (101, 3), (200, 2), (258, 2), (511, 2), (129, 5), (256, 2), (1030, 1), (2048, 1), (2100, 1), (2200, 1)])
def test_inter_residue_geometry_c_abi_inside_sentinels_any_alignment(SB, N, B):
    """The featuriser's per-CU sweep through the C ABI with planes the caller placed anywhere: float planes on 4-byte and
    mask planes on 1-byte boundaries (each plane its own), sentinels in front of, between and behind the planes.  Every
    plane must equal the one-column kernel's (exact_angles = 2: same arithmetic, the plain layout) bit for bit and no
    sentinel may change -- the flat 16-byte mask stores, their byte-store fringes and the 64-floats-per-store layout all
    have to land exactly on their plane."""
    xyz, mask = synth(500 + N, B, N)
    mask[0, N // 2] = False
    mask[B - 1, :, 4] = False            # a structure without CB
    _featuriser_in_sentinels(xyz, mask, [([0] * 6, [0] * 3), ([1, 2, 3, 0, 1, 2], [1, 5, 3]), ([0] * 6, [15, 8, 4])])
    # ... and in the reference's order of operations (round 5: the same sweep, the library's sequences in packed form)
    _featuriser_in_sentinels(xyz, mask, [([0] * 6, [0] * 3), ([1, 2, 3, 0, 1, 2], [1, 5, 3])], faithful=1)


def test_fused_diffuse_frames(SB):
    xyz, mask = synth(88, 5, 37)
    beta = torch.tensor([0.01, 0.2, 0.5, 0.9, 0.999])
    noise = torch.randn(5, 37, 15, 3, generator=torch.Generator().manual_seed(3))
    a = SB.from_xyz(xyz.clone(), mask)
    a.diffuse_xyz(beta, noise=noise)
    ra, ta = a.backbone_orientations_and_translations()
    b = SB.from_xyz(xyz.clone(), mask)
    rb, tb = b.diffuse_xyz_and_frames(beta, noise=noise)
    assert torch.equal(a.get_xyz(), b.get_xyz()) and torch.equal(ra, rb) and torch.equal(ta, tb)
    # sampler path: same seed -> the fused step draws exactly the noise of the unfused one
    c = SB.from_xyz(xyz.clone(), mask).manual_seed(42)
    d = SB.from_xyz(xyz.clone(), mask).manual_seed(42)
    for _ in range(3):
        c.diffuse_xyz(beta)
        rc = c.backbone_orientations()
        rd, td = d.diffuse_xyz_and_frames(beta)
        assert torch.equal(c.get_xyz(), d.get_xyz()) and torch.equal(rc, rd)
        assert torch.equal(td, d.get_xyz()[:, :, 1])
    assert_close(rd, O.backbone_orientations(d.get_xyz().cpu()), bad_frac=1e-3)


@pytest.mark.parametrize("B,N,T", [(2, 37, 5), (3, 130, 4), (1, 7, 3)])
def test_diffusion_trajectory_equals_stepwise(SB, B, N, T):
    """The LDS-resident loop kernel must equal T fused steps bit for bit (same Philox stream)."""
    xyz, mask = synth(99 + N, B, N)
    betas = torch.rand(T, B, generator=torch.Generator().manual_seed(N)) * 0.5 + 0.01
    a = SB.from_xyz(xyz.clone(), mask).manual_seed(5)
    rots, trs, xs = [], [], []
    for t in range(T):
        r, tr = a.diffuse_xyz_and_frames(betas[t])
        rots.append(r); trs.append(tr); xs.append(a.get_xyz().clone())
    b = SB.from_xyz(xyz.clone(), mask).manual_seed(5)
    rot, trans, traj = b.diffuse_trajectory(betas, want_xyz=True)
    assert torch.equal(rot, torch.stack(rots)) and torch.equal(trans, torch.stack(trs))
    assert torch.equal(traj, torch.stack(xs)) and torch.equal(b.get_xyz(), a.get_xyz())
    # the sampler state advanced by T in both: the next draw agrees as well
    a.diffuse_xyz(betas[0]); b.diffuse_xyz(betas[0])
    assert torch.equal(a.get_xyz(), b.get_xyz())


# ----------------------------------------------------------------------------- config 1 + free functions
def test_config1_from_pdb_15c8(SB):
    """BASELINE config 1: StructureBatch.from_pdb('15c8_HL.pdb').pairwise_distance_matrix() (B=1, N=229)."""
    import os
    from tests.conftest import GOLDEN_DIR
    g = load_golden("g10_config1_15c8_HL")
    sb = SB.from_pdb(os.path.join(GOLDEN_DIR, "15c8_HL.pdb"))
    assert sb.get_max_n_residues() == int(g["n_residues"]) == 229
    assert int(sb.get_atom_mask().sum()) == int(g["atom_count"])
    d, m = sb.pairwise_distance_matrix()
    assert d.shape == (1, 229, 229, 15, 15)

    def ulp_close(got, want, k=2):   # protein scale: 1 ulp at 60 A is 3.8e-6 -- gate at k ulp, NaNs identical
        got = got.cpu()
        assert torch.equal(got.isnan(), want.isnan())
        tol = k * torch.finfo(torch.float32).eps * want.abs().clamp_min(1.0)
        assert ((got - want).abs().nan_to_num(0) <= tol.nan_to_num(1)).all()

    ulp_close(d[0, :, :, 1, 1], g["ca_ca"])
    ulp_close(d[0, :, :, 4, 4], g["cb_cb"])
    assert torch.equal(m[0, :, :, 1, 1].cpu(), g["ca_ca_mask"]) and torch.equal(m[0, :, :, 4, 4].cpu(), g["cb_cb_mask"])
    bi, bj = g["block_i"].cuda(), g["block_j"].cuda()
    ulp_close(d[0, bi, bj], g["blocks"])
    assert torch.equal(m[0, bi, bj].cpu(), g["blocks_mask"])
    # reference tests/test_StructureBatch.py:43-66: two chains -> two N- and two C-termini
    assert torch.equal(sb.get_n_terminal_mask().cpu(), g["nterm"]) and int(g["nterm"].sum()) == 2
    assert torch.equal(sb.get_c_terminal_mask().cpu(), g["cterm"]) and int(g["cterm"].sum()) == 2
    dih, dmask = sb.backbone_dihedrals()
    assert_close(dih, g["dihedrals"])
    assert torch.equal(dmask.cpu(), g["dihedral_mask"])
    assert_close(sb.backbone_orientations(), g["rot"], tol=2e-5)
    multi = SB.from_pdb([os.path.join(GOLDEN_DIR, n) for n in ("15c8_HL.pdb", "1ad0_DC.pdb", "6dc4.pdb")])
    assert len(multi.get_xyz()) == 3
    assert (multi.get_n_terminal_mask().sum(1) == 2).all() and (multi.get_c_terminal_mask().sum(1) == 2).all()


def test_from_pdb_termini(SB):
    """reference tests/test_StructureBatch.py:43-66 on the GPU path: from_pdb (single, then the reference's three
    files plus its other fixtures, padded to 448 residues incl. the 7 UNK fillers of 5cjx_HL) -> two N- and two
    C-termini per structure; masks and dihedrals equal the oracle on the reader's tensors."""
    import os
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sb1 = SB.from_pdb(os.path.join(G, "1ad0_DC.pdb"))
    assert len(sb1.get_xyz()) == 1
    assert (sb1.get_n_terminal_mask().sum(axis=1) == 2).all() and (sb1.get_c_terminal_mask().sum(axis=1) == 2).all()
    names = ["15c8_HL.pdb", "1ad0_DC.pdb", "5cjx_HL.pdb", "1a3r_HL.pdb", "1a6v_HL.pdb", "1a6v_JN.pdb"]
    sb = SB.from_pdb([os.path.join(G, n) for n in names])
    assert len(sb.get_xyz()) == 6 and sb.get_max_n_residues() == 448
    nterm, cterm = sb.get_n_terminal_mask(), sb.get_c_terminal_mask()
    assert (nterm.sum(axis=1) == 2).all() and (cterm.sum(axis=1) == 2).all()
    chain, rmask = sb.chain_idx.cpu(), sb.residue_mask.cpu()
    assert torch.equal(nterm.cpu(), O.n_terminal_mask(chain, rmask)) and torch.equal(cterm.cpu(), O.c_terminal_mask(chain, rmask))
    dih, dmask = sb.backbone_dihedrals()
    want, wmask = O.backbone_dihedrals(sb.get_xyz().cpu(), chain, rmask)
    assert torch.equal(dmask.cpu(), wmask)
    valid = wmask & ~torch.isnan(want)
    assert ((dih.cpu() - want).abs()[valid] <= 1e-5).all()
    assert torch.equal(torch.isnan(dih.cpu()), torch.isnan(want))


def test_geometry_free_functions(SB):
    import protstruc_amd.geometry as geom
    g = load_golden("g9_primitives")
    a, b, c, d, c60 = g["a"], g["b"], g["c"], g["d"], g["c60"]
    # analytic answers of the reference's tests/test_geometry.py:35-190, tensor and numpy flavours
    assert torch.allclose(geom.angle(a, b, c, to_degree=True).cpu(), torch.tensor([90.0]))
    assert torch.allclose(geom.angle(a, b, c60, to_degree=True).cpu(), torch.tensor([60.0]), atol=1e-4)
    assert torch.allclose(geom.dihedral(a, b, c, d, to_degree=True).cpu(), torch.tensor([-90.0]))
    out = geom.dihedral(a.numpy(), b.numpy(), c.numpy(), d.numpy(), to_degree=True)
    assert isinstance(out, np.ndarray) and out.shape == (1,) and np.allclose(out, [-90.0])
    assert isinstance(geom.angle(a.numpy().astype(np.float64), b.numpy(), c.numpy()), np.ndarray)
    assert geom.dihedral(a[None], b[None], c[None], d[None]).shape == (1, 1)
    assert geom.dot(torch.tensor([1.0, 2.0, 3.0]), torch.tensor([4.0, 5.0, 6.0])).item() == 32.0
    assert abs(geom.norm(np.array([1.0, 2.0, 3.0])).item() - 14 ** 0.5) < 1e-6
    P = g["P"]
    assert_close(geom.angle(P[0], P[1], P[2]), g["rnd_angle"])
    assert_close(geom.dihedral(P[0], P[1], P[2], P[3]), g["rnd_dihedral"])
    assert_close(geom.dot(P[0], P[1]), g["rnd_dot"])
    assert_close(geom.norm(P[0]), g["rnd_norm"])
    assert_close(geom.unit(P[0]), g["rnd_unit"])
    assert_close(geom.gram_schmidt(P[0], P[1], P[2]), g["rnd_frame"])
    frame = geom.gram_schmidt(torch.randn(16, 30, 3), torch.randn(16, 30, 3), torch.randn(16, 30, 3))
    assert frame.shape == (16, 30, 3, 3)
    ideal = load_golden("g5_frames")["ideal_xyz"]
    eye = geom.gram_schmidt(ideal[:, :, 0], ideal[:, :, 1], ideal[:, :, 2]).cpu()
    assert (eye == torch.eye(3).expand(2, 10, -1, -1)).all()


def test_rigid_body_ops_golden(SB):
    g = load_golden("g11_rigid_ops")
    xyz, mask = g["xyz"], g["atom_mask"]
    T = 1e-5 * 3   # coordinates at 5-15 A scale: one fp32 ulp is ~1e-6, matrix products round differently

    def fresh():
        return SB.from_xyz(xyz.clone(), mask)

    sb = fresh(); sb.translate(g["t_res"]); assert_close(sb.get_xyz(), g["translate_res"], tol=T)
    sb = fresh(); sb.translate(g["t_b"]); assert_close(sb.get_xyz(), g["translate_b"], tol=T)
    sb = fresh(); sb.translate(g["t_atom"], atomwise=True); assert_close(sb.get_xyz(), g["translate_atom"], tol=T)
    sb = fresh(); sb.rotate(g["R_b"]); assert_close(sb.get_xyz(), g["rotate_b"], tol=T)
    sb = fresh(); sb.rotate(g["R_b"][0]); assert_close(sb.get_xyz(), g["rotate_shared"], tol=T)
    assert_close(fresh().center_of_mass(), g["com"], tol=T)
    sb = SB.from_xyz(xyz[:1].clone(), mask[:1]); sb.center_at(); assert_close(sb.get_xyz(), g["center_origin_b1"], tol=T)
    sb = fresh(); sb.center_at(g["centers"]); assert_close(sb.get_xyz(), g["center_b"], tol=T)
    # reference tests/test_StructureBatch.py:258-275: the CA centre lands on the requested point
    assert torch.allclose(sb.center_of_mass().cpu(), g["centers"], rtol=1e-4, atol=1e-5)
    sb = fresh(); sb.center_at(g["centers"][0]); assert_close(sb.get_xyz(), g["center_shared"], tol=T)
    sb = fresh(); sb.center_at(); assert torch.allclose(sb.center_of_mass().cpu(), torch.zeros(4, 3), atol=1e-5)
    with pytest.raises(ValueError):
        fresh().center_at(torch.zeros(2, 3, 3))
    sb2 = SB.from_xyz(g["xyz2"])
    assert_close(sb2.get_local_xyz(), g["local_xyz"], tol=T, bad_frac=1e-3)
    rot, tr = sb2.backbone_orientations(), sb2.backbone_translations()
    for cb in (0, 1):
        sb3 = SB.from_backbone_orientations_translations(rot, tr, include_cb=bool(cb))
        assert sb3.get_max_n_atoms_per_residue() == 15
        assert_close(sb3.get_xyz(), g[f"bb_xyz_cb{cb}"], tol=T, bad_frac=1e-3)
        assert sb3.get_atom_mask().dtype == torch.float32 and torch.equal(sb3.get_atom_mask().cpu(), g[f"bb_mask_cb{cb}"])
        # round trip: frames of the rebuilt backbone are the frames it was built from
        assert_close(sb3.backbone_orientations(), rot.cpu(), tol=T, bad_frac=1e-2)
    import protstruc_amd.geometry as geom
    assert torch.equal(geom.ideal_backbone_coordinates((), True), g["ideal4"])
    assert geom.ideal_backbone_coordinates((16, 30)).shape == (16, 30, 3, 3)


def test_align_topk_select_golden(SB):
    g = load_golden("g12_align_topk")
    xyz, mask, tgt, tmask = g["xyz"], g["atom_mask"], g["target_xyz"], g["target_mask"]
    T = 5e-5   # coordinates at ~10 A scale, rotation from an SVD: a few fp32 ulps
    sb = SB.from_xyz(xyz.clone(), mask)
    R = sb.align(SB.from_xyz(tgt, tmask))
    assert_close(sb.get_xyz(), g["aligned_default_mask"], tol=T)
    assert R.shape == (2, 3, 3)
    eye = torch.eye(3, device=R.device).expand(2, 3, 3)
    assert (R @ R.transpose(-1, -2) - eye).abs().max() < 1e-5 and torch.allclose(torch.linalg.det(R), torch.ones(2, device=R.device), atol=1e-5)
    sb = SB.from_xyz(xyz.clone(), mask)
    sb.align(SB.from_xyz(tgt, tmask), atom_mask=g["ca_sel"])
    assert_close(sb.get_xyz(), g["aligned_ca_only"], tol=T)
    from protstruc_amd import ops
    full = torch.ones(1, 24, 15, dtype=torch.bool)
    R1, t1 = ops.kabsch(xyz[:1].cuda(), tgt[:1].cuda(), full)
    assert_close(R1[0], g["kabsch_R"], tol=5e-6)
    assert_close(t1[0], g["kabsch_t"], tol=T)
    # a single-structure target serves the whole batch; aligning a rotated copy onto the original recovers it
    q = torch.linalg.qr(torch.randn(3, 3, generator=torch.Generator().manual_seed(1)))[0]
    q = q * torch.sign(torch.linalg.det(q))
    moved = torch.einsum("ij,bnaj->bnai", q, xyz) + torch.tensor([3.0, -2.0, 7.0])
    sb = SB.from_xyz(moved, mask)
    sb.align(SB.from_xyz(xyz[:1], mask[:1]), atom_mask=torch.ones(1, 24, 15, dtype=torch.bool))
    assert (sb.get_xyz()[0].cpu() - xyz[0]).abs().max() < 1e-4
    with pytest.raises(ValueError, match="Batch size"):
        SB.from_xyz(xyz, mask).align(SB.from_xyz(torch.cat([tgt, tgt[:1]]), torch.cat([tmask, tmask[:1]])))
    one = SB.from_xyz(xyz[:1], mask[:1], chain_idx=torch.zeros(1, 24), chain_ids=[["A"]])
    assert torch.equal(one.get_topk_nearest_residue_mask(g["query"], k=5).cpu(), g["topk5"])
    assert torch.equal(one.get_topk_nearest_residue_mask(g["query"], k=8, mask=g["topk_user_mask"]).cpu(), g["topk_masked"])
    assert torch.equal(one.get_topk_nearest_residue_mask(g["query"]).cpu(), g["topk_all"])
    with pytest.raises(ValueError, match="batch size > 1"):
        SB.from_xyz(xyz, mask).get_topk_nearest_residue_mask(g["query"])
    sel = one.residue_masked_select(g["pick"])
    assert torch.equal(sel.get_xyz().cpu(), g["picked_xyz"]) and torch.equal(sel.get_atom_mask().cpu(), g["picked_mask"])
    with pytest.raises(ValueError, match="boolean"):
        one.residue_masked_select(g["pick"].float())


def test_input_flavours(SB):
    """numpy / float64 / non-contiguous / integer-mask / already-on-GPU inputs all give the same answer."""
    xyz, mask = synth(321, 2, 20)
    base_d, base_m = SB.from_xyz(xyz, mask).pairwise_distance_matrix()
    flavours = {
        "numpy64": (xyz.double().numpy(), mask.numpy()),
        "gpu": (xyz.cuda(), mask.cuda()),
        "noncontig": (xyz.permute(0, 1, 3, 2).contiguous().permute(0, 1, 3, 2), mask),
        "longmask": (xyz, mask.long()),
        "floatmask": (xyz, mask.float()),
    }
    for name, (x, m) in flavours.items():
        sb = SB.from_xyz(x, m)
        d, dm = sb.pairwise_distance_matrix()
        assert torch.equal(d, base_d), name
        assert torch.equal(dm.bool(), base_m), name
        if name == "longmask":
            assert dm.dtype == torch.long
        dih, _ = sb.backbone_dihedrals()
        assert dih.dtype == torch.float32 and dih.is_cuda
    # chain indices given as float64 numpy
    ci = np.zeros((2, 20)); ci[:, 10:] = 1
    sb = SB.from_xyz(xyz.numpy(), mask.numpy(), chain_idx=ci, chain_ids=[["A", "B"]] * 2)
    assert int(sb.get_n_terminal_mask().sum()) == 4
    # _pairwise_xyz keeps the reference's (B, N*N, n, 3) layout: row p <-> (i = p // N, j = p % N)
    pw = SB.from_xyz(xyz, mask)._pairwise_xyz(["CA", "CB"], ["N"])
    assert pw.shape == (2, 400, 3, 3)
    assert torch.equal(pw[:, 3 * 20 + 7, 0].cpu(), xyz[:, 3, 1]) and torch.equal(pw[:, 3 * 20 + 7, 2].cpu(), xyz[:, 7, 0])
    import protstruc_amd.geometry as geom
    g = load_golden("g12_align_topk")
    R, t = geom.kabsch(g["xyz"][0].reshape(-1, 3), g["target_xyz"][0].reshape(-1, 3))
    assert_close(R, g["kabsch_R"], tol=5e-6)
    assert_close(t, g["kabsch_t"], tol=5e-5)


@pytest.mark.parametrize("B,N", [(0, 8), (2, 0), (0, 0)])
def test_empty_batches(SB, B, N):
    """Empty inputs (no structures, or structures without residues) go through every featuriser: empty outputs of
    the reference's shapes and dtypes, nothing launched, nothing raised."""
    A = 15
    xyz = torch.zeros(B, N, A, 3)
    mask = torch.zeros(B, N, A, dtype=torch.bool)
    sb = SB.from_xyz(xyz, mask)
    d, m = sb.pairwise_distance_matrix()
    assert d.shape == (B, N, N, A, A) and d.dtype == torch.float32 and m.shape == d.shape and m.dtype == torch.bool
    dih, dm = sb.backbone_dihedrals()
    assert dih.shape == (B, N, 3) and dm.shape == (B, N, 3) and dm.dtype == torch.bool
    assert sb.get_n_terminal_mask().shape == (B, N) and sb.get_c_terminal_mask().shape == (B, N)
    assert sb.backbone_orientations().shape == (B, N, 3, 3) and sb.backbone_translations().shape == (B, N, 3)
    assert sb.pairwise_dihedrals(["CA", "CB"], ["CA", "CB"]).shape == (B, N, N)
    assert sb.pairwise_planar_angles(["CA", "CB"], ["CB"]).shape == (B, N, N)
    geo = sb.inter_residue_geometry()
    assert all(v.shape == (B, N, N) for v in geo.values()) and len(geo) == 9
    sb.manual_seed(3).diffuse_xyz(torch.full((B,), 0.1))
    rot, tr = sb.diffuse_xyz_and_frames(torch.full((B,), 0.1))
    assert rot.shape == (B, N, 3, 3) and tr.shape == (B, N, 3)
    r3, t3, x3 = sb.diffuse_trajectory(torch.full((4, B), 0.1), want_xyz=True)
    assert r3.shape == (4, B, N, 3, 3) and t3.shape == (4, B, N, 3) and x3.shape == (4, B, N, A, 3)
    sb.standardize()
    assert sb.mu.shape == (B, 3) and sb.std.shape == (B, 3) and sb.mu.isnan().all()
    sb.unstandardize()
    assert sb.center_of_mass().shape == (B, 3)
    torch.cuda.synchronize()
    assert sb.get_xyz().shape == (B, N, A, 3)


def test_very_large_batch_of_short_structures(SB):
    """70 000 peptides of 17 residues: the flat K1 kernels run on 1-D grids and take any batch size (pairs per launch
    < 2^32), and so do K2 / K3 / K4."""
    B, N, A = 70000, 17, 15
    g = torch.Generator().manual_seed(8)
    xyz = torch.randn(B, N, A, 3, generator=g)
    mask = torch.rand(B, N, A, generator=g) < 0.9
    sb = SB.from_xyz(xyz, mask)
    d, m = sb.pairwise_distance_matrix()
    assert d.shape == (B, N, N, A, A)
    for b in (0, 65535, 65536, B - 1):
        rd, rm = O.pairwise_distance_matrix(xyz[b:b + 1], mask[b:b + 1])
        assert_close(d[b:b + 1], rd)
        assert torch.equal(m[b:b + 1].cpu(), rm)
    assert int(m.view(torch.uint8).max()) <= 1 and not torch.isnan(d).any()
    del d, m
    # K2 / K3 / K4 run on 1-D grids: no batch limit either
    chain = torch.zeros(B, N)
    dih, dmask = sb.backbone_dihedrals()
    om = sb.pairwise_dihedrals(["CA", "CB"], ["CA", "CB"])
    ph = sb.pairwise_planar_angles(["CA", "CB"], ["CB"])
    for b in (0, 65535, 65536, B - 1):
        rdih, rdm = O.backbone_dihedrals(xyz[b:b + 1], chain[b:b + 1], mask[b:b + 1].any(-1))
        assert_close(dih[b:b + 1], rdih, bad_frac=0.02)
        assert torch.equal(dmask[b:b + 1].cpu(), rdm)
        assert_close(om[b:b + 1], O.pairwise_dihedrals(xyz[b:b + 1], [1, 4], [1, 4]), bad_frac=0.01)
        assert_close(ph[b:b + 1], O.pairwise_planar_angles(xyz[b:b + 1], [1, 4], [4]), bad_frac=0.01)
    assert sb.backbone_orientations().shape == (B, N, 3, 3)
    # N = 16: the pattern kernel (N % 16 == 0), also on a 1-D grid
    x16, m16 = xyz[:, :16].contiguous(), mask[:, :16].contiguous()
    d16, k16 = SB.from_xyz(x16, m16).pairwise_distance_matrix()
    for b in (0, 65535, 65536, B - 1):
        rd, rm = O.pairwise_distance_matrix(x16[b:b + 1], m16[b:b + 1])
        assert_close(d16[b:b + 1], rd)
        assert torch.equal(k16[b:b + 1].cpu(), rm)


def test_cpu_batch_raises_instead_of_falling_back(SB):
    xyz, mask = synth(1, 1, 4)
    sb = SB.from_xyz(xyz, mask, device="cpu")
    with pytest.raises(RuntimeError, match="HIP-only"):
        sb.pairwise_distance_matrix()
