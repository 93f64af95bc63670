"""Host-side behaviour of the StructureBatch shell (no GPU): constructor
validation, error conventions of the reference, and the absence of
any CPU fallback."""
import numpy as np
import pytest
import torch

from protstruc_amd import ATOM, MAX_N_ATOMS_PER_RESIDUE, StructureBatch


def make(B=2, N=10, A=15, **kw):
    g = torch.Generator().manual_seed(0)
    xyz = torch.randn(B, N, A, 3, generator=g)
    mask = torch.rand(B, N, A, generator=g) < 0.8
    return StructureBatch.from_xyz(xyz, mask, device="cpu", **kw), xyz, mask


def test_atom_enum_matches_reference_slots():
    # reference general.py:4-20
    assert (ATOM.N, ATOM.CA, ATOM.C, ATOM.O, ATOM.CB) == (0, 1, 2, 3, 4)
    assert ATOM["ca"] == ATOM["Ca"] == ATOM["CA"] == 1 and ATOM["cb"] == 4 and ATOM["o"] == 3
    assert ATOM.is_valid("ca") and ATOM.is_valid("CB") and not ATOM.is_valid("CG")
    with pytest.raises(KeyError):
        ATOM["CX"]
    assert MAX_N_ATOMS_PER_RESIDUE == 15


def test_from_xyz_shapes_and_defaults():
    sb, xyz, mask = make()
    assert sb.get_batch_size() == 2 and sb.get_max_n_residues() == 10 and sb.get_max_n_atoms_per_residue() == 15
    assert sb.get_xyz().dtype == torch.float32 and sb.get_xyz().is_contiguous()
    assert torch.equal(sb.residue_mask, mask.any(-1))
    assert sb.chain_idx.shape == (2, 10) and sb.chain_idx.dtype == torch.float32 and (sb.chain_idx == 0).all()
    assert torch.equal(sb.get_residue_mask(), mask[:, :, 1])          # Q5: the getter is the CA slot
    assert sb.get_chain_idx().dtype == torch.long
    assert sb._standardized is False
    # numpy input, no mask, A = 25 as in the reference's own tests (tests/test_StructureBatch.py:11-21)
    sb2 = StructureBatch.from_xyz(np.random.rand(3, 7, 25, 3), device="cpu")
    assert sb2.get_xyz().shape == (3, 7, 25, 3) and sb2.get_xyz().dtype == torch.float32
    assert sb2.residue_mask.all() and sb2.get_atom_mask() is None


def test_chain_arguments_validation():
    xyz = torch.randn(1, 6, 15, 3)
    chain_idx = torch.tensor([[0., 0., 0., 1., 1., 1.]])
    with pytest.raises(ValueError, match="Both `chain_idx` and `chain_ids`"):
        StructureBatch.from_xyz(xyz, chain_idx=chain_idx, device="cpu")
    with pytest.raises(ValueError, match="Both `chain_idx` and `chain_ids`"):
        StructureBatch.from_xyz(xyz, chain_ids=[["A", "B"]], device="cpu")
    with pytest.raises(AssertionError, match="Chain index should start from zero"):
        StructureBatch.from_xyz(xyz, chain_idx=chain_idx + 1, chain_ids=[["A", "B"]], device="cpu")
    padded = torch.tensor([[0., 0., 1., 1., float("nan"), float("nan")]])
    sb = StructureBatch.from_xyz(xyz, chain_idx=padded, chain_ids=[["A", "B"]], device="cpu")
    assert torch.isnan(sb.chain_idx[0, -1])


def test_errors_raised_before_any_launch():
    sb, xyz, mask = make()
    with pytest.raises(ValueError, match="Atom CG is not valid."):
        sb.pairwise_dihedrals(["CA", "CG"], ["CA", "CB"])
    with pytest.raises(KeyError):
        sb.backbone_orientations("N", "CA", "QQ")
    with pytest.raises(KeyError):
        sb.backbone_translations("QQ")
    with pytest.raises(ValueError, match="Only one of atom_mask and residue_mask"):
        sb.standardize(atom_mask=mask, residue_mask=mask.any(-1))
    with pytest.raises(ValueError, match="Cannot unstandardize"):
        sb.unstandardize()
    sb._standardized = True
    with pytest.raises(ValueError, match="already standardized"):
        sb.standardize()


def test_backbone_translations_is_a_view():
    sb, xyz, _ = make()
    t = sb.backbone_translations("N")
    assert torch.equal(t, xyz[:, :, 0]) and t.data_ptr() == sb.get_xyz()[:, :, 0].data_ptr()


@pytest.mark.parametrize("call", [
    lambda sb: sb.pairwise_distance_matrix(),
    lambda sb: sb.backbone_dihedrals(),
    lambda sb: sb.get_n_terminal_mask(),
    lambda sb: sb.backbone_orientations(),
    lambda sb: sb.pairwise_dihedrals(["CA", "CB"], ["CA", "CB"]),
    lambda sb: sb.pairwise_planar_angles(["CA", "CB"], ["CB"]),
    lambda sb: sb.standardize(),
    lambda sb: sb.diffuse_xyz(torch.full((2,), 0.1)),
    lambda sb: sb.inter_residue_geometry(),
])
def test_no_cpu_fallback(call):
    """A CPU-resident batch must raise, never compute on the host."""
    sb, _, _ = make()
    with pytest.raises(RuntimeError, match="HIP-only"):
        call(sb)
