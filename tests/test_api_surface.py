"""Drop-in surface: every public method of the reference's StructureBatch and every hot-path free function of
protstruc.geometry exists here (names listed as data from protstruc/protstruc.py and protstruc/geometry.py)."""
import inspect

import pytest

from protstruc_amd import StructureBatch
import protstruc_amd.geometry as geom

REFERENCE_STRUCTUREBATCH_METHODS = [
    "from_xyz", "from_pdb", "from_pdb_id", "from_backbone_orientations_translations", "from_dihedrals",
    "get_batch_size", "get_xyz", "get_local_xyz", "get_atom_mask", "get_residue_mask", "get_chain_idx", "get_chain_ids",
    "get_seq", "get_seq_idx", "get_total_lengths", "get_max_n_residues", "get_max_n_atoms_per_residue",
    "get_n_terminal_mask", "get_c_terminal_mask", "pairwise_distance_matrix", "backbone_dihedrals",
    "backbone_orientations", "backbone_translations", "_pairwise_xyz", "pairwise_dihedrals", "pairwise_planar_angles",
    "translate", "rotate", "standardize", "unstandardize", "center_of_mass", "center_at", "inter_residue_geometry",
    "get_topk_nearest_residue_mask", "diffuse_xyz", "align", "residue_masked_select",
]
# signatures of the hot-path methods (argument names and defaults) as in the reference
REFERENCE_SIGNATURES = {
    "pairwise_distance_matrix": "(self)",
    "backbone_dihedrals": "(self)",
    "backbone_orientations": "(self, a1='N', a2='CA', a3='C')",
    "backbone_translations": "(self, atom='CA')",
    "pairwise_dihedrals": "(self, atoms_i, atoms_j)",
    "pairwise_planar_angles": "(self, atoms_i, atoms_j)",
    "standardize": "(self, atom_mask=None, residue_mask=None)",
    "unstandardize": "(self)",
    "translate": "(self, translation, atomwise=False)",
    "rotate": "(self, rotation)",
    "center_at": "(self, center=None)",
    "inter_residue_geometry": "(self)",
    "get_n_terminal_mask": "(self)",
    "get_c_terminal_mask": "(self)",
}
REFERENCE_GEOMETRY_HOT_PATH = ["dot", "norm", "unit", "angle", "dihedral", "gram_schmidt", "ideal_backbone_coordinates", "kabsch"]


def _sig(fn):
    ps = []
    for p in inspect.signature(fn).parameters.values():
        ps.append(p.name if p.default is inspect._empty else f"{p.name}={p.default!r}")
    return "(" + ", ".join(ps) + ")"


@pytest.mark.parametrize("name", REFERENCE_STRUCTUREBATCH_METHODS)
def test_method_exists(name):
    assert callable(getattr(StructureBatch, name))


@pytest.mark.parametrize("name,sig", REFERENCE_SIGNATURES.items())
def test_hot_path_signatures(name, sig):
    assert _sig(getattr(StructureBatch, name)) == sig


def test_constructor_signature():
    params = list(inspect.signature(StructureBatch.__init__).parameters)
    assert params[:7] == ["self", "xyz", "atom_mask", "chain_idx", "chain_ids", "seq", "residue_idx"]
    assert list(inspect.signature(StructureBatch.diffuse_xyz).parameters)[:2] == ["self", "beta"]


@pytest.mark.parametrize("name", REFERENCE_GEOMETRY_HOT_PATH)
def test_geometry_function_exists(name):
    assert callable(getattr(geom, name))


def test_out_of_scope_constructors_say_why():
    with pytest.raises(NotImplementedError, match="from_pdb"):
        StructureBatch.from_pdb_id("1REX")
    with pytest.raises(NotImplementedError, match="TODO"):
        StructureBatch.from_dihedrals(None)
