"""Host-side PDB reader -- CPU tests.  Numeric parity with biotite is unpinned (biotite is
not installed anywhere this runs); the pins are the residue counts the reference's tests / survey state."""
import os

import pytest
import torch

from protstruc_amd import StructureBatch
from protstruc_amd.pdb import ATOM_SLOT, PDB, read_batch

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name,n_res,chains,n_unk", [
    ("15c8_HL.pdb", 229, ["L", "H"], 0),      # reference tests/test_geometry.py:214 (L = 229)
    ("6dc4.pdb", 437, ["H", "L"], 0),         # reference tests/test_AntibodyStructureBatch.py:13
    ("1ad0_DC.pdb", 434, ["C", "D"], 1),      # one numbering gap -> one UNK filler
])
def test_residue_counts(name, n_res, chains, n_unk):
    p = PDB.read_pdb(os.path.join(G, name))
    assert p.n_residues == n_res and p.get_chain_ids() == chains and p.name_of.count("UNK") == n_unk
    xyz, mask = p.get_atom_xyz()
    assert xyz.shape == (n_res, 15, 3) and mask.shape == (n_res, 15) and mask.dtype == torch.bool
    assert torch.equal(torch.isnan(xyz).any(-1), ~mask)          # NaN exactly where the atom is missing
    assert len(p.get_seq()) == n_res and sum(len(v) for v in p.get_seq_dict().values()) == n_res
    unk = [k for k, n in enumerate(p.name_of) if n == "UNK"]
    assert not mask[unk].any()


def test_first_atoms_of_15c8():
    p = PDB.read_pdb(os.path.join(G, "15c8_HL.pdb"))
    xyz, mask = p.get_atom_xyz()
    # first ATOM records of the file: ASP L 1  N (55.694, 23.422, 29.042), CA, C, O, CB
    assert torch.allclose(xyz[0, 0], torch.tensor([55.694, 23.422, 29.042]))
    assert torch.allclose(xyz[0, 4], torch.tensor([56.671, 21.734, 30.535]))
    assert p.name_of[0] == "ASP" and p.get_seq()[0] == "D" and mask[0, :8].all() and not mask[0, 8:14].any()
    assert ATOM_SLOT["TRP"]["CH2"] == 13 and ATOM_SLOT["GLY"].get("CB") is None and ATOM_SLOT["ALA"]["OXT"] == 14


def test_batch_padding_and_from_pdb():
    paths = [os.path.join(G, n) for n in ("15c8_HL.pdb", "1ad0_DC.pdb")]
    xyz, mask, chain_idx, chain_ids, seq, residue_idx = read_batch(paths)
    assert xyz.shape == (2, 434, 15, 3) and chain_ids == [["L", "H"], ["C", "D"]]
    assert (xyz[0, 229:] == 0).all() and not mask[0, 229:].any()          # zero / False padding
    assert torch.isnan(chain_idx[0, 229:]).all() and torch.isnan(residue_idx[0, 229:]).all()
    assert chain_idx[0, 0] == 0 and chain_idx[0, 228] == 1
    sb = StructureBatch.from_pdb(paths, device="cpu")
    assert sb.get_xyz().shape == (2, 434, 15, 3) and sb.get_chain_ids() == chain_ids
    assert sb.get_seq()[0]["L"].startswith("D")
    sb1 = StructureBatch.from_pdb(paths[0], device="cpu")
    assert sb1.get_batch_size() == 1 and sb1.get_max_n_residues() == 229


def test_altloc_hetatm_and_models(tmp_path):
    rec = "{:<6}{:>5} {:<4}{:1}{:>3} {:1}{:>4}{:1}   {:>8.3f}{:>8.3f}{:>8.3f}  1.00  0.00\n"
    lines = [
        "MODEL        1\n",
        rec.format("ATOM", 1, " N", " ", "GLY", "A", 1, " ", 0, 0, 0),
        rec.format("ATOM", 2, " CA", "A", "GLY", "A", 1, " ", 1, 0, 0),
        rec.format("ATOM", 3, " CA", "B", "GLY", "A", 1, " ", 9, 9, 9),     # second altloc: dropped
        rec.format("ATOM", 4, " H", " ", "GLY", "A", 1, " ", 2, 2, 2),      # hydrogen: dropped
        rec.format("HETATM", 5, " N", " ", "MSE", "A", 2, " ", 3, 0, 0),    # selenomethionine -> MET, kept
        rec.format("HETATM", 6, " O", " ", "HOH", "A", 3, " ", 4, 0, 0),    # water: dropped
        rec.format("ATOM", 7, " N", " ", "ALA", "A", 5, " ", 5, 0, 0),      # numbering gap 3,4 -> two UNK
        rec.format("ATOM", 8, " N", " ", "ALA", "A", 5, "A", 6, 0, 0),      # insertion code: its own residue
        rec.format("ATOM", 9, " N", " ", "SER", "B", 1, " ", 7, 0, 0),
        "ENDMDL\n", "MODEL        2\n",
        rec.format("ATOM", 10, " N", " ", "SER", "C", 1, " ", 8, 0, 0),     # model 2: ignored
        "ENDMDL\n",
    ]
    f = tmp_path / "toy.pdb"
    f.write_text("".join(lines))
    p = PDB.read_pdb(str(f))
    assert p.name_of == ["GLY", "MET", "UNK", "UNK", "ALA", "ALA", "SER"]
    assert p.get_chain_ids() == ["A", "B"] and p.chain_idx == [0, 0, 0, 0, 0, 0, 1]
    xyz, mask = p.get_atom_xyz()
    assert xyz[0, 1, 0] == 1.0 and mask[0].sum() == 2 and mask[1, 0] and not mask[2].any()
    assert p.get_seq() == "GMXXAAS"


def test_seq_idx_from_pdb():
    paths = [os.path.join(G, n) for n in ("15c8_HL.pdb", "1ad0_DC.pdb")]
    sb = StructureBatch.from_pdb(paths, device="cpu")
    idx = sb.get_seq_idx()
    assert idx.shape == (2, 434) and idx.dtype == torch.long
    assert idx[0, 0] == 2                      # ASP -> D -> 2 (reference general.py:126-133)
    assert (idx[0, 229:] == 20).all()          # padding is UNK
    assert idx.max() <= 20 and idx.min() >= 0
