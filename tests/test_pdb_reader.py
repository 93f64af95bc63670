"""Host-side PDB reader -- CPU tests.  Numeric parity with biotite is unpinned (biotite is
not installed anywhere this runs); the pins are the residue counts the reference's tests / survey state."""
import math
import os

import numpy as np
import pytest
import torch

from protstruc_amd import StructureBatch
from protstruc_amd.pdb import ATOM_SLOT, PDB, read_batch

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name,n_res,chains,n_unk", [
    ("15c8_HL.pdb", 229, ["L", "H"], 0),      # reference tests/test_geometry.py:214 (L = 229)
    ("6dc4.pdb", 437, ["H", "L"], 0),         # reference tests/test_AntibodyStructureBatch.py:13
    ("1ad0_DC.pdb", 434, ["C", "D"], 1),      # one numbering gap -> one UNK filler
    ("5cjx_HL.pdb", 448, ["H", "L"], 7),      # the hardest fixture: a 1-residue and a 6-residue gap in chain H
    ("1a3r_HL.pdb", 232, ["L", "H"], 0),
    ("1a6v_HL.pdb", 229, ["L", "H"], 1),
    ("1a6v_JN.pdb", 230, ["L", "H"], 1),
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


def test_gap_fill_and_insertion_codes_of_5cjx():
    """reference pdb.py:82-130 on its hardest fixture: residues are keyed by (chain, number, insertion code);
    a numbering gap inside a chain is filled with atom-less UNK residues numbered consecutively; residues that share
    a number and differ by insertion code (antibody CDR numbering) are separate residues and never open a gap."""
    p = PDB.read_pdb(os.path.join(G, "5cjx_HL.pdb"))
    unk = [(p.chain_of[k], p.number_of[k]) for k, n in enumerate(p.name_of) if n == "UNK"]
    assert unk == [("H", 53)] + [("H", n) for n in range(129, 135)]
    assert p.get_seq_dict()["H"].count("X") == 7 and "X" not in p.get_seq_dict()["L"]
    assert {c: len(s) for c, s in p.get_seq_dict().items()} == {"H": 234, "L": 214}
    assert sum(1 for ins in p.insertion_of if ins) == 21
    # every (chain, number, insertion) key resolves to one residue index and the indices tile [0, n)
    assert sorted(p.cri2idx.values()) == list(range(p.n_residues))
    # an insertion-coded run: same number, consecutive indices, all with atoms
    k = next(i for i, ins in enumerate(p.insertion_of) if ins)
    assert p.number_of[k] == p.number_of[k - 1] and p.chain_of[k] == p.chain_of[k - 1]
    xyz, mask = p.get_atom_xyz()
    assert mask[k].any() and mask[k - 1].any()
    # the fillers carry their chain's index (so they do not create chain termini) and no coordinates
    filler = [i for i, n in enumerate(p.name_of) if n == "UNK"]
    assert all(p.chain_idx[i] == 0 for i in filler) and not mask[filler].any() and torch.isnan(xyz[filler]).all()


def test_two_termini_per_structure_like_the_reference_test():
    """reference tests/test_StructureBatch.py:43-66 (from_pdb single + multiple): two chains per file -> exactly two
    N-termini and two C-termini per structure, whatever the padding and the UNK fillers.  Terminus masks here come
    from the CPU oracle on the reader's tensors (the GPU twin is tests/test_gpu_parity.py::test_from_pdb_termini)."""
    from oracle import protstruc_oracle as O
    names = ["15c8_HL.pdb", "1ad0_DC.pdb", "5cjx_HL.pdb", "1a3r_HL.pdb", "1a6v_HL.pdb", "1a6v_JN.pdb", "6dc4.pdb"]
    xyz, mask, chain_idx, chain_ids, seq, residue_idx = read_batch([os.path.join(G, n) for n in names])
    assert xyz.shape == (7, 448, 15, 3) and all(len(c) == 2 for c in chain_ids)
    rmask = mask.any(-1)
    nterm, cterm = O.n_terminal_mask(chain_idx, rmask), O.c_terminal_mask(chain_idx, rmask)
    assert nterm.dtype == torch.bool and (nterm.sum(1) == 2).all() and (cterm.sum(1) == 2).all()
    for b, name in enumerate(names):
        n = PDB.read_pdb(os.path.join(G, name)).n_residues
        first_of_second_chain = int((chain_idx[b, :n] == 1).nonzero()[0])
        assert nterm[b].nonzero().flatten().tolist() == [0, first_of_second_chain]
        assert cterm[b].nonzero().flatten().tolist() == [first_of_second_chain - 1, n - 1]


# ------------------------------------------------------------------------------------------------------------------
# Round 4: every pin the reference holds for the reader.  The tutorial entries under the reference's docs/tutorials/
# (1REX, 4EOT, 4uuj; committed here as data under tests/golden/) are what its network tests and notebooks parse, and
# those state what comes out: tests/test_StructureBatch.py:123-128 (1REX -> 130 residues), :158-163 ([130, 184]),
# docs/tutorials/ramachandran_plot.ipynb cells 3-8 ((2, 184, 3); 128 / 180 residues valid for phi / psi, equal to
# biotite's own count), pairwise_distance_matrix.ipynb cells 3-5, k_nearest_residues.ipynb cell 4 (4uuj -> 545).
# The GPU twins of these are in tests/test_gpu_reference_suite.py.
TUTORIAL = [os.path.join(G, n) for n in ("1REX.pdb", "4EOT.pdb")]


def test_tutorial_entries_counts_the_reference_states():
    xyz, mask, chain_idx, chain_ids, seq, residue_idx = read_batch(TUTORIAL)
    assert xyz.shape == (2, 184, 15, 3)                                  # ramachandran_plot.ipynb cell 3: (2, 184, 3)
    assert chain_ids == [["A"], ["A", "B"]]
    sb = StructureBatch.from_pdb(TUTORIAL, device="cpu")
    assert sb.get_total_lengths().tolist() == [130, 184]                 # test_StructureBatch.py:158-163
    assert PDB.read_pdb(TUTORIAL[0]).n_residues == 130                   # test_StructureBatch.py:123-128
    assert PDB.read_pdb(os.path.join(G, "4uuj.pdb")).n_residues == 545   # k_nearest_residues.ipynb cell 4
    assert PDB.read_pdb(os.path.join(G, "4uuj.pdb")).get_chain_ids() == ["C", "A", "B"]   # file order; cell 1 names A heavy, B light, C antigen


def test_tutorial_entries_valid_phi_psi_counts_through_the_oracle():
    """ramachandran_plot.ipynb cells 3-8: 128 / 180 residues have a valid (phi, psi) pair, the reference's own count
    and biotite's.  Here: the reader's tensors through the CPU oracle's backbone_dihedrals."""
    from oracle import protstruc_oracle as O
    xyz, mask, chain_idx, chain_ids, seq, residue_idx = read_batch(TUTORIAL)
    dih, dmask = O.backbone_dihedrals(xyz, chain_idx, mask.any(-1))
    assert dih.shape == (2, 184, 3) and dmask.shape == (2, 184, 3)
    valid = dmask[:, :, [0, 1]].all(-1)
    assert valid.sum(1).tolist() == [128, 180]
    assert not torch.isnan(dih[:, :, :2][valid]).any()
    assert (dih[:, :, :2][valid].abs() <= math.pi).all()


def _canonical_atom_records(path):
    """The file's model-1, first-altloc ATOM / HETATM records of canonical (or substituted) residues whose atom name is
    in that residue's 15-slot table -- by plain column slicing, independently of protstruc_amd.pdb's parser (only the
    reference's two data tables are shared: the substitution table general.py:109-124 and the slot table :149-171)."""
    from protstruc_amd.pdb import ATOM_SLOT as SLOT, SUBSTITUTIONS as SUB
    out, first_altloc = [], {}
    with open(path) as fh:
        for line in fh:
            if line.startswith("ENDMDL"):
                break                                                # model 1 only (reference pdb.py:66)
            if line[:6] not in ("ATOM  ", "HETATM"):
                continue
            name = line[17:20].strip()
            name = SUB.get(name, name)
            atom = line[12:16].strip()
            if name not in SLOT or atom not in SLOT[name]:
                continue
            res = (line[21], line[22:26], line[26])                 # chain, residue number, insertion code (raw text)
            alt = line[16]
            if alt != " " and first_altloc.setdefault(res, alt) != alt:
                continue
            out.append((res, name, atom, np.array([line[30:38], line[38:46], line[46:54]], dtype=np.float64)))
    return out


@pytest.mark.parametrize("name", ["15c8_HL.pdb", "6dc4.pdb", "1ad0_DC.pdb", "5cjx_HL.pdb", "1a3r_HL.pdb", "1a6v_HL.pdb",
                                  "1a6v_JN.pdb", "1REX.pdb", "4EOT.pdb", "4uuj.pdb"])
def test_every_atom_record_lands_in_exactly_one_slot(name):
    """Conservation (needs no biotite): every qualifying ATOM record of the file appears in exactly one (residue, slot)
    with float32(columns 31-54) as its coordinates, nothing else is set, and residues keep the file's order."""
    from protstruc_amd.pdb import ATOM_SLOT as SLOT
    path = os.path.join(G, name)
    recs = _canonical_atom_records(path)
    p = PDB.read_pdb(path)
    xyz, mask = p.get_atom_xyz()
    assert int(mask.sum()) == len(recs) > 0                          # as many set slots as qualifying records
    seen = torch.zeros_like(mask)
    order, last = [], None
    for res, rname, atom, coord in recs:
        key = (res[0], int(res[1]), res[2].strip())
        idx = p.cri2idx[key]
        slot = SLOT[rname][atom]
        assert p.name_of[idx] == rname and p.chain_of[idx] == res[0] and p.number_of[idx] == int(res[1])
        assert mask[idx, slot] and not seen[idx, slot], (name, res, atom)   # in exactly one slot, no slot hit twice
        seen[idx, slot] = True
        assert np.array_equal(xyz[idx, slot].numpy(), coord.astype(np.float32)), (name, res, atom)
        if idx != last:
            order.append(idx)
            last = idx
    assert torch.equal(seen, mask)                                   # nothing is set that no record accounts for
    assert order == sorted(order) and len(set(order)) == len(order)  # file order, every residue one contiguous run
    with_atoms = mask.any(-1).nonzero().flatten().tolist()
    assert with_atoms == order                                       # residues without records are the UNK fillers only
    assert all(p.name_of[i] == "UNK" for i in range(p.n_residues) if i not in set(order))
