"""Minimal PDB ATOM-record reader feeding ``StructureBatch.from_pdb``.

Host-side, I/O-bound plumbing on the input side of the hot path -- not an
accelerated component.  It restates what the reference obtains from biotite
(``protstruc/pdb.py:24-40, :55-151``; biotite is not installed in the build
container, so this reader is **parity-unpinned numerically**: the only pins the
reference's tests hold are residue counts -- 15c8_HL -> 229, 6dc4 -> 437 --
and the chain-terminus counts of tests/test_StructureBatch.py:43-66):

* model 1 only, ATOM and HETATM records, first alternate location per residue;
* non-standard residue names are mapped to their parent amino acid, then only the
  20 canonical amino acids and their standard heavy-atom names are kept
  (drops waters, ligands, hydrogens);
* residues are keyed by (chain, residue number, insertion code); numbering gaps
  inside a chain are filled with UNK residues that have no atoms;
* every residue gets 15 atom slots (N, CA, C, O, CB, side chain..., OXT in slot
  14), coordinates NaN where an atom is missing, plus a boolean mask;
* chain indices count chains in order of first appearance.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np
import torch

from .general import MAX_N_ATOMS_PER_RESIDUE

# three-letter code -> one-letter code (reference general.py:26-100)
THREE_TO_ONE = {
    "ALA": "A", "CYS": "C", "ASP": "D", "GLU": "E", "PHE": "F", "GLY": "G", "HIS": "H", "ILE": "I", "LYS": "K",
    "LEU": "L", "MET": "M", "ASN": "N", "PRO": "P", "GLN": "Q", "ARG": "R", "SER": "S", "THR": "T", "VAL": "V",
    "TRP": "W", "TYR": "Y", "UNK": "X",
}
CANONICAL = frozenset(k for k in THREE_TO_ONE if k != "UNK")
# one-letter code -> integer code, alphabetical by one-letter code, X last (reference general.py:126-133)
ONE_TO_INDEX = {c: k for k, c in enumerate("ACDEFGHIKLMNPQRSTVWYX")}

# side-chain heavy atoms in slot order 5.. (slots 0-4 are N CA C O CB, slot 14 is OXT); reference general.py:149-171
_SIDE_CHAIN = {
    "ALA": "", "ARG": "CG CD NE CZ NH1 NH2", "ASN": "CG OD1 ND2", "ASP": "CG OD1 OD2", "CYS": "SG",
    "GLN": "CG CD OE1 NE2", "GLU": "CG CD OE1 OE2", "GLY": None, "HIS": "CG ND1 CD2 CE1 NE2",
    "ILE": "CG1 CG2 CD1", "LEU": "CG CD1 CD2", "LYS": "CG CD CE NZ", "MET": "CG SD CE",
    "PHE": "CG CD1 CD2 CE1 CE2 CZ", "PRO": "CG CD", "SER": "OG", "THR": "OG1 CG2",
    "TRP": "CG CD1 CD2 NE1 CE2 CE3 CZ2 CZ3 CH2", "TYR": "CG CD1 CD2 CE1 CE2 CZ OH", "VAL": "CG1 CG2",
}


def _slot_table() -> Dict[str, Dict[str, int]]:
    table = {}
    for res, side in _SIDE_CHAIN.items():
        names = ["N", "CA", "C", "O"] + ([] if side is None else ["CB"] + side.split())  # glycine has no CB
        slots = {name: k for k, name in enumerate(names)}
        slots["OXT"] = MAX_N_ATOMS_PER_RESIDUE - 1
        table[res] = slots
    return table


ATOM_SLOT = _slot_table()
STANDARD_HEAVY_ATOMS = frozenset(a for slots in ATOM_SLOT.values() for a in slots)

# non-standard residue -> parent amino acid (reference general.py:109-124)
_SUBSTITUTIONS = """
2AS:ASP 3AH:HIS 5HP:GLU ACL:ARG AGM:ARG AIB:ALA ALM:ALA ALO:THR ALY:LYS ARM:ARG ASA:ASP ASB:ASP ASK:ASP ASL:ASP
ASQ:ASP AYA:ALA BCS:CYS BHD:ASP BMT:THR BNN:ALA BUC:CYS BUG:LEU C5C:CYS C6C:CYS CAS:CYS CCS:CYS CEA:CYS CGU:GLU
CHG:ALA CLE:LEU CME:CYS CSD:ALA CSO:CYS CSP:CYS CSS:CYS CSW:CYS CSX:CYS CXM:MET CY1:CYS CY3:CYS CYG:CYS CYM:CYS
CYQ:CYS DAH:PHE DAL:ALA DAR:ARG DAS:ASP DCY:CYS DGL:GLU DGN:GLN DHA:ALA DHI:HIS DIL:ILE DIV:VAL DLE:LEU DLY:LYS
DNP:ALA DPN:PHE DPR:PRO DSN:SER DSP:ASP DTH:THR DTR:TRP DTY:TYR DVA:VAL EFC:CYS FLA:ALA FME:MET GGL:GLU GL3:GLY
GLZ:GLY GMA:GLU GSC:GLY HAC:ALA HAR:ARG HIC:HIS HIP:HIS HMR:ARG HPQ:PHE HTR:TRP HYP:PRO IAS:ASP IIL:ILE IYR:TYR
KCX:LYS LLP:LYS LLY:LYS LTR:TRP LYM:LYS LYZ:LYS MAA:ALA MEN:ASN MHS:HIS MIS:SER MLE:LEU MPQ:GLY MSA:GLY MSE:MET
MVA:VAL NEM:HIS NEP:HIS NLE:LEU NLN:LEU NLP:LEU NMC:GLY OAS:SER OCS:CYS OMT:MET PAQ:TYR PCA:GLU PEC:CYS PHI:PHE
PHL:PHE PR3:CYS PRR:ALA PTR:TYR PYX:CYS SAC:SER SAR:GLY SCH:CYS SCS:CYS SCY:CYS SEL:SER SEP:SER SET:SER SHC:CYS
SHR:LYS SMC:CYS SOC:CYS STY:TYR SVA:SER TIH:ALA TPL:TRP TPO:THR TPQ:ALA TRG:LYS TRO:TRP TYB:TYR TYI:TYR TYQ:TYR
TYS:TYR TYY:TYR
"""
SUBSTITUTIONS = dict(item.split(":") for item in _SUBSTITUTIONS.split())


class PDB:
    """One parsed structure: ``atom_xyz (n_res, 15, 3)`` NaN-filled, ``atom_xyz_mask (n_res, 15)``, chain bookkeeping."""

    def __init__(self, residues: List[dict]):
        self._build_lookup(residues)
        self._place_atoms(residues)

    # ---------------------------------------------------------------- parsing
    @classmethod
    def read_pdb(cls, fp) -> "PDB":
        with open(fp) as fh:
            return cls(cls._read_residues(fh))

    @staticmethod
    def _read_residues(lines) -> List[dict]:
        """Residues of model 1 in file order, after tidying (reference pdb.py:24-40)."""
        residues: List[dict] = []
        current_key = None
        in_first_model = True
        for line in lines:
            rec = line[:6]
            if rec.startswith("ENDMDL"):
                in_first_model = False
                continue
            if rec.startswith("MODEL"):
                continue
            if not in_first_model or rec not in ("ATOM  ", "HETATM"):
                continue
            res_name = line[17:20].strip()
            res_name = SUBSTITUTIONS.get(res_name, res_name)
            atom_name = line[12:16].strip()
            if res_name not in CANONICAL or atom_name not in STANDARD_HEAVY_ATOMS:
                continue
            key = (line[21], int(line[22:26]), line[26].strip(), res_name)
            if key != current_key:  # a residue starts where chain / number / insertion code / name changes
                residues.append({"chain": key[0], "number": key[1], "insertion": key[2], "name": res_name,
                                 "altloc": None, "atoms": []})
                current_key = key
            res = residues[-1]
            altloc = line[16]
            if altloc != " ":
                if res["altloc"] is None:
                    res["altloc"] = altloc
                if altloc != res["altloc"]:
                    continue  # only the first alternate location of a residue is kept
            xyz = (float(line[30:38]), float(line[38:46]), float(line[46:54]))
            res["atoms"].append((atom_name, xyz))
        return residues

    # ---------------------------------------------------------------- lookup (reference pdb.py:82-130)
    def _build_lookup(self, residues: List[dict]) -> None:
        self.chain_of: List[str] = []
        self.number_of: List[int] = []
        self.insertion_of: List[str] = []
        self.name_of: List[str] = []
        self.cri2idx: Dict[Tuple[str, int, str], int] = {}
        prev_chain, prev_number = None, None
        for res in residues:
            if prev_chain is None or prev_chain != res["chain"]:
                prev_chain, prev_number = res["chain"], res["number"]
            while prev_number + 1 < res["number"]:  # numbering gap inside a chain -> atom-less UNK residues
                prev_number += 1
                self._append(prev_chain, prev_number, res["insertion"], "UNK")
            self._append(res["chain"], res["number"], res["insertion"], res["name"])
            prev_chain, prev_number = res["chain"], res["number"]
        self.n_residues = len(self.chain_of)
        self.chain_ids: List[str] = list(dict.fromkeys(self.chain_of))  # order of first appearance
        code = {c: k for k, c in enumerate(self.chain_ids)}
        self.chain_idx = [code[c] for c in self.chain_of]

    def _append(self, chain, number, insertion, name) -> None:
        self.cri2idx[(chain, number, insertion)] = len(self.chain_of)  # a repeated key points at its last residue
        self.chain_of.append(chain)
        self.number_of.append(number)
        self.insertion_of.append(insertion)
        self.name_of.append(name)

    # ---------------------------------------------------------------- coordinates (reference pdb.py:132-151)
    def _place_atoms(self, residues: List[dict]) -> None:
        xyz = np.full((self.n_residues, MAX_N_ATOMS_PER_RESIDUE, 3), np.nan, dtype=np.float32)
        mask = np.zeros((self.n_residues, MAX_N_ATOMS_PER_RESIDUE), dtype=bool)
        for res in residues:
            idx = self.cri2idx[(res["chain"], res["number"], res["insertion"])]
            slots = ATOM_SLOT[res["name"]]
            for atom_name, coord in res["atoms"]:
                if atom_name not in slots:  # e.g. a CG on an ALA: the reference raises here, too
                    raise ValueError(f"'{atom_name}' is not in list")
                xyz[idx, slots[atom_name]] = coord
                mask[idx, slots[atom_name]] = True
        self.atom_xyz = torch.from_numpy(xyz)
        self.atom_xyz_mask = torch.from_numpy(mask)

    # ---------------------------------------------------------------- getters (reference pdb.py:153-180)
    def get_atom_xyz(self):
        return self.atom_xyz, self.atom_xyz_mask

    def get_chain_idx(self) -> torch.LongTensor:
        return torch.tensor(self.chain_idx).long()

    def get_chain_ids(self) -> List[str]:
        return list(self.chain_ids)

    def get_residue_idx(self) -> torch.LongTensor:
        return torch.arange(self.n_residues)

    def get_seq(self) -> str:
        return "".join(THREE_TO_ONE[n] for n in self.name_of)

    def get_seq_dict(self) -> Dict[str, str]:
        out = {c: [] for c in self.chain_ids}
        for chain, name in zip(self.chain_of, self.name_of):
            out[chain].append(THREE_TO_ONE[name])
        return {c: "".join(v) for c, v in out.items()}


def read_batch(paths: List[str]):
    """Padding/batching of ``StructureBatch.from_pdb`` (reference protstruc.py:149-187): zero coordinates
    and False mask in the padding, NaN chain / residue indices."""
    parsed = [PDB.read_pdb(p) for p in paths]
    n_max = max(p.n_residues for p in parsed)
    B = len(parsed)
    xyz = torch.zeros(B, n_max, MAX_N_ATOMS_PER_RESIDUE, 3)
    mask = torch.zeros(B, n_max, MAX_N_ATOMS_PER_RESIDUE, dtype=torch.bool)
    chain_idx = torch.full((B, n_max), float("nan"))
    residue_idx = torch.full((B, n_max), float("nan"))
    for i, p in enumerate(parsed):
        n = p.n_residues
        xyz[i, :n], mask[i, :n] = p.get_atom_xyz()
        chain_idx[i, :n] = p.get_chain_idx().float()
        residue_idx[i, :n] = p.get_residue_idx().float()
    chain_ids = [p.get_chain_ids() for p in parsed]
    seq = [p.get_seq_dict() for p in parsed]
    return xyz, mask, chain_idx, chain_ids, seq, residue_idx
