#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the reference itself on seeded inputs.

Runs ONLY in the build container, where the reference checkout is mounted:

    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden.py --reference /root/reference

The reference never travels to the GPU box; what travels are the small .npz
fixtures this script writes (inputs + the reference's outputs).  ``biotite`` is
not installed here, and the hot path never touches it, so empty stand-in modules
are registered for the import to succeed.

Shapes avoid B==3 / N==3 (reference quirk Q6: ``torch.cross`` without ``dim``).
"""
import argparse
import os
import sys
import types

import numpy as np
import torch


def import_reference(path):
    for name in [
        "biotite", "biotite.database", "biotite.database.rcsb", "biotite.structure",
        "biotite.structure.io", "biotite.structure.io.pdb",
    ]:
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["biotite.database.rcsb"].fetch = lambda *a, **k: None
    sys.modules["biotite.structure"].AtomArray = object
    sys.modules["biotite.structure.io.pdb"].PDBFile = object
    sys.path.insert(0, path)
    from protstruc import StructureBatch  # noqa: E402
    import protstruc.geometry as geom  # noqa: E402
    return StructureBatch, geom


def synth(seed, B, N, A=15, p=0.8, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    xyz = torch.randn(B, N, A, 3, generator=g) * scale
    mask = torch.rand(B, N, A, generator=g) < p
    mask[:, :, :3] = True
    return xyz, mask


def npy(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
    args = ap.parse_args()
    SB, geom = import_reference(args.reference)
    os.makedirs(args.out, exist_ok=True)

    def save(name, **arrays):
        np.savez_compressed(os.path.join(args.out, name + ".npz"), **{k: npy(v) for k, v in arrays.items()})
        print("wrote", name, {k: tuple(npy(v).shape) for k, v in arrays.items()})

    # G1 pairwise_distance_matrix ------------------------------------------------
    for tag, (seed, B, N, A, scale) in {
        "b2_n8": (11, 2, 8, 15, 1.0),
        "b1_n21": (12, 1, 21, 15, 1.0),      # N odd: rows are not 16-byte aligned
        "b2_n6_a25": (13, 2, 6, 25, 1.0),    # from_xyz does not fix A at 15
        "b1_n12_protein_scale": (14, 1, 12, 15, 30.0),
    }.items():
        xyz, mask = synth(seed, B, N, A, scale=scale)
        d, m = SB.from_xyz(xyz, mask).pairwise_distance_matrix()
        save(f"g1_dist_{tag}", xyz=xyz, atom_mask=mask, dist=d, dist_mask=m)
    xyz, mask = synth(15, 2, 8)
    fmask = mask.float()
    d, m = SB.from_xyz(xyz, fmask).pairwise_distance_matrix()
    save("g1_dist_floatmask", xyz=xyz, atom_mask=fmask, dist=d, dist_mask=m)
    # NaN coordinates (missing atoms) propagate into dist, mask stays as computed
    xyz, mask = synth(16, 2, 8)
    xyz[~mask] = float("nan")
    d, m = SB.from_xyz(xyz, mask).pairwise_distance_matrix()
    save("g1_dist_nan", xyz=xyz, atom_mask=mask, dist=d, dist_mask=m)

    # G2 backbone_dihedrals + terminal masks ---------------------------------------
    xyz, mask = synth(21, 4, 64)
    chain_idx = torch.zeros(4, 64)
    chain_idx[:, 20:45] = 1.0
    chain_idx[:, 45:] = 2.0
    chain_idx[1, 30:] = 1.0  # structure 1 has two chains only
    sb = SB.from_xyz(xyz, mask, chain_idx=chain_idx, chain_ids=[["A", "B", "C"]] * 4)
    dih, dmask = sb.backbone_dihedrals()
    save("g2_bbdih_chains", xyz=xyz, atom_mask=mask, chain_idx=chain_idx, dihedrals=dih,
         dihedral_mask=dmask, nterm=sb.get_n_terminal_mask(), cterm=sb.get_c_terminal_mask())
    # padded tail (NaN chain_idx, all-False mask, zero xyz) + one fully missing residue (NaN xyz)
    xyz, mask = synth(22, 2, 40)
    chain_idx = torch.zeros(2, 40)
    chain_idx[:, 18:] = 1.0
    xyz[0, 33:] = 0.0
    mask[0, 33:] = False
    chain_idx[0, 33:] = float("nan")
    xyz[1, 10] = float("nan")
    mask[1, 10] = False
    sb = SB.from_xyz(xyz, mask, chain_idx=chain_idx, chain_ids=[["H", "L"]] * 2)
    dih, dmask = sb.backbone_dihedrals()
    save("g2_bbdih_padded_nan", xyz=xyz, atom_mask=mask, chain_idx=chain_idx, dihedrals=dih,
         dihedral_mask=dmask, nterm=sb.get_n_terminal_mask(), cterm=sb.get_c_terminal_mask())
    # default chain_idx (None) and A=25 like the reference's own test (tests/test_StructureBatch.py:68-96)
    g = torch.Generator().manual_seed(23)
    xyz = torch.rand(4, 50, 25, 3, generator=g)
    sb = SB.from_xyz(xyz)
    dih, dmask = sb.backbone_dihedrals()
    save("g2_bbdih_default_a25", xyz=xyz, dihedrals=dih, dihedral_mask=dmask,
         nterm=sb.get_n_terminal_mask(), cterm=sb.get_c_terminal_mask())

    # G3/G4 pairwise dihedrals and planar angles ---------------------------------
    xyz, mask = synth(31, 2, 33)
    sb = SB.from_xyz(xyz, mask)
    out = {"xyz": xyz, "atom_mask": mask}
    for key, (ai, aj) in {
        "dih_CA_CB__CA_CB": (["CA", "CB"], ["CA", "CB"]),
        "dih_N_CA_CB__CB": (["N", "CA", "CB"], ["CB"]),
        "dih_C__N_CA_C": (["C"], ["N", "CA", "C"]),
        "dih_N_CA_C__N": (["N", "CA", "C"], ["N"]),
        "dih_n_ca__cb_o": (["n", "ca"], ["cb", "o"]),
    }.items():
        out[key] = sb.pairwise_dihedrals(ai, aj)
    for key, (ai, aj) in {
        "ang_CA_CB__CB": (["CA", "CB"], ["CB"]),
        "ang_CA__CA_CB": (["CA"], ["CA", "CB"]),
        "ang_N_CA_C__": (["N", "CA", "C"], []),
    }.items():
        out[key] = sb.pairwise_planar_angles(ai, aj)
    save("g3_pairwise_angles", **out)

    # G5 frames --------------------------------------------------------------------
    xyz, mask = synth(41, 4, 30)
    sb = SB.from_xyz(xyz, mask)
    ideal = geom.ideal_backbone_coordinates(size=(2, 10)).contiguous()
    save("g5_frames", xyz=xyz, atom_mask=mask,
         rot_default=sb.backbone_orientations(),
         rot_C_CA_N=sb.backbone_orientations("C", "CA", "N"),
         rot_CB_CA_O=sb.backbone_orientations("CB", "CA", "O"),
         trans_CA=sb.backbone_translations(), trans_N=sb.backbone_translations("N"),
         ideal_xyz=ideal, ideal_rot=SB.from_xyz(ideal).backbone_orientations())

    # G6 standardize at B=1 (the only batch size the reference can run, Q1) ---------
    cases = {}
    for k, (seed, N, scale) in enumerate([(51, 24, 10.0), (52, 40, 25.0), (53, 9, 3.0), (54, 17, 1.0)]):
        xyz, mask = synth(seed, 1, N, scale=scale)
        xyz = xyz + torch.tensor([5.0, -3.0, 11.0])
        if k == 1:  # NaN coordinates at masked-out slots
            xyz[~mask] = float("nan")
        sb = SB.from_xyz(xyz.clone(), mask)
        sb.standardize()
        cases[f"xyz_{k}"] = xyz
        cases[f"atom_mask_{k}"] = mask
        cases[f"std_xyz_{k}"] = sb.get_xyz()
        cases[f"mu_{k}"] = sb.mu
        cases[f"std_{k}"] = sb.std
        sb.unstandardize()
        cases[f"unstd_xyz_{k}"] = sb.get_xyz()
    save("g6_standardize", **cases)

    # G7 diffuse_xyz with the sampler's draw captured ----------------------------------
    xyz, mask = synth(61, 4, 16)
    beta = torch.tensor([0.0001, 0.02, 0.5, 0.999])
    sb = SB.from_xyz(xyz.clone(), mask)
    torch.manual_seed(1234)
    noise = torch.randn_like(xyz)
    torch.manual_seed(1234)
    sb.diffuse_xyz(beta)
    save("g7_diffuse", xyz=xyz, atom_mask=mask, beta=beta, noise=noise, out=sb.get_xyz())

    # G8 inter_residue_geometry ---------------------------------------------------------
    xyz, mask = synth(71, 2, 12)
    geo = SB.from_xyz(xyz, mask).inter_residue_geometry()
    save("g8_inter_residue_geometry", xyz=xyz, atom_mask=mask, **geo)

    # G10 BASELINE config 1: 15c8_HL.pdb parsed by the BUILD's reader (biotite is absent, so the reference's own
    # parser cannot run), then the reference's from_xyz(...) featurisers on those tensors
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    from protstruc_amd.pdb import read_batch
    pxyz, pmask, pchain, pchain_ids, pseq, _ = read_batch([os.path.join(args.out, "15c8_HL.pdb")])
    sb = SB.from_xyz(pxyz, pmask, chain_idx=pchain, chain_ids=pchain_ids)
    d, m = sb.pairwise_distance_matrix()
    g = torch.Generator().manual_seed(101)
    n = pxyz.shape[1]
    bi = torch.randint(0, n, (600,), generator=g)
    bj = torch.randint(0, n, (600,), generator=g)
    dih, dih_mask = sb.backbone_dihedrals()
    save("g10_config1_15c8_HL", n_residues=torch.tensor(n), atom_count=pmask.sum(),
         ca_ca=d[0, :, :, 1, 1], cb_cb=d[0, :, :, 4, 4], ca_ca_mask=m[0, :, :, 1, 1], cb_cb_mask=m[0, :, :, 4, 4],
         block_i=bi, block_j=bj, blocks=d[0, bi, bj], blocks_mask=m[0, bi, bj],
         dihedrals=dih, dihedral_mask=dih_mask, nterm=sb.get_n_terminal_mask(), cterm=sb.get_c_terminal_mask(),
         rot=sb.backbone_orientations(), chain_idx=pchain)

    # G11 rigid-body ops
    xyz, mask = synth(111, 4, 20, scale=5.0)
    xyz[1, 3, :] = float("nan")  # a residue without coordinates: nanmean must skip it
    g = torch.Generator().manual_seed(112)
    q, _ = torch.linalg.qr(torch.randn(4, 3, 3, generator=g))
    t_res = torch.randn(4, 20, 3, generator=g)
    t_b = torch.randn(4, 1, 3, generator=g)
    t_atom = torch.randn(4, 20, 15, 3, generator=g)
    centers = torch.randn(4, 3, generator=g)
    out = {"xyz": xyz, "atom_mask": mask, "R_b": q, "t_res": t_res, "t_b": t_b, "t_atom": t_atom, "centers": centers}
    sb = SB.from_xyz(xyz.clone(), mask); sb.translate(t_res); out["translate_res"] = sb.get_xyz()
    sb = SB.from_xyz(xyz.clone(), mask); sb.translate(t_b); out["translate_b"] = sb.get_xyz()
    sb = SB.from_xyz(xyz.clone(), mask); sb.translate(t_atom, atomwise=True); out["translate_atom"] = sb.get_xyz()
    sb = SB.from_xyz(xyz.clone(), mask); sb.rotate(q); out["rotate_b"] = sb.get_xyz()
    sb = SB.from_xyz(xyz.clone(), mask); sb.rotate(q[0]); out["rotate_shared"] = sb.get_xyz()
    sb = SB.from_xyz(xyz.clone(), mask); out["com"] = sb.center_of_mass()
    # center_at() without argument only runs at B == 1 in the reference (its (1,3) default fails its own check)
    sb = SB.from_xyz(xyz[:1].clone(), mask[:1]); sb.center_at(); out["center_origin_b1"] = sb.get_xyz()
    sb = SB.from_xyz(xyz.clone(), mask); sb.center_at(centers); out["center_b"] = sb.get_xyz()
    sb = SB.from_xyz(xyz.clone(), mask); sb.center_at(centers[0]); out["center_shared"] = sb.get_xyz()
    xyz2, mask2 = synth(113, 2, 10, scale=5.0)
    sb = SB.from_xyz(xyz2, mask2)
    out["xyz2"] = xyz2
    out["local_xyz"] = sb.get_local_xyz()
    rot, tr = sb.backbone_orientations(), sb.backbone_translations()
    for cb in (False, True):
        sb3 = SB.from_backbone_orientations_translations(rot, tr, include_cb=cb)
        out[f"bb_xyz_cb{int(cb)}"] = sb3.get_xyz()
        out[f"bb_mask_cb{int(cb)}"] = sb3.get_atom_mask()
    out["ideal3"] = geom.ideal_backbone_coordinates(size=(), include_cb=False)
    out["ideal4"] = geom.ideal_backbone_coordinates(size=(), include_cb=True)
    save("g11_rigid_ops", **out)

    # G12 align (Kabsch), top-k nearest residues, residue_masked_select
    xyz, mask = synth(121, 2, 24, scale=6.0)
    g = torch.Generator().manual_seed(122)
    q, _ = torch.linalg.qr(torch.randn(2, 3, 3, generator=g))
    q = q * torch.sign(torch.linalg.det(q))[:, None, None]           # proper rotations
    tgt = torch.einsum("bij,bnaj->bnai", q, xyz) + torch.randn(2, 1, 1, 3, generator=g) * 4 + torch.randn(2, 24, 15, 3, generator=g) * 0.3
    tmask = torch.rand(2, 24, 15, generator=g) < 0.9
    out = {"xyz": xyz, "atom_mask": mask, "target_xyz": tgt, "target_mask": tmask}
    sb, tb = SB.from_xyz(xyz.clone(), mask), SB.from_xyz(tgt, tmask)
    sb.align(tb); out["aligned_default_mask"] = sb.get_xyz()
    sel = torch.zeros(2, 24, 15, dtype=torch.bool); sel[:, :, 1] = True  # CA only
    sb = SB.from_xyz(xyz.clone(), mask); sb.align(tb, atom_mask=sel); out["aligned_ca_only"] = sb.get_xyz(); out["ca_sel"] = sel
    r, t = geom.kabsch(xyz[0].reshape(-1, 3), tgt[0].reshape(-1, 3)); out["kabsch_R"] = r; out["kabsch_t"] = t
    one = SB.from_xyz(xyz[:1].clone(), mask[:1])
    query = torch.randn(5, 3, generator=g) * 6
    out["query"] = query
    out["topk5"] = one.get_topk_nearest_residue_mask(query, k=5)
    rm = torch.rand(24, generator=g) < 0.6
    out["topk_masked"] = one.get_topk_nearest_residue_mask(query, k=8, mask=rm); out["topk_user_mask"] = rm
    out["topk_all"] = one.get_topk_nearest_residue_mask(query)
    pick = torch.rand(1, 24, generator=g) < 0.5
    sel_sb = SB.from_xyz(xyz[:1].clone(), mask[:1], chain_idx=torch.zeros(1, 24), chain_ids=[["A"]]).residue_masked_select(pick)
    out["pick"] = pick; out["picked_xyz"] = sel_sb.get_xyz(); out["picked_mask"] = sel_sb.get_atom_mask()
    save("g12_align_topk", **out)

    # G9 free-function known answers, evaluated by the reference ---------------------------
    a = torch.tensor([[1.0, 0.0, 0.0]])
    b = torch.tensor([[0.0, 0.0, 0.0]])
    c = torch.tensor([[0.0, 1.0, 0.0]])
    d = torch.tensor([[0.0, 1.0, 1.0]])
    c60 = torch.tensor([[0.5, 0.8660254037844386, 0.0]])
    g = torch.Generator().manual_seed(91)
    P = torch.randn(4, 257, 3, generator=g)
    save("g9_primitives", a=a, b=b, c=c, d=d, c60=c60,
         angle_abc_deg=geom.angle(a, b, c, to_degree=True),
         angle_abc60_deg=geom.angle(a, b, c60, to_degree=True),
         dihedral_abcd_deg=geom.dihedral(a, b, c, d, to_degree=True),
         P=P, rnd_angle=geom.angle(P[0], P[1], P[2]), rnd_dihedral=geom.dihedral(P[0], P[1], P[2], P[3]),
         rnd_dot=geom.dot(P[0], P[1]), rnd_norm=geom.norm(P[0]), rnd_unit=geom.unit(P[0]),
         rnd_frame=geom.gram_schmidt(P[0], P[1], P[2]))


if __name__ == "__main__":
    main()
