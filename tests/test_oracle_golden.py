"""Pin the CPU oracle (oracle/protstruc_oracle.py) to the reference.

Every expected value here was produced by the reference itself
(tools/make_golden.py, run in the build container) or is an analytic known
answer from the reference's own tests (tests/test_geometry.py:35-190,
:246-262 in the reference checkout).  Float outputs must agree to 1e-6 (they
normally agree bit-for-bit: same ATen / numpy primitives); masks and NaN
positions must be identical.
"""
import math

import pytest
import torch

from oracle import protstruc_oracle as O
from tests.conftest import load_golden

TOL = 1e-6


def close(got, want, tol=TOL):
    assert got.shape == want.shape, (got.shape, want.shape)
    assert got.dtype == want.dtype, (got.dtype, want.dtype)
    assert torch.equal(torch.isnan(got), torch.isnan(want)), "NaN positions differ"
    ok = torch.isclose(got, want, rtol=0, atol=tol, equal_nan=True)
    assert ok.all(), f"max abs diff {(got - want).abs().nan_to_num(0).max().item():.3e}"


@pytest.mark.parametrize("name", [
    "g1_dist_b2_n8", "g1_dist_b1_n21", "g1_dist_b2_n6_a25", "g1_dist_floatmask", "g1_dist_nan",
])
def test_pairwise_distance_matrix(name):
    g = load_golden(name)
    d, m = O.pairwise_distance_matrix(g["xyz"], g["atom_mask"])
    close(d, g["dist"])
    assert m.dtype == g["dist_mask"].dtype
    assert torch.equal(m, g["dist_mask"])
    d2, m2 = O.pairwise_distance_matrix_chunked(g["xyz"], g["atom_mask"])
    assert torch.equal(d2.isnan(), d.isnan()) and torch.equal(d2.nan_to_num(0), d.nan_to_num(0))
    assert torch.equal(m2, m)


def test_pairwise_distance_matrix_protein_scale():
    # at 30 A scale one fp32 ulp is ~4e-6..8e-6: gate at 2 ulp of the value
    g = load_golden("g1_dist_b1_n12_protein_scale")
    d, m = O.pairwise_distance_matrix(g["xyz"], g["atom_mask"])
    ulp = torch.finfo(torch.float32).eps * g["dist"].abs().clamp_min(1.0)
    assert ((d - g["dist"]).abs() <= 2 * ulp).all()
    assert torch.equal(m, g["dist_mask"])


@pytest.mark.parametrize("name", ["g2_bbdih_chains", "g2_bbdih_padded_nan", "g2_bbdih_default_a25"])
def test_backbone_dihedrals(name):
    g = load_golden(name)
    xyz = g["xyz"]
    B, N = xyz.shape[:2]
    if "atom_mask" in g:
        residue_mask = g["atom_mask"].any(dim=-1)
        chain_idx = g["chain_idx"]
    else:
        residue_mask = torch.ones(B, N, dtype=torch.bool)
        chain_idx = torch.zeros(B, N)
    nterm = O.n_terminal_mask(chain_idx, residue_mask)
    cterm = O.c_terminal_mask(chain_idx, residue_mask)
    assert nterm.dtype == torch.bool and torch.equal(nterm, g["nterm"])
    assert cterm.dtype == torch.bool and torch.equal(cterm, g["cterm"])
    dih, dmask = O.backbone_dihedrals(xyz, chain_idx, residue_mask)
    close(dih, g["dihedrals"])
    assert dmask.dtype == torch.bool and torch.equal(dmask, g["dihedral_mask"])


def test_pairwise_dihedrals_and_planar_angles():
    g = load_golden("g3_pairwise_angles")
    xyz = g["xyz"]
    S = {"N": 0, "CA": 1, "C": 2, "O": 3, "CB": 4}

    def slots(txt):
        return [S[t.upper()] for t in txt.split("_") if t]

    n_checked = 0
    for key, want in g.items():
        if not (key.startswith("dih_") or key.startswith("ang_")):
            continue
        left, right = key[4:].split("__")
        fn = O.pairwise_dihedrals if key.startswith("dih_") else O.pairwise_planar_angles
        got = fn(xyz, slots(left), slots(right))
        close(got, want)
        n_checked += 1
    assert n_checked == 8
    # the reference's exact-zero diagonal (identical operands in the first cross product)
    omega = O.pairwise_dihedrals(xyz, [1, 4], [1, 4])
    diag = torch.diagonal(omega, dim1=1, dim2=2)
    assert (diag == 0).all() and not torch.signbit(diag).any()
    # planar-angle diagonal is NaN (0/0) when points 2 and 3 coincide
    phi = O.pairwise_planar_angles(xyz, [1, 4], [4])
    assert torch.diagonal(phi, dim1=1, dim2=2).isnan().all()


def test_frames():
    g = load_golden("g5_frames")
    xyz = g["xyz"]
    close(O.backbone_orientations(xyz), g["rot_default"])
    close(O.backbone_orientations(xyz, 2, 1, 0), g["rot_C_CA_N"])
    close(O.backbone_orientations(xyz, 4, 1, 3), g["rot_CB_CA_O"])
    assert torch.equal(O.backbone_translations(xyz), g["trans_CA"])
    assert torch.equal(O.backbone_translations(xyz, 0), g["trans_N"])
    # reference tests/test_geometry.py:246-262: ideal N/CA/C gives the identity frame, exactly
    ideal = g["ideal_xyz"]
    rot = O.gram_schmidt(ideal[:, :, 0], ideal[:, :, 1], ideal[:, :, 2])
    assert torch.equal(rot, g["ideal_rot"])
    assert (rot == torch.eye(3).expand(2, 10, -1, -1)).all()


def test_standardize_roundtrip():
    g = load_golden("g6_standardize")
    for k in range(4):
        xyz, mask = g[f"xyz_{k}"], g[f"atom_mask_{k}"]
        out, mu, std = O.standardize(xyz, mask)
        close(mu, g[f"mu_{k}"])
        close(std, g[f"std_{k}"])
        close(out, g[f"std_xyz_{k}"])
        close(O.unstandardize(out, mu, std), g[f"unstd_xyz_{k}"])
    # batched call == per-structure calls (the semantics the reference intends)
    xyz = torch.cat([g["xyz_0"], g["xyz_0"].flip(1) * 2.0])
    mask = torch.cat([g["atom_mask_0"], g["atom_mask_0"].flip(1)])
    out, mu, std = O.standardize(xyz, mask)
    for b in range(2):
        o1, m1, s1 = O.standardize(xyz[b:b + 1], mask[b:b + 1])
        assert torch.equal(o1[0], out[b]) and torch.equal(m1[0], mu[b]) and torch.equal(s1[0], std[b])


def test_diffuse_deterministic_part():
    g = load_golden("g7_diffuse")
    got = O.diffuse_xyz(g["xyz"], g["beta"], g["noise"])
    assert torch.equal(got, g["out"])


def test_inter_residue_geometry():
    g = load_golden("g8_inter_residue_geometry")
    geo = O.inter_residue_geometry(g["xyz"], g["atom_mask"])
    assert set(geo) == {"d_ca", "d_ca_mask", "d_cb", "d_cb_mask", "d_no", "d_no_mask", "omega", "theta", "phi"}
    for k, v in geo.items():
        if k.endswith("_mask"):
            assert torch.equal(v, g[k])
        else:
            close(v, g[k])


def test_primitives_known_answers():
    g = load_golden("g9_primitives")
    a, b, c, d, c60 = g["a"], g["b"], g["c"], g["d"], g["c60"]
    # analytic values asserted by the reference's tests/test_geometry.py:35-190
    assert torch.allclose(O.angle(a, b, c, to_degree=True), torch.tensor([90.0]))
    assert torch.allclose(O.angle(a, b, c60, to_degree=True), torch.tensor([60.0]), atol=1e-4)
    assert torch.allclose(O.dihedral(a, b, c, d, to_degree=True), torch.tensor([-90.0]))
    assert O.dihedral(a, b, c, d).shape == (1,)
    assert O.dihedral(a[None], b[None], c[None], d[None]).shape == (1, 1)
    close(O.angle(a, b, c, to_degree=True), g["angle_abc_deg"])
    close(O.dihedral(a, b, c, d, to_degree=True), g["dihedral_abcd_deg"], tol=1e-5)
    assert O.dot(torch.tensor([1.0, 2.0, 3.0]), torch.tensor([4.0, 5.0, 6.0])).item() == 32.0
    assert math.isclose(O.norm(torch.tensor([1.0, 2.0, 3.0])).item(), math.sqrt(14), rel_tol=1e-6)
    P = g["P"]
    close(O.angle(P[0], P[1], P[2]), g["rnd_angle"])
    close(O.dihedral(P[0], P[1], P[2], P[3]), g["rnd_dihedral"])
    close(O.dot(P[0], P[1]), g["rnd_dot"])
    close(O.norm(P[0]), g["rnd_norm"])
    close(O.unit(P[0]), g["rnd_unit"])
    close(O.gram_schmidt(P[0], P[1], P[2]), g["rnd_frame"])


def test_rigid_body_ops():
    g = load_golden("g11_rigid_ops")
    xyz = g["xyz"]
    close(O.translate(xyz, g["t_res"]), g["translate_res"])
    close(O.translate(xyz, g["t_b"]), g["translate_b"])
    close(O.translate(xyz, g["t_atom"], atomwise=True), g["translate_atom"])
    close(O.rotate(xyz, g["R_b"]), g["rotate_b"], tol=5e-6)
    close(O.rotate(xyz, g["R_b"][0]), g["rotate_shared"], tol=5e-6)
    close(O.center_of_mass(xyz), g["com"])
    close(O.center_at(xyz[:1]), g["center_origin_b1"], tol=5e-6)
    close(O.center_at(xyz, g["centers"]), g["center_b"], tol=5e-6)
    close(O.center_at(xyz, g["centers"][0]), g["center_shared"], tol=5e-6)
    close(O.get_local_xyz(g["xyz2"]), g["local_xyz"], tol=5e-6)
    assert torch.equal(O.ideal_backbone(False), g["ideal3"]) and torch.equal(O.ideal_backbone(True), g["ideal4"])
    rot, tr = O.backbone_orientations(g["xyz2"]), O.backbone_translations(g["xyz2"])
    for cb in (0, 1):
        x, m = O.frames_to_backbone(rot, tr, include_cb=bool(cb))
        close(x, g[f"bb_xyz_cb{cb}"], tol=5e-6)
        assert m.dtype == g[f"bb_mask_cb{cb}"].dtype and torch.equal(m, g[f"bb_mask_cb{cb}"])


def test_align_topk_select():
    g = load_golden("g12_align_topk")
    xyz, mask, tgt, tmask = g["xyz"], g["atom_mask"], g["target_xyz"], g["target_mask"]
    close(O.align(xyz, tgt, mask & tmask), g["aligned_default_mask"], tol=2e-5)
    close(O.align(xyz, tgt, g["ca_sel"]), g["aligned_ca_only"], tol=2e-5)
    r, t = O.kabsch(xyz[0].reshape(-1, 3), tgt[0].reshape(-1, 3))
    close(r, g["kabsch_R"], tol=2e-6)
    close(t, g["kabsch_t"], tol=2e-5)
    rmask = mask[0].any(-1)
    assert torch.equal(O.topk_nearest_residue_mask(xyz[0], rmask, g["query"], 5), g["topk5"])
    assert torch.equal(O.topk_nearest_residue_mask(xyz[0], rmask, g["query"], 8, g["topk_user_mask"]), g["topk_masked"])
    assert torch.equal(O.topk_nearest_residue_mask(xyz[0], rmask, g["query"]), g["topk_all"])
    assert torch.equal(xyz[:1][g["pick"]].unsqueeze(0), g["picked_xyz"])


ATOM_COUNTS = [("g13_dist_atom_counts", f"a{A}") for A in (14, 37, 25, 3, 4, 5, 8, 16)] + \
              [("g14_dist_small_atom_counts", t) for t in ("a1", "a2", "a6", "a7", "a10", "a13", "a1n7", "a9")]


@pytest.mark.parametrize("fixture,t", ATOM_COUNTS)
def test_oracle_matches_reference_at_other_atom_counts(fixture, t):
    """G13 / G14 (tools/make_golden_atom_counts.py): the reference's pairwise_distance_matrix at atom14 / atom37 / 25,
    the backbone-only layouts, single atoms (CA traces), atom pairs and the other small counts, at lengths of every
    alignment phase and below 16 -- sampled whole blocks, exact per-pair mask counts, per-pair distance sums."""
    g = load_golden(fixture)
    xyz, mask = g[f"{t}_xyz"], g[f"{t}_atom_mask"]
    d, m = O.pairwise_distance_matrix(xyz, mask)
    b, i, j = g[f"{t}_b"].long(), g[f"{t}_i"].long(), g[f"{t}_j"].long()
    got, want = d[b, i, j], g[f"{t}_dist_blocks"]
    assert torch.equal(torch.isnan(got), torch.isnan(want))
    assert (got - want).abs().nan_to_num(0).max() <= 1e-6
    assert torch.equal(m[b, i, j], g[f"{t}_mask_blocks"])
    assert torch.equal(m.sum((3, 4)).to(torch.int32), g[f"{t}_mask_row_sums"])
    sums = torch.nan_to_num(d, nan=0.0).double().sum((3, 4)).float()
    assert torch.allclose(sums, g[f"{t}_dist_row_nansum"], rtol=1e-6, atol=1e-5)
