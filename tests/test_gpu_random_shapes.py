"""Seeded random-shape sweep of every kernel against the CPU oracle (edge cases: tiny / odd / ragged sizes,
atom counts other than 15, NaN atoms, padded chains, row shards).  -m gpu."""
import numpy as np
import pytest
import torch

from oracle import protstruc_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def SB():
    assert torch.cuda.is_available()
    from protstruc_amd import StructureBatch
    return StructureBatch


def rand_batch(rng, B, N, A, nan_frac=0.0):
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    xyz = torch.randn(B, N, A, 3, generator=g) * float(rng.choice([1.0, 3.0]))
    mask = torch.rand(B, N, A, generator=g) < 0.8
    mask[:, :, :min(3, A)] = True
    if nan_frac:
        drop = torch.rand(B, N, generator=g) < nan_frac
        xyz[drop] = float("nan")
        mask[drop] = False
    return xyz, mask


def eq_nan(a, b, tol):
    a, b = a.cpu(), b
    assert a.shape == b.shape
    assert torch.equal(a.isnan(), b.isnan())
    return ((a - b).abs().nan_to_num(0) > tol).float().mean().item()


def test_random_shapes_k1():
    from protstruc_amd import ops
    rng = np.random.default_rng(20261003)
    for trial in range(40):
        B = int(rng.integers(1, 4))
        N = int(rng.choice([1, 2, 3, 5, 7, 12, 15, 16, 17, 31, 32, 33, 48, 63, 64, 65, 80, 100, 127, 128, 130]))
        A = int(rng.choice([15, 15, 15, 1, 4, 14, 16, 25]))
        xyz, mask = rand_batch(rng, B, N, A, nan_frac=0.1 if trial % 3 == 0 else 0.0)
        use_mask = trial % 5 != 0
        rd, rm = O.pairwise_distance_matrix(xyz, mask if use_mask else torch.ones_like(mask))
        d, m = ops.pairwise_distance(xyz.cuda(), mask.cuda() if use_mask else None)
        assert eq_nan(d, rd, 1e-5 * 3) == 0.0, (trial, B, N, A)
        assert torch.equal(m.cpu(), rm), (trial, B, N, A)
        if N >= 2:
            r0 = int(rng.integers(0, N - 1)); r1 = int(rng.integers(r0 + 1, N + 1))
            cd, cm = ops.pairwise_distance(xyz.cuda(), mask.cuda() if use_mask else None, row_begin=r0, row_end=r1,
                                           compact=True)
            assert torch.equal(cd, d[:, r0:r1]) or torch.equal(cd.nan_to_num(7.0), d[:, r0:r1].nan_to_num(7.0))
            assert torch.equal(cm, m[:, r0:r1])


def test_random_shapes_per_residue_kernels(SB):
    rng = np.random.default_rng(7)
    for trial in range(30):
        B = int(rng.integers(1, 5)); N = int(rng.integers(1, 200)); A = int(rng.choice([3, 5, 15, 25]))
        xyz, mask = rand_batch(rng, B, N, A, nan_frac=0.05 if trial % 2 else 0.0)
        # chains: random break points, optional NaN-padded tail
        chain_idx = torch.zeros(B, N)
        for b in range(B):
            cuts = np.sort(rng.choice(np.arange(1, max(N, 2)), size=min(int(rng.integers(0, 4)), max(N - 1, 0)), replace=False)) if N > 1 else []
            for c in cuts:
                chain_idx[b, c:] += 1
            if N > 4 and trial % 4 == 0:
                pad = int(rng.integers(1, N // 2))
                chain_idx[b, N - pad:] = float("nan"); mask[b, N - pad:] = False; xyz[b, N - pad:] = 0.0
        sb = SB.from_xyz(xyz, mask, chain_idx=chain_idx, chain_ids=[["A"]] * B)
        rmask = mask.any(-1)
        dih, dm = sb.backbone_dihedrals()
        rdih, rdm = O.backbone_dihedrals(xyz, chain_idx, rmask)
        assert eq_nan(dih, rdih, 1e-5) <= 2e-3, (trial, B, N, A)      # a few ill-conditioned torsions at most
        assert torch.equal(dm.cpu(), rdm)
        assert torch.equal(sb.get_n_terminal_mask().cpu(), O.n_terminal_mask(chain_idx, rmask))
        assert torch.equal(sb.get_c_terminal_mask().cpu(), O.c_terminal_mask(chain_idx, rmask))
        rot = sb.backbone_orientations()
        assert eq_nan(rot, O.backbone_orientations(xyz), 3e-5) <= 5e-3
        if A >= 5:
            si, sj = [1, 4], [1, 4]
            got = sb.pairwise_dihedrals(["CA", "CB"], ["CA", "CB"])
            assert eq_nan(got, O.pairwise_dihedrals(xyz, si, sj), 1e-5) <= 5e-3
            geo = sb.inter_residue_geometry()
            rgeo = O.inter_residue_geometry(xyz, mask)
            for k in ("d_ca", "d_cb", "d_no"):
                assert eq_nan(geo[k], rgeo[k], 3e-5) == 0.0
                assert torch.equal(geo[k + "_mask"].cpu(), rgeo[k + "_mask"])


def test_random_shapes_diffusion(SB):
    rng = np.random.default_rng(11)
    for trial in range(20):
        B = int(rng.integers(1, 6)); N = int(rng.integers(1, 150)); A = int(rng.choice([3, 15, 25]))
        xyz, mask = rand_batch(rng, B, N, A)
        beta = torch.rand(B) * 0.9 + 0.01
        noise = torch.randn(B, N, A, 3)
        sb = SB.from_xyz(xyz.clone(), mask)
        sb.diffuse_xyz(beta, noise=noise)
        assert torch.allclose(sb.get_xyz().cpu(), O.diffuse_xyz(xyz, beta, noise), rtol=3e-7, atol=1e-6)
        a = SB.from_xyz(xyz.clone(), mask).manual_seed(trial)
        b = SB.from_xyz(xyz.clone(), mask).manual_seed(trial)
        T = 3
        betas = torch.rand(T, B) * 0.5 + 0.01
        for t in range(T):
            a.diffuse_xyz(betas[t])
        rot, trans, _ = b.diffuse_trajectory(betas)
        assert torch.equal(a.get_xyz(), b.get_xyz()), (trial, B, N, A)
        assert torch.equal(rot[-1], a.backbone_orientations()) and torch.equal(trans[-1], a.get_xyz()[:, :, 1])
        xs = xyz.clone()
        s = SB.from_xyz(xs, mask); s.standardize()
        out, mu, std = O.standardize(xyz, mask)
        assert torch.allclose(s.mu.cpu(), mu, rtol=1e-5, atol=2e-5) and torch.allclose(s.std.cpu(), std, rtol=1e-5, atol=2e-5)
