import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from oracle import protstruc_oracle as O
from protstruc_amd import StructureBatch as SB

def synth(seed, B, N, A=15, p=0.9, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    xyz = torch.randn(B, N, A, 3, generator=g) * scale
    mask = torch.rand(B, N, A, generator=g) < p
    mask[:, :, :3] = True
    return xyz, mask
N = 16
xyz, mask = synth(600 + N, 3, N)
beta = torch.tensor([0.1, 0.5, 0.9])
noise = torch.randn(3, N, 15, 3, generator=torch.Generator().manual_seed(1))
sb = SB.from_xyz(xyz.clone(), mask)
sb.diffuse_xyz(beta, noise=noise)
got = sb.get_xyz().cpu()
want = O.diffuse_xyz(xyz, beta, noise)
bad = (got != want)
print("mismatch count", bad.sum().item(), "of", bad.numel(), "per struct", bad.reshape(3, -1).sum(1).tolist())
x = xyz.numpy(); e = noise.numpy(); b = beta.numpy().reshape(3, 1, 1, 1)
keep = np.sqrt(np.float32(1) - b); add = np.sqrt(b)
emu = (keep * x).astype(np.float32) + (e * add).astype(np.float32)
print("cpu numpy-unfused == oracle:", np.array_equal(emu, want.numpy()), " == gpu:", np.array_equal(emu, got.numpy()))
print("keep/add cpu:", keep.ravel().view(np.uint32), add.ravel().view(np.uint32))
idx = bad.nonzero()[:5]
for i in idx:
    i = tuple(i.tolist())
    print(i, got[i].item().hex(), want[i].item().hex(), "x", x[i].item().hex(), "eps", e[i].item().hex())
# K4 check
xyz, mask = synth(400, 256, 384)
rot = SB.from_xyz(xyz, mask).backbone_orientations().cpu()
ref = O.backbone_orientations(xyz)
err = (rot - ref).abs()
print("K4 max err", err.max().item(), "frac>1e-5", (err > 1e-5).float().mean().item(), "nan", rot.isnan().sum().item(), ref.isnan().sum().item())
w = err.reshape(-1, 9).max(1).values.argmax().item()
print("worst residue", w, "\n got", rot.reshape(-1, 3, 3)[w], "\n ref", ref.reshape(-1, 3, 3)[w], "\n atoms", xyz.reshape(-1, 15, 3)[w, :3])
