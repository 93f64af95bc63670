import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protstruc_amd import StructureBatch
B, N = 256, 384
g = torch.Generator().manual_seed(2)
xyz = torch.randn(B, N, 15, 3, generator=g)
sb = StructureBatch.from_xyz(xyz).manual_seed(1)
beta = torch.full((B,), 0.01, device="cuda")
noise = torch.randn(B, N, 15, 3, device="cuda")
rot = torch.empty(B, N, 3, 3, device="cuda"); tr = torch.empty(B, N, 3, device="cuda")
for _ in range(20):
    sb.diffuse_xyz(beta)
for _ in range(20):
    sb.diffuse_xyz(beta, noise=noise)
for _ in range(20):
    sb.backbone_orientations()
for _ in range(20):
    sb.diffuse_xyz_and_frames(beta, out_rot=rot, out_trans=tr)
for _ in range(20):
    sb._standardized = False; sb.standardize()
for _ in range(20):
    sb.backbone_dihedrals()
torch.cuda.synchronize()
# the whole loop in one launch (T = 300), with and without the per-step coordinates
betas = torch.full((300, B), 0.01, device="cuda")
for _ in range(5):
    sb.diffuse_trajectory(betas)
for _ in range(3):
    sb.diffuse_trajectory(betas, want_xyz=True)
torch.cuda.synchronize()
