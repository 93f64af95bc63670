import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protstruc_amd import ops
mode = sys.argv[1]
B, N, A = 64, 512, 15
g = torch.Generator().manual_seed(0)
xyz = torch.randn(B, N, A, 3, generator=g).cuda()
mask = (torch.rand(B, N, A, generator=g) < 0.9); mask[:, :, :3] = True; mask = mask.cuda()
d = torch.empty(B, N, N, A, A, device="cuda"); m = torch.empty(B, N, N, A, A, dtype=torch.bool, device="cuda")
if mode == "touch":
    d.zero_(); m.zero_()
elif mode == "touch2":
    for _ in range(2): d.zero_(); m.zero_()
torch.cuda.synchronize()
def step(): ops.pairwise_distance(xyz, mask, out_dist=d, out_mask=m)
for _ in range(3): step()
torch.cuda.synchronize()
res = []
for blk in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(20): step()
    e1.record(); torch.cuda.synchronize(); res.append(e0.elapsed_time(e1) / 20)
print(mode, " ".join(f"{r:.3f}" for r in res))
