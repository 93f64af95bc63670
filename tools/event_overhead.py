import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protstruc_amd import ops
B, N, A = 64, 512, 15
g = torch.Generator().manual_seed(0)
xyz = torch.randn(B, N, A, 3, generator=g).cuda()
mask = (torch.rand(B, N, A, generator=g) < 0.9); mask[:, :, :3] = True; mask = mask.cuda()
d = torch.empty(B, N, N, A, A, device="cuda"); m = torch.empty(B, N, N, A, A, dtype=torch.bool, device="cuda")
def step(): ops.pairwise_distance(xyz, mask, out_dist=d, out_mask=m)
for _ in range(3): step()
K = 20
for rnd in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s = [torch.cuda.Event(enable_timing=True) for _ in range(K)]; e = [torch.cuda.Event(enable_timing=True) for _ in range(K)]
    for k in range(K):
        s[k].record(); step(); e[k].record()
    torch.cuda.synchronize(); wall_a = (time.perf_counter() - t0) / K * 1e3
    per = sum(a.elapsed_time(b) for a, b in zip(s, e)) / K
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(K): step()
    e1.record(); torch.cuda.synchronize(); wall_b = (time.perf_counter() - t0) / K * 1e3
    print(f"round {rnd}: per-launch events: kernel {per:.3f} ms wall {wall_a:.3f} ms/step | one event pair: {e0.elapsed_time(e1)/K:.3f} ms wall {wall_b:.3f} ms/step")
