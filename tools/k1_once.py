#!/usr/bin/env python3
"""Run the default K1 configuration a few times (profiling target for rocprofv3).  K1_B / K1_N in the environment
choose the shape (default 64 x 512); arguments: repetitions, then key=value tuning."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os
os.environ.pop("PROTSTRUC_AMD_AUTOTUNE", None)   # no implicit tuning while measuring
import torch
from protstruc_amd import _lib, ops

B, N, A = int(os.environ.get("K1_B", "64")), int(os.environ.get("K1_N", "512")), 15
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    _lib.set_tuning(k, int(v))
g = torch.Generator().manual_seed(0)
xyz = torch.randn(B, N, A, 3, generator=g).cuda()
mask = (torch.rand(B, N, A, generator=g) < 0.9)
mask[:, :, :3] = True
mask = mask.cuda()
dist = torch.empty(B, N, N, A, A, device="cuda")
dmask = torch.empty(B, N, N, A, A, dtype=torch.bool, device="cuda")
for _ in range(reps):
    ops.pairwise_distance(xyz, mask, out_dist=dist, out_mask=dmask)
torch.cuda.synchronize()
print("done")
