#!/usr/bin/env python3
"""K1 store rate as a function of the padded length N (which launch path each N takes, and what it costs).

N % 16 == 0 takes the pattern kernel, every other N >= 16 the flat pattern kernel (k1_flat=0 sends those to the
slot-decode kernel instead).  Arguments: lengths, and key=value K1 tuning (flat=0, flat_cpw=2, ...).  Prints ms per launch and TB/s of algorithmic stores for each N at a fixed
number of output bytes (B chosen so that B*N*N stays near 64*512*512; K1_B=<n> in the environment fixes B instead)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protstruc_amd import _lib, ops

A = 15
for kv in [v for v in sys.argv[1:] if "=" in v]:   # e.g. flat_cpw=2 flat=0; A=5 sets the atom count
    k, v = kv.split("=")
    if k == "A":
        A = int(v)
    else:
        _lib.set_tuning("k1_" + k, int(v))
lengths = [int(v) for v in sys.argv[1:] if "=" not in v] or [512, 511, 510, 508, 504, 500, 496, 437, 448, 256, 250, 128,
                                                             100, 64, 50, 33, 17]
g = torch.Generator().manual_seed(0)
for N in lengths:
    B = int(os.environ["K1_B"]) if "K1_B" in os.environ else max(1, round(64 * 512 * 512 * 225 / (N * N * A * A) * (0.5 if A != 15 else 1)))
    xyz = torch.randn(B, N, A, 3, generator=g).cuda()
    mask = (torch.rand(B, N, A, generator=g) < 0.9)
    mask[:, :, :3] = True
    mask = mask.cuda()
    dist = torch.empty(B, N, N, A, A, device="cuda")
    dmask = torch.empty(B, N, N, A, A, dtype=torch.bool, device="cuda")
    for _ in range(40):
        ops.pairwise_distance(xyz, mask, out_dist=dist, out_mask=dmask)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
    for a, b in ev:
        a.record()
        ops.pairwise_distance(xyz, mask, out_dist=dist, out_mask=dmask)
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    med = ts[len(ts) // 2]
    nbytes = B * N * N * A * A * 5
    print(f"A={A:3d} N={N:5d} B={B:5d}  med {med:7.3f} ms  {nbytes / med / 1e9:6.2f} TB/s  [{_lib.k1_plan(B, N, A)['kernel']}]",
          flush=True)
    del xyz, mask, dist, dmask
print("autotune:", ops.k1_autotune_result())
