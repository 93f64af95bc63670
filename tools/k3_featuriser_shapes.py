#!/usr/bin/python3
"""The fused featuriser (inter_residue_geometry) against the padded length: us per launch (median / min of `reps`, one HIP-event
pair per launch, host-paced) at a fixed number of residue pairs (2^25, BASELINE config 3's), and the rate in G pairs/s.
python3 tools/k3_featuriser_shapes.py [reps] [N ...]      (PS_K3_FAITHFUL=1: the reference's order of operations)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from protstruc_amd import StructureBatch, ops

if os.environ.get("PS_K3_FAITHFUL"):
    ops.set_exact_angles(True)

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
lengths = [int(v) for v in sys.argv[2:]] or [512, 511, 510, 500, 384, 383, 256, 255, 200, 129, 128, 101, 100, 99, 64]
g = torch.Generator().manual_seed(1)
for N in lengths:
    B = max(1, round(2 ** 25 / (N * N)))
    xyz = torch.randn(B, N, 15, 3, generator=g).cuda()
    mask = (torch.rand(B, N, 15, generator=g) < 0.9).cuda()
    sb = StructureBatch.from_xyz(xyz, mask)
    for _ in range(3):
        out = sb.inter_residue_geometry()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = sb.inter_residue_geometry(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    med = ts[len(ts) // 2]
    print(f"N={N:4d} B={B:5d}  {med:7.1f} / {ts[0]:7.1f} us  {B * N * N / med / 1e3:7.1f} G pairs/s  ({27 * B * N * N / med / 1e6:5.2f} TB/s of its 27 B/pair)",
          flush=True)
    del sb, xyz, mask, out
