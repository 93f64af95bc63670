#!/usr/bin/python3
"""K3 at BASELINE config 3 (B=128, N=512), HIP events around every launch: per feature the median / min over `reps`
launches into (a) whatever torch.empty returns per call, (b) each of `nbuf` distinct preallocated outputs held at once
(does the allocation matter as it does for K1?).   python3 tools/k3_time.py [reps] [nbuf]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from protstruc_amd import StructureBatch, ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
nbuf = int(sys.argv[2]) if len(sys.argv) > 2 else 4
B, N = 128, 512
g = torch.Generator().manual_seed(1)
xyz = torch.randn(B, N, 15, 3, generator=g).cuda()
sb = StructureBatch.from_xyz(xyz)


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


feats = {"dihedral (2,2) CA,CB|CA,CB": (4, [1, 4], [1, 4]), "dihedral (3,1) N,CA,CB|CB": (4, [0, 1, 4], [4]),
         "dihedral (1,3) C|N,CA,C": (4, [2], [0, 1, 2]), "planar (2,1) CA,CB|CB": (3, [1, 4], [4])}
bufs = [torch.empty(B, N, N, device="cuda") for _ in range(nbuf)]
for name, (npts, si, sj) in feats.items():
    med, mn = timed(lambda: ops.pairwise_angles(xyz, si, sj, npts))
    line = f"{name:30s} torch.empty per call: {med:6.1f} / {mn:6.1f} us |"
    for k, buf in enumerate(bufs):
        med, mn = timed(lambda: ops.pairwise_angles(xyz, si, sj, npts, out=buf))
        line += f" buf{k}: {med:6.1f} / {mn:6.1f}"
    print(line, flush=True)
med, mn = timed(lambda: sb.inter_residue_geometry())
print(f"{'inter_residue_geometry':30s} {med:6.1f} / {mn:6.1f} us  (median / min)")
for k, buf in enumerate(bufs):
    med, mn = timed(lambda: buf.fill_(1.0))
    print(f"fill_ buf{k}: {med:6.1f} / {mn:6.1f} us = {B * N * N * 4 / med / 1e6:.2f} TB/s")
