#!/usr/bin/env python3
"""Host-side cost per call of the Python shell (wall clock over many asynchronous calls at a tiny shape, where the
GPU work is a few microseconds): what a single-structure user pays per method call."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protstruc_amd import StructureBatch, ops

g = torch.Generator().manual_seed(0)
B, N = 1, 64
xyz = torch.randn(B, N, 15, 3, generator=g)
mask = torch.rand(B, N, 15, generator=g) < 0.9
mask[:, :, :3] = True
sb = StructureBatch.from_xyz(xyz, mask)
xg, mg = sb.get_xyz(), sb.get_atom_mask()
d = torch.empty(B, N, N, 15, 15, device="cuda")
m = torch.empty(B, N, N, 15, 15, dtype=torch.bool, device="cuda")
beta = torch.full((B,), 0.01, device="cuda")


def per_call(fn, n=2000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return (t1 - t0) / n * 1e6


rows = [
    ("ops.pairwise_distance (preallocated outputs)", lambda: ops.pairwise_distance(xg, mg, out_dist=d, out_mask=m)),
    ("ops.pairwise_distance (allocating)", lambda: ops.pairwise_distance(xg, mg)),
    ("sb.pairwise_distance_matrix()", lambda: sb.pairwise_distance_matrix()),
    ("sb.backbone_dihedrals()", lambda: sb.backbone_dihedrals()),
    ("sb.backbone_orientations()", lambda: sb.backbone_orientations()),
    ("sb.pairwise_dihedrals(CA,CB|CA,CB)", lambda: sb.pairwise_dihedrals(["CA", "CB"], ["CA", "CB"])),
    ("sb.inter_residue_geometry()", lambda: sb.inter_residue_geometry()),
    ("sb.diffuse_xyz(beta)", lambda: sb.diffuse_xyz(beta)),
    ("torch.empty + torch.add (reference point)", lambda: torch.add(xg, 1.0)),
]
for name, fn in rows:
    print(f"{name:50s} {per_call(fn):8.1f} us/call", flush=True)
