#!/usr/bin/python3
"""K3 across chain lengths at a fixed number of residue pairs (2^25): the dispatcher's pick (per-CU sweep kernels for even N)
against the one-column kernel (bit 1 of exact_angles), HIP events over a train of launches; PS_K3_FAITHFUL=1: the reference's order of operations.
   python3 tools/k3_shapes.py [reps] [N ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from protstruc_amd import _lib, ops

if os.environ.get("PS_K3_FAITHFUL"):
    ops.set_exact_angles(True)

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


print("us per launch, 2^25 residue pairs (134 MB) per launch;  dihedral (2,2) / dihedral (3,1) / planar (2,1)")
for N in ([int(v) for v in sys.argv[2:]] or [512, 384, 256, 200, 128, 100, 64, 48, 32, 16, 511, 255]):
    B = max(1, (1 << 25) // (N * N))
    g = torch.Generator().manual_seed(N)
    xyz = torch.randn(B, N, 15, 3, generator=g).cuda()
    out = torch.empty(B, N, N, device="cuda")
    row = f"N={N:4d} B={B:6d}  "
    for npts, si, sj in ((4, [1, 4], [1, 4]), (4, [0, 1, 4], [4]), (3, [1, 4], [4])):
        a = timed(lambda: ops.pairwise_angles(xyz, si, sj, npts, out=out))
        b = timed(lambda: ops.pairwise_angles(xyz, si, sj, npts, out=out, _one_column=True))
        fam = _lib.k3_plan(B, N, 15, si, sj, npts, exact_angles=int(ops.get_exact_angles()))["family"]
        row += f"  {a:6.1f} {fam} (one-column {b:6.1f})"
    print(row, flush=True)
