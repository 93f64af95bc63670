import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import protstruc_oracle as O
from protstruc_amd import ops
g = torch.Generator().manual_seed(3)
for N in (5, 300):
    xyz = torch.randn(2, N, 15, 3, generator=g)
    xg = xyz.cuda()
    ops.set_exact_angles(True)
    for si, sj in (([], [0, 1, 2, 3]), ([0], [1, 2, 3])):
        one = ops.pairwise_angles(xg, si, sj, 4, _one_column=True).cpu()
        got = ops.pairwise_angles(xg, si, sj, 4).cpu()
        ref = O.pairwise_dihedrals(xyz, si, sj)
        d = (one != got)
        print("N", N, si, sj, "differing entries", int(d.sum()), "of", d.numel())
        idx = d.nonzero()[:6]
        for b, i, j in idx.tolist():
            print("  ", (b, i, j), "one-column %.9g  sweep %.9g  oracle %.9g" % (one[b, i, j], got[b, i, j], ref[b, i, j]), hex(one[b,i,j].view(torch.int32).item() & 0xffffffff), hex(got[b,i,j].view(torch.int32).item() & 0xffffffff))
    ops.set_exact_angles(False)
