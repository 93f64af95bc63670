"""Error statistics of the GPU dihedral / planar angle against the oracle and against fp64 (for DESIGN.md)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import protstruc_oracle as O
from protstruc_amd import StructureBatch as SB, ops
g = torch.Generator().manual_seed(0)
B, N = 8, 256
xyz = torch.randn(B, N, 15, 3, generator=g)
sb = SB.from_xyz(xyz)
off = ~torch.eye(N, dtype=torch.bool).expand(B, N, N)
for mode in (False, True):
  ops.set_exact_angles(mode)
  print("== exact_angles =", mode, "(the reference's order of operations)" if mode else "(default: fast arithmetic)")
  for name, ai, aj, si, sj, npts in [("omega (2,2)", ["CA", "CB"], ["CA", "CB"], [1, 4], [1, 4], 4), ("theta (3,1)", ["N", "CA", "CB"], ["CB"], [0, 1, 4], [4], 4),
                                     ("phi planar", ["CA", "CB"], ["CB"], [1, 4], [4], 3)]:
      got = (sb.pairwise_dihedrals if npts == 4 else sb.pairwise_planar_angles)(ai, aj).cpu()
      ref = (O.pairwise_dihedrals if npts == 4 else O.pairwise_planar_angles)(xyz, si, sj)
      p = O.pairwise_points(xyz.double(), si, sj)
      truth = (O.dihedral(p[:, :, 0], p[:, :, 1], p[:, :, 2], p[:, :, 3]) if npts == 4 else O.angle(p[:, :, 0], p[:, :, 1], p[:, :, 2])).reshape(B, N, N)
      wrap = (lambda d: torch.minimum(d.abs(), (2 * np.pi - d.abs()).abs())) if npts == 4 else (lambda d: d.abs())
      ok = off & ~(got.isnan() | ref.isnan())
      e_gr, e_gt, e_rt = wrap(got - ref)[ok], wrap(got.double() - truth)[ok], wrap(ref.double() - truth)[ok]
      print(f"{name:12s} gpu-vs-oracle: max {e_gr.max():.2e} frac>1e-5 {(e_gr > 1e-5).float().mean():.2e} | gpu-vs-fp64: max {e_gt.max():.2e} median {e_gt.median():.2e} frac>1e-5 {(e_gt > 1e-5).float().mean():.2e} | oracle-vs-fp64: max {e_rt.max():.2e} median {e_rt.median():.2e} frac>1e-5 {(e_rt > 1e-5).float().mean():.2e}")
