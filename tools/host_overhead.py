#!/usr/bin/env python3
"""Host-side cost per call of the Python shell (wall clock over many asynchronous calls at a tiny shape, where the
GPU work is a few microseconds): what a single-structure user pays per method call."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protstruc_amd import StructureBatch, _lib, ops

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
    # config 5's eager step: diffuse_xyz + backbone_orientations
    ("config-5 eager step (diffuse_xyz + backbone_orientations)", lambda: (sb.diffuse_xyz(beta), sb.backbone_orientations())),
    # where the time goes
    ("  part: torch.empty (1 tensor, cuda)", lambda: torch.empty(B, N, 3, 3, device="cuda")),
    ("  part: ops._stream(xg)", lambda: ops._stream(xg)),
    ("  part: ctypes call of the K1 entry with B = 0 (no launch)", lambda: _K1(xg.data_ptr(), 0, d.data_ptr(), 0, 0, N, 15, 0, N, N, 0, _REF, 0)),
    ("  part: ops._f32c + shape + _u8c", lambda: (ops._f32c(xg, "xyz"), xg.shape[:3], ops._u8c(mg, "m"))),
    # the floor of any launch from Python: the C entry point called directly with prebuilt arguments (ctypes marshalling
    # + hipLaunchKernel), no validation, no allocation
    ("  floor: bare ctypes launch of ps_frames_f32 (K4)", lambda: _K4(_xp, _rp, None, B, N, 15, 0, 1, 2, 1, 0)),
    ("  floor: bare ctypes launch of the K1 entry", lambda: _K1(_xp, _mp, _dp, _mmp, B, N, 15, 0, N, N, 0, _REF, 0)),
]
_K4 = _lib.load().ps_frames_f32
_rot = torch.empty(B, N, 3, 3, device="cuda")
_xp, _rp, _mp, _dp, _mmp = xg.data_ptr(), _rot.data_ptr(), mg.data_ptr(), d.data_ptr(), m.data_ptr()
_K1 = _lib.load().ps_pairwise_distance_cfg_f32
_REF = _lib.k1_config_ref(0)
for name, fn in rows:
    print(f"{name:50s} {per_call(fn):8.1f} us/call", flush=True)
