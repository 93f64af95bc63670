#!/usr/bin/python3
"""K3 and the featuriser at BASELINE config 3 (B=128, N=512) in BOTH arithmetic modes (fast / the reference's order of
operations), HIP events: median over `reps` single launches (host-paced) and the mean of a back-to-back train, with the
kernel the dispatcher picked (ps_k3_plan_f32).   python3 tools/k3_modes_time.py [reps] [B] [N]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from protstruc_amd import StructureBatch, _lib, ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
N = int(sys.argv[3]) if len(sys.argv) > 3 else 512
g = torch.Generator().manual_seed(1)
xyz = torch.randn(B, N, 15, 3, generator=g).cuda()
sb = StructureBatch.from_xyz(xyz)
cus = torch.cuda.get_device_properties(0).multi_processor_count


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
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return ts[len(ts) // 2], ts[0], e0.elapsed_time(e1) * 1e3 / reps


feats = {"dihedral (2,2) CA,CB|CA,CB": (4, [1, 4], [1, 4]), "dihedral (3,1) N,CA,CB|CB": (4, [0, 1, 4], [4]),
         "dihedral (1,3) C|N,CA,C": (4, [2], [0, 1, 2]), "planar (2,1) CA,CB|CB": (3, [1, 4], [4])}
out = torch.empty(B, N, N, device="cuda")
print(f"B={B} N={N}  us per launch: median / min of single launches, mean of a {reps}-launch train   (PS_K3_NC={os.environ.get('PS_K3_NC', '-')})")
for mode in (0, 1):
    ops.set_exact_angles(bool(mode))
    for name, (npts, si, sj) in feats.items():
        med, mn, train = timed(lambda: ops.pairwise_angles(xyz, si, sj, npts, out=out))
        k = _lib.k3_plan(B, N, 15, si, sj, npts, exact_angles=mode, cu_count=cus)["kernel"]
        print(f"mode {mode} {name:28s} {med:6.1f} / {mn:6.1f} / {train:6.1f}   {k}", flush=True)
    med, mn, train = timed(lambda: sb.inter_residue_geometry())
    k = _lib.featuriser_plan(B, N, 15, exact_angles=mode, cu_count=cus)["kernel"]
    print(f"mode {mode} {'inter_residue_geometry':28s} {med:6.1f} / {mn:6.1f} / {train:6.1f}   {k}", flush=True)
ops.set_exact_angles(False)
