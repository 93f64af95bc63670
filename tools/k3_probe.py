#!/usr/bin/python3
"""What bounds the K3 kernels and the featuriser (the -DPS_K3_AB build): each launch (2^25 pairs) timed as it is and with its
stores sent out of range (the buffer range check drops them: the arithmetic and the store issue alone, no memory traffic); the
featuriser's tile kernel also storing constants instead of computing (its store pattern alone).
PROTSTRUC_AMD_LIB=protstruc_amd/lib/libprotstruc_hip_ab.so python3 tools/k3_probe.py [N ...]      (default 512 160 64)"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from protstruc_amd import _lib, ops
from protstruc_amd.structure_batch import StructureBatch

lib = _lib.load()
lib.ps_k3f_debug_probe.restype = ctypes.c_int
lib.ps_k3f_debug_probe.argtypes = [ctypes.c_int]
cus = torch.cuda.get_device_properties(0).multi_processor_count
SPLITS = {"dihedral (2,2)": (4, [1, 4], [1, 4]), "dihedral (3,1)": (4, [0, 1, 4], [4]), "planar (2,1)": (3, [1, 4], [4])}


def timed(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    us = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        us.append(e0.elapsed_time(e1) * 1e3)
    return f"{np.median(us):6.1f} / {min(us):6.1f}"


for N in [int(a) for a in sys.argv[1:]] or [512, 160, 64]:
    B = (1 << 25) // (N * N)
    g = torch.Generator().manual_seed(1)
    xyz = torch.randn(B, N, 15, 3, generator=g).cuda()
    sb = StructureBatch.from_xyz(xyz, (torch.rand(B, N, 15, generator=g) < 0.9).cuda())
    out = torch.empty(B, N, N, device="cuda")
    for faithful in (False, True):
        ops.set_exact_angles(faithful)
        launches = {name: ((lambda s=s: ops.pairwise_angles(xyz, s[1], s[2], s[0], out=out)),
                           _lib.k3_plan(B, N, 15, s[1], s[2], s[0], exact_angles=int(faithful), cu_count=cus)) for name, s in SPLITS.items()}
        launches["featuriser"] = (sb.inter_residue_geometry, _lib.featuriser_plan(B, N, 15, exact_angles=int(faithful), cu_count=cus))
        for name, (fn, plan) in launches.items():
            row = [f"N={N:4d} B={B:5d} {'faithful' if faithful else 'fast    '} {name:15s} {plan['kernel'][:52]:52s}"]
            modes = [(0, "as it is"), (2, "arithmetic only")] + ([(1, "stores only")] if plan["family"] == "featurise_tiles" else []) + [(0, "as it is")]
            for mode, label in modes:
                assert lib.ps_k3f_debug_probe(mode) == 0
                row.append(f"{label}: {timed(fn)} us")
            print("  ".join(row), flush=True)
    ops.set_exact_angles(False)
lib.ps_k3f_debug_probe(0)
