#!/usr/bin/python3
"""Where the featuriser's tile kernel spends its time at short chains: wave time stamps of one k3_featurise_tiles launch
(100 MHz wall clock, the -DPS_K3_AB build: PROTSTRUC_AMD_LIB=protstruc_amd/lib/libprotstruc_hip_ab.so python3
tools/k3f_stamps.py [N ...]), 2^25 pairs per launch.  Per wave: entry, and for each of the first five staging passes: pass
begun (past the top barrier), rows staged (past the second barrier), the wave's last task of the pass done."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from protstruc_amd import _lib
from protstruc_amd.structure_batch import StructureBatch

lib = _lib.load()
lib.ps_k3f_debug_stamps.restype = ctypes.c_int
lib.ps_k3f_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
cus = torch.cuda.get_device_properties(0).multi_processor_count
for N in [int(a) for a in sys.argv[1:]] or [64, 48, 160]:
    B = (1 << 25) // (N * N)
    g = torch.Generator().manual_seed(1)
    xyz = torch.randn(B, N, 15, 3, generator=g).cuda()
    mask = (torch.rand(B, N, 15, generator=g) < 0.9).cuda()
    sb = StructureBatch.from_xyz(xyz, mask)
    plan = _lib.featuriser_plan(B, N, 15, cu_count=cus)
    if plan["family"] != "featurise_tiles":
        print(f"N={N}: {plan['family']}, not the tile kernel"); continue
    for _ in range(5):
        sb.inter_residue_geometry()
    torch.cuda.synchronize()
    us = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); sb.inter_residue_geometry(); e1.record(); e1.synchronize()
        us.append(e0.elapsed_time(e1) * 1e3)
    buf = np.zeros(512 * 8 * 16, dtype=np.uint64)
    assert lib.ps_k3f_debug_stamps(buf.ctypes.data, buf.size) == 0
    waves = plan["threads_per_workgroup"] // 64
    wgs = min(plan["n_workgroups"], 512)
    st = buf.reshape(512, 8, 16)[:wgs, :waves].astype(np.int64)
    rel = (st - st[:, :, 0].min()) / 100.0                    # us
    ks = plan["structures_per_segment"]
    q = lambda a: "min %6.1f  p10 %6.1f  median %6.1f  p90 %6.1f  max %6.1f" % (a.min(), np.percentile(a, 10), np.median(a), np.percentile(a, 90), a.max())
    print(f"== N={N} B={B}  {plan['kernel']}  {plan['n_workgroups']} workgroups x {waves} waves, {ks} structures per pass, {plan['lds_bytes']} B LDS;"
          f"  HIP events: median {np.median(us):.1f} us, min {min(us):.1f}")
    print("  wave entry                     ", q(rel[:, :, 0]))
    tot = {"barrier wait": 0.0, "staging": 0.0, "tasks": 0.0}
    prev = rel[:, :, 0]
    for p in range(5):
        if (st[:, :, 3 + 3 * p] == 0).any():
            break
        begun, staged, done = rel[:, :, 1 + 3 * p], rel[:, :, 2 + 3 * p], rel[:, :, 3 + 3 * p]
        print(f"  pass {p}: wait at the top barrier ", q(begun - prev))
        print(f"          staging (to 2nd barrier)", q(staged - begun))
        print(f"          the wave's tasks        ", q(done - staged), f"  spread of the ends inside a workgroup: median {np.median(done.max(axis=1) - done.min(axis=1)):.1f}")
        tot["barrier wait"] += (begun - prev).mean(); tot["staging"] += (staged - begun).mean(); tot["tasks"] += (done - staged).mean()
        prev = done
    d0 = (rel[:, :, 3] - rel[:, :, 2]).mean(axis=1)          # pass 0, per workgroup
    print("  pass-0 task time by workgroup % 8 (XCD):   " + "  ".join(f"{d0[k::8].mean():5.1f}" for k in range(8)))
    print("  pass-0 task time by (workgroup / 8) % 8:   " + "  ".join(f"{d0[[w for w in range(wgs) if (w // 8) % 8 == k]].mean():5.1f}" for k in range(8)))
    print("  pass-0 task time by (workgroup / 64) % 8:  " + "  ".join(f"{d0[[w for w in range(wgs) if (w // 64) % 8 == k]].mean():5.1f}" for k in range(8)))
    print("  pass-0 task time, workgroups 0..31:        " + " ".join(f"{x:.0f}" for x in d0[:32]))
    if os.environ.get("K3F_STAMPS_DUMP"):
        np.save(os.path.join(os.environ["K3F_STAMPS_DUMP"], f"k3f_stamps_N{N}.npy"), st)
    end = prev.max(axis=1)
    print("  workgroup end (stamped passes) ", q(end), f"  (last - median: {end.max() - np.median(end):.1f} us)")
    print("  mean per wave over the stamped passes: " + ", ".join(f"{k} {v:.1f} us" for k, v in tot.items()))
