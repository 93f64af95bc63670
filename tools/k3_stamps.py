#!/usr/bin/python3
"""Where K3's time goes at BASELINE config 3 (B=128, N=512): wave time stamps of one k3_sweep launch (100 MHz wall clock, the
-DPS_K3_AB build: PROTSTRUC_AMD_LIB=protstruc_amd/lib/libprotstruc_hip_ab.so python3 tools/k3_stamps.py [faithful]).
Per wave: entry, after the first staging barrier, after its first task, after its last task.  Printed: the launch's HIP-event
time and, relative to the first wave's entry, the distribution over the 4 096 waves of each stamp."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from protstruc_amd import _lib, ops

faithful = len(sys.argv) > 1 and sys.argv[1] == "faithful"
B, N = 128, 512
xyz = torch.randn(B, N, 15, 3, generator=torch.Generator().manual_seed(1)).cuda()
out = torch.empty(B, N, N, device="cuda")
lib = _lib.load()
lib.ps_k3_debug_stamps.restype = ctypes.c_int
lib.ps_k3_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
ops.set_exact_angles(faithful)
for name, (npts, si, sj) in {"dihedral (2,2)": (4, [1, 4], [1, 4]), "dihedral (3,1)": (4, [0, 1, 4], [4]), "planar (2,1)": (3, [1, 4], [4])}.items():
    plan = _lib.k3_plan(B, N, 15, si, sj, npts, exact_angles=int(faithful), cu_count=torch.cuda.get_device_properties(0).multi_processor_count)
    waves = plan["threads_per_workgroup"] // 64
    for _ in range(5):
        ops.pairwise_angles(xyz, si, sj, npts, out=out)
    torch.cuda.synchronize()
    us = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.pairwise_angles(xyz, si, sj, npts, out=out); e1.record(); e1.synchronize()
        us.append(e0.elapsed_time(e1) * 1e3)
    buf = np.zeros(512 * 16 * 4, dtype=np.uint64)
    assert lib.ps_k3_debug_stamps(buf.ctypes.data, buf.size) == 0
    st = buf.reshape(512, 16, 4)[:plan["n_workgroups"], :waves].astype(np.int64)
    t0 = st[:, :, 0].min()
    rel = (st - t0) / 100.0                                   # us
    q = lambda a: "min %5.1f  p10 %5.1f  median %5.1f  p90 %5.1f  max %5.1f" % (a.min(), np.percentile(a, 10), np.median(a), np.percentile(a, 90), a.max())
    print(f"== {name}  {plan['kernel']}  {plan['n_workgroups']} workgroups x {waves} waves;  HIP events: median {np.median(us):.1f} us, min {min(us):.1f}")
    print("  wave entry            ", q(rel[:, :, 0]))
    print("  rows staged (barrier) ", q(rel[:, :, 1]))
    print("  first task done       ", q(rel[:, :, 2]))
    print("  last task done        ", q(rel[:, :, 3]))
    print("  staging per workgroup ", q((rel[:, :, 1] - rel[:, :, 0]).max(axis=1)))
    wg_end = rel[:, :, 3].max(axis=1)
    print("  workgroup end         ", q(wg_end), f"  (last - median: {wg_end.max() - np.median(wg_end):.1f} us)")
    per_wave = rel[:, :, 3] - rel[:, :, 1]
    print("  compute per wave      ", q(per_wave))
    print("  within-workgroup spread of the waves' ends", q(rel[:, :, 3].max(axis=1) - rel[:, :, 3].min(axis=1)))
ops.set_exact_angles(False)
