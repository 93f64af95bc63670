#!/usr/bin/env python3
"""K1 experiments: where is the time going?  math modes x tile x rows, interleaved rounds.
Needs the experiments build (store-only mode writes WRONG values and is not compiled into the product library):
    python -m protstruc_amd.build --experiments
    PROTSTRUC_AMD_LIB=protstruc_amd/lib/libprotstruc_hip_experiments.so python tools/k1_probe2.py
m = 0: correctly rounded sqrt, 1: hardware sqrt (product default), 2: store-only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os
os.environ.pop("PROTSTRUC_AMD_AUTOTUNE", None)   # no implicit tuning while measuring
import torch
from protstruc_amd import _lib, ops
B, N, A = 64, 512, 15
g = torch.Generator().manual_seed(0)
xyz = torch.randn(B, N, A, 3, generator=g).cuda()
mask = (torch.rand(B, N, A, generator=g) < 0.9); mask[:, :, :3] = True; mask = mask.cuda()
dist = torch.empty(B, N, N, A, A, device="cuda")
dmask = torch.empty(B, N, N, A, A, dtype=torch.bool, device="cuda")
pairs = B * N * N
def run(wd, wm):
    ops.pairwise_distance(xyz, mask, out_dist=dist if wd else None, out_mask=dmask if wm else None, want_dist=wd, want_mask=wm)
res = {}
variants = [(m, jt, rows) for m in (0, 1, 2) for jt in (64, 128) for rows in (1, 2)]
for rnd in range(3):
    for m, jt, rows in variants:
        _lib.set_tuning("k1_exact_sqrt", int(m == 0)); _lib.set_tuning("k1_experiment", 2 if m == 2 else 0); _lib.set_tuning("k1_jt", jt); _lib.set_tuning("k1_rows_per_block", rows)
        for what in ("both", "dist", "mask"):
            if what == "mask" and m: continue
            wd, wm = what != "mask", what != "dist"
            for _ in range(3): run(wd, wm)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): run(wd, wm)
            e1.record(); torch.cuda.synchronize()
            res.setdefault((m, jt, rows, what), []).append(e0.elapsed_time(e1) / 10)
for (m, jt, rows, what), v in sorted(res.items()):
    ms = min(v); nb = pairs * {"both": 1125, "dist": 900, "mask": 225}[what]
    print(f"math={m} jt={jt:3d} rows={rows} {what:5s} min {ms:6.3f} ms  {nb/ms/1e9:5.2f} TB/s")
