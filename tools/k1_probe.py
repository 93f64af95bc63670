#!/usr/bin/env python3
"""Quick K1 timing probe on the GPU box: variants x rounds, interleaved in one process."""
import json
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os
os.environ.pop("PROTSTRUC_AMD_AUTOTUNE", None)   # no implicit tuning while measuring
import torch
from protstruc_amd import _lib, ops

B, N, A = 64, 512, 15
g = torch.Generator().manual_seed(0)
xyz = torch.randn(B, N, A, 3, generator=g).cuda()
mask = (torch.rand(B, N, A, generator=g) < 0.9)
mask[:, :, :3] = True
mask = mask.cuda()
dist = torch.empty(B, N, N, A, A, device="cuda")
dmask = torch.empty(B, N, N, A, A, dtype=torch.bool, device="cuda")
pairs = B * N * N

def run(want_dist=True, want_mask=True):
    ops.pairwise_distance(xyz, mask, out_dist=dist if want_dist else None, out_mask=dmask if want_mask else None,
                          want_dist=want_dist, want_mask=want_mask)

variants = []
import itertools
ROWS = [int(r) for r in os.environ.get("K1_ROWS", "1,2,4").split(",")]
for var in (0, 1):
    for jt in (64, 128):
        for nt in (1, 0):
            for rows in ROWS:
                variants.append((var, jt, nt, rows))
res = {}
for rnd in range(3):
    for var, jt, nt, rows in variants:
        _lib.set_tuning("k1_variant", var)
        _lib.set_tuning("k1_jt", jt)
        _lib.set_tuning("k1_store_nt", nt)
        _lib.set_tuning("k1_rows_per_block", rows)
        for what in ("both", "dist", "mask"):
            wd, wm = what != "mask", what != "dist"
            run(wd, wm)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                run(wd, wm)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            res.setdefault((var, jt, nt, rows, what), []).append(ms)
for (var, jt, nt, rows, what), v in sorted(res.items()):
    ms = min(v)
    nbytes = pairs * {"both": 1125, "dist": 900, "mask": 225}[what]
    print(f"var={var} jt={jt:3d} nt={nt} rows={rows:2d} {what:5s} min {ms:7.3f} ms  med {sorted(v)[len(v)//2]:7.3f} ms  {nbytes/ms/1e9:7.2f} TB/s  {pairs/ms/1e6:7.2f} Gpairs/s")
# memset / copy ceilings for context
buf = torch.empty(pairs * 1125 // 4, dtype=torch.float32, device="cuda")
for name, fn in (("fill", lambda: buf.fill_(1.0)),):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"torch {name}: {ms:.3f} ms  {buf.numel()*4/ms/1e9:.2f} TB/s (write-only ceiling reference)")
