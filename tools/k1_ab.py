#!/usr/bin/env python3
"""Interleaved A/B of K1 tuning configurations in ONE process (rule: never compare across runs/boxes).
usage: k1_ab.py "variant=0,jt=64,rows_per_block=1,unroll=0" "variant=0,jt=64,rows_per_block=1,unroll=1" ..."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.pop("PROTSTRUC_AMD_AUTOTUNE", None)   # this harness sets every knob itself
import torch
from protstruc_amd import _lib, ops
B, N, A = int(os.environ.get("K1_B", "64")), int(os.environ.get("K1_N", "512")), 15
g = torch.Generator().manual_seed(0)
xyz = torch.randn(B, N, A, 3, generator=g).cuda()
mask = (torch.rand(B, N, A, generator=g) < 0.9); mask[:, :, :3] = True; mask = mask.cuda()
dist = torch.empty(B, N, N, A, A, device="cuda")
dmask = torch.empty(B, N, N, A, A, dtype=torch.bool, device="cuda")
DEFAULT = dict(variant=0, jt=0, rows_per_block=1, store_nt=0, math=0, unroll=0, lds_pad_kb=0, xcd_remap=1, flat=1, exact_sqrt=0,
               flat_cpw=1)
cfgs = []
for arg in sys.argv[1:]:
    c = dict(DEFAULT)
    for kv in arg.split(","):
        k, v = kv.split("="); c[k] = int(v)
    cfgs.append((arg, c))
rounds = int(os.environ.get("ROUNDS", "7"))
res = {a: [] for a, _ in cfgs}
for r in range(rounds):
    for a, c in cfgs:
        for k, v in c.items():
            _lib.set_tuning("k1_" + k, v)
        for _ in range(2):
            ops.pairwise_distance(xyz, mask, out_dist=dist, out_mask=dmask)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.pairwise_distance(xyz, mask, out_dist=dist, out_mask=dmask)
        e1.record(); torch.cuda.synchronize()
        res[a].append(e0.elapsed_time(e1) / 10)
nb = B * N * N * 1125
for a, v in res.items():
    print(f"{a:60s} min {min(v):6.3f} med {statistics.median(v):6.3f} ms   {nb/statistics.median(v)/1e9:5.2f} TB/s (med)")
