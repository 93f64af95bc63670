#!/usr/bin/python3
"""Rows per lane of K1's row-phase kernel (ps_k1_config.rows_per_block > 1 overrides the default): TB/s for several
values on the SAME buffers in one process, interleaved, with the default dispatch and torch.fill_ alongside.
Usage: python3 tools/k1_rowphase_rows.py [A:N ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from protstruc_amd import _lib, ops

shapes = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or [
    (1, 512), (1, 501), (2, 512), (3, 512), (3, 500), (5, 512), (5, 500), (5, 501), (7, 500), (10, 500), (13, 250), (20, 125),
    (33, 100), (25, 128), (37, 128), (64, 64)]
if os.environ.get("K1_ROWPHASE"):         # 1: the row-phase kernel also where a fixed-A flat kernel is the default
    _lib.set_tuning("k1_rowphase", int(os.environ["K1_ROWPHASE"]))
ROWS = (0, 4, 6, 8, 12, 16, 24, 32)      # 0 = the library's default
if os.environ.get("K1_ROWS"):
    ROWS = tuple(int(v) for v in os.environ["K1_ROWS"].split(","))
g = torch.Generator().manual_seed(0)
print("rows per lane:      " + "  ".join(f"{'dflt' if r == 0 else r:>5}" for r in ROWS) + "   fill   (TB/s)")
for A, N in shapes:
    B = max(1, int(8e9 / (N * N * A * A * 5)))
    xyz = torch.randn(B, N, A, 3, generator=g).cuda()
    mask = (torch.rand(B, N, A, generator=g) < 0.9).cuda()
    d = torch.empty(B, N, N, A, A, device="cuda")
    m = torch.empty(B, N, N, A, A, dtype=torch.bool, device="cuda")
    best = {r: float("inf") for r in ROWS}
    best["fill"] = float("inf")
    for rnd in range(3):
        for r in list(ROWS) + ["fill"]:
            if r == "fill":
                run = lambda: (d.fill_(0.0), m.fill_(False))
            else:
                _lib.set_tuning("k1_rows_per_block", r if r else 1)
                run = lambda: ops.pairwise_distance(xyz, mask, out_dist=d, out_mask=m)
            run(); run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                run()
            e1.record(); torch.cuda.synchronize()
            best[r] = min(best[r], e0.elapsed_time(e1) / 4)
    _lib.set_tuning("k1_rows_per_block", 1)
    nb = B * N * N * A * A * 5
    print(f"A={A:3d} N={N:4d} B={B:5d}  " + "  ".join(f"{nb / best[r] / 1e9:5.2f}" for r in ROWS) + f"  {nb / best['fill'] / 1e9:5.2f}"
          + f"   [{_lib.k1_plan(B, N, A)['kernel']}]", flush=True)
    del xyz, mask, d, m
