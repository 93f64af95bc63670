"""K1 store rate for atom counts other than 15 (any-A flat kernel; flat=0 selects the element-per-lane kernel).
Arguments: key=value K1 tuning, e.g. flat=0 or anya_fl_log2=7."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protstruc_amd import _lib, ops
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    _lib.set_tuning("k1_" + k, int(v))
g = torch.Generator().manual_seed(0)
for A, N in [(15, 256), (14, 256), (4, 512), (5, 512), (8, 256), (3, 512), (1, 1024), (25, 128), (37, 128), (16, 256)]:
    B = max(1, int(8e9 / (N * N * A * A * 5)))
    xyz = torch.randn(B, N, A, 3, generator=g).cuda()
    mask = (torch.rand(B, N, A, generator=g) < 0.9).cuda()
    d = torch.empty(B, N, N, A, A, device="cuda"); m = torch.empty(B, N, N, A, A, dtype=torch.bool, device="cuda")
    try:
        for _ in range(5): ops.pairwise_distance(xyz, mask, out_dist=d, out_mask=m)
    except Exception as exc:   # e.g. a forced chunk length whose LDS image does not fit
        print(f"A={A:3d} N={N:5d}: {exc}"); continue
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.pairwise_distance(xyz, mask, out_dist=d, out_mask=m)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"A={A:3d} N={N:5d} B={B:5d}  {ms:8.3f} ms  {B*N*N*A*A*5/ms/1e9:6.2f} TB/s", flush=True)
    del xyz, mask, d, m
