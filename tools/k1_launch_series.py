#!/usr/bin/python3
"""Launch-to-launch spread of K1 at the headline shape: 40 back-to-back launches of each configuration on each of
`n_buffers` output-buffer pairs, one HIP event between every two launches, the whole series printed.  The tuner keeps
each candidate's MINIMUM; the bench's timed region reports the MEAN of 20 launches -- this shows how far apart the two
are per configuration.  Usage: python3 tools/k1_launch_series.py [n_buffers] [launches]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.pop("PROTSTRUC_AMD_AUTOTUNE", None)
import torch

from protstruc_amd import _lib, ops

nbuf = int(sys.argv[1]) if len(sys.argv) > 1 else 3
L = int(sys.argv[2]) if len(sys.argv) > 2 else 40
B, N, A = 64, 512, 15
g = torch.Generator().manual_seed(0)
xyz = torch.randn(B, N, A, 3, generator=g).cuda()
mask = (torch.rand(B, N, A, generator=g) < 0.9).cuda()
cfgs = {
    "default (jt32 +20KB)": dict(k1_jt=0, k1_lds_pad_kb=20),
    "jt128 +8KB": dict(k1_jt=128),
    "jt128 +0KB": dict(k1_jt=128, k1_lds_pad_kb=0),
    "jt64 +8KB": dict(k1_jt=64),
    "jt32 +0KB": dict(k1_jt=32, k1_lds_pad_kb=0),
    "jt32 +18KB": dict(k1_jt=32, k1_lds_pad_kb=18),
    "jt32 +24KB": dict(k1_jt=32, k1_lds_pad_kb=24),
    "jt32 +32KB": dict(k1_jt=32, k1_lds_pad_kb=32),
    "jt32 +24KB noremap": dict(k1_jt=32, k1_lds_pad_kb=24, k1_xcd_remap=0),
    "jt32 +24KB r2": dict(k1_jt=32, k1_lds_pad_kb=24, k1_rows_per_block=2),
}
DEFAULTS = dict(k1_rows_per_block=1, k1_jt=0, k1_lds_pad_kb=8, k1_flat=1, k1_xcd_remap=1)
bufs = [(torch.empty(B, N, N, A, A, device="cuda"), torch.empty(B, N, N, A, A, dtype=torch.bool, device="cuda"))
        for _ in range(nbuf)]
for _ in range(40):
    ops.pairwise_distance(xyz, mask, out_dist=bufs[0][0], out_mask=bufs[0][1])
torch.cuda.synchronize()
nb = B * N * N * A * A * 5
for k, (d, m) in enumerate(bufs):
    for name, c in cfgs.items():
        for kk, v in {**DEFAULTS, **c}.items():
            _lib.set_tuning(kk, v)
        for _ in range(3):
            ops.pairwise_distance(xyz, mask, out_dist=d, out_mask=m)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(L + 1)]
        ev[0].record()
        for i in range(L):
            ops.pairwise_distance(xyz, mask, out_dist=d, out_mask=m)
            ev[i + 1].record()
        torch.cuda.synchronize()
        ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(L)]
        s = sorted(ms)
        print(f"buf{k} {name:22s} min {s[0]:.3f} med {s[L // 2]:.3f} mean {sum(ms) / L:.3f} max {s[-1]:.3f}  "
              f"TB/s mean {nb / (sum(ms) / L) / 1e9:.2f} min-based {nb / s[0] / 1e9:.2f} | " + " ".join(f"{v:.2f}" for v in ms), flush=True)
