#!/usr/bin/python3
"""Is one launch per plane faster than the fused launch?  Headline shape, several buffers held at once; per buffer the
fused launch at two configurations against "distance plane with 32-residue tiles, then mask plane with 128-residue tiles"
(two calls of ops.pairwise_distance with want_mask / want_dist False).  Usage: python3 tools/k1_split_probe.py [n_buffers]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.pop("PROTSTRUC_AMD_AUTOTUNE", None)
import torch

from protstruc_amd import _lib, ops

nbuf = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B, N, A = 64, 512, 15
g = torch.Generator().manual_seed(0)
xyz = torch.randn(B, N, A, 3, generator=g).cuda()
mask = (torch.rand(B, N, A, generator=g) < 0.9).cuda()
bufs = [(torch.empty(B, N, N, A, A, device="cuda"), torch.empty(B, N, N, A, A, dtype=torch.bool, device="cuda")) for _ in range(nbuf)]
nb = B * N * N * A * A * 5


def setc(jt, pad):
    _lib.set_tuning("k1_jt", jt); _lib.set_tuning("k1_lds_pad_kb", pad)


def timed(fn, reps=10):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for _ in range(40):
    ops.pairwise_distance(xyz, mask, out_dist=bufs[0][0], out_mask=bufs[0][1])
for k, (d, m) in enumerate(bufs):
    def fused(jt, pad):
        setc(jt, pad)
        return timed(lambda: ops.pairwise_distance(xyz, mask, out_dist=d, out_mask=m))

    def split(jt, pad):
        def step():
            setc(jt, pad)
            ops.pairwise_distance(xyz, mask, out_dist=d, want_mask=False)
            setc(128, 8)
            ops.pairwise_distance(xyz, mask, out_mask=m, want_dist=False)
        return timed(step)
    res = {"fused jt32+20": fused(32, 20), "fused jt32+36": fused(32, 36), "fused jt128+8": fused(128, 8),
           "split jt32+20 | jt128+8": split(32, 20), "split jt32+36 | jt128+8": split(32, 36), "split jt128+8 | jt128+8": split(128, 8)}
    print(f"buf{k}  " + "  ".join(f"{n} {nb / t / 1e9:.2f}" for n, t in res.items()), flush=True)
