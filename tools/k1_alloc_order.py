#!/usr/bin/python3
"""Does the ORDER of allocation decide a buffer's speed class?  Headline shape, default K1 configuration.  Round 1: four
(dist, mask) pairs allocated one after the other and all held; each timed (mean of 10 launches) and filled.  Then
everything is released (torch.cuda.empty_cache) and round 2 allocates four pairs again.  Prints the device addresses.
Usage: python3 tools/k1_alloc_order.py [pairs per round] [rounds]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.pop("PROTSTRUC_AMD_AUTOTUNE", None)
import torch

from protstruc_amd import ops

npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
B, N, A = 64, 512, 15
g = torch.Generator().manual_seed(0)
xyz = torch.randn(B, N, A, 3, generator=g).cuda()
mask = (torch.rand(B, N, A, generator=g) < 0.9).cuda()
nb = B * N * N * A * A


def timed(fn, reps=10):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print(f"free / total before anything: {[round(x / 2**30, 1) for x in torch.cuda.mem_get_info()]} GiB", flush=True)
for r in range(rounds):
    pairs = []
    for k in range(npairs):
        d = torch.empty(B, N, N, A, A, device="cuda")
        m = torch.empty(B, N, N, A, A, dtype=torch.bool, device="cuda")
        pairs.append((d, m))
        if r == 0 and k == 0:
            for _ in range(30):
                ops.pairwise_distance(xyz, mask, out_dist=d, out_mask=m)     # clock ramp
        t = timed(lambda: ops.pairwise_distance(xyz, mask, out_dist=d, out_mask=m))
        f = timed(lambda: (d.fill_(0.0), m.fill_(False)), 4)
        print(f"round {r} pair {k}: dist {d.data_ptr():#x} mask {m.data_ptr():#x}  K1 {nb * 5 / t / 1e9:.2f} TB/s  fill {nb * 5 / f / 1e9:.2f} TB/s", flush=True)
    # a second look at the first pair of the round, after the others exist
    d, m = pairs[0]
    t = timed(lambda: ops.pairwise_distance(xyz, mask, out_dist=d, out_mask=m))
    print(f"round {r} pair 0 again: K1 {nb * 5 / t / 1e9:.2f} TB/s", flush=True)
    del pairs, d, m
    torch.cuda.empty_cache()
