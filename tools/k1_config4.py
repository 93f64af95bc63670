#!/usr/bin/env python3
"""BASELINE config 4 (B=32, N=2048: 134.2 M residue pairs, 151 GB of output) on ONE MI355X.

Times the kernel part of the residue-sharded run for P = 1, 2, 4, 8 ranks -- rank r of P computes rows
[r*N/P, (r+1)*N/P) of every structure -- by running one rank's shard here (compact buffer, so P = 8 needs 18.9 GB
and P = 1 the whole 151 GB), and checks whole-tensor properties of the P = 1 result.  The all-gather that follows on
a real node is not part of this script (bench.py --gpus N measures it on N GPUs).  Writes JSON to gpurun_out/."""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protstruc_amd import ops

B, N, A = 32, 2048, 15
g = torch.Generator().manual_seed(0)
xyz = torch.randn(B, N, A, 3, generator=g)
mask = torch.rand(B, N, A, generator=g) < 0.9
mask[:, :, :3] = True
xg, mg = xyz.cuda(), mask.cuda()
out = {"device": torch.cuda.get_device_name(0), "B": B, "N": N, "shards": []}


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


# warm the clocks / run the per-device autotune on the headline shape first
ops.pairwise_distance(xg[:, :512].contiguous(), mg[:, :512].contiguous())
torch.cuda.synchronize()
torch.cuda.empty_cache()
for P in (8, 1, 2, 4, 8):
    rows = N // P
    d = torch.empty(B, rows, N, A, A, device="cuda")
    m = torch.empty(B, rows, N, A, A, dtype=torch.bool, device="cuda")
    for r in sorted({0, P // 2}):
        med, mn = timed(lambda: ops.pairwise_distance(xg, mg, row_begin=r * rows, row_end=(r + 1) * rows, compact=True,
                                                      out_dist=d, out_mask=m), 10 if P > 1 else 5)
        pairs = B * rows * N
        out["shards"].append({"P": P, "rank": r, "rows": rows, "pairs": pairs, "ms_med": med, "ms_min": mn,
                              "TBps": pairs * 1125 / med / 1e9, "Gpairs_per_s": pairs / med / 1e6})
        print(out["shards"][-1], flush=True)
    if P == 1:
        # properties of the full 151 GB result
        per_struct = mask.reshape(B, -1).sum(1).to(torch.int64)
        got_cnt = torch.stack([torch.count_nonzero(m[b]) for b in range(B)]).cpu()   # no 8-byte temporaries
        ok_mask = torch.equal(got_cnt, per_struct * per_struct)
        gs = torch.Generator().manual_seed(1)
        bs = torch.randint(0, B, (256,), generator=gs)
        is_ = torch.cat([torch.randint(0, N, (252,), generator=gs), torch.tensor([0, N - 1, N - 1, 0])])
        js = torch.cat([torch.randint(0, N, (252,), generator=gs), torch.tensor([0, N - 1, 0, N - 1])])
        want = torch.norm(xyz[bs, is_][:, :, None, :] - xyz[bs, js][:, None, :, :], dim=-1)
        got = d[bs.cuda(), is_.cuda(), js.cuda()].cpu()
        err = float((got - want).abs().max())
        sym = torch.equal(d[B - 1, :256, :256], d[B - 1, :256, :256].permute(1, 0, 3, 2))
        last = torch.equal(d[B - 1, N - 1, N - 1].cpu(),
                           torch.norm(xyz[B - 1, N - 1][:, None, :] - xyz[B - 1, N - 1][None, :, :], dim=-1)) or \
            float((d[B - 1, N - 1, N - 1].cpu() -
                   torch.norm(xyz[B - 1, N - 1][:, None, :] - xyz[B - 1, N - 1][None, :, :], dim=-1)).abs().max()) <= 1e-5
        out["full_checks"] = {"mask_checksum_exact": bool(ok_mask), "sampled_blocks_max_abs_err": err,
                              "symmetric_block": bool(sym), "last_block_ok": bool(last)}
        print(out["full_checks"], flush=True)
        assert ok_mask and err <= 1e-5 and sym and last
    del d, m
    torch.cuda.empty_cache()
out["autotune"] = ops.k1_autotune_result()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/k1_config4.json", "w"), indent=1)
print("ok")
