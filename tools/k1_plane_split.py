#!/usr/bin/python3
"""Does the slow allocation class come from the two output streams (fp32 plane + byte plane) being written together?
Headline shape, `n_buffers` output-buffer pairs in one process; per pair: K1 both planes in one launch, the distance plane
alone, the mask plane alone (ps_pairwise_distance_f32 with the other pointer NULL), the two single-plane launches back to back,
and torch.fill_ of each plane.  Usage: python3 tools/k1_plane_split.py [n_buffers]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.pop("PROTSTRUC_AMD_AUTOTUNE", None)
import torch

from protstruc_amd import _lib, ops

nbuf = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B, N, A = 64, 512, 15
g = torch.Generator().manual_seed(0)
xyz = torch.randn(B, N, A, 3, generator=g).cuda()
mask = (torch.rand(B, N, A, generator=g) < 0.9).cuda()
bufs = [(torch.empty(B, N, N, A, A, device="cuda"), torch.empty(B, N, N, A, A, dtype=torch.bool, device="cuda"))
        for _ in range(nbuf)]
cfgs = {"jt128+8KB": dict(k1_jt=128, k1_lds_pad_kb=8, k1_xcd_remap=1), "jt32+20KB": dict(k1_jt=32, k1_lds_pad_kb=20, k1_xcd_remap=1),
        # without the XCD-contiguous map consecutive workgroups (= consecutive runs) go to different XCDs, like torch.fill_'s;
        # the 32-residue tile's mask runs (7200 B) then share 128-byte lines across XCDs, the 128-residue tile's (28800 B) do not
        "jt32+20KB noremap": dict(k1_jt=32, k1_lds_pad_kb=20, k1_xcd_remap=0),
        "jt32+0KB noremap": dict(k1_jt=32, k1_lds_pad_kb=0, k1_xcd_remap=0),
        "jt128+8KB noremap": dict(k1_jt=128, k1_lds_pad_kb=8, k1_xcd_remap=0),
        "jt128+0KB noremap": dict(k1_jt=128, k1_lds_pad_kb=0, k1_xcd_remap=0)}


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
torch.cuda.synchronize()
for k, (d, m) in enumerate(bufs):
    fd = timed(lambda: d.fill_(1.0)); fm = timed(lambda: m.fill_(True))
    print(f"buf{k} fill: dist {fd:.3f} ms ({d.numel() * 4 / fd / 1e9:.2f} TB/s)  mask {fm:.3f} ms ({m.numel() / fm / 1e9:.2f} TB/s)  sum {fd + fm:.3f}", flush=True)
    for name, c in cfgs.items():
        for kk, v in c.items():
            _lib.set_tuning(kk, v)
        both = timed(lambda: ops.pairwise_distance(xyz, mask, out_dist=d, out_mask=m))
        donly = timed(lambda: ops.pairwise_distance(xyz, mask, out_dist=d, want_mask=False))
        monly = timed(lambda: ops.pairwise_distance(xyz, mask, out_mask=m, want_dist=False))
        two = timed(lambda: (ops.pairwise_distance(xyz, mask, out_dist=d, want_mask=False),
                             ops.pairwise_distance(xyz, mask, out_mask=m, want_dist=False)))
        nb = d.numel()
        print(f"buf{k} {name:18s} both {both:.3f} ms ({nb * 5 / both / 1e9:.2f} TB/s) | dist only {donly:.3f} ({nb * 4 / donly / 1e9:.2f}) | "
              f"mask only {monly:.3f} ({nb / monly / 1e9:.2f}) | two launches {two:.3f} ({nb * 5 / two / 1e9:.2f})", flush=True)

# second part (argv[2] == "grid"): the single-plane launches over tile length x idle LDS, first and last buffer only
if len(sys.argv) > 2 and sys.argv[2] == "grid":
    for k in sorted({0, nbuf - 1}):
        d, m = bufs[k]
        for jt in (32, 64, 128):
            for pad in (0, 8, 16, 24, 32, 48):
                _lib.set_tuning("k1_jt", jt); _lib.set_tuning("k1_lds_pad_kb", pad)
                both = timed(lambda: ops.pairwise_distance(xyz, mask, out_dist=d, out_mask=m), 6)
                donly = timed(lambda: ops.pairwise_distance(xyz, mask, out_dist=d, want_mask=False), 6)
                monly = timed(lambda: ops.pairwise_distance(xyz, mask, out_mask=m, want_dist=False), 6)
                nb = d.numel()
                print(f"buf{k} jt{jt:3d} +{pad:2d}KB  both {both:.3f} ({nb * 5 / both / 1e9:.2f}) | dist only {donly:.3f} ({nb * 4 / donly / 1e9:.2f}) | "
                      f"mask only {monly:.3f} ({nb / monly / 1e9:.2f}) | sum {donly + monly:.3f} ({nb * 5 / (donly + monly) / 1e9:.2f})", flush=True)
