#!/usr/bin/python3
"""Which K1 launch configuration is fastest on WHICH output buffer?  Allocates several output-buffer pairs of the
headline shape in one process (they land in physical memory of different "speed classes") and
times a list of configurations on each, interleaved (unnamed knobs: the 128-residue tile + 8 KB that was the default
until late round 3).  Informs the autotuner's candidate list; results are identical
for every configuration.  Usage: python3 tools/k1_ab_buffers.py [n_buffers]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.pop("PROTSTRUC_AMD_AUTOTUNE", None)
import torch

from protstruc_amd import _lib, ops

nbuf = int(sys.argv[1]) if len(sys.argv) > 1 else 6
B, N, A = 64, 512, 15
g = torch.Generator().manual_seed(0)
xyz = torch.randn(B, N, A, 3, generator=g).cuda()
mask = (torch.rand(B, N, A, generator=g) < 0.9).cuda()
cfgs = {
    "r1 jt128": dict(k1_rows_per_block=1, k1_jt=128, k1_lds_pad_kb=0, k1_flat=1),
    "r1 jt64": dict(k1_rows_per_block=1, k1_jt=64, k1_lds_pad_kb=0, k1_flat=1),
    "r1 jt128 +8KB": dict(k1_rows_per_block=1, k1_jt=128, k1_lds_pad_kb=8, k1_flat=1),
    "r1 jt64 +8KB": dict(k1_rows_per_block=1, k1_jt=64, k1_lds_pad_kb=8, k1_flat=1),
    "r2 jt128": dict(k1_rows_per_block=2, k1_jt=128, k1_lds_pad_kb=0, k1_flat=1),
    "r2 jt64": dict(k1_rows_per_block=2, k1_jt=64, k1_lds_pad_kb=0, k1_flat=1),
    "r4 jt64": dict(k1_rows_per_block=4, k1_jt=64, k1_lds_pad_kb=0, k1_flat=1),
    "flat": dict(k1_rows_per_block=1, k1_jt=128, k1_lds_pad_kb=0, k1_flat=2),
    # stronger residency caps: 36 KB + pad of LDS per workgroup -> 2 workgroups per CU from 18 KB, 1 from 45 KB
    "r1 jt128 +20KB": dict(k1_rows_per_block=1, k1_jt=128, k1_lds_pad_kb=20, k1_flat=1),
    "r1 jt128 +48KB": dict(k1_rows_per_block=1, k1_jt=128, k1_lds_pad_kb=48, k1_flat=1),
    "r1 jt64 +36KB": dict(k1_rows_per_block=1, k1_jt=64, k1_lds_pad_kb=36, k1_flat=1),
    # round 3, the bounded small-granule retry: the flat pattern kernel (today's inner loop: hardware sqrt, 34 VALU per
    # slot) with 64 / 32 / 16 pairs per chunk = 72 / 36 / 18 KB of output per short-lived workgroup instead of 144 KB
    "flat 64p": dict(k1_flat=2, k1_flat_fl_log2=6), "flat 32p": dict(k1_flat=2, k1_flat_fl_log2=5),
    "flat 16p": dict(k1_flat=2, k1_flat_fl_log2=4),
    "flat 32p +8KB": dict(k1_flat=2, k1_flat_fl_log2=5, k1_flat_lds_pad_kb=8),
    "flat 16p +8KB": dict(k1_flat=2, k1_flat_fl_log2=4, k1_flat_lds_pad_kb=8),
    "flat 16p x2": dict(k1_flat=2, k1_flat_fl_log2=4, k1_flat_cpw=2),
    "flat 16p x4": dict(k1_flat=2, k1_flat_fl_log2=4, k1_flat_cpw=4),
    # round 3: the PATTERN kernel with 32- / 16-residue tiles: 36 / 18 KB of contiguous output per short-lived workgroup,
    # consecutive workgroups at consecutive addresses (the store pattern of torch.fill_), 7 workgroups per CU
    "pat jt32": dict(k1_jt=32, k1_lds_pad_kb=0), "pat jt32 +8KB": dict(k1_jt=32, k1_lds_pad_kb=8),
    "pat jt32 noremap": dict(k1_jt=32, k1_lds_pad_kb=0, k1_xcd_remap=0), 
    "pat jt64": dict(k1_jt=64, k1_lds_pad_kb=0),
    # ... with residency caps (idle LDS): 8.4 KB + pad per workgroup of the 160 KB per CU
    "cap jt32 +16KB": dict(k1_jt=32, k1_lds_pad_kb=16), "cap jt32 +24KB": dict(k1_jt=32, k1_lds_pad_kb=24),
    "cap jt32 +32KB": dict(k1_jt=32, k1_lds_pad_kb=32), "cap jt32 +44KB": dict(k1_jt=32, k1_lds_pad_kb=44),
    "cap jt32 +64KB": dict(k1_jt=32, k1_lds_pad_kb=64), "cap jt64 +16KB": dict(k1_jt=64, k1_lds_pad_kb=16),
    "cap jt64 +24KB": dict(k1_jt=64, k1_lds_pad_kb=24), "cap jt128 +24KB": dict(k1_jt=128, k1_lds_pad_kb=24),
    "fine jt32 +18KB": dict(k1_jt=32, k1_lds_pad_kb=18), "fine jt32 +20KB": dict(k1_jt=32, k1_lds_pad_kb=20),
    "fine jt32 +22KB": dict(k1_jt=32, k1_lds_pad_kb=22), "fine jt32 +24KB": dict(k1_jt=32, k1_lds_pad_kb=24),
    "fine jt32 +26KB": dict(k1_jt=32, k1_lds_pad_kb=26), "fine jt32 +28KB": dict(k1_jt=32, k1_lds_pad_kb=28),
    "fine r2 jt32 +24KB": dict(k1_jt=32, k1_lds_pad_kb=24, k1_rows_per_block=2),
    "fine r2 jt32 +16KB": dict(k1_jt=32, k1_lds_pad_kb=16, k1_rows_per_block=2),
    "fine jt64 +20KB": dict(k1_jt=64, k1_lds_pad_kb=20), "fine jt64 +12KB": dict(k1_jt=64, k1_lds_pad_kb=12),
    # round 3: the row-phase kernel at A = 15 (k1_rowphase = 1), rows per lane 4 .. 32: 8 KB + 2 KB pieces of that many rows
    "rowphase r4": dict(k1_rowphase=1, k1_rows_per_block=4), "rowphase r6": dict(k1_rowphase=1, k1_rows_per_block=6),
    "rowphase r8": dict(k1_rowphase=1, k1_rows_per_block=8), "rowphase r12": dict(k1_rowphase=1, k1_rows_per_block=12),
    "rowphase r16": dict(k1_rowphase=1, k1_rows_per_block=16), "rowphase r32": dict(k1_rowphase=1, k1_rows_per_block=32),
}
DEFAULTS = dict(k1_rows_per_block=1, k1_jt=128, k1_lds_pad_kb=8, k1_flat=1, k1_flat_fl_log2=0, k1_flat_lds_pad_kb=0, k1_flat_cpw=1,
                k1_rowphase=0, k1_xcd_remap=1)
if len(sys.argv) > 2:      # a comma-separated subset of configuration names
    cfgs = {k: v for k, v in cfgs.items() if any(k.startswith(p) for p in sys.argv[2].split(","))}
bufs = [(torch.empty(B, N, N, A, A, device="cuda"), torch.empty(B, N, N, A, A, dtype=torch.bool, device="cuda"))
        for _ in range(nbuf)]


def apply(c):
    for k, v in {**DEFAULTS, **c}.items():
        _lib.set_tuning(k, v)


for _ in range(30):
    ops.pairwise_distance(xyz, mask, out_dist=bufs[0][0], out_mask=bufs[0][1])
torch.cuda.synchronize()
best = {(n, k): float("inf") for n in cfgs for k in range(nbuf)}
for rnd in range(3):
    for k, (d, m) in enumerate(bufs):
        for name, c in cfgs.items():
            apply(c)
            ops.pairwise_distance(xyz, mask, out_dist=d, out_mask=m)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                ops.pairwise_distance(xyz, mask, out_dist=d, out_mask=m)
            e1.record(); torch.cuda.synchronize()
            best[(name, k)] = min(best[(name, k)], e0.elapsed_time(e1) / 3)
nb = B * N * N * A * A * 5
# the class of every buffer, by the rate torch.fill_ writes it
fill = []
for d, m in bufs:
    d.fill_(0.0); m.fill_(False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        d.fill_(0.0); m.fill_(False)
    e1.record(); torch.cuda.synchronize()
    fill.append(nb / (e0.elapsed_time(e1) / 3) / 1e9)
print(f"{'config':16s}" + "".join(f"  buf{k}" for k in range(nbuf)) + "   (TB/s)")
print(f"{'torch.fill_':16s}" + "".join(f" {f:5.2f}" for f in fill), flush=True)
for name in cfgs:
    print(f"{name:16s}" + "".join(f" {nb / best[(name, k)] / 1e9:5.2f}" for k in range(nbuf)), flush=True)
