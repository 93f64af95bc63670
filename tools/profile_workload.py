#!/usr/bin/python3
"""Workloads for rocprofv3 (run as `rocprofv3 ... -- python3 tools/profile_workload.py <what> [reps]`):

    featshort  the featuriser at 2^25 pairs for N = 64, 48, 160 (tile kernel), 128 and 512 (sweep)
    k3flat  K3 dihedral (2,2) and planar (2,1) at 2^25 pairs for N = 64, 99, 48 (flat kernel) and 512 (sweep)
    k3p     the same as k3, host-paced (a synchronise after every launch)
    k3      BASELINE config 3 (B=128, N_res=512): pairwise_dihedrals (2,2) CA,CB|CA,CB and (3,1) N,CA,CB|CB,
            pairwise_planar_angles (2,1) CA,CB|CB, and the fused inter_residue_geometry -- in both arithmetic modes
    k1a     K1 at atom14 (N=256) and atom37 (N=128), ~8 GB of output each: default dispatch (fixed-A flat pattern
            kernel / row-phase kernel) and the row-phase kernel forced (k1_rowphase=1)
    k1      the headline K1 launch (B=64, N=512, A=15)
    k1s     K1 at the small atom counts, ~4 GB of output each: (A, N) = (5, 512), (5, 500), (5, 501), (3, 501), (1, 512),
            (2, 512) through the default dispatch (row-phase kernel)
    k1u     K1 at the unaligned shapes (A, N) = (5, 501), (1, 501), (37, 100) and the aligned (5, 512), ~4 GB each, three
            modes per shape in this order: both planes, distance plane only, mask plane only (read with `pmcseq`)
    k1f     the flat short-chain kernels of K1, ~4 GB of output each: (A, N) = (1, 128), (1, 16), (3, 16), (4, 32), (13, 8)
    k5      K5 / K6 / K4 at BASELINE config 5's shape (B=256, N=384): diffuse_xyz (in-kernel Philox), diffuse_xyz with
            injected noise, standardize, backbone_orientations, the fused diffuse + frames step
Every kernel is launched `reps` times (default 10) after 2 warm-ups, nothing else runs on the GPU."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.pop("PROTSTRUC_AMD_AUTOTUNE", None)   # no implicit tuning while measuring
import torch

from protstruc_amd import StructureBatch, _lib, ops

what = sys.argv[1] if len(sys.argv) > 1 else "k3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
g = torch.Generator().manual_seed(0)


def synth(B, N, A=15):
    xyz = torch.randn(B, N, A, 3, generator=g)
    mask = torch.rand(B, N, A, generator=g) < 0.9
    mask[:, :, :3] = True
    return xyz.cuda(), mask.cuda()


def repeat(fn):
    for _ in range(2 + reps):
        fn()
    torch.cuda.synchronize()


if what in ("k3", "k3p"):
    if what == "k3p":        # host-paced: every launch waits for the previous one (a back-to-back train runs at a sagging clock, and
        def repeat(fn):      # what follows the featuriser inherits it: profiles/r05_k3_timeline_back_to_back.csv)
            for _ in range(2 + reps):
                fn()
                torch.cuda.synchronize()
    xyz, mask = synth(128, 512)
    sb = StructureBatch.from_xyz(xyz, mask)
    for faithful in (False, True):      # the fast arithmetic, then the reference's order of operations (kernel names differ: FAITHFUL)
        ops.set_exact_angles(faithful)
        repeat(lambda: sb.pairwise_dihedrals(["CA", "CB"], ["CA", "CB"]))
        repeat(lambda: sb.pairwise_dihedrals(["N", "CA", "CB"], ["CB"]))
        repeat(lambda: sb.pairwise_planar_angles(["CA", "CB"], ["CB"]))
        repeat(lambda: sb.inter_residue_geometry())
    ops.set_exact_angles(False)
elif what == "k3flat":      # the tile K3 kernel at 2^25 pairs: N = 64 (four-column tiles), 99 and 48 (two-column tiles), N = 512 (the sweep) beside it
    for n in (64, 99, 48, 512):
        b = (1 << 25) // (n * n)
        xyz, _ = synth(b, n)
        out = torch.empty(b, n, n, device="cuda")
        repeat(lambda: ops.pairwise_angles(xyz, [1, 4], [1, 4], 4, out=out))
        repeat(lambda: ops.pairwise_angles(xyz, [1, 4], [4], 3, out=out))
elif what == "featshort":   # the featuriser at 2^25 pairs: N = 64, 48, 160 (tile kernel), 128, 512 (sweep)
    for n in (64, 48, 160, 128, 512):
        b = (1 << 25) // (n * n)
        xyz, mask = synth(b, n)
        sbn = StructureBatch.from_xyz(xyz, mask)
        repeat(lambda: sbn.inter_residue_geometry())
elif what == "k5":
    xyz, mask = synth(256, 384)
    sb = StructureBatch.from_xyz(xyz, mask).manual_seed(1)
    beta = torch.full((256,), 0.01, device="cuda")
    noise = torch.randn(256, 384, 15, 3, device="cuda")
    rot = torch.empty(256, 384, 3, 3, device="cuda")
    tr = torch.empty(256, 384, 3, device="cuda")
    repeat(lambda: sb.diffuse_xyz(beta))
    repeat(lambda: sb.diffuse_xyz(beta, noise=noise))

    def restd():
        sb._standardized = False
        sb.standardize()
    repeat(restd)
    repeat(lambda: sb.backbone_orientations())
    repeat(lambda: sb.diffuse_xyz_and_frames(beta, out_rot=rot, out_trans=tr))
elif what == "k1a":
    for A, N in ((14, 256), (37, 128)):
        B = max(1, int(8e9 / (N * N * A * A * 5)))
        xyz, mask = synth(B, N, A)
        d = torch.empty(B, N, N, A, A, device="cuda")
        m = torch.empty(B, N, N, A, A, dtype=torch.bool, device="cuda")
        for rp in (0, 1):
            _lib.set_tuning("k1_rowphase", rp)
            repeat(lambda: ops.pairwise_distance(xyz, mask, out_dist=d, out_mask=m))
        _lib.set_tuning("k1_rowphase", 0)
        del xyz, mask, d, m
elif what == "k1s":
    for A, N in ((5, 512), (5, 500), (5, 501), (3, 501), (1, 512), (2, 512)):
        B = max(1, int(4e9 / (N * N * A * A * 5)))
        xyz, mask = synth(B, N, A)
        d = torch.empty(B, N, N, A, A, device="cuda")
        m = torch.empty(B, N, N, A, A, dtype=torch.bool, device="cuda")
        print("k1s", A, N, B, _lib.k1_plan(B, N, A), flush=True)
        repeat(lambda: ops.pairwise_distance(xyz, mask, out_dist=d, out_mask=m))
        del xyz, mask, d, m
elif what == "k1u":
    for A, N in ((5, 501), (1, 501), (37, 100), (5, 512)):
        B = max(1, int(4e9 / (N * N * A * A * 5)))
        xyz, mask = synth(B, N, A)
        d = torch.empty(B, N, N, A, A, device="cuda")
        m = torch.empty(B, N, N, A, A, dtype=torch.bool, device="cuda")
        print("k1u", A, N, B, _lib.k1_plan(B, N, A), f"{2 + reps} dispatches per mode: both, dist only, mask only", flush=True)
        repeat(lambda: ops.pairwise_distance(xyz, mask, out_dist=d, out_mask=m))
        repeat(lambda: ops.pairwise_distance(xyz, mask, out_dist=d, want_mask=False))
        repeat(lambda: ops.pairwise_distance(xyz, mask, out_mask=m, want_dist=False))
        del xyz, mask, d, m
elif what == "k1f":
    for A, N in ((1, 128), (1, 16), (3, 16), (4, 32), (13, 8)):
        B = max(1, int(4e9 / (N * N * A * A * 5)))
        g2 = torch.Generator().manual_seed(A * 1000 + N)
        xyz = torch.randn(B, N, A, 3, generator=g2).cuda()
        mask = (torch.rand(B, N, A, generator=g2) < 0.9).cuda()
        d = torch.empty(B, N, N, A, A, device="cuda")
        m = torch.empty(B, N, N, A, A, dtype=torch.bool, device="cuda")
        print("k1f", A, N, B, _lib.k1_plan(B, N, A), "algorithmic bytes per launch", B * N * N * A * A * 5, flush=True)
        repeat(lambda: ops.pairwise_distance(xyz, mask, out_dist=d, out_mask=m))
        del xyz, mask, d, m
elif what == "k1":
    xyz, mask = synth(64, 512)
    d = torch.empty(64, 512, 512, 15, 15, device="cuda")
    m = torch.empty(64, 512, 512, 15, 15, dtype=torch.bool, device="cuda")
    repeat(lambda: ops.pairwise_distance(xyz, mask, out_dist=d, out_mask=m))
else:
    raise SystemExit(f"unknown workload {what!r}")
print("done", what, reps)
