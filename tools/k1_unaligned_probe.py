"""Where do the unaligned K1 shapes (N = 501, atom37 at N = 100) lose against N = 512: the buffers or the shape?
In ONE process, on the two output buffers of an unaligned shape: torch.fill_ of each plane separately, torch.fill_ of a
prefix of 2^22-element granularity, K1 of the unaligned shape, and K1 of an ALIGNED shape of the same atom count written
into the same memory (a prefix of the same two allocations).  Best of 3 interleaved rounds of 5 launches each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.pop("PROTSTRUC_AMD_AUTOTUNE", None)
import torch
from protstruc_amd import _lib, ops

shapes = [(5, 501, 512), (1, 501, 512), (37, 100, 128), (3, 501, 512), (7, 125, 128), (1, 125, 128)]
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    if k == "shapes": shapes = [tuple(int(x) for x in sh.split(":")) for sh in v.split(",")]
    else: _lib.set_tuning("k1_" + k, int(v))
g = torch.Generator().manual_seed(0)


ab_flags = [1]   # row-phase A/B switches (cfg.rowphase bits 4..): 1 = seams written element-wise from both rows (round 3)


def with_flags(fl, run):
    _lib.set_tuning("k1_rowphase", 16 * fl)
    try:
        run()
    finally:
        _lib.set_tuning("k1_rowphase", 0)


def timed(run):
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5


for A, N, Nal in shapes:
    AA = A * A
    B = max(1, int(8e9 / (N * N * AA * 5)))
    n = B * N * N * AA
    Bal = n // (Nal * Nal * AA)
    nal = Bal * Nal * Nal * AA
    xyz = torch.randn(B, N, A, 3, generator=g).cuda()
    mask = (torch.rand(B, N, A, generator=g) < 0.9).cuda()
    xyz_al = torch.randn(Bal, Nal, A, 3, generator=g).cuda()
    mask_al = (torch.rand(Bal, Nal, A, generator=g) < 0.9).cuda()
    d = torch.empty(B, N, N, A, A, device="cuda"); m = torch.empty(B, N, N, A, A, dtype=torch.bool, device="cuda")
    df, mf = d.view(-1), m.view(-1)
    k = n >> 22 << 22
    d_al, m_al = df[:nal].view(Bal, Nal, Nal, A, A), mf[:nal].view(Bal, Nal, Nal, A, A)
    m32 = mf[:k].view(torch.int32)
    runs = {
        "fill_dist": (4 * n, lambda: d.fill_(0.0)),
        "fill_mask": (n, lambda: m.fill_(False)),
        "fill_dist_prefix": (4 * k, lambda: df[:k].fill_(0.0)),
        "fill_mask_prefix": (k, lambda: mf[:k].fill_(False)),
        "fill_mask_prefix_as_int32": (k, lambda: m32.fill_(0)),
        "k1_unaligned": (5 * n, lambda: ops.pairwise_distance(xyz, mask, out_dist=d, out_mask=m)),
        "k1_unaligned_dist_only": (4 * n, lambda: ops.pairwise_distance(xyz, mask, out_dist=d, want_mask=False)),
        "k1_unaligned_flags1": (5 * n, lambda: with_flags(1, lambda: ops.pairwise_distance(xyz, mask, out_dist=d, out_mask=m))),
        "k1_unaligned_dist_only_flags1": (4 * n, lambda: with_flags(1, lambda: ops.pairwise_distance(xyz, mask, out_dist=d, want_mask=False))),
        "k1_aligned_same_memory": (5 * nal, lambda: ops.pairwise_distance(xyz_al, mask_al, out_dist=d_al, out_mask=m_al)),
        "k1_aligned_dist_only": (4 * nal, lambda: ops.pairwise_distance(xyz_al, mask_al, out_dist=d_al, want_mask=False)),
    }
    best = {name: float("inf") for name in runs}
    for rnd in range(3):
        for name, (nb, run) in runs.items():
            best[name] = min(best[name], timed(run))
    print(f"A={A} N={N} B={B} (numel {n} = 2^{(n & -n).bit_length() - 1} x odd; aligned twin N={Nal} B={Bal})", flush=True)
    for name, (nb, run) in runs.items():
        print(f"    {name:28s} {best[name]:8.4f} ms  {nb / best[name] / 1e9:5.2f} TB/s", flush=True)
    del xyz, mask, d, m, df, mf, d_al, m_al, m32, xyz_al, mask_al, runs
    torch.cuda.empty_cache()
