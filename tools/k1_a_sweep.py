"""K1 store rate for atom counts other than 15, every kernel that can serve the shape timed in ONE process in
interleaved rounds (never compare across runs / boxes): the default dispatch, the row-phase kernel where another kernel
is the default (rowphase=1), flat=0 (element-per-lane kernel);
plus the default dispatch with only the distance plane / only the mask plane.
Round 3: "fill" = torch.fill_ on the same two buffers (which class of allocation the shape drew).  (The "r2path" column
of profiles/r03_k1_a_sweep_rowphase*.log was the round-2 dispatch -- odd row-tile kernels + k1_mask_rows, small fixed-A
flat kernels -- timed next to the row-phase kernel that replaced them; those kernels were removed afterwards.)
Arguments: key=value K1 tuning applied to every run (e.g. flat_cpw=2), `json=path` writes the table,
`shapes=A:N,A:N,...` replaces the shape list."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.pop("PROTSTRUC_AMD_AUTOTUNE", None)   # no implicit tuning while measuring
import torch
from protstruc_amd import _lib, ops
out_json = None
shapes = [(14, 256), (14, 250), (37, 128), (37, 100), (15, 256), (4, 512), (4, 500), (5, 512), (5, 500), (5, 501), (8, 256),
          (3, 512), (3, 500), (3, 501), (16, 256), (25, 128), (1, 512), (1, 500), (1, 501), (2, 512), (2, 501), (7, 512),
          (7, 500), (10, 512), (10, 500), (6, 501), (13, 250), (20, 128), (20, 125), (33, 100), (24, 128), (27, 128), (32, 128),
          (64, 64), (15, 512), (15, 500)]
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    if k == "json": out_json = v
    elif k == "shapes": shapes = [tuple(int(x) for x in sh.split(":")) for sh in v.split(",")]
    else: _lib.set_tuning("k1_" + k, int(v))
g = torch.Generator().manual_seed(0)
rows = []
for A, N in shapes:
    B = max(1, int(8e9 / (N * N * A * A * 5)))
    xyz = torch.randn(B, N, A, 3, generator=g).cuda()
    mask = (torch.rand(B, N, A, generator=g) < 0.9).cuda()
    d = torch.empty(B, N, N, A, A, device="cuda"); m = torch.empty(B, N, N, A, A, dtype=torch.bool, device="cuda")
    variants = {"default": (1, True, True, 0), "rowphase": (1, True, True, 1),
                "element": (0, True, True, 0), "default_dist_only": (1, True, False, 0),
                "default_mask_only": (1, False, True, 0), "fill": None}
    best = {k: float("inf") for k in variants}
    for rnd in range(3):
        for name, var in variants.items():
            if name == "element" and rnd > 0: continue    # slow; once is enough
            if var is None:
                run = lambda: (d.fill_(0.0), m.fill_(False))
            else:
                flat, wd, wm, rowphase = var
                _lib.set_tuning("k1_flat", flat)
                _lib.set_tuning("k1_rowphase", rowphase)
                run = lambda: ops.pairwise_distance(xyz, mask, out_dist=d if wd else None, out_mask=m if wm else None,
                                                    want_dist=wd, want_mask=wm)
            try:
                for _ in range(4): run()      # (the variant that follows `fill` reads ~3 % low with only two warm-up launches)
            except Exception as exc:          # e.g. B > 65535 for the simple kernels (structure on grid.z)
                if rnd == 0: print(f"   [{name}: {type(exc).__name__}: {exc}]", flush=True)
                continue
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): run()
            e1.record(); torch.cuda.synchronize()
            best[name] = min(best[name], e0.elapsed_time(e1) / 5)
    _lib.set_tuning("k1_flat", 1)
    _lib.set_tuning("k1_rowphase", 0)
    nbytes = {"default": 5, "rowphase": 5, "element": 5, "default_dist_only": 4, "default_mask_only": 1, "fill": 5}
    row = {"A": A, "N": N, "B": B, "kernel": _lib.k1_plan(B, N, A)["kernel"], **{k: {"ms": round(v, 4), "TBps": round(B * N * N * A * A * nbytes[k] / v / 1e9, 3)}
                                     for k, v in best.items()}}
    rows.append(row)
    print(f"A={A:3d} N={N:4d} B={B:5d} " + "  ".join(f"{k} {v['TBps']:5.2f}" for k, v in row.items() if isinstance(v, dict))
          + f"  [{row['kernel']}]", flush=True)
    del xyz, mask, d, m
if out_json:
    with open(out_json, "w") as f: json.dump(rows, f, indent=1)
