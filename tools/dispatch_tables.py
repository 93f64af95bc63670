#!/usr/bin/python3
"""Markdown tables of which kernel a launch takes, straight from the library's own dispatchers in record-only mode
(ps_k1_plan_f32, ps_k3_plan_f32, ps_featuriser_plan_f32; no GPU needed) -- what DESIGN.md section 4 shows.
    python3 tools/dispatch_tables.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protstruc_amd import _lib

print("| shape | K1 kernel (family) | workgroups | LDS / workgroup |")
print("|---|---|---|---|")
for name, B, N, A, rows in (("config 1: 15c8_HL", 1, 229, 15, None), ("config 2", 64, 256, 15, None), ("headline", 64, 512, 15, None),
                            ("config 4, full", 32, 2048, 15, None), ("config 4, one of 8 row shards", 32, 2048, 15, (256, 512)),
                            ("N = 500", 64, 500, 15, None), ("CA trace, N = 512", 8, 512, 1, None), ("CA trace, N = 128", 64, 128, 1, None),
                            ("atom37, N = 128", 8, 128, 37, None), ("atom14, N = 256", 8, 256, 14, None), ("peptides, N = 12", 4096, 12, 15, None)):
    r0, r1 = rows if rows else (0, N)
    p = _lib.k1_plan(B, N, A, r0, r1, device=0)
    print(f"| {name}: B={B}, N={N}, A={A} | `{p['kernel']}` ({p['family']}) | {p['n_workgroups']} | {p['lds_bytes']} B |")
print()
feats = {"(2,2) CA,CB\\|CA,CB": (4, [1, 4], [1, 4]), "(3,1) N,CA,CB\\|CB": (4, [0, 1, 4], [4]), "planar (2,1) CA,CB\\|CB": (3, [1, 4], [4])}
print("| chain length (2^25 pairs) | split | fast arithmetic | faithful arithmetic |")
print("|---|---|---|---|")
for N in (16, 32, 48, 64, 99, 128, 140, 160, 256, 300, 512, 2048):
    B = max(1, (1 << 25) // (N * N))
    for fname, (npts, si, sj) in feats.items():
        cells = []
        for mode in (0, 1):
            p = _lib.k3_plan(B, N, 15, si, sj, npts, exact_angles=mode, cu_count=256)
            extra = f", {p['workgroups_per_cu']} wg/CU" if p["workgroups_per_cu"] else ""
            extra += f", {p['structures_per_segment']} structures/pass" if p["family"].startswith("flat") else ""
            cells.append(f"`{p['kernel']}` x{p['threads_per_workgroup']}{extra}")
        print(f"| N={N}, B={B} | {fname} | {cells[0]} | {cells[1]} |")
print()
print("| chain length | featuriser, fast | featuriser, faithful |")
print("|---|---|---|")
for N in (33, 48, 64, 100, 128, 255, 256, 500, 512, 2048):
    B = max(1, (1 << 25) // (N * N))
    cells = []
    for mode in (0, 1):
        p = _lib.featuriser_plan(B, N, 15, exact_angles=mode, cu_count=256)
        extra = f", {p['workgroups_per_cu']} wg/CU, {p['structures_per_segment']} structures/pass" if p["family"] == "featurise" else ""
        cells.append(f"`{p['kernel']}` x{p['threads_per_workgroup']}{extra}")
    print(f"| N={N}, B={B} | {cells[0]} | {cells[1]} |")
