#!/usr/bin/env python3
"""Timings of the non-headline BASELINE configs on one MI355X (writes JSON).

  config 2: B=64, N=256   K1 + K2
  config 3: B=128, N=512  K3 dihedrals (2,2), (3,1) and planar angles (2,1)
  config 5: B=256, N=384  T=300 steps of diffuse_xyz + backbone_orientations, eager vs hipGraph
  per-kernel: K4, K5, K6 at the config-5 shape
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from protstruc_amd import StructureBatch, ops


def synth(seed, B, N, A=15):
    g = torch.Generator().manual_seed(seed)
    xyz = torch.randn(B, N, A, 3, generator=g)
    mask = torch.rand(B, N, A, generator=g) < 0.9
    mask[:, :, :3] = True
    return xyz, mask


def gpu_time(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    return min(ts)


def graph_time(fn, n=50):
    """GPU time per call with host launch overhead removed: capture n calls, replay, divide."""
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts) / n * 1e6  # us


out = {"device": torch.cuda.get_device_name(0)}

# ---- config 2 ----
B, N = 64, 256
xyz, mask = synth(0, B, N)
chain_idx = torch.zeros(B, N)
chain_idx[:, N // 2:] = 1
sb = StructureBatch.from_xyz(xyz, mask, chain_idx=chain_idx, chain_ids=[["A", "B"]] * B)
d = torch.empty(B, N, N, 15, 15, device="cuda")
m = torch.empty(B, N, N, 15, 15, dtype=torch.bool, device="cuda")
ms = gpu_time(lambda: ops.pairwise_distance(sb.xyz, sb.atom_mask, out_dist=d, out_mask=m))
out["config2_K1_B64_N256"] = {"ms": ms, "pairs_per_s": B * N * N / ms * 1e3, "TBps": B * N * N * 1125 / ms / 1e9}
ms = gpu_time(lambda: sb.backbone_dihedrals(), reps=100)
out["config2_K2_B64_N256"] = {"eager_us_host_bound": ms * 1e3, "graph_us": graph_time(lambda: sb.backbone_dihedrals()),
                              "residues": B * N}
del d, m

# ---- config 3 ----
B, N = 128, 512
xyz, mask = synth(1, B, N)
sb = StructureBatch.from_xyz(xyz, mask)
for name, fn in {
    "dihedral_CA_CB__CA_CB": lambda: sb.pairwise_dihedrals(["CA", "CB"], ["CA", "CB"]),
    "dihedral_N_CA_CB__CB": lambda: sb.pairwise_dihedrals(["N", "CA", "CB"], ["CB"]),
    "planar_CA_CB__CB": lambda: sb.pairwise_planar_angles(["CA", "CB"], ["CB"]),
}.items():
    ms = gpu_time(fn, reps=20)
    out["config3_K3_" + name] = {"ms": ms, "pairs_per_s": B * N * N / ms * 1e3, "GBps_written": B * N * N * 4 / ms / 1e6}

ms = gpu_time(lambda: sb.inter_residue_geometry(), reps=10)
out["config3_fused_inter_residue_geometry_B128_N512"] = {"ms": ms, "pairs_per_s": B * N * N / ms * 1e3,
                                                         "GBps_written": B * N * N * 27 / ms / 1e6}

# ---- config 5 ----
B, N, T = 256, 384, 300
xyz, mask = synth(2, B, N)
sb = StructureBatch.from_xyz(xyz.clone(), mask).manual_seed(1234)
sb.standardize()
s = 8e-3
tt = torch.arange(T + 1, dtype=torch.float64)
f = torch.cos((tt / T + s) / (1 + s) * torch.pi / 2) ** 2
betas = (1 - f[1:] / f[:-1]).clamp(max=0.999).float()
beta_dev = [betas[t].expand(B).contiguous().cuda() for t in range(T)]
out["config5_K6_standardize_once_us"] = None
sb2 = StructureBatch.from_xyz(xyz.clone(), mask)
t0 = time.perf_counter(); sb2.standardize(); torch.cuda.synchronize(); out["config5_K6_standardize_once_us"] = (time.perf_counter() - t0) * 1e6


def eager_loop():
    for t in range(T):
        sb.diffuse_xyz(beta_dev[t])
        sb.backbone_orientations()


eager_loop(); torch.cuda.synchronize()
eager_runs = []
for _ in range(3):
    t0 = time.perf_counter(); eager_loop(); torch.cuda.synchronize(); eager_runs.append(time.perf_counter() - t0)
eager_s = min(eager_runs)
# hipGraph: capture the whole T-step loop once (beta lives in device buffers, rng offset on the device)
graph = torch.cuda.CUDAGraph()
rots = []
with torch.cuda.graph(graph):
    for t in range(T):
        sb.diffuse_xyz(beta_dev[t])
        rots.append(sb.backbone_orientations())
graph.replay(); torch.cuda.synchronize()
t0 = time.perf_counter(); graph.replay(); torch.cuda.synchronize(); graph_s = time.perf_counter() - t0
out["config5_loop_B256_N384_T300"] = {"eager_us_per_step": eager_s / T * 1e6, "eager_us_per_step_runs": [r / T * 1e6 for r in eager_runs], "hipgraph_us_per_step": graph_s / T * 1e6,
                                      "kernels_per_step": 3}
# fused step: one launch (+ the 1-thread rng advance) per step, outputs written into static buffers
rot_buf = torch.empty(B, N, 3, 3, device="cuda"); tr_buf = torch.empty(B, N, 3, device="cuda")
graph2 = torch.cuda.CUDAGraph()
sb.diffuse_xyz_and_frames(beta_dev[0], out_rot=rot_buf, out_trans=tr_buf); torch.cuda.synchronize()
with torch.cuda.graph(graph2):
    for t in range(T):
        sb.diffuse_xyz_and_frames(beta_dev[t], out_rot=rot_buf, out_trans=tr_buf)
graph2.replay(); torch.cuda.synchronize()
t0 = time.perf_counter(); graph2.replay(); torch.cuda.synchronize(); fused_s = time.perf_counter() - t0
out["config5_loop_B256_N384_T300"]["hipgraph_fused_us_per_step"] = fused_s / T * 1e6
betas_TB = torch.stack([b for b in beta_dev])  # (T, B) on the device
sb.diffuse_trajectory(betas_TB); torch.cuda.synchronize()
ts = []
for _ in range(3):
    t0 = time.perf_counter(); sb.diffuse_trajectory(betas_TB); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
out["config5_loop_B256_N384_T300"]["lds_resident_trajectory_us_per_step"] = min(ts) / T * 1e6
ts = []
for _ in range(3):
    t0 = time.perf_counter(); sb.diffuse_trajectory(betas_TB, want_xyz=True); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
out["config5_loop_B256_N384_T300"]["lds_resident_trajectory_with_xyz_us_per_step"] = min(ts) / T * 1e6
us = graph_time(lambda: sb.diffuse_xyz(beta_dev[0]))
out["K5_diffuse_B256_N384"] = {"graph_us": us, "GBps_rw": B * N * 45 * 8 / us / 1e3}
us = graph_time(lambda: sb.backbone_orientations())
out["K4_frames_B256_N384"] = {"graph_us": us}
us = graph_time(lambda: sb.diffuse_xyz_and_frames(beta_dev[0], out_rot=rot_buf, out_trans=tr_buf))
out["K54_fused_step_B256_N384"] = {"graph_us": us}
sb3 = StructureBatch.from_xyz(xyz.clone(), mask)
def std_roundtrip():
    sb3._standardized = False
    sb3.standardize()
us = graph_time(std_roundtrip, n=20)
out["K6_standardize_B256_N384"] = {"graph_us": us, "GBps": B * N * 45 * 4 * 4 / us / 1e3}
# ---- CPU oracle (the reference's op sequence) on this box's host cores, bounded samples ----
from oracle import protstruc_oracle as O


def cpu_time(fn, min_s=2.0):
    fn()
    n, t0 = 0, time.perf_counter()
    while True:
        fn(); n += 1
        if time.perf_counter() - t0 > min_s:
            break
    return (time.perf_counter() - t0) / n


cpu = {"threads": torch.get_num_threads(), "logical_cpus": os.cpu_count()}
x2, m2 = synth(0, 64, 256)
ci = torch.zeros(64, 256); ci[:, 128:] = 1
cpu["config2_K2_backbone_dihedrals_B64_N256_ms"] = cpu_time(lambda: O.backbone_dihedrals(x2, ci, m2.any(-1))) * 1e3
x3, m3 = synth(1, 4, 512)   # 4 of the 128 structures of config 3
t = cpu_time(lambda: O.pairwise_dihedrals(x3, [1, 4], [1, 4]))
cpu["config3_K3_dihedral_pairs_per_s"] = 4 * 512 * 512 / t
t = cpu_time(lambda: O.pairwise_planar_angles(x3, [1, 4], [4]))
cpu["config3_K3_planar_pairs_per_s"] = 4 * 512 * 512 / t
x5, m5 = synth(2, 256, 384)
beta5 = torch.full((256,), 0.01)
t = cpu_time(lambda: O.diffuse_xyz(x5, beta5, torch.randn_like(x5)))
cpu["config5_diffuse_xyz_us"] = t * 1e6
t = cpu_time(lambda: O.backbone_orientations(x5))
cpu["config5_backbone_orientations_us"] = t * 1e6
t = cpu_time(lambda: O.standardize(x5, m5))
cpu["config5_standardize_us"] = t * 1e6
out["cpu_oracle_on_this_host"] = cpu
print(json.dumps(out, indent=1))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/bench_configs.json", "w"), indent=1)
