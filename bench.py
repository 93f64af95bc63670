#!/usr/bin/env python3
"""Benchmark of the geometry hot path on MI355X -- prints ONE JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): residue-pairs/s of ``pairwise_distance_matrix`` at the
headline shape B=64, N_res=512, N_atom=15 (synthetic random xyz, bool mask).
A "step" is one launch of K1 over one batch; inputs and the pre-allocated
outputs are resident in HBM before the timed region.

* N = 1: the headline workload on one GPU.
* N > 1: weak scaling -- every rank owns its own B=64 batch (structures are
  independent; no data-path collective); value = all pairs of all ranks divided
  by the slowest rank's time.  After the timed region (and guarded by a
  watchdog) the residue-sharded variant of the north star -- rows [r*N/P,
  (r+1)*N/P) per rank into one full-size buffer, then an RCCL all-gather over
  xGMI -- is timed separately and reported under "rowshard_allgather".

``roofline.achieved`` = algorithmic bytes per launch (1125 B per residue pair:
225 fp32 distances + 225 mask bytes) / mean launch duration
measured with HIP events on the launch stream inside the timed region.
``cpu_baseline`` (N = 1 only) times the CPU oracle -- the same ATen op sequence
as the reference -- on a bounded sample of the same workload on the host cores.
"""
import argparse
import json
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B, N_RES, N_ATOM = 64, 512, 15
BYTES_PER_PAIR = N_ATOM * N_ATOM * 4 + N_ATOM * N_ATOM  # 900 + 225
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip table)


def synth(seed, b=B, n=N_RES, a=N_ATOM):
    """SURVEY 8(d) synthetic inputs: unit-scale random-normal xyz, p=0.9 bool mask with the backbone present."""
    g = torch.Generator().manual_seed(seed)
    xyz = torch.randn(b, n, a, 3, generator=g, dtype=torch.float32)
    mask = torch.rand(b, n, a, generator=g) < 0.9
    mask[:, :, :3] = True
    return xyz, mask


def cpu_baseline(xyz, mask, budget_s=12.0):
    """Oracle (PyTorch-CPU restatement of reference protstruc.py:477-483) on a bounded sample."""
    from oracle import protstruc_oracle as O

    threads = torch.get_num_threads()
    O.pairwise_distance_matrix(xyz[:1], mask[:1])  # warm-up (allocator, threads)
    done, t0 = 0, time.perf_counter()
    while done < xyz.shape[0]:
        O.pairwise_distance_matrix(xyz[done:done + 1], mask[done:done + 1])
        done += 1
        if time.perf_counter() - t0 > budget_s and done >= 2:
            break
    dt = time.perf_counter() - t0
    n = xyz.shape[1]
    model = "?"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {
        "value": done * n * n / dt, "unit": "residue-pairs/s", "cores": threads, "kind": "port",
        "sample": f"first {done} of {xyz.shape[0]} structures (N_res={n}), one structure per call, {dt:.1f} s",
        "host_cpu": model, "host_logical_cpus": os.cpu_count(),
    }


def load_traffic():
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/k1_traffic.json), or None."""
    p = os.path.join(ROOT, "profiles", "k1_traffic.json")
    try:
        with open(p) as f:
            t = json.load(f)
        if t.get("B") == B and t.get("N_res") == N_RES:
            return t.get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        pass
    return None


def rowshard_allgather(dev, rank, world, max_over_ranks, steps=3):
    """North-star config 4 in miniature: residue-sharded K1 into a full-size buffer + RCCL all-gather."""
    import torch.distributed as dist
    from protstruc_amd.distributed import pairwise_distance_matrix_sharded

    b, n = 8, 2048
    xyz, mask = synth(1234, b, n)  # same seed on every rank: inputs are replicated, only outputs are sharded
    xyz, mask = xyz.to(dev), mask.to(dev)
    out_d = torch.empty(b, n, n, N_ATOM, N_ATOM, device=dev)
    out_m = torch.empty(b, n, n, N_ATOM, N_ATOM, dtype=torch.bool, device=dev)
    res = {}
    for gather in (False, "recompute", True):
        for _ in range(1):
            pairwise_distance_matrix_sharded(xyz, mask, gather=gather, out_dist=out_d, out_mask=out_m)
        torch.cuda.synchronize(dev)
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            pairwise_distance_matrix_sharded(xyz, mask, gather=gather, out_dist=out_d, out_mask=out_m)
        torch.cuda.synchronize(dev)
        dist.barrier()
        dt = max_over_ranks([(time.perf_counter() - t0) / steps])[0]
        key = {False: "kernel_only_ms", True: "kernel_plus_allgather_ms", "recompute": "full_matrix_recomputed_per_rank_ms"}
        res[key[gather]] = dt * 1e3
    pairs = b * n * n
    res.update({
        "workload": f"B={b}, N_res={n}, rows sharded over {world} ranks",
        "pairs_per_s_kernel_only": pairs / (res["kernel_only_ms"] * 1e-3),
        "pairs_per_s_with_allgather": pairs / (res["kernel_plus_allgather_ms"] * 1e-3),
        "allgather_GBps_per_rank_ingress": pairs * BYTES_PER_PAIR * (world - 1) / world
        / max((res["kernel_plus_allgather_ms"] - res["kernel_only_ms"]) * 1e-3, 1e-9) / 1e9,
    })
    return res


def main():
    # Anything written to fd 1 by native libraries (RCCL prints a version banner to stdout on init) must not
    # precede the one JSON line: send stdout to stderr for the whole run and restore it only to emit the result.
    sys.stdout.flush()
    saved_stdout_fd = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rowshard", action="store_true")
    ap.add_argument("--shop-allocations", type=int, default=0,
                    help="opt-in experiment: pick the fastest of K output allocations (reported in config); default off")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsals)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")

    n_dev = torch.cuda.device_count()
    if args.backend == "nccl" and world > n_dev:
        raise SystemExit(f"{world} ranks but only {n_dev} GPUs visible (RCCL needs one GPU per rank)")
    dev = torch.device("cuda", local_rank % n_dev)  # ranks share a GPU only in gloo rehearsals
    torch.cuda.set_device(dev)
    dist = None
    force_dist = bool(os.environ.get("PS_BENCH_FORCE_DIST"))  # rehearsal: run the collective code path with one rank
    if world > 1 or force_dist:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {"device_id": dev} if args.backend == "nccl" else {}
        dist.init_process_group(args.backend, timeout=datetime.timedelta(seconds=300), **kw)

    def max_over_ranks(values):
        """Element-wise max of a list of floats over all ranks (object collective: works on any backend)."""
        if not dist:
            return values
        gathered = [None] * world
        dist.all_gather_object(gathered, values)
        return [max(v[i] for v in gathered) for i in range(len(values))]

    from protstruc_amd import _lib, ops

    _lib.load()  # fails loudly if the HIP library is missing
    xyz_cpu, mask_cpu = synth(seed=rank)  # rank 0 / N=1: seed 0 as in SURVEY 8(d)
    xyz, mask = xyz_cpu.to(dev), mask_cpu.to(dev)
    shop_report = None
    if args.shop_allocations > 1:
        # opt-in, OFF by default: keep the fastest of K output allocations (ops.allocate_fast_outputs; DESIGN.md
        # section 4, "fast and slow allocations").  The default run takes whatever torch.empty returns.
        out_d, out_m, shop_report = ops.allocate_fast_outputs(xyz, mask, candidates=args.shop_allocations)
    else:
        out_d = torch.empty(B, N_RES, N_RES, N_ATOM, N_ATOM, device=dev)
        out_m = torch.empty(B, N_RES, N_RES, N_ATOM, N_ATOM, dtype=torch.bool, device=dev)

    def step():
        ops.pairwise_distance(xyz, mask, out_dist=out_d, out_mask=out_m)

    ops.autotune_pairwise_distance(xyz, mask, out_d, out_m)  # one-time per-device library initialisation (not a step)
    for _ in range(args.warmup):
        step()
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]

    torch.cuda.synchronize(dev)
    if dist:
        dist.barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        starts[k].record()  # recorded on torch's current stream == the stream K1 is launched on
        step()
        ends[k].record()
    torch.cuda.synchronize(dev)
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0

    kernel_ms = sum(s.elapsed_time(e) for s, e in zip(starts, ends)) / args.steps
    elapsed, kernel_ms_max = max_over_ranks([elapsed, kernel_ms])

    pairs_per_step = B * N_RES * N_RES * world
    value = pairs_per_step * args.steps / elapsed
    achieved = B * N_RES * N_RES * BYTES_PER_PAIR / (kernel_ms_max * 1e-3) / 1e9
    traffic = load_traffic()
    result = {
        "metric": "residue-pairs/sec on pairwise_distance_matrix (B=64,N=512); % HBM roofline",
        "value": value, "unit": "residue-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "pairwise_distance_matrix B=64 N_res=512 N_atom=15 (dist fp32 + bool mask), per GPU",
                   "global_batch": B * world, "n_res": N_RES, "n_atom": N_ATOM,
                   "parallelism": "replicas-of-batch" if world > 1 else "single-gpu",
                   "k1_tuning": {k: _lib.get_tuning(k) for k in ("k1_variant", "k1_jt", "k1_rows_per_block", "k1_store_nt",
                                                                 "k1_xcd_remap", "k1_exact_sqrt", "k1_lds_pad_kb")},
                   "k1_sqrt": ("correctly rounded" if _lib.get_tuning("k1_exact_sqrt")
                               else "hardware v_sqrt_f32 (exact for 85 % of inputs, 1 ulp off otherwise; parity gate 1e-5 abs)"),
                   "k1_autotune": ops.k1_autotune_result(dev),
                   "output_allocation": ("torch.empty (default)" if shop_report is None
                                         else {"best_of": args.shop_allocations, **shop_report})},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                     "kernel": "k1_pairdist_a15_pat", "kernel_ms": kernel_ms_max,
                     "algorithmic_bytes_per_launch": B * N_RES * N_RES * BYTES_PER_PAIR,
                     "frac_of_measured_write_ceiling_6.88TBps": achieved / 6880.0},
        "pct_hbm_roofline": 100.0 * achieved / HBM_PEAK_GBPS,
    }

    printed = threading.Event()

    def emit():
        if rank == 0 and not printed.is_set():
            printed.set()
            sys.stdout.flush()
            os.dup2(saved_stdout_fd, 1)
            print(json.dumps(result), flush=True)

    if world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(xyz_cpu, mask_cpu)
        result["gpu_over_cpu"] = value / result["cpu_baseline"]["value"]

    if (world > 1 or force_dist) and not args.no_rowshard:
        # The aux section must never cost the main line: a watchdog prints it and exits if RCCL stalls.
        def watchdog():
            time.sleep(240)
            result["rowshard_allgather"] = {"error": "timed out after 240 s"}
            emit()
            os._exit(0)

        threading.Thread(target=watchdog, daemon=True).start()
        del out_d, out_m
        torch.cuda.empty_cache()
        try:
            result["rowshard_allgather"] = rowshard_allgather(dev, rank, world, max_over_ranks)
        except Exception as exc:  # noqa: BLE001 -- report, do not lose the main measurement
            result["rowshard_allgather"] = {"error": f"{type(exc).__name__}: {exc}"}
    emit()
    if dist:
        try:
            dist.destroy_process_group()
        except Exception:  # noqa: BLE001
            pass


if __name__ == "__main__":
    main()
