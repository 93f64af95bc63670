#!/usr/bin/env python3
"""Benchmark of the geometry hot path on MI355X -- prints ONE JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Both forms work for N > 1.  The control plane (barriers, max over ranks, verdict exchange) is a gloo group; only the
all-gather of the row-shard section runs on RCCL (its own group), so the headline line does not depend on RCCL.  Started WITHOUT a launcher (WORLD_SIZE unset), ``bench.py --gpus N`` is its own launcher:
the parent -- which never touches the GPU -- starts N fresh rank processes of itself (RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_ADDR=127.0.0.1 / MASTER_PORT in their environment, one rank per GPU over RCCL), relays rank 0's single JSON line
to its own stdout and exits with the largest child exit code.  It never replaces itself with another program.

Metric (BASELINE.json): residue-pairs/s of ``pairwise_distance_matrix`` at the
headline shape B=64, N_res=512, N_atom=15 (synthetic random xyz, bool mask).
A "step" is one launch of K1 over one batch; inputs and the pre-allocated
outputs are resident in HBM before the timed region.

* N = 1: the headline workload on one GPU.
* N > 1: weak scaling -- every rank owns its own B=64 batch (structures are
  independent; no data-path collective); value = all pairs of all ranks divided
  by the slowest rank's time.  After the timed region (guarded by a watchdog that
  makes a stall a NON-ZERO exit) the residue-sharded form of the north star --
  BASELINE config 4 at full size, B=32, N_res=2048: rows [r*N/P, (r+1)*N/P) per
  rank into one full-size buffer, then the RCCL all-gather over xGMI -- is timed
  with HIP events; the strong-scaling figures are TOP-LEVEL keys of the line
  (``config4_kernel_only_pairs_per_s``, ``config4_kernel_only_efficiency_vs_1gpu``,
  ``config4_allgather_ingress_GBps_per_rank`` next to ``config4_xgmi_ingress_bound_GBps_per_rank``,
  ``config4_end_to_end_ms``) and every detail is under "rowshard_allgather".
  One gather implementation failing while the other completes and the gathered
  matrix checks out is a top-level ``rowshard_warning`` (exit 0); a failed check
  or no working gather is ``rowshard_error`` (exit 4).
  Time budget: the watchdog allows that section 420 s (at P = 2 three gathers of
  0.49 s or more per implementation plus warm-ups, 151 GB of allocations and the
  checks take well under a minute); the self-launcher gives the whole run 570 s,
  inside the driver's 600 s.

After the timed region the buffers that were just timed are CHECKED (sampled
blocks against the fp32 formula, exact mask checksum per structure, symmetry of
one structure); a failed check prints no result line and exits non-zero.

``roofline.achieved`` = algorithmic bytes per launch (1125 B per residue pair:
225 fp32 distances + 225 mask bytes) / mean launch duration
measured with HIP events on the launch stream inside the timed region.
``roofline.kernel`` is what the library's own dispatcher reports for this launch
(``ps_k1_plan_f32``), and ``roofline.buffer_fill_GBps`` is the rate at which
``torch.fill_`` writes the SAME two output buffers, measured after the timed
region: MI355X allocations come in a faster and a slower class,
and this is how the line shows which one this run drew; ``roofline.allocation_lottery``
(N = 1, informational, measured after everything else) times the same kernel on
four fresh allocations of the process and then times the fastest of them exactly
like the headline (``kept_pair_timed_like_the_headline``: what a caller who uses
``ops.allocate_fast_outputs`` gets); the headline itself is always measured on the
first allocation ``torch.empty`` returned.  ``roofline.wall_minus_kernel_ms_per_step``
is the host-side gap between the wall clock of the timed region and the kernels'
own event time (timing events are created before the timed region).
``cpu_baseline`` (N = 1 only) times the CPU oracle -- the same ATen op sequence
as the reference -- on a bounded sample of the same workload on the host cores,
at the default thread count and at one thread.
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
PROCESS_T0 = time.time()

B, N_RES, N_ATOM = 64, 512, 15
BYTES_PER_PAIR = N_ATOM * N_ATOM * 4 + N_ATOM * N_ATOM  # 900 + 225
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip table)
XGMI_LINK_GBPS = 153.0  # per xGMI link and direction (MI355X_MICROARCH.md); a GPU has 7 links
C4_B, C4_N = 32, 2048   # BASELINE config 4


def synth(seed, b=B, n=N_RES, a=N_ATOM):
    """SURVEY 8(d) synthetic inputs: unit-scale random-normal xyz, p=0.9 bool mask with the backbone present."""
    g = torch.Generator().manual_seed(seed)
    xyz = torch.randn(b, n, a, 3, generator=g, dtype=torch.float32)
    mask = torch.rand(b, n, a, generator=g) < 0.9
    mask[:, :, :3] = True
    return xyz, mask


def host_cpu_info():
    """Model name, logical CPUs, physical cores and sockets from /proc/cpuinfo (no external tools)."""
    model, cores, sockets = "?", set(), set()
    try:
        phys = core = None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name") and model == "?":
                    model = line.split(":", 1)[1].strip()
                elif line.startswith("physical id"):
                    phys = line.split(":", 1)[1].strip()
                    sockets.add(phys)
                elif line.startswith("core id"):
                    core = line.split(":", 1)[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        cores.add((phys, core))
                    phys = core = None
    except OSError:
        pass
    return {"host_cpu": model, "host_logical_cpus": os.cpu_count(), "host_physical_cores": len(cores) or None,
            "host_sockets": len(sockets) or None}


def _time_oracle(xyz, mask, budget_s, min_structs):
    from oracle import protstruc_oracle as O

    done, t0 = 0, time.perf_counter()
    while done < xyz.shape[0]:
        O.pairwise_distance_matrix(xyz[done:done + 1], mask[done:done + 1])
        done += 1
        if time.perf_counter() - t0 > budget_s and done >= min_structs:
            break
    return done, time.perf_counter() - t0


def cpu_baseline(xyz, mask, budget_s=10.0, budget_1t_s=6.0):
    """Oracle (PyTorch-CPU restatement of reference protstruc.py:477-483) on a bounded sample of the same workload:
    one structure per call (the monolithic call needs ~64 GB of temporaries), at torch's default thread count and
    at ONE thread (BASELINE.md section 4 asks for both)."""
    from oracle import protstruc_oracle as O

    n = xyz.shape[1]
    threads = torch.get_num_threads()
    O.pairwise_distance_matrix(xyz[:1], mask[:1])  # warm-up (allocator, thread pool)
    done, dt = _time_oracle(xyz, mask, budget_s, 2)
    out = {
        "value": done * n * n / dt, "unit": "residue-pairs/s", "cores": threads, "kind": "port",
        "sample": f"first {done} of {xyz.shape[0]} structures (N_res={n}), one structure per call, {dt:.1f} s",
    }
    try:
        torch.set_num_threads(1)
        done1, dt1 = _time_oracle(xyz, mask, budget_1t_s, 1)
        out["single_thread"] = {"value": done1 * n * n / dt1, "unit": "residue-pairs/s", "cores": 1,
                                "sample": f"first {done1} of {xyz.shape[0]} structures, {dt1:.1f} s"}
    finally:
        torch.set_num_threads(threads)
    out.update(host_cpu_info())
    return out


def load_traffic(kernel):
    """HBM bytes per launch of `kernel` (the name ps_k1_plan_f32 reports, e.g. "k1_pairdist_a15_pat<32>") from the committed
    rocprofv3 PMC passes (profiles/k1_traffic.json), or None when that kernel was not in the passes."""
    p = os.path.join(ROOT, "profiles", "k1_traffic.json")
    try:
        with open(p) as f:
            t = json.load(f)
        if t.get("B") == B and t.get("N_res") == N_RES:
            stem = kernel.rstrip(">")                      # "k1_pairdist_a15_pat<32" matches "k1_pairdist_a15_pat<32, false, 0, false>"
            for name, e in t.get("kernels", {}).items():
                if name == kernel or name.startswith(stem + ",") or name.startswith(stem + ">"):
                    return e.get("hbm_bytes_per_launch"), (f"profiles/k1_traffic.json [{name}]: rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE "
                                                           "passes of this command, committed; NOT re-measured inside this run")
    except (OSError, ValueError):
        pass
    return None, None


def check_outputs(xyz, mask, out_d, out_m, n_blocks=64, seed=7):
    """Verify, OUTSIDE the timed region, the buffers the timed launches wrote.  Returns a list of failures."""
    dev = out_d.device
    b_, n_ = xyz.shape[:2]
    fails = []
    g = torch.Generator().manual_seed(seed)
    bs = torch.randint(0, b_, (n_blocks,), generator=g).to(dev)
    i_s = torch.randint(0, n_, (n_blocks,), generator=g).to(dev)
    js = torch.randint(0, n_, (n_blocks,), generator=g).to(dev)
    got = out_d[bs, i_s, js]                                                    # (n_blocks, A, A)
    diff = xyz[bs, i_s][:, :, None, :] - xyz[bs, js][:, None, :, :]
    want = (diff * diff).sum(-1).sqrt()                                         # the fp32 formula, on the device
    err = (got - want).abs().max().item()
    if not err <= 1e-5:
        fails.append(f"sampled blocks: max |d - formula| = {err:.3e} > 1e-5")
    want_m = mask[bs, i_s][:, :, None] & mask[bs, js][:, None, :]
    if not torch.equal(out_m[bs, i_s, js], want_m):
        fails.append("sampled mask blocks differ")
    # exact checksum of the whole mask plane, per structure: sum = (number of present atoms)^2
    # (count_nonzero per structure: a sum with dtype=int64 would first cast the whole 1-byte plane to 8-byte words)
    count = mask.sum((1, 2), dtype=torch.int64)
    bytes_ = out_m.view(torch.uint8)
    got_sum = torch.stack([torch.count_nonzero(bytes_[b]) for b in range(b_)])
    if not torch.equal(got_sum, count * count):
        fails.append("mask checksum per structure differs from (present atoms)^2")
    if int(bytes_.max()) > 1:
        fails.append("mask plane holds bytes other than 0 / 1")
    # symmetry of one whole structure: d[i,j,a,c] == d[j,i,c,a] bit for bit
    bsym = int(bs[0])
    if not torch.equal(out_d[bsym], out_d[bsym].permute(1, 0, 3, 2)):
        fails.append(f"structure {bsym} is not symmetric")
    if torch.isnan(out_d[bsym]).any():
        fails.append(f"structure {bsym} holds NaN")
    return fails


def all_ranks_early(dist, world, obj):
    """``obj`` of every rank (object collective on the control group; a no-op list at world 1 without a group)."""
    if dist is None or not dist.is_initialized():
        return [obj]
    got = [None] * world
    dist.all_gather_object(got, obj)
    return got


def rowshard_allgather(dev, rank, world, max_over_ranks, backend, shared_gpu, steps=2, group=None):
    """BASELINE config 4: residue-sharded K1 into a full-size buffer + all-gather (native RCCL and torch paths),
    each part timed with HIP events on the launch stream; max over ranks.  Every verdict (an exception in a step, a
    failed check) is exchanged between the ranks before anyone acts on it, so all ranks take the same path and
    rank 0's line carries every rank's failures."""
    import torch.distributed as dist
    from protstruc_amd import distributed as D
    from protstruc_amd import ops

    # ranks sharing one card (gloo rehearsal on a 1-GPU box) cannot hold 151 GB each, and gloo moves data through the host
    b, n = (C4_B, C4_N) if not shared_gpu else (2, C4_N // 2)
    xyz, mask = synth(1234, b, n)  # same seed on every rank: inputs are replicated, only outputs are sharded
    xyz, mask = xyz.to(dev), mask.to(dev)
    # pre-flight: every rank holds the FULL matrix (the all-gather destination), 151 GB at config 4, plus RCCL's own
    # buffers.  A rank that cannot must say so as a clean error on every rank, not die in an OOM traceback while its
    # peers wait in a collective.
    need = b * n * n * BYTES_PER_PAIR + (2 << 30)
    free, total = torch.cuda.mem_get_info(dev)
    short = [s_ for s_ in all_ranks_early(dist, world, None if free >= need else
                                          f"rank {rank} ({dev}): {free / 1e9:.1f} GB free of {total / 1e9:.1f}, "
                                          f"the full matrix + workspace needs {need / 1e9:.1f} GB") if s_]
    if short:
        raise RuntimeError("not enough device memory for the row-sharded matrix: " + "; ".join(short))
    out_d = torch.empty(b, n, n, N_ATOM, N_ATOM, device=dev)
    out_m = torch.empty(b, n, n, N_ATOM, N_ATOM, dtype=torch.bool, device=dev)
    lo, hi = D.shard_rows(n, rank, world)
    pairs = b * n * n
    total_bytes = pairs * BYTES_PER_PAIR

    def all_ranks(obj):
        got = [None] * world
        dist.all_gather_object(got, obj)
        return got

    def timed(fn, reps):
        """Mean HIP-event time of ``fn`` over ``reps`` (after one untimed call), max over ranks.  After every call the
        ranks exchange their status (that exchange is also the barrier between steps): an exception on ANY rank is
        re-raised on ALL of them, so no rank is left waiting in a collective for a peer that gave up."""
        ms = []
        for k in range(reps + 1):
            err = None
            try:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                e1.synchronize()
                if k:                      # k == 0 is the warm-up (communicator creation, first-touch of the buffers)
                    ms.append(e0.elapsed_time(e1))
            except Exception as exc:  # noqa: BLE001 -- exchanged below, then raised on every rank
                err = f"rank {rank}: {type(exc).__name__}: {exc}"
            errs = [e for e in all_ranks(err) if e]
            if errs:
                raise RuntimeError("; ".join(errs))
        return max_over_ranks([sum(ms) / len(ms)])[0]

    res = {"workload": f"B={b}, N_res={n}, N_atom={N_ATOM}; rows sharded over {world} ranks ({hi - lo} rows on rank {rank})",
           "backend": backend, "algorithmic_bytes_total": total_bytes}
    kernel = lambda: ops.pairwise_distance(xyz, mask, row_begin=lo, row_end=hi, out_dist=out_d, out_mask=out_m)
    res["kernel_only_ms"] = timed(kernel, steps + 1)
    res["kernel_only_pairs_per_s"] = pairs / (res["kernel_only_ms"] * 1e-3)
    res["kernel_only_GBps_per_rank"] = total_bytes / world / (res["kernel_only_ms"] * 1e-3) / 1e9
    # explicit implementations: they issue the collectives at ANY world size (world 1 = self-gather through the same calls)
    impls = ("native", "torch") if backend == "nccl" else ("torch",)
    ingress = total_bytes * (world - 1) / world
    for impl in impls:
        def gather_only(impl=impl):
            D.allgather_rows(out_d, group, impl=impl)
            D.allgather_rows(out_m, group, impl=impl)
        key = f"allgather_{impl}"
        try:
            res[key + "_ms"] = timed(gather_only, steps)
            res[key + "_ingress_GBps_per_rank"] = ingress / (res[key + "_ms"] * 1e-3) / 1e9
        except Exception as exc:  # noqa: BLE001 -- same on every rank (see timed); keep the other variant's numbers
            res[key + "_error"] = f"{type(exc).__name__}: {exc}"
    res["xgmi_ingress_bound_GBps_per_rank"] = XGMI_LINK_GBPS * min(world - 1, 7)
    done = [(res[f"allgather_{i}_ms"], i) for i in impls if f"allgather_{i}_ms" in res]
    if done:
        res["allgather_best_ms"], res["allgather_best_impl"] = min(done)
        res["allgather_best_ingress_GBps_per_rank"] = ingress / (res["allgather_best_ms"] * 1e-3) / 1e9
    if not done:
        raise RuntimeError("no all-gather implementation completed: " +
                           "; ".join(f"{k}: {v}" for k, v in res.items() if k.endswith("_error")))
    # end to end with the implementation that worked best (normally the native one)
    e2e = lambda: D.pairwise_distance_matrix_sharded(xyz, mask, group=group, gather=True, impl=res["allgather_best_impl"],
                                                     out_dist=out_d, out_mask=out_m)
    res["end_to_end_impl"] = res["allgather_best_impl"]
    res["end_to_end_ms"] = timed(e2e, steps)
    res["end_to_end_pairs_per_s"] = pairs / (res["end_to_end_ms"] * 1e-3)
    # after the last gather every rank must hold the whole matrix: check blocks from every rank's rows + checksum,
    # on EVERY rank, and combine the verdicts before anyone decides an exit code
    fails = check_outputs(xyz, mask, out_d, out_m, n_blocks=128, seed=11 + rank)
    all_fails = [f"rank {r}: {f}" for r, fl in enumerate(all_ranks(fails)) for f in fl]
    res["check_after_gather"] = "ok" if not all_fails else all_fails
    recompute = lambda: D.pairwise_distance_matrix_sharded(xyz, mask, group=group, gather="recompute", out_dist=out_d,
                                                           out_mask=out_m)
    res["full_matrix_recomputed_per_rank_ms"] = timed(recompute, steps)
    # strong scaling of the kernel alone: the same matrix written by ONE GPU (the recompute form, timed just above in
    # this very run) against 1/P of its rows per GPU
    res["kernel_only_efficiency_vs_1gpu"] = res["full_matrix_recomputed_per_rank_ms"] / (res["kernel_only_ms"] * world)
    return res


def launch_ranks(n, argv):
    """``bench.py --gpus N`` without a launcher: start N rank processes of this script, relay rank 0's JSON line.
    This parent makes no GPU call (``torch.cuda.device_count()`` does not initialise the GPU on this image) and
    never execs; it ends exactly the children it started if they overrun."""
    import socket
    import subprocess

    with socket.socket() as sock:          # a free rendezvous port
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    # HSA_ENABLE_IPC_MODE_LEGACY=0: the host driver of this pool supports only dmabuf IPC; with the legacy mode RCCL's
    # (and torch's) cross-process buffer sharing fails at hipIpcGetMemHandle ("invalid argument") as soon as two ranks
    # map each other's memory.  The pool exports 0 already; an inherited value is kept, an unset one becomes 0.
    base = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(n),
                HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, os.path.abspath(__file__)] + argv
    procs, last_err, out0_chunks, relays = [], [b""] * n, [], []

    def relay_stderr(r, pipe):              # pass a child's stderr through, remembering its last non-empty line
        for raw in iter(pipe.readline, b""):
            sys.stderr.buffer.write(raw)
            sys.stderr.buffer.flush()
            if raw.strip():
                last_err[r] = raw.strip()
        pipe.close()

    def collect_stdout(pipe):
        for raw in iter(lambda: pipe.read(65536), b""):
            out0_chunks.append(raw)
        pipe.close()

    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr.fileno(), stderr=subprocess.PIPE)
        procs.append(p)
        relays.append(threading.Thread(target=relay_stderr, args=(r, p.stderr), daemon=True))
        if r == 0:
            relays.append(threading.Thread(target=collect_stdout, args=(p.stdout,), daemon=True))
    for t in relays:
        t.start()
    deadline = time.time() + float(os.environ.get("PS_BENCH_LAUNCH_TIMEOUT", "570"))   # inside the driver's 600 s
    codes, timed_out = [], False
    for r, p in enumerate(procs):
        try:
            p.wait(timeout=max(1.0, deadline - time.time()))
        except subprocess.TimeoutExpired:
            timed_out = True
            p.kill()                        # exactly the child we started
            p.wait()
        codes.append(p.returncode)
    for t in relays:
        t.join(timeout=5.0)
    out0 = b"".join(out0_chunks)
    line = None
    for cand in out0.decode(errors="replace").splitlines():
        cand = cand.strip()
        if cand.startswith("{") and '"metric"' in cand:
            line = cand
    if line is not None:
        print(line, flush=True)
    if line is None or any(codes):
        # say which rank failed how: exit code and the last thing it wrote to stderr
        print(f"bench.py launcher: {'rank 0 printed no result line; ' if line is None else ''}exit codes {codes}"
              f"{' (timed out)' if timed_out else ''}", file=sys.stderr, flush=True)
        for r, c in enumerate(codes):
            if c != 0 or line is None:
                print(f"bench.py launcher:   rank {r}: exit code {c}; last stderr line: "
                      f"{last_err[r].decode(errors='replace')[:400] or '(none)'}", file=sys.stderr, flush=True)
    worst = max((c if c >= 0 else 128 - c) for c in codes)
    if timed_out:
        worst = max(worst, 3)
    if line is None:
        worst = max(worst, 1)
    sys.exit(worst)


def fill_rate_GBps(out_d, out_m, reps=5):
    """Rate at which torch.fill_ writes the two output buffers (HIP events, mean of ``reps`` after one warm-up): the
    store rate THESE allocations accept from the simplest possible kernel (one aligned 16-byte store per lane, 4 KB
    per short-lived workgroup)."""
    nbytes = out_d.numel() * 4 + out_m.numel()
    out_d.fill_(0.0)
    out_m.fill_(False)
    ms = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out_d.fill_(0.0)
        out_m.fill_(False)
        e1.record()
        e1.synchronize()
        ms.append(e0.elapsed_time(e1))
    return nbytes / (sum(ms) / len(ms) * 1e-3) / 1e9



# ---------------------------------------------------------------------------------------------------------------------
# BASELINE configs 2, 3 and 5 (N = 1 only, after the headline and its check, informational: `value` is untouched).
# Reference: protstruc.py:486-541 (backbone_dihedrals), :620-660 (pairwise_dihedrals / planar angles), :864-878
# (diffuse_xyz), :543-571 (frames).  Every leg is timed with HIP events on the launch stream and VERIFIED outside its timed
# region: sampled outputs against the fp32 formula or the CPU oracle on <= 4 structures.  A failed verification is reported
# in the leg's "check" (it does not void the headline, whose own check has passed).
def _event_times_us(fn, reps, warm=3):
    """`fn` launched `reps` times, one HIP-event pair per launch (host-paced, like the headline); returns the list in us."""
    for _ in range(warm):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * reps)]
    for e in ev:
        e.record()
    torch.cuda.synchronize()
    for k in range(reps):
        ev[2 * k].record()
        fn()
        ev[2 * k + 1].record()
    torch.cuda.synchronize()
    return [ev[2 * k].elapsed_time(ev[2 * k + 1]) * 1e3 for k in range(reps)]


def _train_us(fn, reps):
    """Mean over a back-to-back train of `reps` launches, one event pair around the train."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def _graph_replay_us(graph, n_calls, reps=3):
    """Device time per captured call: HIP events around a replay, best of `reps`."""
    graph.replay()
    torch.cuda.synchronize()
    best = None
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        graph.replay()
        e1.record()
        e1.synchronize()
        t = e0.elapsed_time(e1) * 1e3 / n_calls
        best = t if best is None else min(best, t)
    return best


def load_k3_valu_bounds():
    """VALU-issue bounds of the K3 / featuriser kernels at config 3 from the committed SQ counter pass
    (profiles/k3_valu_bound.json: SQ_ACTIVE_INST_VALU x 4 cycles / SIMDs / clock per launch), or {}."""
    try:
        with open(os.path.join(ROOT, "profiles", "k3_valu_bound.json")) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def other_configs(dev, steps):
    from oracle import protstruc_oracle as O          # the checker (outside every timed region), never the thing measured
    from protstruc_amd import StructureBatch, _lib, ops

    import numpy as np
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    out = {"what": "BASELINE configs 2, 3, 5 on this GPU, measured after the headline; HIP events on the launch stream; every leg "
                   "verified outside its timed region (see each leg's check / check_what)",
           "steps": steps}

    def guarded(name, fn):
        try:
            out[name] = fn()
        except Exception as exc:  # noqa: BLE001 -- informational section: report, do not lose the headline
            out[name] = {"error": f"{type(exc).__name__}: {exc}"}

    # ---- config 2: B=64, N=256: pairwise_distance_matrix + backbone_dihedrals ----
    def config2():
        b, n = 64, 256
        xyz_c, mask_c = synth(2, b, n)
        chain = torch.zeros(b, n)
        chain[:, n // 2:] = 1
        xyz, mask = xyz_c.to(dev), mask_c.to(dev)
        d = torch.empty(b, n, n, N_ATOM, N_ATOM, device=dev)
        m = torch.empty(b, n, n, N_ATOM, N_ATOM, dtype=torch.bool, device=dev)
        d.fill_(float("nan")); m.fill_(False)
        ts = _event_times_us(lambda: ops.pairwise_distance(xyz, mask, out_dist=d, out_mask=m), steps)
        fails = check_outputs(xyz, mask, d, m, n_blocks=64, seed=21)
        ms = sum(ts) / len(ts) / 1e3
        nbytes = b * n * n * BYTES_PER_PAIR
        plan = _lib.k1_plan(b, n, N_ATOM, dist_misalign=d.data_ptr() % 16, mask_misalign=m.data_ptr() % 16, device=dev)
        res = {"workload": f"B={b}, N_res={n}, N_atom={N_ATOM}",
               "K1_pairwise_distance_matrix": {"kernel": plan["kernel"], "ms": ms, "ms_min_max": [min(ts) / 1e3, max(ts) / 1e3],
                                               "pairs_per_s": b * n * n / (ms * 1e-3), "TBps": nbytes / (ms * 1e-3) / 1e12,
                                               "frac_of_hbm_peak": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                               "check": "ok" if not fails else fails,
                                               "check_what": "as the headline: 64 sampled blocks vs the fp32 formula <= 1e-5, mask blocks "
                                                             "and per-structure mask checksum exact, one structure symmetric"}}
        del d, m
        sb = StructureBatch.from_xyz(xyz_c, mask_c, chain_idx=chain, chain_ids=[["A", "B"]] * b, device=dev)
        dih, dmask = sb.backbone_dihedrals()
        torch.cuda.synchronize()
        n_calls = 50
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(n_calls):
                dih, dmask = sb.backbone_dihedrals()
        us_graph = _graph_replay_us(g, n_calls)
        ts = _event_times_us(lambda: sb.backbone_dihedrals(), steps)
        pick = [0, 21, 42, 63]
        t0 = time.perf_counter()
        ref, refm = O.backbone_dihedrals(xyz_c[pick], chain[pick], mask_c[pick].any(-1))
        cpu_s = time.perf_counter() - t0
        err = (dih[pick].cpu() - ref).abs().max().item()
        ok = err <= 1e-5 and torch.equal(dmask[pick].cpu(), refm.bool())
        res["K2_backbone_dihedrals"] = {"us_in_graph": us_graph, "us_eager_host_paced": sum(ts) / len(ts), "residues": b * n,
                                        "check": "ok" if ok else f"max |dihedral - oracle| = {err:.3e} or mask mismatch",
                                        "check_what": "4 structures (captured-graph output) vs the CPU oracle: dihedrals <= 1e-5, masks exact",
                                        "cpu_oracle_us_4_structures": cpu_s * 1e6}
        return res

    # ---- config 3: B=128, N=512: pairwise_dihedrals / pairwise_planar_angles and the fused featuriser, both arithmetics ----
    def config3():
        b, n = 128, 512
        xyz_c, mask_c = synth(3, b, n)
        xyz = xyz_c.to(dev)
        sb = StructureBatch.from_xyz(xyz_c, mask_c, device=dev)
        pick = [0, 37, 64, 127]
        feats = {"dihedral_CA_CB__CA_CB": (4, [1, 4], [1, 4]), "dihedral_N_CA_CB__CB": (4, [0, 1, 4], [4]),
                 "planar_CA_CB__CB": (3, [1, 4], [4])}
        bounds = load_k3_valu_bounds()
        refs, cpu_pairs_per_s = {}, {}
        for key, (npts, si, sj) in feats.items():
            t0 = time.perf_counter()
            refs[key] = (O.pairwise_dihedrals if npts == 4 else O.pairwise_planar_angles)(xyz_c[pick], si, sj)
            cpu_pairs_per_s[key] = len(pick) * n * n / (time.perf_counter() - t0)
        off = ~torch.eye(n, dtype=torch.bool).expand(len(pick), n, n)
        res = {"workload": f"B={b}, N_res={n}: 4 B written per residue pair (27 B by the featuriser)",
               "cpu_oracle_pairs_per_s_4_structures": cpu_pairs_per_s, "cpu_oracle_threads": torch.get_num_threads(),
               "valu_bound_source": bounds.get("source")}
        outbuf = torch.empty(b, n, n, device=dev)
        for mode, mname in ((0, "fast"), (1, "faithful")):
            ops.set_exact_angles(bool(mode), dev)
            legs = {}
            try:
                for key, (npts, si, sj) in feats.items():
                    outbuf.fill_(float("inf"))
                    fn = lambda: ops.pairwise_angles(xyz, si, sj, npts, out=outbuf)
                    ts = _event_times_us(fn, steps)
                    train = _train_us(fn, steps)
                    got, ref = outbuf[pick].cpu(), refs[key]
                    wrap = (lambda dd: torch.minimum(dd.abs(), (2 * np.pi - dd.abs()).abs())) if npts == 4 else (lambda dd: dd.abs())
                    both = ~(got.isnan() | ref.isnan())
                    err = wrap(got - ref)
                    frac_bad = (err[both & off] > 1e-5).float().mean().item()
                    nan_equal = bool(torch.equal(got.isnan(), ref.isnan()))
                    if mode == 1 and npts == 4:
                        ok = nan_equal and (err[both] > 1e-5).sum().item() == 0
                    else:
                        ok = frac_bad <= 1e-4 and (torch.isnan(got) != torch.isnan(ref))[off].float().mean().item() <= 1e-5
                    ok = ok and not torch.isinf(outbuf).any().item()
                    us = sum(ts) / len(ts)
                    plan = _lib.k3_plan(b, n, N_ATOM, si, sj, npts, out_misalign=outbuf.data_ptr() % 16, exact_angles=mode, cu_count=cus)
                    vb = bounds.get("kernels", {}).get(plan["kernel"], {}).get("valu_bound_us")
                    ib = bounds.get("kernels", {}).get(plan["kernel"], {}).get("issue_bound_us")
                    legs[key] = {"kernel": plan["kernel"], "us": us, "us_min_max": [min(ts), max(ts)], "us_back_to_back_train": train,
                                 "G_pairs_per_s": b * n * n / us / 1e3, "frac_of_hbm_peak": b * n * n * 4 / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                                 "valu_bound_us": vb, "frac_of_valu_bound": (vb / us) if vb else None,
                                 "issue_bound_us": ib, "frac_of_issue_bound": (ib / us) if ib else None,
                                 "check": "ok" if ok else f"frac > 1e-5: {frac_bad:.2e}, NaN positions equal: {nan_equal}",
                                 "max_abs_err_vs_oracle": err[both].max().item(), "frac_off_diagonal_beyond_1e-5": frac_bad,
                                 "nan_positions_equal_to_oracle": nan_equal}
                fn = lambda: sb.inter_residue_geometry()
                ts = _event_times_us(fn, max(5, steps // 2))
                train = _train_us(fn, max(5, steps // 2))
                geo = sb.inter_residue_geometry()
                ok = True
                for gkey, fkey in (("omega", "dihedral_CA_CB__CA_CB"), ("theta", "dihedral_N_CA_CB__CB"), ("phi", "planar_CA_CB__CB")):
                    npts, si, sj = feats[fkey]
                    k3 = ops.pairwise_angles(xyz, si, sj, npts)
                    ok = ok and torch.equal(geo[gkey].isnan(), k3.isnan()) and torch.equal(geo[gkey].nan_to_num(0), k3.nan_to_num(0))
                d_ca = geo["d_ca"][pick].cpu()
                ca = xyz_c[pick][:, :, 1]
                want = (ca[:, :, None] - ca[:, None, :]).square().sum(-1).sqrt()
                ok = ok and (d_ca - want).abs().max().item() <= 1e-5
                mk = mask_c[pick][:, :, 1]
                ok = ok and torch.equal(geo["d_ca_mask"][pick].cpu().bool(), mk[:, :, None] & mk[:, None, :])
                us = sum(ts) / len(ts)
                plan = _lib.featuriser_plan(b, n, N_ATOM, exact_sqrt=_lib.get_tuning("k1_exact_sqrt", dev), exact_angles=mode, cu_count=cus)
                vb = bounds.get("kernels", {}).get(plan["kernel"], {}).get("valu_bound_us")
                ib = bounds.get("kernels", {}).get(plan["kernel"], {}).get("issue_bound_us")
                legs["inter_residue_geometry"] = {"kernel": plan["kernel"], "us": us, "us_min_max": [min(ts), max(ts)], "us_back_to_back_train": train,
                                                  "G_pairs_per_s": b * n * n / us / 1e3,
                                                  "frac_of_hbm_peak": b * n * n * 27 / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                                                  "valu_bound_us": vb, "frac_of_valu_bound": (vb / us) if vb else None,
                                                  "issue_bound_us": ib, "frac_of_issue_bound": (ib / us) if ib else None,
                                                  "check": "ok" if ok else "a plane differs from its K3 launch / the distance formula / the mask",
                                                  "check_what": "omega / theta / phi bit-identical to the K3 launches of the same mode (all 128 "
                                                                "structures); CA-CA distance and mask planes of 4 structures vs the formula"}
            finally:
                ops.set_exact_angles(False, dev)
            res[mname] = legs
        res["check_what"] = ("4 structures (0, 37, 64, 127) of the timed buffer vs the CPU oracle.  fast: <= 1e-4 of off-diagonal entries "
                             "beyond 1e-5 (SURVEY hard part 3), NaN positions equal off the diagonal up to 1e-5 of entries; faithful dihedrals: "
                             "NO entry beyond 1e-5 (diagonal included) and NaN positions equal; the buffer was inf-filled before the launches")
        return res

    # ---- config 5: B=256, N=384, T=300: diffuse_xyz + backbone_orientations, four ways ----
    def config5():
        b, n, T = 256, 384, 300
        xyz_c, mask_c = synth(5, b, n)
        s_ = 8e-3
        tt = torch.arange(T + 1, dtype=torch.float64)
        f = torch.cos((tt / T + s_) / (1 + s_) * torch.pi / 2) ** 2
        betas = torch.cat([torch.zeros(1, dtype=torch.float64), (1 - f[1:] / f[:-1]).clamp(min=1e-5, max=0.999)])[:T].float()
        betas_TB = betas[:, None].expand(T, b).contiguous().to(dev)
        beta_dev = [betas_TB[t] for t in range(T)]
        sb = StructureBatch.from_xyz(xyz_c.clone(), mask_c, device=dev).manual_seed(1234)
        sb.standardize()
        xyz0, state0 = sb.get_xyz().clone(), sb._rng_state.clone()

        def reset():
            sb.get_xyz().copy_(xyz0)
            sb._rng_state.copy_(state0)

        def loop_events(fn):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            fn()
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1) * 1e3 / T

        def eager():
            for t in range(T):
                sb.diffuse_xyz(beta_dev[t])
                sb.backbone_orientations()

        eager()                                                   # warm-up (library, allocator)
        reset()
        us_eager = loop_events(eager)
        graph = torch.cuda.CUDAGraph()
        rots = []
        reset()
        with torch.cuda.graph(graph):
            for t in range(T):
                sb.diffuse_xyz(beta_dev[t])
                rots.append(sb.backbone_orientations())
        reset()
        us_graph = loop_events(graph.replay)
        final_graph, rot_last_graph = sb.get_xyz().clone(), rots[T - 1].clone()
        del graph, rots
        rot_buf = torch.empty(b, n, 3, 3, device=dev)
        tr_buf = torch.empty(b, n, 3, device=dev)
        sb.diffuse_xyz_and_frames(beta_dev[0], out_rot=rot_buf, out_trans=tr_buf)
        graph2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph2):
            for t in range(T):
                sb.diffuse_xyz_and_frames(beta_dev[t], out_rot=rot_buf, out_trans=tr_buf)
        reset()
        us_fused = loop_events(graph2.replay)
        ok_fused = torch.equal(sb.get_xyz(), final_graph) and torch.equal(rot_buf, rot_last_graph)
        del graph2
        reset()
        sb.diffuse_trajectory(betas_TB)                           # warm-up
        reset()
        held = {}
        us_traj = loop_events(lambda: held.update(r=sb.diffuse_trajectory(betas_TB)))
        rot_t = held["r"][0]
        ok_traj = torch.equal(sb.get_xyz(), final_graph) and torch.equal(rot_t[T - 1], rot_last_graph)
        pick = [0, 100, 200, 255]
        fin = final_graph[pick].cpu()
        t0 = time.perf_counter()
        want = O.backbone_orientations(fin)
        cpu_s = time.perf_counter() - t0
        bad = ((rot_last_graph[pick].cpu() - want).abs() > 1e-5).float().mean().item()
        z = final_graph
        w = mask_c.to(dev).unsqueeze(-1).float()
        cnt = w.sum((1, 2))
        mean = (z * w).sum((1, 2)) / cnt
        var = (((z - mean[:, None, None]) ** 2) * w).sum((1, 2)) / cnt
        stats_ok = mean.abs().max().item() < 0.08 and (var - 1).abs().max().item() < 0.12
        ok = ok_fused and ok_traj and bad <= 1e-3 and stats_ok
        return {"workload": f"B={b}, N_res={n}, N_atom={N_ATOM}, T={T} (cosine schedule), standardize once, then diffuse_xyz + backbone_orientations per step",
                "us_per_step": {"eager_api_loop": us_eager, "hipgraph_of_the_api_loop": us_graph,
                                "hipgraph_of_fused_diffuse_and_frames": us_fused, "lds_resident_trajectory_kernel": us_traj},
                "steps_per_s_trajectory_kernel": 1e6 / us_traj,
                "check": "ok" if ok else {"fused_equals_graph": bool(ok_fused), "trajectory_equals_graph": bool(ok_traj),
                                          "frames_frac_beyond_1e-5_vs_oracle": bad, "final_statistics_ok": bool(stats_ok)},
                "check_what": "same seed, same start: the fused-step graph and the trajectory kernel end in the captured API loop's final "
                              "coordinates and step-300 frames BIT FOR BIT; those frames (4 structures) vs the CPU oracle on the final "
                              "coordinates <= 1e-5 (<= 1e-3 of entries beyond); masked mean / variance of the final coordinates ~ (0, 1)",
                "cpu_oracle_backbone_orientations_us_4_structures": cpu_s * 1e6}

    # ---- not a BASELINE config: K3 and the featuriser away from N = 512 (2^25 pairs per launch: the tile kernels' lengths) ----
    def chain_lengths():
        res = {"workload": "2^25 residue pairs per launch at N = 64, 192 and 320 (the lengths of the tile kernels): the (2,2) dihedral, 4 B "
                           "per pair, and the fused featuriser, 27 B per pair; fast arithmetic; host-paced HIP events",
               "check_what": "3 structures of each timed output vs the CPU oracle: dihedral <= 1e-4 of off-diagonal entries beyond 1e-5; "
                             "featuriser omega bit-identical to that launch, CA-CA distances <= 1e-5, mask plane exact"}
        for n in (64, 192, 320):
            b = (1 << 25) // (n * n)
            xyz_c, mask_c = synth(10 + n, b, n)
            xyz = xyz_c.to(dev)
            sb = StructureBatch.from_xyz(xyz_c, mask_c, device=dev)
            pick = [0, b // 2, b - 1]
            outbuf = torch.full((b, n, n), float("inf"), device=dev)
            fn = lambda: ops.pairwise_angles(xyz, [1, 4], [1, 4], 4, out=outbuf)
            ts = _event_times_us(fn, max(5, steps // 2))
            got = outbuf[pick].cpu()
            ref = O.pairwise_dihedrals(xyz_c[pick], [1, 4], [1, 4])
            off = ~torch.eye(n, dtype=torch.bool).expand(len(pick), n, n)
            both = off & ~torch.isnan(ref) & ~torch.isnan(got)
            frac_bad = ((got - ref).abs()[both] > 1e-5).float().mean().item()
            ok = frac_bad <= 1e-4 and not torch.isinf(outbuf).any().item()
            plan = _lib.k3_plan(b, n, N_ATOM, [1, 4], [1, 4], 4, out_misalign=outbuf.data_ptr() % 16, cu_count=cus)
            us = sum(ts) / len(ts)
            leg = {"B": b, "dihedral_CA_CB__CA_CB": {"kernel": plan["kernel"], "family": plan["family"], "us": us, "us_min_max": [min(ts), max(ts)],
                                                     "G_pairs_per_s": b * n * n / us / 1e3,
                                                     "check": "ok" if ok else f"frac beyond 1e-5: {frac_bad:.2e}"}}
            ts = _event_times_us(lambda: sb.inter_residue_geometry(), max(5, steps // 2))
            geo = sb.inter_residue_geometry()
            ca = xyz_c[pick][:, :, 1]
            want = (ca[:, :, None] - ca[:, None, :]).square().sum(-1).sqrt()
            mk = mask_c[pick][:, :, 1]
            ok = (torch.equal(geo["omega"].isnan(), outbuf.isnan()) and torch.equal(geo["omega"].nan_to_num(0), outbuf.nan_to_num(0))
                  and (geo["d_ca"][pick].cpu() - want).abs().max().item() <= 1e-5
                  and torch.equal(geo["d_ca_mask"][pick].cpu().bool(), mk[:, :, None] & mk[:, None, :]))
            plan = _lib.featuriser_plan(b, n, N_ATOM, exact_sqrt=_lib.get_tuning("k1_exact_sqrt", dev), cu_count=cus)
            us = sum(ts) / len(ts)
            leg["inter_residue_geometry"] = {"kernel": plan["kernel"], "family": plan["family"], "us": us, "us_min_max": [min(ts), max(ts)],
                                             "G_pairs_per_s": b * n * n / us / 1e3,
                                             "frac_of_hbm_peak": b * n * n * 27 / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                                             "check": "ok" if ok else "a plane differs from its K3 launch / the distance formula / the mask"}
            res[f"N={n}"] = leg
            del outbuf, geo, sb, xyz
            torch.cuda.empty_cache()
        return res

    guarded("config2", config2)
    torch.cuda.empty_cache()
    guarded("config3", config3)
    torch.cuda.empty_cache()
    guarded("config5", config5)
    torch.cuda.empty_cache()
    guarded("chain_lengths", chain_lengths)
    torch.cuda.empty_cache()
    checks = []
    for cname in ("config2", "config3", "config5", "chain_lengths"):
        def walk(o):
            if isinstance(o, dict):
                if "error" in o:
                    checks.append(False)
                if "check" in o:
                    checks.append(o["check"] == "ok")
                for v in o.values():
                    walk(v)
        walk(out.get(cname))
    out["all_checks_ok"] = bool(checks) and all(checks)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rowshard", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="skip the post-run verification of the timed buffers")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the timing of BASELINE configs 2, 3 and 5 after the headline")
    ap.add_argument("--no-lottery", action="store_true",
                    help="skip the informational timing of K1 on four fresh output allocations after the measurement")
    ap.add_argument("--shop-allocations", type=int, default=0,
                    help="opt-in experiment: pick the fastest of K output allocations (reported in config); default off")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsals)")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as `python bench.py --gpus N` with no launcher around it: be the launcher (no GPU call in this process)
        n_dev = torch.cuda.device_count()
        if args.backend == "nccl" and args.gpus > n_dev:
            raise SystemExit(f"--gpus {args.gpus} but only {n_dev} GPUs visible (RCCL needs one GPU per rank)")
        launch_ranks(args.gpus, sys.argv[1:])
        return

    # Anything written to fd 1 by native libraries (RCCL prints a version banner to stdout on init) must not
    # precede the one JSON line: send stdout to stderr for the whole run and restore it only to emit the result.
    sys.stdout.flush()
    saved_stdout_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
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
        os.environ.setdefault("MASTER_PORT", "29511")
        # Control plane (barriers, object collectives, max over ranks) on gloo: the headline measurement needs nothing
        # from RCCL, so a communicator problem on a node surfaces in the row-shard section below (watchdog, loud error)
        # and cannot take the headline line with it.  The data plane -- the all-gather -- gets its own RCCL group.
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")     # one node: loopback (the hostname may not resolve)
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))

    def max_over_ranks(values):
        """Element-wise max of a list of floats over all ranks (object collective: works on any backend)."""
        if not dist:
            return values
        gathered = [None] * world
        dist.all_gather_object(gathered, values)
        return [max(v[i] for v in gathered) for i in range(len(values))]

    from protstruc_amd import _lib, ops

    _lib.load()  # fails loudly if the HIP library is missing, stale or of another ABI version
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
    # The check below must see what the TIMED launches wrote, not what warm-up left behind -- and nothing but warm-up steps
    # may sit between the warm-up and the timed region (round 4: an 18.9 GB NaN fill placed there left the first 1-8 timed
    # launches of a fresh box 5-20 % slow, profiles/r04_bench_per_step_pattern.log).  So: poison the buffers FIRST, then warm
    # up on a DECOY input of the same shape (coordinates doubled, mask inverted): whatever a timed launch failed to
    # overwrite would hold doubled distances / an inverted mask and fail the check exactly as a NaN would.
    out_d.fill_(float("nan"))
    out_m.fill_(False)
    xyz_decoy, mask_decoy = (xyz * 2.0).contiguous(), (~mask).contiguous()
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    # torch creates the underlying HIP event at the first record(): do that here, outside the timed region, so that the
    # timed loop allocates nothing (a first-use allocation of timing events in front of the FIRST launch leaves the GPU
    # idle for its duration -- seen once as 11 ms of wall clock that no kernel accounted for)
    for ev in starts + ends:
        ev.record()
    for _ in range(args.warmup):
        ops.pairwise_distance(xyz_decoy, mask_decoy, out_dist=out_d, out_mask=out_m)

    torch.cuda.synchronize(dev)
    if dist:
        dist.barrier()
    ordinal_timed = ops.K1_DISPATCHES[0]     # the timed launches are K1 dispatches [ordinal_timed, ordinal_timed + steps) of this process
    t0 = time.perf_counter()
    for k in range(args.steps):
        starts[k].record()  # recorded on torch's current stream == the stream K1 is launched on
        step()
        ends[k].record()
    torch.cuda.synchronize(dev)
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0

    per_step_ms = [s.elapsed_time(e) for s, e in zip(starts, ends)]
    kernel_ms = sum(per_step_ms) / args.steps
    per_rank_kernel_ms = [kernel_ms]
    if dist:   # every rank's own figure: the spread between ranks is the fast / slow output-buffer lottery
        per_rank_kernel_ms = [None] * world
        dist.all_gather_object(per_rank_kernel_ms, kernel_ms)
    elapsed, kernel_ms_max = max_over_ranks([elapsed, kernel_ms])

    # ---- verification of the timed buffers (outside the timed region) ----
    if args.no_check:
        check = "skipped (--no-check)"
    else:
        fails = check_outputs(xyz, mask, out_d, out_m)
        all_fails = fails
        if dist:
            gathered = [None] * world
            dist.all_gather_object(gathered, fails)
            all_fails = [f"rank {r}: {f}" for r, fl in enumerate(gathered) for f in fl]
        if all_fails:
            print("bench.py: the timed outputs are WRONG -- no result line is emitted:\n  " + "\n  ".join(all_fails),
                  file=sys.stderr, flush=True)
            os._exit(5)
        check = "ok"

    # ---- what the library DEFAULT launch configuration does on the same buffers (the timed launches above ran the explicit
    # tuner's pick; StructureBatch.pairwise_distance_matrix() without a tuner call launches candidate 0).  Timed like the
    # headline -- mean of `steps` launches, one HIP-event pair each -- after the timed region and the check. ----
    default_cfg = dict(ops._K1_CANDIDATE_PATTERN[0])
    tuned_cfg = {f: _lib.get_tuning("k1_" + f, dev) for f in default_cfg}
    default_kernel_ms = kernel_ms
    if tuned_cfg != default_cfg:
        try:
            for f, v in default_cfg.items():
                _lib.set_tuning("k1_" + f, v, dev)
            for _ in range(max(1, args.warmup)):
                step()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * args.steps)]
            for e in ev:
                e.record()
            torch.cuda.synchronize(dev)
            for k in range(args.steps):
                ev[2 * k].record()
                step()
                ev[2 * k + 1].record()
            torch.cuda.synchronize(dev)
            default_kernel_ms = sum(ev[2 * k].elapsed_time(ev[2 * k + 1]) for k in range(args.steps)) / args.steps
        finally:
            for f, v in tuned_cfg.items():
                _lib.set_tuning("k1_" + f, v, dev)
    default_kernel_ms = max_over_ranks([default_kernel_ms])[0]

    # ---- which class of allocation did this run draw?  fill rate of the very buffers that were timed (they have been
    # checked; their contents are no longer needed) ----
    fill_GBps = fill_rate_GBps(out_d, out_m)
    fill_all = max_over_ranks([-fill_GBps])
    fill_GBps_min = -fill_all[0]      # the slowest rank's buffers, like kernel_ms_max
    plan = _lib.k1_plan(B, N_RES, N_ATOM, dist_misalign=out_d.data_ptr() % 16, mask_misalign=out_m.data_ptr() % 16,
                        has_atom_mask=True, device=dev)

    pairs_per_step = B * N_RES * N_RES * world
    value = pairs_per_step * args.steps / elapsed
    achieved = B * N_RES * N_RES * BYTES_PER_PAIR / (kernel_ms_max * 1e-3) / 1e9
    traffic, traffic_source = load_traffic(plan["kernel"])
    result = {
        "metric": "residue-pairs/sec on pairwise_distance_matrix (B=64,N=512); % HBM roofline",
        "value": value, "unit": "residue-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "check": check,
        "check_what": "64 sampled (b,i,j) blocks of the timed buffers vs the fp32 formula <= 1e-5 and their mask blocks "
                      "exactly; exact mask checksum of every structure; bitwise symmetry of one whole structure; the "
                      "buffers were NaN/False-filled and then warmed up on a decoy input (doubled coordinates, inverted mask): "
                      "anything the timed launches had not overwritten would fail these checks",
        "config": {"workload": "pairwise_distance_matrix B=64 N_res=512 N_atom=15 (dist fp32 + bool mask), per GPU",
                   "global_batch": B * world, "n_res": N_RES, "n_atom": N_ATOM,
                   "parallelism": "replicas-of-batch" if world > 1 else "single-gpu",
                   "k1_tuning": _lib.all_tuning(dev),     # every knob the timed launches ran with
                   "library": {"abi": _lib.load().ps_abi_version(), "experiments_compiled_in": bool(_lib.load().ps_has_experiments())},
                   "k1_sqrt": ("correctly rounded" if _lib.get_tuning("k1_exact_sqrt", dev)
                               else "hardware v_sqrt_f32 (exact for 85 % of inputs, 1 ulp off otherwise; parity gate 1e-5 abs)"),
                   "k1_autotune": ops.k1_autotune_result(dev),
                   "output_allocation": ("torch.empty (default)" if shop_report is None
                                         else {"best_of": args.shop_allocations, **shop_report})},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                     "kernel": plan["kernel"], "kernel_family": plan["family"],
                     "kernel_source": "ps_k1_plan_f32: the library's own dispatcher in record-only mode, for the timed "
                                      "buffers' shape, alignment and this device's configuration",
                     "kernel_workgroups": plan["n_workgroups"], "kernel_lds_bytes_per_workgroup": plan["lds_bytes"],
                     "kernel_ms": kernel_ms_max,
                     # for cutting a rocprofv3 kernel trace of this process at the timed launches (tools/summarize_rocprof.py ranges)
                     "k1_dispatch_ordinals_timed_region": [ordinal_timed, ordinal_timed + args.steps],
                     "kernel_ms_min_max_this_rank": [min(per_step_ms), max(per_step_ms)],
                     "kernel_ms_per_step_this_rank": [round(t, 4) for t in per_step_ms],
                     "wall_minus_kernel_ms_per_step": elapsed / args.steps * 1e3 - kernel_ms_max,
                     "algorithmic_bytes_per_launch": B * N_RES * N_RES * BYTES_PER_PAIR,
                     "buffer_fill_GBps": fill_GBps_min, "frac_of_buffer_fill": achieved / fill_GBps_min,
                     "buffer_fill_what": "torch.fill_ on the two timed output buffers, HIP events, mean of 5 after one "
                                         "warm-up, measured after the timed region and the check"},
        "pct_hbm_roofline": 100.0 * achieved / HBM_PEAK_GBPS,
        # the timed launches ran `config.k1_autotune`'s pick; these three say what the library default does on the same buffers
        "default_config": ops._cand_label(default_cfg) + (" (= the tuner's pick: the headline IS the default launch)"
                                                           if tuned_cfg == default_cfg else ""),
        "default_config_kernel_ms": default_kernel_ms,
        "default_config_frac_of_hbm_peak": B * N_RES * N_RES * BYTES_PER_PAIR / (default_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
        "tuner_gain": default_kernel_ms / kernel_ms_max,
    }
    if dist:
        result["rccl_ranks"] = world if args.backend == "nccl" else 0
        result["dist_backend"] = args.backend
        # value / roofline use the SLOWEST rank (max over ranks); the per-rank list shows how much of any shortfall
        # against N = 1 is the per-allocation store rate of each rank's output buffers rather than the scaling
        result["per_rank_kernel_ms"] = per_rank_kernel_ms
        result["per_rank_frac_of_hbm_peak"] = [B * N_RES * N_RES * BYTES_PER_PAIR / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS
                                               for ms in per_rank_kernel_ms]

    printed = threading.Event()

    def emit():
        if rank == 0 and not printed.is_set():
            printed.set()
            sys.stdout.flush()
            os.dup2(saved_stdout_fd, 1)
            print(json.dumps(result), flush=True)

    if world == 1:
        # The headline is complete and checked.  What follows at N = 1 is informational (allocation lottery: 4 x 18.9 GB of
        # allocations and ~60 launches; the CPU baseline: ~20 s of host work): a hang or a hard fault in it must not lose the
        # result line -- the watchdog prints it and ends the process with a non-zero code, like the N > 1 path below.
        def watchdog_n1():
            # the driver gives a run 600 s: fire at 540 s of PROCESS time, while the checked headline can still be printed
            time.sleep(max(5.0, 540.0 - (time.time() - PROCESS_T0)))
            result["informational_sections_error"] = ("timed out 540 s after process start (other_configs / allocation lottery / "
                                                      "cpu_baseline); the headline above is complete and checked")
            emit()
            os._exit(3)

        threading.Thread(target=watchdog_n1, daemon=True).start()

    if world == 1 and not args.no_other_configs:
        # BASELINE configs 2, 3 and 5, timed and checked on this GPU (informational; `value` is the headline's)
        result["other_configs"] = other_configs(dev, args.steps)

    if world == 1 and not args.no_lottery and shop_report is None:
        # Informational, AFTER the measurement above and not part of it: the same kernel timed on four fresh output
        # allocations of this process.  MI355X allocations come in a faster and a slower class; the line
        # above reports whatever torch.empty handed out, this shows the spread a caller can choose from with
        # ops.allocate_fast_outputs.
        out_d = out_m = None
        torch.cuda.empty_cache()
        try:
            _d, _m, lot = ops.allocate_fast_outputs(xyz, mask, candidates=4)
            # the pair the helper kept, timed exactly like the headline (mean of `steps` launches, one event pair each)
            for _ in range(max(1, args.warmup)):
                ops.pairwise_distance(xyz, mask, out_dist=_d, out_mask=_m)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
            ordinal_kept = ops.K1_DISPATCHES[0]
            ev[0].record()
            for k in range(args.steps):
                ops.pairwise_distance(xyz, mask, out_dist=_d, out_mask=_m)
                ev[k + 1].record()
            torch.cuda.synchronize(dev)
            best_ms = sum(ev[k].elapsed_time(ev[k + 1]) for k in range(args.steps)) / args.steps
            del _d, _m
            torch.cuda.empty_cache()
            nb = B * N_RES * N_RES * BYTES_PER_PAIR
            result["roofline"]["allocation_lottery"] = {
                "what": "informational, measured after and outside the timed region: K1 on 4 fresh (dist, mask) allocations "
                        "of this process, mean of 3 interleaved rounds of 3 launches each; NOT the figures above",
                "ms_per_candidate": lot["ms_per_candidate"],
                "frac_of_hbm_peak_per_candidate": [nb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS for ms in lot["ms_per_candidate"]],
                "kept_pair_timed_like_the_headline": {
                    "what": "the pair ops.allocate_fast_outputs kept, mean of `steps` launches with one HIP-event pair each, "
                            "same configuration as the timed region",
                    "kernel_ms": best_ms, "frac_of_hbm_peak": nb / (best_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                    "k1_dispatch_ordinals": [ordinal_kept, ordinal_kept + args.steps]}}
        except Exception as exc:  # noqa: BLE001 -- informational only
            result["roofline"]["allocation_lottery"] = {"error": f"{type(exc).__name__}: {exc}"}

    if world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(xyz_cpu, mask_cpu)
        result["gpu_over_cpu"] = value / result["cpu_baseline"]["value"]
        result["gpu_over_cpu_single_thread"] = value / result["cpu_baseline"]["single_thread"]["value"]

    exit_code = 0
    if (world > 1 or force_dist) and not args.no_rowshard:
        # A stalled collective or hung kernel must not look like success: the watchdog prints the main line (the
        # headline measurement above is complete and valid) and then ends EVERY rank with a non-zero code.
        def watchdog():
            time.sleep(420)
            result["rowshard_allgather"] = None
            result["rowshard_error"] = "timed out after 420 s (stalled collective or kernel)"
            emit()
            os._exit(3)

        threading.Thread(target=watchdog, daemon=True).start()
        out_d = out_m = None
        torch.cuda.empty_cache()
        try:
            data_group = None        # gloo rehearsals: the control group carries the data as well
            if args.backend != "gloo":
                data_group = dist.new_group(backend=args.backend, timeout=datetime.timedelta(seconds=300), device_id=dev)
            rs = rowshard_allgather(dev, rank, world, max_over_ranks, args.backend, shared_gpu=world > n_dev,
                                    group=data_group)
            result["rowshard_allgather"] = rs
            # the strong-scaling figures of the north star, where a parser finds them
            result["config4_workload"] = rs["workload"]
            result["config4_kernel_only_ms"] = rs["kernel_only_ms"]
            result["config4_kernel_only_pairs_per_s"] = rs["kernel_only_pairs_per_s"]
            result["config4_kernel_only_efficiency_vs_1gpu"] = rs["kernel_only_efficiency_vs_1gpu"]
            result["config4_allgather_ms"] = rs.get("allgather_best_ms")
            result["config4_allgather_impl"] = rs.get("allgather_best_impl")
            result["config4_allgather_ingress_GBps_per_rank"] = rs.get("allgather_best_ingress_GBps_per_rank")
            result["config4_xgmi_ingress_bound_GBps_per_rank"] = rs["xgmi_ingress_bound_GBps_per_rank"]
            result["config4_end_to_end_ms"] = rs["end_to_end_ms"]
            result["config4_end_to_end_pairs_per_s"] = rs["end_to_end_pairs_per_s"]
            result["config4_full_matrix_on_one_gpu_ms"] = rs["full_matrix_recomputed_per_rank_ms"]
            # one of the two gather implementations failing while the other completed and the gathered matrix checks out
            # is reported (loudly, top level) but does not void the run; a failed check does
            partial = [f"{k}: {v}" for k, v in rs.items() if k.endswith("_error")]
            if partial:
                result["rowshard_warning"] = "; ".join(partial)
            if rs.get("check_after_gather") != "ok":
                result["rowshard_error"] = f"check_after_gather: {rs.get('check_after_gather')}"
                exit_code = 4
        except Exception as exc:  # noqa: BLE001 -- the main measurement is still printed, the failure is not hidden
            result["rowshard_allgather"] = None
            result["rowshard_error"] = f"{type(exc).__name__}: {exc}"
            exit_code = 4
    emit()
    if dist and exit_code == 0:
        try:
            from protstruc_amd import distributed as D
            D.destroy_native_comms()
            dist.destroy_process_group()
        except Exception:  # noqa: BLE001
            pass
    if exit_code:
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(exit_code)     # on every rank; no teardown of a possibly wedged communicator


if __name__ == "__main__":
    main()
