#!/usr/bin/env python3
"""Rehearsal of the row-sharded multi-rank path with the REAL HIP kernels on however few GPUs there are.

    python tools/rehearse_rowshard.py --world 2 --backend gloo --out gpurun_out/rehearse.json

The launcher (this process) never touches the GPU: it picks a free port, starts ``--world`` child ranks of itself
and collects their verdicts.  Every rank uses GPU ``rank % device_count`` (so on a 1-GPU box all ranks share the one
card, which only the gloo backend allows -- RCCL refuses two ranks on one device), initialises
``torch.distributed`` and runs ``pairwise_distance_matrix_sharded`` and ``pairwise_angles_sharded`` for
``gather`` in {False, True, "recompute"} at residue counts that do and do not divide by the world size, comparing
bit for bit with the single-GPU kernels.  With ``--backend nccl`` the gathers go through the native
``ps_allgather_rows`` (RCCL); on one GPU that is only possible with ``--world 1``.
Exit code 0 = every rank passed every case.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rank_main(args):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    n_dev = torch.cuda.device_count()
    dev = torch.device("cuda", args.rank % n_dev)
    torch.cuda.set_device(dev)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(args.port)
    import datetime
    kw = {"device_id": dev} if args.backend == "nccl" else {}
    dist.init_process_group(args.backend, rank=args.rank, world_size=args.world,
                            timeout=datetime.timedelta(seconds=120), **kw)
    from protstruc_amd import distributed as D
    from protstruc_amd import ops

    results, ok_all = [], True
    try:
        def same(x, y):
            return torch.equal(x.isnan(), y.isnan()) and torch.equal(x.nan_to_num(7.0), y.nan_to_num(7.0))

        cases = [(2, 64, 15), (3, 51, 15), (2, 40, 5), (1, 33, 37)] if not args.quick else [(2, 64, 15), (2, 51, 15)]
        for (B, N, A) in cases:
            g = torch.Generator().manual_seed(1000 + N)      # same inputs on every rank: inputs are replicated
            xyz = torch.randn(B, N, A, 3, generator=g).to(dev)
            mask = (torch.rand(B, N, A, generator=g) < 0.85).to(dev)
            ref_d, ref_m = ops.pairwise_distance(xyz, mask)   # the single-GPU result
            lo, hi = D.shard_rows(N, args.rank, args.world)
            outside = torch.ones(N, dtype=torch.bool, device=dev)
            outside[lo:hi] = False
            for gather in (False, True, "recompute"):
                for impl in (("native", "torch") if args.backend == "nccl" and gather is True else ("auto",)):
                    od = torch.full((B, N, N, A, A), -5.0, device=dev)
                    om = torch.zeros((B, N, N, A, A), dtype=torch.bool, device=dev)
                    d, m, rng = D.pairwise_distance_matrix_sharded(xyz, mask, gather=gather, impl=impl, out_dist=od,
                                                                   out_mask=om)
                    torch.cuda.synchronize(dev)
                    if gather:
                        ok = same(d, ref_d) and torch.equal(m, ref_m)
                    else:
                        ok = same(d[:, lo:hi], ref_d[:, lo:hi]) and torch.equal(m[:, lo:hi], ref_m[:, lo:hi]) \
                            and bool((d[:, outside] == -5.0).all()) and not bool(m[:, outside].any())
                    ok = ok and tuple(rng) == ((0, N) if gather == "recompute" else (lo, hi))
                    results.append({"op": "distance", "B": B, "N": N, "A": A, "gather": str(gather), "impl": impl, "ok": bool(ok)})
                    ok_all &= bool(ok)
            if A >= 5:
                for npts, si, sj in ((4, [1, 4], [1, 4]), (3, [1, 4], [4])):
                    want = ops.pairwise_angles(xyz, si, sj, npts)
                    for gather in (False, True, "recompute"):
                        buf = torch.full((B, N, N), 77.0, device=dev)
                        a, rng = D.pairwise_angles_sharded(xyz, si, sj, npts, gather=gather, out=buf)
                        torch.cuda.synchronize(dev)
                        if gather:
                            ok = same(a, want)
                        else:
                            ok = same(a[:, lo:hi], want[:, lo:hi]) and bool((a[:, outside] == 77.0).all())
                        results.append({"op": f"angles{npts}", "B": B, "N": N, "gather": str(gather), "ok": bool(ok)})
                        ok_all &= bool(ok)
        # The exchange itself, called directly with an EXPLICIT implementation (which issues the collectives at any world
        # size, world 1 included -- a self-gather): on RCCL this is the body of ps_allgather_rows_ex, i.e.
        # ncclGroupStart / in-place ncclAllGather per structure / ncclGroupEnd, and with force_broadcast the
        # per-(structure, owner) in-place ncclBroadcast form that uneven splits take.  Rows owned by other ranks are
        # poisoned first, so a gather that does not deliver them cannot pass.
        for (B, N, A) in ((3, 48, 15), (2, 51, 5)):
            g = torch.Generator().manual_seed(2000 + N)
            xyz = torch.randn(B, N, A, 3, generator=g).to(dev)
            mask = (torch.rand(B, N, A, generator=g) < 0.85).to(dev)
            ref_d, ref_m = ops.pairwise_distance(xyz, mask)
            lo, hi = D.shard_rows(N, args.rank, args.world)
            for impl in (("native", "torch") if args.backend == "nccl" else ("torch",)):
                for fb in (False, True):
                    d, m = ref_d.clone(), ref_m.clone()
                    d[:, :lo] = -3.0; d[:, hi:] = -3.0
                    m[:, :lo] = False; m[:, hi:] = False
                    D.allgather_rows(d, impl=impl, force_broadcast=fb)
                    D.allgather_rows(m, impl=impl, force_broadcast=fb)
                    torch.cuda.synchronize(dev)
                    ok = same(d, ref_d) and torch.equal(m, ref_m)
                    results.append({"op": "allgather_rows", "B": B, "N": N, "A": A, "impl": impl,
                                    "force_broadcast": fb, "gather": "True", "ok": bool(ok)})
                    ok_all &= bool(ok)
        # the same through the StructureBatch methods
        from protstruc_amd import StructureBatch
        g = torch.Generator().manual_seed(77)
        xyz = torch.randn(2, 34, 15, 3, generator=g)
        mask = torch.rand(2, 34, 15, generator=g) < 0.9
        sb = StructureBatch.from_xyz(xyz, mask, device=dev)
        want_d, want_m = sb.pairwise_distance_matrix()
        d, m, _ = sb.pairwise_distance_matrix_sharded(gather=True)
        om, _ = sb.pairwise_dihedrals_sharded(["CA", "CB"], ["CA", "CB"], gather=True)
        ph, _ = sb.pairwise_planar_angles_sharded(["CA", "CB"], ["CB"], gather=True)
        torch.cuda.synchronize(dev)
        ok = (same(d, want_d) and torch.equal(m, want_m) and same(om, sb.pairwise_dihedrals(["CA", "CB"], ["CA", "CB"]))
              and same(ph, sb.pairwise_planar_angles(["CA", "CB"], ["CB"])))
        results.append({"op": "StructureBatch.*_sharded", "B": 2, "N": 34, "gather": "True", "ok": bool(ok)})
        ok_all &= bool(ok)
        dist.barrier()
    finally:
        D.destroy_native_comms()
        dist.destroy_process_group()
    with open(f"{args.out}.rank{args.rank}", "w") as f:
        json.dump({"rank": args.rank, "ok": ok_all, "device": str(dev), "cases": results}, f)
    return 0 if ok_all else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=2)
    ap.add_argument("--backend", default="gloo")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "rehearse_rowshard.json"))
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--rank", type=int, default=-1)
    ap.add_argument("--port", type=int, default=0)
    args = ap.parse_args()
    if args.rank >= 0:
        sys.exit(rank_main(args))

    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with socket.socket() as s:          # a free rendezvous port
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    # dmabuf IPC only on this pool's host driver: with the legacy IPC mode cross-process buffer sharing (RCCL, torch) fails
    # at hipIpcGetMemHandle; the pool exports 0, an unset variable becomes 0
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, os.path.abspath(__file__), "--world", str(args.world), "--backend", args.backend,
           "--out", args.out, "--port", str(port)] + (["--quick"] if args.quick else [])
    t0 = time.time()
    procs = [subprocess.Popen(cmd + ["--rank", str(r)], env=env) for r in range(args.world)]
    codes = []
    for p in procs:
        try:
            codes.append(p.wait(timeout=300))
        except subprocess.TimeoutExpired:
            p.kill()                       # exactly the child we started
            codes.append(-9)
    ranks = []
    for r in range(args.world):
        try:
            with open(f"{args.out}.rank{r}") as f:
                ranks.append(json.load(f))
            os.remove(f"{args.out}.rank{r}")
        except OSError:
            ranks.append({"rank": r, "ok": False, "error": "no verdict written"})
    ok = all(c == 0 for c in codes) and all(r.get("ok") for r in ranks)
    summary = {"ok": ok, "world": args.world, "backend": args.backend, "exit_codes": codes,
               "seconds": round(time.time() - t0, 1), "n_cases": sum(len(r.get("cases", [])) for r in ranks),
               "failed": [c for r in ranks for c in r.get("cases", []) if not c["ok"]], "ranks": ranks}
    with open(args.out, "w") as f:
        json.dump(summary, f, indent=1)
    print(json.dumps({k: summary[k] for k in ("ok", "world", "backend", "exit_codes", "seconds", "n_cases", "failed")}))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
