"""The multi-rank row-sharded path with the REAL HIP kernels, rehearsed on the one GPU of the test box.

The ranks are separate processes started by tests/conftest.py at session start (before pytest touched the GPU):
* 2 ranks, backend gloo, both on cuda:0: ``pairwise_distance_matrix_sharded`` / ``pairwise_angles_sharded`` for
  gather in {False, True, "recompute"}, residue counts divisible and not divisible by the world size, A in
  {15, 5, 37}, every result ``torch.equal`` to the single-GPU kernels (tools/rehearse_rowshard.py);
* 1 rank, backend nccl: the same calls through the native ``ps_allgather_rows`` / RCCL communicator (RCCL refuses two
  ranks on one device, so world > 1 over RCCL needs the driver's multi-GPU node; ``bench.py --gpus N`` runs it there).
Reference being sharded: protstruc.py:455-484 and :620-660.
"""
import json

import pytest

from tests import conftest

pytestmark = pytest.mark.gpu


def _verdict(name, timeout=330):
    code, log = conftest.wait_rehearsal(name, timeout)
    try:
        with open(conftest.REHEARSALS[name]["out"]) as f:
            summary = json.load(f)
    except OSError:
        pytest.fail(f"rehearsal {name} wrote no verdict (exit {code}); log:\n{log[-3000:]}")
    return code, summary, log


def test_two_ranks_gloo_real_kernels_bit_identical():
    code, s, log = _verdict("gloo_world2")
    assert code == 0 and s["ok"], (s["failed"], s["exit_codes"], log[-2000:])
    assert s["world"] == 2 and s["exit_codes"] == [0, 0]
    ops_seen = {(c["op"], c["gather"]) for r in s["ranks"] for c in r["cases"]}
    for op in ("distance", "angles4", "angles3"):
        for gather in ("False", "True", "recompute"):
            assert (op, gather) in ops_seen
    ns = {c["N"] for r in s["ranks"] for c in r["cases"]}
    assert any(n % 2 == 0 for n in ns) and any(n % 2 == 1 for n in ns)
    assert s["n_cases"] >= 2 * 30
    assert any(c["op"] == "StructureBatch.*_sharded" and c["ok"] for r in s["ranks"] for c in r["cases"])


def test_one_rank_rccl_native_gather():
    code, s, log = _verdict("rccl_world1")
    assert code == 0 and s["ok"], (s["failed"], s["exit_codes"], log[-2000:])
    impls = {c.get("impl") for r in s["ranks"] for c in r["cases"] if c["op"] == "distance" and c["gather"] == "True"}
    assert impls == {"native", "torch"}
    # the body of ps_allgather_rows_ex has executed on RCCL: grouped in-place ncclAllGather AND the per-(structure, owner)
    # ncclBroadcast form, through the native library and through torch.distributed, bits intact
    direct = {(c["impl"], c["force_broadcast"]) for r in s["ranks"] for c in r["cases"]
              if c["op"] == "allgather_rows" and c["ok"]}
    assert direct == {("native", False), ("native", True), ("torch", False), ("torch", True)}


def test_bench_gpus2_self_launch_as_the_driver_invokes_it():
    """``python3 bench.py --gpus 2 --backend gloo --steps 2 --warmup 1 --no-cpu-baseline`` with no launcher around it
    (the command form the driver uses for N > 1): the parent starts two rank processes of itself, relays ONE JSON line
    and exits 0; the headline check and the check after the gather are both ok and the strong-scaling keys are
    top-level."""
    code, log = conftest.wait_rehearsal("bench_gpus2", 750)
    out = open(conftest.REHEARSALS["bench_gpus2"]["stdout"]).read()
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert code == 0, (code, out[-500:], log[-3000:])
    assert len(lines) == 1, lines
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["warmup"] == 1 and r["scaling"] == "weak"
    assert r["check"] == "ok" and r["rowshard_allgather"]["check_after_gather"] == "ok" and "rowshard_error" not in r
    assert r["dist_backend"] == "gloo" and r["rccl_ranks"] == 0
    assert r["roofline"]["kernel"].startswith("k1_pairdist_a15_pat") and r["roofline"]["kernel_family"] == "pattern"
    assert r["roofline"]["buffer_fill_GBps"] > 0 and 0 < r["roofline"]["frac_of_buffer_fill"] < 2
    for key in ("config4_kernel_only_pairs_per_s", "config4_kernel_only_efficiency_vs_1gpu",
                "config4_allgather_ingress_GBps_per_rank", "config4_xgmi_ingress_bound_GBps_per_rank",
                "config4_end_to_end_ms"):
        assert isinstance(r[key], float) and r[key] > 0, key
    assert len(r["per_rank_kernel_ms"]) == 2
