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


def _verdict(name, timeout=420):
    if name not in conftest.REHEARSALS:
        pytest.fail("the rehearsal launchers were not started (conftest.pytest_sessionstart found no GPU?)")
    r = conftest.REHEARSALS[name]
    try:
        code = r["proc"].wait(timeout=timeout)
    except Exception:  # noqa: BLE001
        r["proc"].kill()
        pytest.fail(f"rehearsal {name} did not finish in {timeout} s; log:\n" + open(r["log"]).read()[-3000:])
    log = open(r["log"]).read()
    try:
        with open(r["out"]) as f:
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
