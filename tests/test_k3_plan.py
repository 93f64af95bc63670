"""``ps_k3_plan_f32`` / ``ps_featuriser_plan_f32`` -- which kernel a K3 / featuriser launch takes -- are pure host queries
(the library runs its own dispatchers in record-only mode, no HIP call), so the mapping shape -> dispatch arm is testable
without a GPU.  The table (tests/k3_families.py) is also what the GPU parity tests launch against the oracle."""
import ctypes

import pytest

from protstruc_amd import _lib
from tests.k3_families import (FEATURISER_ARM_KEYS, FEATURISER_SHAPES, K3_ARM_KEYS, K3_SHAPES, arm_key)


def _k3(entry):
    B, N, (npts, si, sj), rows, compact, mis, mode, want, _gpu = entry
    r0, r1 = rows if rows else (0, N)
    return _lib.k3_plan(B, N, 15, si, sj, npts, r0, r1, compact=compact, out_misalign=mis, exact_angles=mode, cu_count=256), want


def _feat(entry):
    B, N, fmis, mmis, mode, want, _gpu = entry
    return _lib.featuriser_plan(B, N, 15, float_misalign=fmis, mask_misalign=mmis, exact_angles=mode, cu_count=256), want


@pytest.mark.parametrize("entry", K3_SHAPES, ids=lambda e: f"B{e[0]}-N{e[1]}-np{e[2][0]}-i{len(e[2][1])}-mis{e[5]}-mode{e[6]}")
def test_every_k3_table_entry_selects_its_arm(entry):
    plan, want = _k3(entry)
    for k, v in want.items():
        assert plan[k] == v, (k, plan)
    assert plan["n_launches"] == 1 and plan["n_workgroups"] >= 1 and plan["kernel"].startswith("k3_")
    if plan["family"] == "sweep":
        assert plan["threads_per_workgroup"] == (512 if plan["faithful"] and plan["columns_per_lane"] == 4 else 1024) // plan["workgroups_per_cu"]
        assert plan["lds_bytes"] > 64 * 1024 and plan["rows_per_task"] in (2, 4, 6, 8)
        assert plan["n_workgroups"] <= 256 * plan["workgroups_per_cu"]
        assert plan["n_workgroups"] * plan["tasks_per_workgroup"] >= plan["n_tasks"]


@pytest.mark.parametrize("entry", FEATURISER_SHAPES, ids=lambda e: f"B{e[0]}-N{e[1]}-f{e[2]}-m{e[3]}-mode{e[4]}")
def test_every_featuriser_table_entry_selects_its_arm(entry):
    plan, want = _feat(entry)
    for k, v in want.items():
        assert plan[k] == v, (k, plan)
    assert plan["n_launches"] == 1 and plan["kernel"].startswith("k3_")
    if plan["family"] == "featurise":
        full = 512 if (plan["columns_per_lane"] == 4 or plan["faithful"]) else 1024
        assert plan["threads_per_workgroup"] == full // plan["workgroups_per_cu"]
        short = entry[1] <= 64 and not plan["vector_stores"]       # one column group: eight rows, a lane's chains are four row pairs
        assert plan["rows_per_task"] == (4 if plan["mask_store_mode"] == 2 else 8 if short else 2)


def test_the_tables_cover_every_arm_the_dispatchers_report():
    """Sweep chain lengths, batch sizes, point splits, alignments and modes through both dispatchers: every arm that comes out
    must be in the table (so: has a launch that is held to the oracle on the GPU)."""
    have = {arm_key(_k3(e)[0], K3_ARM_KEYS) for e in K3_SHAPES}
    splits = [(4, [1, 4], [1, 4]), (4, [0, 1, 4], [4]), (4, [2], [0, 1, 2]), (4, [0, 1], [2, 3]), (4, [0, 1, 2, 3], []), (4, [], [0, 1, 2, 3]),
              (3, [1, 4], [4]), (3, [1], [1, 4]), (3, [], [0, 1, 2]), (3, [4, 1, 0], [])]
    seen = set()
    for N in (1, 7, 16, 17, 32, 33, 64, 99, 100, 101, 128, 130, 140, 200, 255, 256, 300, 301, 302, 384, 450, 511, 512, 516, 1000, 2048, 4608):
        for B in (1, 3, 600, 2048, 5000):
            if B * N * N > (1 << 33):
                continue
            for npts, si, sj in splits:
                for mis in (0, 4, 8):
                    for mode in (0, 1, 2, 3):
                        seen.add(arm_key(_lib.k3_plan(B, N, 15, si, sj, npts, out_misalign=mis, exact_angles=mode, cu_count=256), K3_ARM_KEYS))
    assert seen <= have, sorted(seen - have)
    fhave = {arm_key(_feat(e)[0], FEATURISER_ARM_KEYS) for e in FEATURISER_SHAPES}
    fseen = set()
    for N in (5, 8, 33, 48, 63, 64, 70, 80, 96, 97, 100, 101, 110, 111, 112, 128, 129, 140, 160, 176, 192, 200, 255, 256, 258, 272, 288, 300, 301, 320, 336, 383, 384, 480, 496, 500, 511, 512, 516, 1030, 2048, 2100):
        for B in (1, 3, 600, 1024, 3000):
            for fmis, mmis in ((0, 0), (4, 0), (8, 0), (0, 5), (16, 16), (64, 0), (4, 3)):
                for mode in (0, 1, 2, 3):
                    fseen.add(arm_key(_lib.featuriser_plan(B, N, 15, float_misalign=fmis, mask_misalign=mmis, exact_angles=mode, cu_count=256),
                                      FEATURISER_ARM_KEYS))
    assert fseen <= fhave, sorted(fseen - fhave)


def test_plan_baseline_config3():
    """BASELINE config 3 (B=128, N=512): the three features and the featuriser, both arithmetic modes."""
    for mode, faithful in ((0, 0), (1, 1)):
        p = _lib.k3_plan(128, 512, 15, [1, 4], [1, 4], 4, exact_angles=mode, cu_count=256)
        nc = 4 if faithful else 2        # (the fast (2,2) split keeps its registers at two columns per lane)
        assert p["kernel"] == f"k3_sweep<NP=4,SRC=12,NC={nc},VEC=1,FAITHFUL={faithful}>" and p["n_workgroups"] == 256
        assert p["n_tasks"] == 128 * (512 // (64 * nc)) * 64 and p["tasks_per_workgroup"] == p["n_tasks"] // 256 and p["rows_per_task"] == 8
        p = _lib.k3_plan(128, 512, 15, [0, 1, 4], [4], 4, exact_angles=mode, cu_count=256)
        assert p["kernel"] == f"k3_sweep<NP=4,SRC=8,NC=4,VEC=1,FAITHFUL={faithful}>"
        p = _lib.k3_plan(128, 512, 15, [1, 4], [4], 3, exact_angles=mode, cu_count=256)
        assert p["kernel"] == f"k3_sweep<NP=3,SRC=4,NC=4,VEC=1,FAITHFUL={faithful}>"
    assert _lib.featuriser_plan(128, 512)["kernel"] == "k3_featurise<EXACT=0,NC=4,VEC=1,M16=1,WT=1,FAITHFUL=0>"
    assert _lib.featuriser_plan(128, 512, exact_sqrt=1, exact_angles=1)["kernel"] == "k3_featurise<EXACT=1,NC=2,VEC=1,M16=1,WT=1,FAITHFUL=1>"
    # the grid follows the device's CU count (a query argument: the plan makes no HIP call)
    assert _lib.k3_plan(128, 512, 15, [1, 4], [1, 4], 4, cu_count=304)["n_workgroups"] == 304
    assert _lib.k3_plan(0, 512, 15, [1, 4], [1, 4], 4)["family"] == "empty"
    assert _lib.featuriser_plan(3, 0)["family"] == "empty"


def test_plan_argument_errors():
    lib = _lib.load()
    plan = _lib.K3Plan(struct_size=ctypes.sizeof(_lib.K3Plan))
    arr = ctypes.c_int * 4
    src, atom = arr(0, 0, 1, 1), arr(1, 4, 1, 4)
    ok = (2, 256, 15, 4, src, atom, 0, 256, 256, 0, 0, 0, 256)
    assert lib.ps_k3_plan_f32(*ok, ctypes.byref(plan)) == 0 and plan.family == b"sweep"
    assert lib.ps_k3_plan_f32(*ok, None) == 1
    bad_size = _lib.K3Plan(struct_size=8)
    assert lib.ps_k3_plan_f32(*ok, ctypes.byref(bad_size)) == 1
    for bad in ((2, 256, 15, 5, src, atom, 0, 256, 256, 0, 0, 0, 256),       # n_points
                (2, 256, 15, 4, src, atom, 9, 3, 256, 0, 0, 0, 256),         # row_begin > row_end
                (2, 256, 15, 4, src, atom, 0, 257, 257, 0, 0, 0, 256),       # row_end > N
                (2, 256, 15, 4, src, atom, 0, 256, 256, 0, 2, 0, 256),       # a float plane cannot be 2 bytes off
                (2, 256, 15, 4, src, atom, 0, 256, 256, 0, 16, 0, 256),      # misalignment is an address modulo 16
                (2, 256, 15, 4, src, atom, 0, 256, 256, 0, 0, 4, 256),       # exact_angles outside 0..3
                (2, 256, 4, 4, src, atom, 0, 256, 256, 0, 0, 0, 256),        # atom slot 4 of a 4-atom residue
                (2, 256, 15, 4, None, atom, 0, 256, 256, 0, 0, 0, 256)):
        assert lib.ps_k3_plan_f32(*bad, ctypes.byref(plan)) == 1, bad
        assert plan.family == b"empty" and plan.n_launches == 0
    assert lib.ps_featuriser_plan_f32(2, 256, 15, 0, 0, 0, 0, 256, ctypes.byref(plan)) == 0 and plan.family == b"featurise"
    for bad in ((2, 256, 4, 0, 0, 0, 0, 256), (2, 256, 15, 0, 0, 2, 0, 256), (2, 256, 15, 0, 0, 0, 5, 256), (2, 256, 15, 128, 0, 0, 0, 256),
                (-1, 256, 15, 0, 0, 0, 0, 256)):
        assert lib.ps_featuriser_plan_f32(*bad, ctypes.byref(plan)) == 1, bad
