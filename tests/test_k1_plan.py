"""``ps_k1_plan_f32`` -- which kernel a K1 launch takes -- is a pure host query (the library runs its own dispatcher in
record-only mode), so the mapping shape -> kernel family is testable without a GPU."""
import ctypes

import pytest

from protstruc_amd import _lib
from tests.k1_families import ALL_FAMILIES, FAMILY_SHAPES


def _plan(B, N, A, rows, compact, overrides, **kw):
    r0, r1 = rows if rows else (0, N)
    return _lib.k1_plan(B, N, A, r0, r1, compact=compact, device=0, **{k[3:]: v for k, v in overrides.items()}, **kw)


@pytest.mark.parametrize("entry", FAMILY_SHAPES, ids=lambda e: f"B{e[0]}-N{e[1]}-A{e[2]}-{e[6].replace(' ', '')}")
def test_every_table_entry_selects_its_family(entry):
    B, N, A, rows, compact, overrides, family = entry
    p = _plan(B, N, A, rows, compact, overrides)
    assert p["family"] == family, p
    assert p["n_launches"] == 1 and p["n_workgroups"] >= 1 and p["threads_per_workgroup"] == 256
    assert p["kernel"].startswith("k1_")


def test_the_table_covers_every_family_the_dispatcher_reports():
    assert {e[6] for e in FAMILY_SHAPES} == ALL_FAMILIES
    # and the dispatcher reports nothing outside that set, over a sweep of shapes, settings and alignments
    seen = set()
    for A in list(range(1, 18)) + [20, 25, 32, 37, 40, 64, 65, 70]:
        for N in (1, 5, 15, 16, 17, 20, 32, 33, 100, 512):
            for overrides in ({}, {"variant": 1}, {"flat": 0}, {"flat": 2}, {"flat": 4}, {"rowphase": 1}, {"rowphase": 2}):
                for mis in (0, 4):
                    p = _lib.k1_plan(2, N, A, device=0, dist_misalign=mis, mask_misalign=mis, **overrides)
                    seen.add(p["family"])
    assert seen <= ALL_FAMILIES, seen - ALL_FAMILIES
    assert seen == ALL_FAMILIES, ALL_FAMILIES - seen


def test_plan_headline_and_config_shapes():
    """The BASELINE shapes: headline / config 2 / config 4 (full and one of 8 row shards) take the pattern kernel."""
    for B, N, rows in ((64, 512, None), (64, 256, None), (32, 2048, None), (32, 2048, (256, 512))):
        p = _plan(B, N, 15, rows, False, {})
        assert p["family"] == "pattern" and p["kernel"] == "k1_pairdist_a15_pat<32>"
        n_rows = (rows[1] - rows[0]) if rows else N
        assert p["n_workgroups"] == B * n_rows * (N // 32)
    assert _lib.k1_plan(64, 500, 15, device=0)["family"] == "flat"
    assert _lib.k1_plan(8, 512, 1, device=0)["family"] == "rowphase"          # a CA trace
    assert _lib.k1_plan(8, 128, 1, device=0)["family"] == "ca_flat"           # a short one
    assert _lib.k1_plan(0, 512, 15, device=0)["family"] == "empty"


def test_plan_argument_errors_and_planes():
    lib = _lib.load()
    plan = _lib.K1Plan(struct_size=ctypes.sizeof(_lib.K1Plan))
    cfg = _lib.k1_config(0)
    ok = (2, 64, 15, 0, 64, 64, 0, 0, 0, 1)
    assert lib.ps_k1_plan_f32(*ok, ctypes.byref(cfg), ctypes.byref(plan)) == 0 and plan.family == b"pattern"
    assert lib.ps_k1_plan_f32(*ok, None, ctypes.byref(plan)) == 0 and plan.family == b"pattern"      # NULL = defaults
    assert lib.ps_k1_plan_f32(*ok, ctypes.byref(cfg), None) == 1
    bad_size = _lib.K1Plan(struct_size=4)
    assert lib.ps_k1_plan_f32(*ok, ctypes.byref(cfg), ctypes.byref(bad_size)) == 1
    for bad in ((2, 64, 0, 0, 64, 64, 0, 0, 0, 1),        # A = 0
                (2, 64, 15, 5, 3, 64, 0, 0, 0, 1),        # row_begin > row_end
                (2, 64, 15, 0, 65, 64, 0, 0, 0, 1),       # row_end > N
                (2, 64, 15, 0, 64, 64, 0, -1, -1, 1),     # neither plane requested
                (2, 64, 15, 0, 64, 64, 0, 3, 0, 1),       # a float plane cannot be 3 bytes off
                (2, 64, 15, 0, 64, 64, 0, 16, 0, 1)):     # misalignment is an address modulo 16
        assert lib.ps_k1_plan_f32(*bad, ctypes.byref(cfg), ctypes.byref(plan)) == 1, bad
    # only one plane requested / misaligned planes: still a valid launch, other kernels
    assert _lib.k1_plan(2, 64, 15, device=0, mask_misalign=-1)["family"] == "pattern"
    assert _lib.k1_plan(2, 64, 15, device=0, dist_misalign=4, mask_misalign=1)["family"] == "slot_decode"
    assert _lib.k1_plan(2, 40, 5, device=0, dist_misalign=8)["family"] == "element"
