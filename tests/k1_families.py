"""One table of K1 launches, one per kernel family behind ``ps_pairwise_distance_cfg_f32``.

Used twice: ``tests/test_k1_plan.py`` (CPU) asserts through the library's own dispatcher (``ps_k1_plan_f32``) that each
entry selects the family it names and that every family the dispatcher can report appears in the table;
``tests/test_gpu_parity.py::test_k1_every_kernel_family_vs_oracle`` runs exactly these launches on the GPU against the
oracle.  Together: every kernel family is reached by a shape that is held to the oracle.

Entry: (B, N, A, (row_begin, row_end) or None, compact, {K1 tuning overrides}, expected family)."""

FAMILY_SHAPES = [
    # A = 15: pattern (N % 16 == 0), flat pattern (any other N >= 16), slot-decode (the simple variant; N < 16 without the
    # row-phase kernel, which is the default there)
    (2, 64, 15, None, False, {}, "pattern"),
    (2, 256, 15, (64, 128), False, {}, "pattern"),
    (2, 96, 15, None, False, {"k1_jt": 128, "k1_lds_pad_kb": 8}, "pattern"),
    (2, 96, 15, None, False, {"k1_jt": 16, "k1_lds_pad_kb": 0}, "pattern"),
    (3, 50, 15, None, False, {}, "flat"),
    (2, 37, 15, (5, 30), True, {}, "flat"),
    (2, 64, 15, None, False, {"k1_flat": 2, "k1_flat_fl_log2": 5}, "flat"),
    (2, 12, 15, None, False, {"k1_rowphase": 2}, "slot_decode"),
    (3, 12, 15, (2, 9), False, {}, "rowphase"),
    (2, 48, 15, None, False, {"k1_variant": 1}, "slot_decode"),
    # fixed-A flat pattern kernels (the even counts 14, 16, 24, 32; A = 15 with k1_flat = 4 as the cross-check)
    (2, 40, 14, None, False, {}, "flatA"),
    (1, 20, 32, (3, 17), False, {}, "flatA"),
    (2, 33, 24, None, False, {}, "flatA"),
    (2, 19, 16, (2, 19), True, {}, "flatA"),
    (2, 35, 15, None, False, {"k1_flat": 4}, "flatA"),
    # row-tile kernels of A = 4, 8
    (2, 129, 4, None, False, {}, "rowtile"),
    (2, 33, 8, (1, 32), False, {}, "rowtile"),
    # row-phase kernel (round 3): every other atom count up to 13, any length
    (3, 501, 1, None, False, {}, "rowphase"),
    (3, 33, 1, (2, 30), False, {}, "rowphase"),
    (4, 100, 1, None, False, {"k1_rowphase": 1}, "rowphase"),
    (2, 200, 2, None, False, {}, "rowphase"),
    (2, 18, 2, (3, 18), False, {}, "rowphase"),
    (2, 21, 5, (4, 20), True, {}, "rowphase"),
    (2, 40, 7, None, False, {}, "rowphase"),
    (1, 30, 13, None, False, {}, "rowphase"),
    (2, 100, 3, None, False, {}, "rowphase"),
    (2, 6, 3, None, False, {"k1_rowphase": 1}, "rowphase"),
    # ... and, through its run-time atom-count instantiations (even / odd), every other count up to 64
    (2, 40, 20, None, False, {}, "rowphase"),
    (2, 17, 33, (0, 9), True, {}, "rowphase"),
    (1, 9, 64, None, False, {}, "rowphase"),
    (2, 10, 21, None, False, {}, "rowphase"),
    (2, 33, 14, None, False, {"k1_rowphase": 1}, "rowphase"),
    (1, 20, 37, (3, 17), False, {}, "rowphase"),
    (2, 33, 25, None, False, {}, "rowphase"),
    # CA traces (A = 1) of 8 .. 255 residues, full matrices (round 4): the flat kernel
    (3, 33, 1, None, False, {}, "ca_flat"),
    (40, 9, 1, None, False, {}, "ca_flat"),
    (2, 255, 1, None, False, {}, "ca_flat"),
    # short chains of 2 .. 16 atoms per residue (up to 64 residues at A <= 4, 16 at A = 9 .. 13, 7 at 14 .. 16), full matrices (round 4):
    # the generic flat kernel
    (3, 5, 15, None, False, {}, "small_flat"),
    (2, 18, 2, None, False, {}, "small_flat"),
    (5, 6, 3, None, False, {}, "small_flat"),
    (3, 16, 4, None, False, {}, "small_flat"),
    (2, 3, 13, None, False, {}, "small_flat"),
    (7, 2, 5, None, False, {}, "small_flat"),
    # element-per-lane kernel: A > 64, or nothing else eligible (N < 16 without the row-phase kernel; the simple variant)
    (1, 8, 70, None, False, {}, "element"),
    (2, 10, 20, None, False, {"k1_rowphase": 2}, "element"),
    (2, 40, 7, None, False, {"k1_variant": 1}, "element"),
]

ALL_FAMILIES = {"pattern", "flat", "slot_decode", "flatA", "rowtile", "rowphase", "ca_flat", "small_flat", "element"}
