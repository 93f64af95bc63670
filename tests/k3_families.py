"""One table of K3 / featuriser launches, one (or more) per ARM of the dispatchers behind ``ps_pairwise_angles_f32`` and
``ps_inter_residue_geometry_f32`` (round 5; K1 has the same arrangement: tests/k1_families.py).

Used twice: ``tests/test_k3_plan.py`` (CPU) asserts through the library's own dispatchers in record-only mode
(``ps_k3_plan_f32`` / ``ps_featuriser_plan_f32``) that each entry selects the arm it names and that no arm the dispatchers can
report over a sweep of shapes is missing from the table; ``tests/test_gpu_parity.py::test_k3_every_dispatch_arm_vs_oracle``
/ ``::test_featuriser_every_dispatch_arm_vs_oracle`` run exactly these launches on the GPU against the ORACLE (fast
arithmetic: the conditioning gate of SURVEY hard part 3; faithful: no dihedral beyond 1e-5) inside sentinel buffers.

An arm = (family, columns_per_lane, vector_stores, skips_dead_groups, faithful, workgroups_per_cu) for K3, plus
(mask_store_mode, write_through) for the featuriser (families featurise, featurise_tiles, one_column).  All plans are taken at 256 compute units (MI355X)."""

CA_CB__CA_CB = (4, [1, 4], [1, 4])       # (2,2) dihedral, SRC = 12
N_CA_CB__CB = (4, [0, 1, 4], [4])        # (3,1) dihedral, SRC = 8
C__N_CA_C = (4, [2], [0, 1, 2])          # (1,3) dihedral, SRC = 14: two columns per lane in the fast arithmetic (registers), like (2,2)
CA_CB__CB = (3, [1, 4], [4])             # (2,1) planar angle
ALL_I = (4, [0, 1, 2, 3], [])            # every point from the row residue (SRC = 0)


def arm(family, nc=1, vec=0, skips=0, faithful=0, wgs=0):
    return {"family": family, "columns_per_lane": nc, "vector_stores": vec, "skips_dead_groups": skips, "faithful": faithful,
            "workgroups_per_cu": wgs}


# (B, N, feature, rows or None, compact, out_misalign, exact_angles, expected arm, run on the GPU)
K3_SHAPES = [
    # ---- fast arithmetic ----
    (3, 16, CA_CB__CA_CB, None, False, 0, 0, arm("small", nc=16), True),
    (5, 20, N_CA_CB__CB, (3, 17), True, 0, 0, arm("small", nc=32), True),
    (2, 32, CA_CB__CB, None, False, 4, 0, arm("small", nc=32), True),
    (3, 64, CA_CB__CA_CB, None, False, 0, 0, arm("flat_tiles", nc=4, vec=2, skips=1, wgs=2), True),
    (700, 60, N_CA_CB__CB, None, False, 4, 0, arm("flat_tiles", nc=4, vec=0, skips=1, wgs=2), True),   # several staging passes per workgroup
    (5, 57, C__N_CA_C, (4, 31), True, 0, 0, arm("flat_tiles", nc=4, vec=0, skips=1, wgs=2), True),
    (700, 48, N_CA_CB__CB, None, False, 4, 0, arm("flat_tiles", nc=4, vec=0, skips=1, wgs=2), True),
    (5, 33, C__N_CA_C, (4, 31), True, 0, 0, arm("flat_tiles", nc=4, vec=0, skips=1, wgs=2), True),
    (2, 140, CA_CB__CB, None, False, 0, 0, arm("flat_tiles", nc=4, vec=2, skips=1, wgs=2), True),            # three column groups for 140 columns: flat
    (2, 130, CA_CB__CA_CB, (1, 130), False, 0, 0, arm("flat_tiles", nc=4, vec=1, skips=1, wgs=2), True),
    (2, 99, CA_CB__CB, (1, 98), False, 0, 0, arm("flat_tiles", nc=4, vec=0, skips=1, wgs=2), True),
    (700, 65, N_CA_CB__CB, None, False, 4, 0, arm("flat_tiles", nc=4, vec=0, skips=1, wgs=2), True),
    (5, 80, C__N_CA_C, (4, 77), True, 0, 0, arm("flat_tiles", nc=4, vec=2, skips=1, wgs=2), True),
    (300, 100, CA_CB__CA_CB, (3, 98), False, 0, 0, arm("flat_tiles", nc=4, vec=2, skips=1, wgs=2), True),   # four-column tiles, a row range inside a full-size plane
    (3, 64, CA_CB__CB, None, False, 8, 0, arm("flat_tiles", nc=4, vec=1, skips=1, wgs=2), True),             # N % 4 == 0 but only 8-byte aligned: two-column tiles
    (900, 36, CA_CB__CA_CB, None, False, 0, 0, arm("flat_tiles", nc=4, vec=1, skips=1, wgs=2), True),       # N % 4 == 0 but 81 wide tiles would fill 2 tasks of 64: two-column tiles
    (3, 64, CA_CB__CA_CB, None, False, 0, 2, arm("one_column"), True),
    (2, 256, N_CA_CB__CB, None, False, 0, 2, arm("one_column"), True),                       # the diagnostic bit
    (1, 50000, CA_CB__CB, (0, 13000), True, 0, 0, arm("one_column"), False),                 # 32-bit store offsets would overflow
    (1, 14000, N_CA_CB__CB, None, False, 0, 0, arm("one_column"), False),                    # the rows do not fit the LDS
    (2, 512, CA_CB__CA_CB, None, False, 0, 0, arm("sweep", nc=2, vec=1, wgs=1), True),       # BASELINE config 3's layout, (2,2) split
    (2, 512, N_CA_CB__CB, None, False, 0, 0, arm("sweep", nc=4, vec=1, wgs=1), True),        # ... the (3,1) split
    (2, 256, CA_CB__CB, (10, 200), True, 0, 0, arm("sweep", nc=4, vec=1, wgs=1), True),
    (2, 512, C__N_CA_C, None, False, 0, 0, arm("sweep", nc=2, vec=1, wgs=1), True),
    (2, 254, N_CA_CB__CB, None, False, 0, 0, arm("sweep", nc=2, vec=1, wgs=1), True),        # N % 4 != 0
    (2, 256, CA_CB__CA_CB, None, False, 8, 0, arm("sweep", nc=2, vec=1, wgs=1), True),       # an 8-byte aligned output
    (2, 301, CA_CB__CB, None, False, 0, 0, arm("sweep", nc=4, vec=0, skips=1, wgs=1), True),
    (2, 301, N_CA_CB__CB, (7, 290), False, 0, 0, arm("sweep", nc=4, vec=0, skips=1, wgs=1), True),
    (2, 511, CA_CB__CA_CB, None, False, 0, 0, arm("sweep", nc=2, vec=0, skips=1, wgs=1), True),
    (2, 301, CA_CB__CA_CB, None, False, 0, 0, arm("sweep", nc=2, vec=0, skips=1, wgs=1), True),
    (3, 127, CA_CB__CB, None, False, 0, 0, arm("sweep", nc=2, vec=0, skips=1, wgs=1), True),
    (2, 256, CA_CB__CB, None, False, 4, 0, arm("sweep", nc=4, vec=0, skips=1, wgs=1), True),  # a 4-byte misaligned output
    (2, 190, CA_CB__CB, None, False, 0, 0, arm("sweep", nc=4, vec=0, skips=1, wgs=1), True),  # even N, but three groups instead of four
    (2048, 128, CA_CB__CA_CB, None, False, 0, 0, arm("sweep", nc=2, vec=1, wgs=2), True),     # two workgroups per CU
    (2048, 127, CA_CB__CB, None, False, 0, 0, arm("sweep", nc=2, vec=0, skips=1, wgs=2), True),
    (2, 512, ALL_I, None, False, 0, 0, arm("sweep", nc=4, vec=1, wgs=1), True),
    (2048, 256, CA_CB__CB, None, False, 0, 0, arm("sweep", nc=4, vec=1, wgs=2), True),
    (2048, 255, CA_CB__CA_CB, None, False, 0, 0, arm("sweep", nc=2, vec=0, skips=1, wgs=2), True),
    (2048, 255, CA_CB__CB, None, False, 0, 0, arm("sweep", nc=4, vec=0, skips=1, wgs=2), True),
    # ---- the reference's order of operations (bit 0 of exact_angles) ----
    (3, 16, CA_CB__CA_CB, None, False, 0, 1, arm("small", nc=16, faithful=1), True),
    (2, 31, CA_CB__CB, None, False, 0, 1, arm("small", nc=32, faithful=1), True),
    (3, 64, N_CA_CB__CB, None, False, 0, 1, arm("flat_tiles", nc=4, vec=2, skips=1, faithful=1, wgs=2), True),
    (40, 120, N_CA_CB__CB, (5, 118), True, 0, 1, arm("flat_tiles", nc=4, vec=2, skips=1, faithful=1, wgs=2), True),   # four-column tiles, compact row range
    (700, 62, CA_CB__CB, (3, 50), False, 0, 1, arm("flat_tiles", nc=4, vec=1, skips=1, faithful=1, wgs=2), True),
    (700, 50, CA_CB__CB, (3, 50), False, 0, 1, arm("flat_tiles", nc=4, vec=1, skips=1, faithful=1, wgs=2), True),
    (300, 90, CA_CB__CA_CB, (3, 90), False, 0, 1, arm("flat_tiles", nc=4, vec=1, skips=1, faithful=1, wgs=2), True),
    (40, 77, N_CA_CB__CB, None, False, 4, 1, arm("flat_tiles", nc=4, vec=0, skips=1, faithful=1, wgs=2), True),       # odd N: dword stores
    (2, 256, CA_CB__CA_CB, None, False, 0, 3, arm("one_column", faithful=1), True),
    (2, 512, CA_CB__CA_CB, None, False, 0, 1, arm("sweep", nc=4, vec=1, faithful=1, wgs=1), True),
    (2, 512, C__N_CA_C, None, False, 0, 1, arm("sweep", nc=4, vec=1, faithful=1, wgs=1), True),
    (2, 256, CA_CB__CB, None, False, 0, 1, arm("sweep", nc=4, vec=1, faithful=1, wgs=1), True),
    (2, 254, N_CA_CB__CB, None, False, 0, 1, arm("sweep", nc=2, vec=1, faithful=1, wgs=1), True),       # N % 4 != 0
    (2, 491, CA_CB__CA_CB, (5, 300), True, 0, 1, arm("sweep", nc=4, vec=0, skips=1, faithful=1, wgs=1), True),
    (2, 301, CA_CB__CA_CB, (5, 300), True, 0, 1, arm("flat_tiles", nc=4, vec=0, skips=1, faithful=1, wgs=2), True),
    (3, 127, CA_CB__CB, None, False, 0, 1, arm("sweep", nc=2, vec=0, skips=1, faithful=1, wgs=1), True),
    (2048, 128, N_CA_CB__CB, None, False, 0, 1, arm("sweep", nc=2, vec=1, faithful=1, wgs=2), True),
    (2048, 256, CA_CB__CA_CB, None, False, 0, 1, arm("sweep", nc=4, vec=1, faithful=1, wgs=2), True),
    (2048, 255, CA_CB__CA_CB, None, False, 0, 1, arm("sweep", nc=4, vec=0, skips=1, faithful=1, wgs=2), True),
    (2048, 127, CA_CB__CB, None, False, 0, 1, arm("sweep", nc=2, vec=0, skips=1, faithful=1, wgs=2), True),
]


def farm(family, nc=1, vec=0, mask=0, wt=0, faithful=0, wgs=0):
    return {"family": family, "columns_per_lane": nc, "vector_stores": vec, "mask_store_mode": mask, "write_through": wt,
            "faithful": faithful, "workgroups_per_cu": wgs}


# (B, N, float plane misalignment in bytes, mask plane misalignment, exact_angles, expected arm, run on the GPU)
FEATURISER_SHAPES = [
    (3, 5, 0, 0, 0, farm("one_column"), True),   # below the tile kernel's minimum
    (2, 256, 0, 0, 2, farm("one_column"), True),   # the diagnostic bit
    (1, 2200, 0, 0, 0, farm("one_column"), True),   # rows + column points + masks beyond the LDS
    (3, 5, 0, 0, 1, farm("one_column", faithful=1), True),
    (2, 256, 0, 0, 3, farm("one_column", faithful=1), True),
    (3, 48, 0, 0, 0, farm("featurise_tiles", nc=4, vec=1, mask=3, wt=0, wgs=2), True),   # 8-byte float stores, 2-byte mask stores
    (700, 64, 0, 0, 0, farm("featurise_tiles", nc=4, vec=2, mask=4, wt=0, wgs=2), True),   # several staging passes per workgroup
    (3, 33, 0, 0, 0, farm("featurise_tiles", nc=4, vec=0, mask=0, wt=0, wgs=2), True),   # odd N: dword and byte stores
    (3, 80, 4, 3, 0, farm("featurise_tiles", nc=4, vec=0, mask=0, wt=0, wgs=2), True),   # planes on 4-byte / odd boundaries
    (3, 48, 0, 5, 0, farm("featurise_tiles", nc=4, vec=1, mask=0, wt=0, wgs=2), True),
    (3, 48, 4, 0, 0, farm("featurise_tiles", nc=4, vec=0, mask=3, wt=0, wgs=2), True),
    (2, 160, 0, 0, 0, farm("featurise_tiles", nc=4, vec=2, mask=4, wt=0, wgs=2), True),   # even N, 83 % of a sweep's lanes
    (2, 200, 0, 0, 0, farm("featurise_tiles", nc=4, vec=2, mask=4, wt=0, wgs=2), True),
    (2, 192, 0, 0, 0, farm("featurise_tiles", nc=4, vec=2, mask=4, wt=0, wgs=2), True),   # three column groups: the sweep would write dwords
    (2, 300, 0, 0, 0, farm("featurise_tiles", nc=4, vec=2, mask=4, wt=0, wgs=2), True),   # five column groups
    (2, 288, 0, 0, 1, farm("featurise_tiles", nc=4, vec=2, mask=4, wt=0, faithful=1, wgs=2), True),
    (3, 48, 0, 0, 1, farm("featurise_tiles", nc=4, vec=1, mask=3, wt=0, faithful=1, wgs=2), True),
    (700, 64, 0, 0, 1, farm("featurise_tiles", nc=4, vec=2, mask=4, wt=0, faithful=1, wgs=2), True),
    (3, 80, 4, 3, 1, farm("featurise_tiles", nc=4, vec=0, mask=0, wt=0, faithful=1, wgs=2), True),
    (3, 48, 0, 5, 1, farm("featurise_tiles", nc=4, vec=1, mask=0, wt=0, faithful=1, wgs=2), True),
    (3, 60, 4, 0, 1, farm("featurise_tiles", nc=4, vec=0, mask=3, wt=0, faithful=1, wgs=2), True),
    (3, 39, 0, 0, 1, farm("featurise_tiles", nc=4, vec=0, mask=0, wt=0, faithful=1, wgs=2), True),
    (2, 512, 0, 0, 0, farm("featurise", nc=4, vec=1, mask=2, wt=1, wgs=1), True),   # BASELINE config 3's layout
    (2, 496, 0, 0, 0, farm("featurise", nc=4, vec=1, mask=2, wt=0, wgs=1), True),
    (2, 500, 0, 0, 0, farm("featurise", nc=4, vec=1, mask=1, wt=0, wgs=1), True),
    (2, 512, 0, 5, 0, farm("featurise", nc=4, vec=1, mask=1, wt=0, wgs=1), True),   # mask planes off the 16-byte grid
    (2, 301, 0, 0, 0, farm("featurise", nc=4, vec=0, mask=1, wt=0, wgs=1), True),
    (2, 512, 4, 0, 0, farm("featurise", nc=4, vec=0, mask=1, wt=0, wgs=1), True),   # float planes on a 4-byte boundary only
    (1024, 191, 0, 0, 0, farm("featurise", nc=4, vec=0, mask=1, wt=0, wgs=2), True),   # two workgroups per CU
    (3, 129, 0, 0, 0, farm("featurise_tiles", nc=4, vec=0, mask=0, wt=0, wgs=2), True),   # odd N, 67 % of a sweep's lanes
    (1024, 256, 0, 5, 0, farm("featurise", nc=4, vec=1, mask=1, wt=0, wgs=2), True),
    (1024, 256, 16, 16, 0, farm("featurise", nc=4, vec=1, mask=2, wt=0, wgs=2), True),
    (1024, 256, 0, 0, 0, farm("featurise", nc=4, vec=1, mask=2, wt=1, wgs=2), True),
    (3, 128, 0, 0, 0, farm("featurise", nc=2, vec=1, mask=2, wt=1, wgs=1), True),
    (3, 112, 0, 0, 0, farm("featurise", nc=2, vec=1, mask=2, wt=0, wgs=1), True),
    (3, 110, 0, 0, 0, farm("featurise", nc=2, vec=1, mask=1, wt=0, wgs=1), True),
    (3, 101, 0, 0, 0, farm("featurise", nc=2, vec=0, mask=1, wt=0, wgs=1), True),
    (1024, 128, 0, 0, 0, farm("featurise", nc=2, vec=1, mask=2, wt=1, wgs=2), True),
    (1024, 112, 0, 0, 0, farm("featurise", nc=2, vec=1, mask=2, wt=0, wgs=2), True),
    (1024, 110, 0, 0, 0, farm("featurise", nc=2, vec=1, mask=1, wt=0, wgs=2), True),
    (1024, 101, 0, 0, 0, farm("featurise", nc=2, vec=0, mask=1, wt=0, wgs=2), True),
    (2, 512, 0, 0, 1, farm("featurise", nc=2, vec=1, mask=2, wt=1, faithful=1, wgs=1), True),
    (2, 496, 0, 0, 1, farm("featurise_tiles", nc=4, vec=2, mask=4, wt=0, faithful=1, wgs=2), True),
    (2, 500, 0, 0, 1, farm("featurise_tiles", nc=4, vec=2, mask=4, wt=0, faithful=1, wgs=2), True),
    (2, 301, 0, 0, 1, farm("featurise", nc=2, vec=0, mask=1, wt=0, faithful=1, wgs=1), True),
    (1024, 128, 0, 0, 1, farm("featurise_tiles", nc=4, vec=2, mask=4, wt=0, faithful=1, wgs=2), True),
    (1024, 112, 0, 0, 1, farm("featurise_tiles", nc=4, vec=2, mask=4, wt=0, faithful=1, wgs=2), True),
    (1024, 110, 0, 0, 1, farm("featurise", nc=2, vec=1, mask=1, wt=0, faithful=1, wgs=2), True),
    (1024, 101, 0, 0, 1, farm("featurise", nc=2, vec=0, mask=1, wt=0, faithful=1, wgs=2), True),
    (2, 498, 0, 0, 1, farm("featurise", nc=2, vec=1, mask=1, wt=0, faithful=1, wgs=1), True),   # faithful sweep (no four-column tiles here)
    (1100, 128, 8, 0, 1, farm("featurise", nc=2, vec=1, mask=2, wt=0, faithful=1, wgs=2), True),   # faithful sweep (no four-column tiles here)
    (2, 112, 8, 0, 1, farm("featurise", nc=2, vec=1, mask=2, wt=0, faithful=1, wgs=1), True),   # faithful sweep: planes 8-byte aligned only
    (1024, 512, 0, 0, 1, farm("featurise", nc=2, vec=1, mask=2, wt=1, faithful=1, wgs=2), False),   # (config-3 length in a large batch: plan only, 27 GB of planes)
]


def arm_key(plan, keys):
    return tuple(plan[k] for k in keys)


K3_ARM_KEYS = ("family", "columns_per_lane", "vector_stores", "skips_dead_groups", "faithful", "workgroups_per_cu")
FEATURISER_ARM_KEYS = ("family", "columns_per_lane", "vector_stores", "mask_store_mode", "write_through", "faithful", "workgroups_per_cu")
