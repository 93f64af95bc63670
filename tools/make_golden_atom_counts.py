#!/usr/bin/env python3
"""Golden vectors for pairwise_distance_matrix at the atom counts that have their own (fixed-A) kernels and at shapes
those kernels actually take (N >= 16): atom14, atom37, the reference test's 25, and backbone-only 3 / 4 / 5 / 8.

Runs ONLY in the build container (the reference is imported exactly as tools/make_golden.py does it):

    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden_atom_counts.py --reference /root/reference

For the large atom counts the full (B,N,N,A,A) output would be megabytes, so the fixture keeps the inputs, every
mask-plane checksum, and the reference's output for a seeded sample of (b, i, j) blocks (whole A x A blocks, incl. the
diagonal and the first / last rows, so row changes inside a chunk are covered)."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import import_reference, npy, synth  # noqa: E402


# G13 (round 2): the atom counts with fixed-A flat / row-tile kernels.  G14 (round 3): single atoms (CA traces), atom
# pairs and the other small counts the row-phase kernel takes, at lengths of every alignment phase and below 16.
SETS = {
    "g13_dist_atom_counts": (130, [(2, 19, 14), (1, 17, 37), (2, 16, 25), (2, 23, 3), (2, 21, 4), (3, 18, 5),
                                   (1, 33, 8), (1, 20, 16)]),
    "g14_dist_small_atom_counts": (150, [(2, 37, 1), (2, 18, 2), (2, 9, 6), (2, 21, 7), (1, 30, 10), (1, 11, 13),
                                         (3, 7, 1), (1, 19, 9)]),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
    args = ap.parse_args()
    SB, _ = import_reference(args.reference)
    for name, (seed0, shapes) in SETS.items():
        write_set(SB, args.out, name, seed0, shapes)


def write_set(SB, outdir, name, seed0, shapes):
    out, seen = {}, set()
    for seed, (B, N, A) in enumerate(shapes, start=seed0):
        xyz, mask = synth(seed, B, N, A, p=0.85)
        if A >= 14:
            xyz[0, N // 2, A - 1] = float("nan")      # a missing atom: NaN distances, mask untouched
        d, m = SB.from_xyz(xyz, mask).pairwise_distance_matrix()
        assert d.shape == (B, N, N, A, A) and m.dtype == torch.bool
        g = torch.Generator().manual_seed(seed)
        n_blocks = 48
        bs = torch.randint(0, B, (n_blocks,), generator=g)
        i_s = torch.randint(0, N, (n_blocks,), generator=g)
        js = torch.randint(0, N, (n_blocks,), generator=g)
        # always include the corners and one diagonal block
        i_s[:4] = torch.tensor([0, 0, N - 1, N - 1]); js[:4] = torch.tensor([0, N - 1, 0, N - 1]); js[4] = i_s[4]
        tag = f"a{A}" if A not in seen else f"a{A}n{N}"     # a second shape of one atom count carries its length
        seen.add(A)
        out.update({f"{tag}_xyz": xyz, f"{tag}_atom_mask": mask, f"{tag}_b": bs, f"{tag}_i": i_s, f"{tag}_j": js,
                    f"{tag}_dist_blocks": d[bs, i_s, js], f"{tag}_mask_blocks": m[bs, i_s, js],
                    f"{tag}_mask_row_sums": m.sum((3, 4)).to(torch.int32),          # (B,N,N) exact counts
                    f"{tag}_dist_row_nansum": torch.nan_to_num(d, nan=0.0).double().sum((3, 4)).float()})
        print(tag, tuple(d.shape))
    np.savez_compressed(os.path.join(outdir, name + ".npz"), **{k: npy(v) for k, v in out.items()})
    print("wrote", name + ".npz")


if __name__ == "__main__":
    main()
