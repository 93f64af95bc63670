"""Atom-slot vocabulary of the geometry hot path.

Only the data the hot path needs is restated here (reference:
protstruc/general.py:4-20 for the backbone atom slots and their spelling
aliases, protstruc/constants/__init__.py:1 for the 15-slot residue width).
"""
import enum

MAX_N_ATOMS_PER_RESIDUE = 15


class ATOM(enum.IntEnum):
    """Slot of each backbone atom inside a residue's atom axis.

    ``ATOM["ca"]``, ``ATOM["Ca"]`` and ``ATOM["CA"]`` all resolve to slot 1,
    an unknown name raises ``KeyError`` (reference: general.py:4-16).
    """

    N = 0
    n = 0
    CA = 1
    Ca = 1
    ca = 1
    C = 2
    c = 2
    O = 3  # noqa: E741
    o = 3
    CB = 4
    Cb = 4
    cb = 4

    @classmethod
    def is_valid(cls, value):
        # reference: general.py:18-20 -- validity is judged on the upper-cased name
        return value.upper() in cls._member_names_

    def __str__(self):
        return self.name
