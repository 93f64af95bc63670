"""Atom-slot vocabulary of the geometry hot path.

Only the data the hot path needs is restated here, as a table: the five backbone atom slots with the spellings the
reference accepts for each (reference protstruc/general.py:4-20) and the 15-slot residue width
(protstruc/constants/__init__.py:1).
"""
import enum

MAX_N_ATOMS_PER_RESIDUE = 15

# slot -> (canonical name, accepted alternative spellings); the slot is the index on a residue's atom axis
_BACKBONE = (
    ("N", ("n",)),
    ("CA", ("Ca", "ca")),
    ("C", ("c",)),
    ("O", ("o",)),
    ("CB", ("Cb", "cb")),
)


class _AtomSlot(enum.IntEnum):
    """Behaviour of the enumeration below (an IntEnum without members can be extended through the functional API)."""

    @classmethod
    def is_valid(cls, name):
        # as in the reference, validity is judged on the upper-cased spelling against the canonical names only
        return name.upper() in cls._member_names_

    def __str__(self):
        return self.name


# ATOM["ca"], ATOM["Ca"] and ATOM["CA"] all resolve to slot 1 (later spellings of a slot become aliases of the
# canonical member); an unknown name raises KeyError; int(ATOM.CB) == 4.
ATOM = _AtomSlot("ATOM", [(spelling, slot) for slot, (canonical, others) in enumerate(_BACKBONE)
                          for spelling in (canonical,) + others])
