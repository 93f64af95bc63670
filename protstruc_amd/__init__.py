"""protstruc_amd -- MI355X-native geometry hot path behind protstruc's StructureBatch API.

    from protstruc_amd import StructureBatch

The featurisers run in hand-written HIP kernels for gfx950 loaded from
``protstruc_amd/lib/libprotstruc_hip.so`` (C ABI: include/protstruc_hip.h).
Build it with ``python -m protstruc_amd.build``.
"""
from .general import ATOM, MAX_N_ATOMS_PER_RESIDUE  # noqa: F401
from .structure_batch import StructureBatch  # noqa: F401

__version__ = "0.1.0"
