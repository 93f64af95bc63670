"""ctypes binding of libprotstruc_rccl.so (C ABI: include/protstruc_rccl.h) -- the RCCL all-gather of row shards.

Loaded only by the multi-GPU path (``protstruc_amd.distributed``).  Import PyTorch before loading: the library
names RCCL by SONAME (librccl.so.1), so inside a PyTorch process it binds to the RCCL PyTorch already loaded and the
process holds one copy.
"""
import ctypes
import os

from ._lib import HipLibraryError

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libprotstruc_rccl.so")
EXPECTED_ABI = 2
COMM_ID_BYTES = 128
GATHER_FORCE_BROADCAST = 1   # PS_GATHER_FORCE_BROADCAST

_vp, _int = ctypes.c_void_p, ctypes.c_int
SIGNATURES = {
    "ps_rccl_abi_version": (_int, []),
    "ps_rccl_version": (_int, [ctypes.POINTER(_int)]),
    "ps_comm_unique_id": (_int, [_vp]),
    "ps_comm_create": (_int, [ctypes.POINTER(_vp), _vp, _int, _int]),
    "ps_comm_destroy": (_int, [_vp]),
    "ps_comm_rank": (_int, [_vp, ctypes.POINTER(_int), ctypes.POINTER(_int)]),
    "ps_shard_rows": (_int, [_int, _int, _int, ctypes.POINTER(_int), ctypes.POINTER(_int)]),
    "ps_allgather_rows": (_int, [_vp, _vp, _int, _int, ctypes.c_longlong, _vp]),
    "ps_allgather_rows_ex": (_int, [_vp, _vp, _int, _int, ctypes.c_longlong, _int, _vp]),
    "ps_comm_error_string": (ctypes.c_char_p, [_int]),
}

_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401 -- first: its RCCL is then the one librccl.so.1 resolves to

    from . import build

    if build.rccl_is_stale() and os.path.exists(build.HIPCC) and not os.environ.get("PROTSTRUC_AMD_NO_AUTOBUILD"):
        try:
            build.build_rccl(force=True, verbose=True)
        except Exception as exc:  # noqa: BLE001 -- reported below as "missing / stale"
            print(f"[protstruc_amd] in-tree build of libprotstruc_rccl.so failed: {exc}", flush=True)
    if not os.path.exists(LIB_PATH) or build.rccl_is_stale():
        raise HipLibraryError(f"{LIB_PATH} is missing or older than its sources: run `python -m protstruc_amd.build`")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.ps_rccl_abi_version() != EXPECTED_ABI:
        raise HipLibraryError(f"{LIB_PATH} has ABI version {lib.ps_rccl_abi_version()}, expected {EXPECTED_ABI}")
    _lib = lib
    return lib


def check(code, what):
    if code != 0:
        msg = load().ps_comm_error_string(code)
        raise HipLibraryError(f"{what} failed: code {code} ({msg.decode() if msg else '?'})")


def rccl_version():
    v = _int(0)
    check(load().ps_rccl_version(ctypes.byref(v)), "ps_rccl_version")
    return v.value
