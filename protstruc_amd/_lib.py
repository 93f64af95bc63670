"""ctypes binding of libprotstruc_hip.so (C ABI: include/protstruc_hip.h).

There is deliberately no fallback: if the shared library is missing or a launch
fails, the caller gets an exception -- never a silent CPU / eager-PyTorch path.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libprotstruc_hip.so")

_c_f32p = ctypes.c_void_p
_c_u8p = ctypes.c_void_p
_c_int = ctypes.c_int
_c_stream = ctypes.c_void_p

# name -> (restype, argtypes); mirrors include/protstruc_hip.h one to one
SIGNATURES = {
    "ps_abi_version": (_c_int, []),
    "ps_error_string": (ctypes.c_char_p, [_c_int]),
    "ps_set_tuning": (_c_int, [ctypes.c_char_p, _c_int]),
    "ps_get_tuning": (_c_int, [ctypes.c_char_p, ctypes.POINTER(_c_int)]),
    "ps_pairwise_distance_f32": (_c_int, [_c_f32p, _c_u8p, _c_f32p, _c_u8p, _c_int, _c_int, _c_int, _c_int, _c_int,
                                          _c_int, _c_int, _c_stream]),
    "ps_backbone_dihedrals_f32": (_c_int, [_c_f32p, _c_f32p, _c_u8p, _c_f32p, _c_u8p, _c_u8p, _c_u8p, _c_int, _c_int,
                                           _c_int, _c_stream]),
    "ps_pairwise_angles_f32": (_c_int, [_c_f32p, _c_f32p, _c_int, _c_int, _c_int, _c_int, ctypes.POINTER(_c_int),
                                        ctypes.POINTER(_c_int), _c_int, _c_int, _c_int, _c_int, _c_stream]),
    "ps_frames_f32": (_c_int, [_c_f32p, _c_f32p, _c_f32p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                               _c_stream]),
    "ps_pointwise_f32": (_c_int, [_c_int, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, ctypes.c_longlong, _c_stream]),
    "ps_diffuse_f32": (_c_int, [_c_f32p, _c_f32p, _c_int, _c_int, ctypes.c_void_p, _c_f32p, _c_stream]),
    "ps_diffuse_frames_f32": (_c_int, [_c_f32p, _c_f32p, _c_int, _c_int, _c_int, ctypes.c_void_p, _c_f32p, _c_f32p, _c_f32p,
                                       _c_int, _c_int, _c_int, _c_int, _c_stream]),
    "ps_diffusion_trajectory_f32": (_c_int, [_c_f32p, _c_f32p, _c_int, _c_int, _c_int, _c_int, ctypes.c_void_p, _c_f32p,
                                             _c_f32p, _c_f32p, _c_int, _c_int, _c_int, _c_int, _c_stream]),
    "ps_inter_residue_geometry_f32": (_c_int, [_c_f32p, _c_u8p] + [_c_f32p] * 6 + [_c_u8p] * 3 + [_c_int, _c_int, _c_int,
                                                                                                  _c_stream]),
    "ps_rigid_f32": (_c_int, [_c_f32p, _c_f32p, _c_f32p, _c_int, _c_int, _c_f32p, _c_int, _c_int, _c_int, _c_int, _c_stream]),
    "ps_center_of_mass_f32": (_c_int, [_c_f32p, _c_f32p, _c_int, _c_int, _c_int, _c_int, _c_stream]),
    "ps_frames_to_backbone_f32": (_c_int, [_c_f32p, _c_f32p, _c_f32p, _c_int, _c_f32p, _c_int, _c_int, _c_int, _c_stream]),
    "ps_kabsch_f32": (_c_int, [_c_f32p, _c_f32p, _c_u8p, _c_f32p, _c_f32p, _c_int, _c_int, _c_int, _c_int, _c_stream]),
    "ps_min_dist_to_points_f32": (_c_int, [_c_f32p, _c_f32p, _c_f32p, _c_int, _c_int, _c_int, _c_int, _c_stream]),
    "ps_standardize_f32": (_c_int, [_c_f32p, _c_u8p, _c_f32p, _c_f32p, _c_int, _c_int, _c_int, _c_stream]),
    "ps_affine_f32": (_c_int, [_c_f32p, _c_f32p, _c_f32p, _c_int, _c_int, _c_stream]),
}

_lib = None


class HipLibraryError(RuntimeError):
    pass


def _try_build_in_tree():
    """The .so is a git-ignored build artefact; if it did not travel with the tree, compile it in place
    (hipcc is part of the ROCm image).  Never a fallback to another code path: if this fails, load() raises."""
    from . import build

    if os.path.abspath(LIB_PATH) != os.path.abspath(build.LIB_PATH) or os.environ.get("PROTSTRUC_AMD_NO_AUTOBUILD"):
        return
    if not os.path.exists(build.HIPCC):
        return
    try:
        build.build(force=True, verbose=True)
    except Exception as exc:  # noqa: BLE001 -- reported by the caller as "library missing"
        print(f"[protstruc_amd] in-tree build failed: {exc}", flush=True)


def load():
    """Load the shared library once and type every entry point."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        _try_build_in_tree()
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} is missing: build it with `python -m protstruc_amd.build` "
            "(hipcc --offload-arch=gfx950). protstruc_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    # K1 uses the hardware square root (<= 1 ulp) unless the user asks for the correctly rounded one
    if os.environ.get("PROTSTRUC_AMD_EXACT_SQRT", "0") not in ("", "0"):
        lib.ps_set_tuning(b"k1_exact_sqrt", 1)
    return lib


def check(code, what):
    if code != 0:
        msg = load().ps_error_string(code)
        raise HipLibraryError(f"{what} failed: hipError {code} ({msg.decode() if msg else '?'})")


def set_tuning(key, value):
    check(load().ps_set_tuning(key.encode(), int(value)), f"ps_set_tuning({key})")


def get_tuning(key):
    v = ctypes.c_int(0)
    check(load().ps_get_tuning(key.encode(), ctypes.byref(v)), f"ps_get_tuning({key})")
    return v.value
