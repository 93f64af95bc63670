"""The C-ABI library loads (no GPU needed) and exports every symbol include/*.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "protstruc_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ps_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib_path():
    from protstruc_amd import build
    return build.build(force=False, verbose=False)  # hipcc cross-compiles gfx950 without a GPU


def test_header_declares_the_hot_path():
    names = declared_symbols()
    for must in ["ps_pairwise_distance_f32", "ps_backbone_dihedrals_f32", "ps_pairwise_angles_f32", "ps_frames_f32",
                 "ps_diffuse_f32", "ps_standardize_f32", "ps_affine_f32", "ps_abi_version", "ps_error_string"]:
        assert must in names


def test_library_exports_every_declared_symbol(lib_path):
    lib = ctypes.CDLL(lib_path)
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} is declared in include/protstruc_hip.h but not exported"
    lib.ps_abi_version.restype = ctypes.c_int
    assert lib.ps_abi_version() >= 1


def test_python_binding_matches_header(lib_path):
    from protstruc_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    lib = _lib.load()
    assert lib.ps_abi_version() >= 1
    # tuning knobs are host-side state: usable without a GPU
    old = _lib.get_tuning("k1_rows_per_block")
    _lib.set_tuning("k1_rows_per_block", 4)
    assert _lib.get_tuning("k1_rows_per_block") == 4
    _lib.set_tuning("k1_rows_per_block", old)
    with pytest.raises(_lib.HipLibraryError):
        _lib.set_tuning("no_such_knob", 1)


def test_argument_errors_are_reported_before_any_launch(lib_path):
    """Invalid arguments return hipErrorInvalidValue (1) without touching a device."""
    from protstruc_amd import _lib
    lib = _lib.load()
    assert lib.ps_pairwise_distance_f32(None, None, None, None, 1, 4, 15, 0, 4, 4, 0, None) == 1
    assert lib.ps_frames_f32(None, None, None, 1, 4, 15, 0, 1, 2, 1, None) == 1
    assert lib.ps_diffuse_f32(None, None, 1, 4, None, None, None) == 1
    assert b"invalid" in lib.ps_error_string(1).lower()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from protstruc_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libprotstruc_hip.so"))
    with pytest.raises(_lib.HipLibraryError, match="no CPU fallback"):
        _lib.load()
