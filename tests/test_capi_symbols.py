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
    header = open(HEADER).read()
    assert lib.ps_abi_version() == int(re.search(r"#define PS_ABI_VERSION (\d+)", header).group(1))
    assert lib.ps_has_experiments() == 0, "the product library must not contain the timing experiments"
    for gone in ("ps_set_tuning", "ps_get_tuning", "ps_k1_set_tuning", "ps_k1_get_tuning"):
        assert not hasattr(lib, gone), f"{gone}: the library must not hold tuning state any more"


def test_python_binding_matches_header(lib_path):
    from protstruc_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    lib = _lib.load()
    assert lib.ps_abi_version() == _lib.EXPECTED_ABI
    # the struct the binding passes is the struct the header declares, field for field
    header = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    body = re.search(r"typedef struct ps_k1_config \{(.*?)\} ps_k1_config;", header, flags=re.S).group(1)
    assert re.findall(r"int\s+(\w+)\s*;", body) == [f for f, _ in _lib.K1Config._fields_]
    cfg = _lib.K1Config()
    lib.ps_k1_config_default(ctypes.byref(cfg))
    assert cfg.struct_size == ctypes.sizeof(_lib.K1Config)
    assert (cfg.flat, cfg.rows_per_block, cfg.flat_cpw, cfg.xcd_remap, cfg.exact_sqrt, cfg.experiment) == (1, 1, 1, 1, 0, 0)
    assert cfg.lds_pad_kb == -1          # by chain length: 36 KB from 256 residues on, 20 KB below


def test_tuning_is_a_per_device_host_table(lib_path):
    """K1 knobs live in a per-device table on the Python side and reach the library as a per-call argument: setting
    one device's knob never changes another device's, and unknown keys / values are refused."""
    from protstruc_amd import _lib
    old0, old1 = _lib.get_tuning("k1_rows_per_block", 0), _lib.get_tuning("k1_rows_per_block", 1)
    try:
        _lib.set_tuning("k1_rows_per_block", 4, device=1)
        assert _lib.get_tuning("k1_rows_per_block", 1) == 4 and _lib.get_tuning("k1_rows_per_block", 0) == old0
        assert _lib.k1_config(1).rows_per_block == 4 and _lib.k1_config(0).rows_per_block == old0
        assert _lib.k1_config(0, rows_per_block=2).rows_per_block == 2 and _lib.get_tuning("k1_rows_per_block", 0) == old0
        assert set(_lib.all_tuning(0)) == set(_lib._K1_KEYS)
    finally:
        _lib.set_tuning("k1_rows_per_block", old1, device=1)
    with pytest.raises(_lib.HipLibraryError):
        _lib.set_tuning("no_such_knob", 1)
    with pytest.raises(_lib.HipLibraryError):
        _lib.set_tuning("k1_rows_per_block", 33)
    with pytest.raises(_lib.HipLibraryError):
        _lib.set_tuning("k1_jt", 96)
    with pytest.raises(_lib.HipLibraryError, match="experiments"):
        _lib.set_tuning("k1_experiment", 2)          # store-only timing mode: not in the product library


def test_k1_config_is_validated_before_any_launch(lib_path):
    """A malformed ps_k1_config is refused (hipErrorInvalidValue) before anything is launched; B = 0 keeps the call
    off the device, so this runs without a GPU (the pointers are never dereferenced)."""
    from protstruc_amd import _lib
    lib = _lib.load()
    fake = ctypes.c_void_p(0x1000)

    def call(cfg):
        return lib.ps_pairwise_distance_cfg_f32(fake, None, fake, fake, 0, 32, 15, 0, 32, 32, 0,
                                                None if cfg is None else ctypes.byref(cfg), None)

    assert call(None) == 0 and call(_lib.k1_config(0)) == 0
    for field, value in [("struct_size", 8), ("struct_size", 0), ("experiment", 2), ("experiment", 1), ("variant", 2),
                         ("flat", 5), ("rows_per_block", 0), ("rows_per_block", 33), ("flat_cpw", 0), ("jt", 96),
                         ("lds_pad_kb", 121), ("lds_pad_kb", -2), ("flat_fl_log2", 3), ("flat_fl_log2", 8), ("flat", 3), ("rowphase", 3),
                         ("flat_lds_pad_kb", -1)]:
        assert call(_lib.k1_config(0, **{field: value})) == 1, (field, value)


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


def test_stale_or_foreign_library_is_refused(monkeypatch, tmp_path, lib_path):
    """A library built from other sources than the tree's (digest mismatch) is rebuilt or refused, never loaded; a
    library with another ABI version is refused."""
    import shutil
    from protstruc_amd import _lib, build
    assert not build.is_stale(lib_path)
    # (1) same library, but the recorded source digest says it was built from something else
    fake_lib = tmp_path / "libprotstruc_hip.so"
    shutil.copy(lib_path, fake_lib)
    (tmp_path / "libprotstruc_hip.so.srchash").write_text("0" * 64 + "\n")
    assert build.is_stale(str(fake_lib))
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(fake_lib))
    monkeypatch.setattr(build, "LIB_PATH", str(fake_lib))
    monkeypatch.setenv("PROTSTRUC_AMD_NO_AUTOBUILD", "1")
    with pytest.raises(_lib.HipLibraryError, match="older than its sources"):
        _lib.load()
    # (2) right digest, wrong ABI number expected by the binding
    (tmp_path / "libprotstruc_hip.so.srchash").write_text(build.source_hash() + "\n")
    monkeypatch.setattr(_lib, "EXPECTED_ABI", _lib.EXPECTED_ABI + 1)
    with pytest.raises(_lib.HipLibraryError, match="ABI version"):
        _lib.load()
