"""ctypes binding of libprotstruc_hip.so (C ABI: include/protstruc_hip.h).

There is deliberately no fallback: if the shared library is missing or a launch
fails, the caller gets an exception -- never a silent CPU / eager-PyTorch path.
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# PROTSTRUC_AMD_LIB selects another build of the same sources (tools/ use the -DPS_EXPERIMENTS one)
LIB_PATH = os.environ.get("PROTSTRUC_AMD_LIB") or os.path.join(_HERE, "lib", "libprotstruc_hip.so")
EXPECTED_ABI = 5  # PS_ABI_VERSION of include/protstruc_hip.h; bumped together with any signature change


class K1Config(ctypes.Structure):
    """``ps_k1_config`` of include/protstruc_hip.h, field for field."""
    _fields_ = [(name, ctypes.c_int) for name in (
        "struct_size", "exact_sqrt", "variant", "flat", "rows_per_block", "lds_pad_kb", "flat_cpw",
        "flat_lds_pad_kb", "jt", "xcd_remap", "store_nt", "flat_fl_log2", "rowphase", "experiment")]


class K1Plan(ctypes.Structure):
    """``ps_k1_plan`` of include/protstruc_hip.h, field for field."""
    _fields_ = [("struct_size", ctypes.c_int), ("n_launches", ctypes.c_int), ("family", ctypes.c_char * 48),
                ("kernel", ctypes.c_char * 96), ("n_workgroups", ctypes.c_uint), ("lds_bytes", ctypes.c_uint),
                ("threads_per_workgroup", ctypes.c_int), ("n_workgroups_2", ctypes.c_uint), ("lds_bytes_2", ctypes.c_uint)]


class K3Plan(ctypes.Structure):
    """``ps_k3_plan`` of include/protstruc_hip.h, field for field."""
    _fields_ = [("struct_size", ctypes.c_int), ("n_launches", ctypes.c_int), ("family", ctypes.c_char * 32),
                ("kernel", ctypes.c_char * 96)] + \
               [(name, ctypes.c_int) for name in ("columns_per_lane", "vector_stores", "skips_dead_groups", "mask_store_mode",
                                                  "write_through", "faithful", "rows_per_task", "workgroups_per_cu",
                                                  "structures_per_segment")] + \
               [("n_workgroups", ctypes.c_uint), ("threads_per_workgroup", ctypes.c_int), ("lds_bytes", ctypes.c_uint),
                ("n_tasks", ctypes.c_uint), ("tasks_per_workgroup", ctypes.c_uint)]


_c_f32p = ctypes.c_void_p
_c_u8p = ctypes.c_void_p
_c_int = ctypes.c_int
_c_stream = ctypes.c_void_p

# name -> (restype, argtypes); mirrors include/protstruc_hip.h one to one
SIGNATURES = {
    "ps_abi_version": (_c_int, []),
    "ps_error_string": (ctypes.c_char_p, [_c_int]),
    "ps_has_experiments": (_c_int, []),
    "ps_k1_config_default": (None, [ctypes.POINTER(K1Config)]),
    "ps_pairwise_distance_f32": (_c_int, [_c_f32p, _c_u8p, _c_f32p, _c_u8p, _c_int, _c_int, _c_int, _c_int, _c_int,
                                          _c_int, _c_int, _c_stream]),
    "ps_pairwise_distance_cfg_f32": (_c_int, [_c_f32p, _c_u8p, _c_f32p, _c_u8p, _c_int, _c_int, _c_int, _c_int, _c_int,
                                              _c_int, _c_int, ctypes.POINTER(K1Config), _c_stream]),
    "ps_k1_plan_f32": (_c_int, [_c_int] * 10 + [ctypes.POINTER(K1Config), ctypes.POINTER(K1Plan)]),
    "ps_backbone_dihedrals_f32": (_c_int, [_c_f32p, _c_f32p, _c_u8p, _c_f32p, _c_u8p, _c_u8p, _c_u8p, _c_int, _c_int,
                                           _c_int, _c_stream]),
    "ps_pairwise_angles_f32": (_c_int, [_c_f32p, _c_f32p, _c_int, _c_int, _c_int, _c_int, ctypes.POINTER(_c_int),
                                        ctypes.POINTER(_c_int), _c_int, _c_int, _c_int, _c_int, _c_int, _c_stream]),
    "ps_k3_plan_f32": (_c_int, [_c_int, _c_int, _c_int, _c_int, ctypes.POINTER(_c_int), ctypes.POINTER(_c_int)] + [_c_int] * 7 +
                       [ctypes.POINTER(K3Plan)]),
    "ps_featuriser_plan_f32": (_c_int, [_c_int] * 8 + [ctypes.POINTER(K3Plan)]),
    "ps_frames_f32": (_c_int, [_c_f32p, _c_f32p, _c_f32p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                               _c_stream]),
    "ps_pointwise_f32": (_c_int, [_c_int, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, ctypes.c_longlong, _c_stream]),
    "ps_diffuse_f32": (_c_int, [_c_f32p, _c_f32p, _c_int, _c_int, ctypes.c_void_p, _c_f32p, _c_stream]),
    "ps_diffuse_frames_f32": (_c_int, [_c_f32p, _c_f32p, _c_int, _c_int, _c_int, ctypes.c_void_p, _c_f32p, _c_f32p, _c_f32p,
                                       _c_int, _c_int, _c_int, _c_int, _c_stream]),
    "ps_diffusion_trajectory_f32": (_c_int, [_c_f32p, _c_f32p, _c_int, _c_int, _c_int, _c_int, ctypes.c_void_p, _c_f32p,
                                             _c_f32p, _c_f32p, _c_int, _c_int, _c_int, _c_int, _c_stream]),
    "ps_inter_residue_geometry_f32": (_c_int, [_c_f32p, _c_u8p] + [_c_f32p] * 6 + [_c_u8p] * 3 + [_c_int, _c_int, _c_int,
                                                                                                  _c_int, _c_int, _c_stream]),
    "ps_rigid_f32": (_c_int, [_c_f32p, _c_f32p, _c_f32p, _c_int, _c_int, _c_f32p, _c_int, _c_int, _c_int, _c_int, _c_stream]),
    "ps_center_of_mass_f32": (_c_int, [_c_f32p, _c_f32p, _c_int, _c_int, _c_int, _c_int, _c_stream]),
    "ps_frames_to_backbone_f32": (_c_int, [_c_f32p, _c_f32p, _c_f32p, _c_int, _c_f32p, _c_int, _c_int, _c_int, _c_stream]),
    "ps_kabsch_f32": (_c_int, [_c_f32p, _c_f32p, _c_u8p, _c_f32p, _c_f32p, _c_int, _c_int, _c_int, _c_int, _c_stream]),
    "ps_min_dist_to_points_f32": (_c_int, [_c_f32p, _c_f32p, _c_f32p, _c_int, _c_int, _c_int, _c_int, _c_stream]),
    "ps_standardize_f32": (_c_int, [_c_f32p, _c_u8p, _c_f32p, _c_f32p, _c_int, _c_int, _c_int, _c_stream]),
    "ps_standardize_variant_f32": (_c_int, [_c_f32p, _c_u8p, _c_f32p, _c_f32p, _c_int, _c_int, _c_int, _c_int, _c_stream]),
    "ps_affine_f32": (_c_int, [_c_f32p, _c_f32p, _c_f32p, _c_int, _c_int, _c_stream]),
}

_lib = None


class HipLibraryError(RuntimeError):
    pass


def _try_build_in_tree():
    """The .so is a git-ignored build artefact; if it did not travel with the tree, compile it in place
    (hipcc is part of the ROCm image).  Never a fallback to another code path: if this fails, load() raises."""
    from . import build

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
    from . import build

    in_tree = os.path.abspath(LIB_PATH) == os.path.abspath(build.LIB_PATH)
    if in_tree and build.is_stale() and not os.environ.get("PROTSTRUC_AMD_NO_AUTOBUILD"):
        # missing, or older than csrc/*.hip / include/*.h: rebuild in place (hipcc is part of the ROCm image).
        # Without hipcc a stale library is refused below rather than loaded.
        _try_build_in_tree()
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} is missing: build it with `python -m protstruc_amd.build` "
            "(hipcc --offload-arch=gfx950). protstruc_amd has no CPU fallback.")
    if in_tree and build.is_stale():
        raise HipLibraryError(
            f"{LIB_PATH} is older than its sources (csrc/*.hip, include/*.h) and could not be rebuilt here: "
            "run `python -m protstruc_amd.build --force` where hipcc is available.")
    lib = ctypes.CDLL(LIB_PATH)
    try:
        lib.ps_abi_version.restype = _c_int
        abi = lib.ps_abi_version()
    except AttributeError as exc:
        raise HipLibraryError(f"{LIB_PATH} does not export ps_abi_version: not a protstruc_amd library") from exc
    if abi != EXPECTED_ABI:
        raise HipLibraryError(f"{LIB_PATH} has ABI version {abi}, this package binds version {EXPECTED_ABI}: "
                              "rebuild with `python -m protstruc_amd.build --force`")
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(code, what):
    if code != 0:
        msg = load().ps_error_string(code)
        raise HipLibraryError(f"{what} failed: hipError {code} ({msg.decode() if msg else '?'})")


# ---- K1 launch configuration: a per-device table on the HOST side --------------------------------------------------
# The library itself is stateless (every launch takes its configuration as an argument).  What the Python shell keeps
# is one small dict per device -- the autotuner's choice for that device's output buffers, plus whatever a test or
# tool set explicitly -- guarded by a lock and snapshotted into a fresh struct for every launch, so a thread that
# changes a knob can never tear the configuration another thread is launching with.
_K1_KEYS = {   # tuning key -> (struct field, lowest, highest)
    "k1_exact_sqrt": ("exact_sqrt", 0, 1), "k1_variant": ("variant", 0, 1), "k1_flat": ("flat", 0, 4),
    "k1_rows_per_block": ("rows_per_block", 1, 32), "k1_lds_pad_kb": ("lds_pad_kb", -1, 120),
    "k1_flat_cpw": ("flat_cpw", 1, 64), "k1_flat_lds_pad_kb": ("flat_lds_pad_kb", 0, 100),
    "k1_jt": ("jt", 0, 128), "k1_xcd_remap": ("xcd_remap", 0, 1), "k1_store_nt": ("store_nt", 0, 1),
    "k1_flat_fl_log2": ("flat_fl_log2", 0, 7), "k1_rowphase": ("rowphase", 0, 255), "k1_experiment": ("experiment", 0, 31),
}
_k1_lock = threading.RLock()
_k1_table = {}   # device index -> {field: value}


def _device_index(device=None):
    idx = getattr(device, "index", None)     # torch.device with an explicit index: the hot path
    if idx is not None:
        return idx
    if device is None:
        import torch
        return torch.cuda.current_device() if torch.cuda.is_available() else 0
    if isinstance(device, int):
        return device
    import torch
    d = torch.device(device)
    return d.index if d.index is not None else (torch.cuda.current_device() if torch.cuda.is_available() else 0)


def _k1_defaults():
    cfg = K1Config()
    load().ps_k1_config_default(ctypes.byref(cfg))
    d = {f: getattr(cfg, f) for f, _ in K1Config._fields_}
    # K1 uses the hardware square root (<= 1 ulp) unless the user asks for the correctly rounded one
    if os.environ.get("PROTSTRUC_AMD_EXACT_SQRT", "0") not in ("", "0"):
        d["exact_sqrt"] = 1
    return d


def _k1_entry(device):
    idx = _device_index(device)
    with _k1_lock:
        if idx not in _k1_table:
            _k1_table[idx] = _k1_defaults()
        return _k1_table[idx]


_k1_structs = {}   # device index -> ps_k1_config built from the table entry; REPLACED (never mutated) by set_tuning


def k1_config(device=None, **overrides):
    """Snapshot of ``device``'s K1 configuration as a ``ps_k1_config`` struct (fields overridden by keyword).
    Without overrides the struct is the cached one of the device's current settings: it is never modified in place
    (a change of settings installs a new struct), and the library copies it by value before launching, so sharing
    it between launches and threads keeps the snapshot semantics."""
    idx = _device_index(device)
    if not overrides:
        cfg = _k1_structs.get(idx)
        if cfg is not None:
            return cfg
    with _k1_lock:
        d = dict(_k1_entry(idx))
        if not overrides:
            cfg = _k1_structs[idx] = K1Config(**d)
            return cfg
    d.update(overrides)
    return K1Config(**d)


def k1_plan(B, N, A, row_begin=0, row_end=None, *, compact=False, dist_misalign=0, mask_misalign=0, has_atom_mask=True,
            device=None, **overrides):
    """Which kernel ``ps_pairwise_distance_cfg_f32`` takes for this shape under ``device``'s current configuration
    (fields overridden by keyword): a dict with ``family``, ``kernel``, ``n_launches``, ``n_workgroups``, ``lds_bytes``.
    Pure host query (``ps_k1_plan_f32``): the library runs its own dispatcher in record-only mode."""
    row_end = N if row_end is None else row_end
    out_rows, origin = (row_end - row_begin, row_begin) if compact else (N, 0)
    plan = K1Plan(struct_size=ctypes.sizeof(K1Plan))
    cfg = k1_config(device, **overrides)
    check(load().ps_k1_plan_f32(B, N, A, row_begin, row_end, out_rows, origin, dist_misalign, mask_misalign,
                                int(bool(has_atom_mask)), ctypes.byref(cfg), ctypes.byref(plan)), "ps_k1_plan_f32")
    return {"family": plan.family.decode(), "kernel": plan.kernel.decode(), "n_launches": plan.n_launches,
            "n_workgroups": plan.n_workgroups, "lds_bytes": plan.lds_bytes,
            "threads_per_workgroup": plan.threads_per_workgroup,
            **({"n_workgroups_2": plan.n_workgroups_2, "lds_bytes_2": plan.lds_bytes_2} if plan.n_launches > 1 else {})}


def _k3_plan_dict(plan):
    d = {name: getattr(plan, name) for name, _ in K3Plan._fields_ if name not in ("struct_size", "family", "kernel")}
    d["family"], d["kernel"] = plan.family.decode(), plan.kernel.decode()
    return d


def k3_plan(B, N, A, slots_i, slots_j, n_points, row_begin=0, row_end=None, *, compact=False, out_misalign=0, exact_angles=0,
            cu_count=0):
    """Which kernel ``ps_pairwise_angles_f32`` takes for this launch: a dict with ``family`` ("sweep", "flat_tiles",
    "small", "one_column", "empty"), ``kernel`` (name with template arguments), the layout (``columns_per_lane``,
    ``vector_stores`` -- flat_tiles: 2 = four-column tiles with 16-byte rows, 1 = two-column tiles, 0 = dword stores --,
    ``skips_dead_groups``, ``faithful``), ``rows_per_task``, ``workgroups_per_cu``, grid, workgroup size and LDS bytes.
    Pure host query (``ps_k3_plan_f32``): the library runs its own dispatcher in record-only mode; no GPU needed.
    ``cu_count`` <= 0 means 256 (MI355X)."""
    row_end = N if row_end is None else row_end
    out_rows, origin = (row_end - row_begin, row_begin) if compact else (N, 0)
    slots = [int(v) for v in slots_i] + [int(v) for v in slots_j]
    src = [0] * len(slots_i) + [1] * len(slots_j)
    arr = ctypes.c_int * n_points
    plan = K3Plan(struct_size=ctypes.sizeof(K3Plan))
    check(load().ps_k3_plan_f32(B, N, A, n_points, arr(*src[:n_points]), arr(*slots[:n_points]), row_begin, row_end, out_rows,
                                origin, out_misalign, int(exact_angles), cu_count, ctypes.byref(plan)), "ps_k3_plan_f32")
    return _k3_plan_dict(plan)


def featuriser_plan(B, N, A=15, *, float_misalign=0, mask_misalign=0, exact_sqrt=0, exact_angles=0, cu_count=0):
    """Which kernel ``ps_inter_residue_geometry_f32`` takes (``ps_featuriser_plan_f32``; see ``k3_plan``)."""
    plan = K3Plan(struct_size=ctypes.sizeof(K3Plan))
    check(load().ps_featuriser_plan_f32(B, N, A, float_misalign, mask_misalign, int(exact_sqrt), int(exact_angles), cu_count,
                                        ctypes.byref(plan)), "ps_featuriser_plan_f32")
    return _k3_plan_dict(plan)


_k1_refs = {}      # device index -> (struct, ctypes.byref(struct)) of the device's current settings


def k1_config_ref(idx):
    """``ctypes.byref`` of the cached configuration struct of device ``idx`` (the launch hot path: one dict lookup).
    The pair (struct, reference) is replaced, never mutated, when a setting changes; the library copies the struct by
    value before launching."""
    ent = _k1_refs.get(idx)      # the hit path takes no lock: entries are replaced, never mutated
    if ent is None:
        with _k1_lock:           # look up, build and store under the lock set_tuning holds while it drops the entry,
            ent = _k1_refs.get(idx)   # so a struct built from the settings before a change can never be cached after it
            if ent is None:
                cfg = k1_config(idx)
                ent = _k1_refs[idx] = (cfg, ctypes.byref(cfg))
    return ent[1]


def set_tuning(key, value, device=None):
    """Set one K1 knob for ``device`` (default: the current device).  Host-side state only."""
    value = int(value)
    if key not in _K1_KEYS:
        raise HipLibraryError(f"unknown tuning key {key!r} (known: {', '.join(sorted(_K1_KEYS))})")
    field, lo, hi = _K1_KEYS[key]
    if not lo <= value <= hi or (key == "k1_jt" and value not in (0, 16, 32, 64, 128)) \
            or (key == "k1_flat_fl_log2" and value in (1, 2, 3)) or (key == "k1_flat" and value == 3):
        raise HipLibraryError(f"tuning value {key}={value} outside its range")
    if key == "k1_experiment" and value and not load().ps_has_experiments():
        raise HipLibraryError("timing experiments are not compiled into the product library "
                              "(build with `python -m protstruc_amd.build --experiments` and set PROTSTRUC_AMD_LIB)")
    with _k1_lock:
        idx = _device_index(device)
        _k1_entry(idx)[field] = value
        _k1_structs.pop(idx, None)     # the next launch builds a fresh struct; structs in flight stay as they were
        _k1_refs.pop(idx, None)


# ---- K3 arithmetic mode: per device, host side (the library takes it per call, like exact_sqrt) ----------------------
_angle_mode = {}   # device index -> 0 fast / 1 the reference's order of operations


def set_exact_angles(flag, device=None):
    with _k1_lock:
        _angle_mode[_device_index(device)] = 1 if flag else 0


def get_exact_angles(device=None):
    idx = _device_index(device)
    v = _angle_mode.get(idx)
    if v is None:
        v = 1 if os.environ.get("PROTSTRUC_AMD_EXACT_ANGLES", "0") not in ("", "0") else 0
        with _k1_lock:
            v = _angle_mode.setdefault(idx, v)
    return v


def get_tuning(key, device=None):
    if key not in _K1_KEYS:
        raise HipLibraryError(f"unknown tuning key {key!r}")
    with _k1_lock:
        return _k1_entry(device)[_K1_KEYS[key][0]]


def all_tuning(device=None):
    """Every K1 knob of ``device`` as a dict (what ``bench.py`` reports)."""
    with _k1_lock:
        e = dict(_k1_entry(device))
    return {key: e[field] for key, (field, _, _) in sorted(_K1_KEYS.items())}
