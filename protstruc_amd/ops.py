"""Tensor-level wrappers over the C ABI (include/protstruc_hip.h).

Each function takes device tensors, allocates the outputs with PyTorch (the
caller owns every buffer; the library never allocates) and launches on
PyTorch's current HIP stream, so calls order with surrounding torch ops and can
be captured by ``torch.cuda.graph``.  Inputs on the CPU raise: there is no
CPU path in this package.
"""
from __future__ import annotations

import ctypes
import threading
import time
from typing import Optional, Sequence, Tuple

import torch

from . import _lib


RNG_STATE_WORDS = 528  # PS_RNG_STATE_WORDS of include/protstruc_hip.h


def _require_device(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"protstruc_amd: `{name}` lives on {t.device}; the geometry kernels are HIP-only "
            "(no CPU fallback). Move the batch to the GPU first (StructureBatch(..., device='cuda')).")


def _f32c(t: torch.Tensor, name: str) -> torch.Tensor:
    _require_device(t, name)
    if t.dtype != torch.float32:
        t = t.to(torch.float32)
    return t.contiguous()


def _u8c(t: Optional[torch.Tensor], name: str) -> Optional[torch.Tensor]:
    """A mask as a contiguous one-byte-per-entry tensor for the kernels, which test every byte against zero (C ABI:
    "any non-zero input byte counts as true"): bool and uint8 tensors are passed as they are -- only their address is
    used, so not even a reinterpreting view is created -- anything else is reduced to its truth value first."""
    if t is None:
        return None
    _require_device(t, name)
    if t.dtype == torch.bool or t.dtype == torch.uint8:
        return t.contiguous()
    return (t != 0).contiguous()


def _check_rng_state(rng_state: Optional[torch.Tensor], device: torch.device) -> None:
    """The kernels atomically update words 1, 2 and 16 + 16 r (r < 32) of ``rng_state`` on the device: anything but a
    contiguous int64 tensor of RNG_STATE_WORDS words on the coordinates' own GPU would be an out-of-bounds (or
    host-pointer) access, so it is refused before the launch."""
    if rng_state is None:
        return
    if not isinstance(rng_state, torch.Tensor) or rng_state.dtype != torch.int64:
        raise ValueError(f"rng_state must be an int64 tensor of {RNG_STATE_WORDS} words")
    if not rng_state.is_cuda or rng_state.device != device:
        raise ValueError(f"rng_state lives on {rng_state.device}, the coordinates on {device}")
    if rng_state.ndim != 1 or rng_state.numel() < RNG_STATE_WORDS or not rng_state.is_contiguous():
        raise ValueError(f"rng_state must be a contiguous 1-D int64 tensor of {RNG_STATE_WORDS} words "
                         f"(got shape {tuple(rng_state.shape)})")


def _same_device(ref: torch.Tensor, **tensors) -> None:
    for name, t in tensors.items():
        if t is not None and t.device != ref.device:
            raise ValueError(f"`{name}` lives on {t.device}, the coordinates on {ref.device}")


def _check_out(t: Optional[torch.Tensor], shape, name: str, device: torch.device) -> None:
    if t is None:
        return
    if (not isinstance(t, torch.Tensor) or tuple(t.shape) != tuple(shape) or t.dtype != torch.float32
            or not t.is_contiguous() or t.device != device):
        raise ValueError(f"{name} must be a contiguous float32 tensor of shape {tuple(shape)} on {device}")


def _ptr(t: Optional[torch.Tensor]):
    # a plain int is what ctypes wants for a c_void_p argument (None = NULL); no wrapper object per pointer
    return None if t is None else t.data_ptr()


# torch's current HIP stream of a device as a raw handle: the private accessor torch's own kernel launchers use
# (0.1 us) where it exists, else the public Stream object (1.2 us per call)
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(t: torch.Tensor):
    if _raw_stream is not None:
        return _raw_stream(t.device.index)
    return torch.cuda.current_stream(t.device).cuda_stream


class _NoSwitch:
    """`with` target used when the tensor's device is already the current one (the common case): entering
    torch.cuda.device() costs two device switches, about a quarter of a small call's host time."""

    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_SWITCH = _NoSwitch()


def _on(device: torch.device):
    """Context that makes ``device`` current for allocations and launches."""
    return _NO_SWITCH if device.index == torch.cuda.current_device() else torch.cuda.device(device)


# ---- optional, per-device choice of K1's output granule per workgroup -----------------------------------------
# How fast K1's store stream is absorbed depends on the physical memory behind the output buffers (DESIGN.md section 4,
# "the two classes of allocation": 6.2-7.3 TB/s for the same kernel on different allocations of one process), and so
# does which launch configuration is the fastest.  The library default (ps_k1_config_default: 32-residue tiles, idle LDS by
# chain length -- 36 KB = 3 workgroups per CU from 256 residues on, 20 KB = 5 below) is the configuration that won on seven
# of eight boxes in round 4 (profiles/r04_k1_tuner_tables.log);
# nothing is timed behind the caller's back.  The tuner is an explicit call -- ops.autotune_pairwise_distance(), which
# bench.py makes before its warm-up and reports -- or, with PROTSTRUC_AMD_AUTOTUNE=1 (read at import;
# ops.set_implicit_autotune at run time), runs on the first large call of each kind per device.  It writes that DEVICE's
# entry of the host-side table in _lib.py (a per-call argument to the library; other devices and threads are never
# affected) and never runs during stream capture.  History of the candidates: NOTES.md.
_K1_TUNED = {}
_K1_TUNE_LOCK = threading.Lock()
import os as _os
_AUTOTUNE_ENV = bool(_os.environ.get("PROTSTRUC_AMD_AUTOTUNE"))   # read once, at import: the launch path does no environment lookups


def set_implicit_autotune(flag: bool) -> None:
    """Opt in to (or out of) tuning K1's launch configuration on the first large call of each kind per device -- what
    ``PROTSTRUC_AMD_AUTOTUNE=1`` in the environment at import time selects.  Off by default."""
    global _AUTOTUNE_ENV
    _AUTOTUNE_ENV = bool(flag)


def _k1_launch_entry(*args):
    """First call: bind the library's entry point, then replace this trampoline with it."""
    global _K1_LAUNCH
    _K1_LAUNCH = _lib.load().ps_pairwise_distance_cfg_f32
    return _K1_LAUNCH(*args)


_K1_LAUNCH = _k1_launch_entry
# K1 kernel dispatches this process has issued through this module (every launch is exactly one kernel dispatch, the
# autotuner's included): bench.py reports the ordinals of its timed launches so that a rocprofv3 kernel trace of the same
# process can be cut at exactly those dispatches (tools/summarize_rocprof.py ranges)
K1_DISPATCHES = [0]
# Candidates of the explicit tuner, as ps_k1_config fields.  The default comes first: the choice moves away from it
# only for a clear (>= 1 %) gain in the MEAN launch time.  Pattern kernel (N % 16 == 0): KB of idle LDS per workgroup
# (only lowers the number of resident workgroups per CU: fewer concurrent store streams) and column residues per tile.
# Around the default -- 32-residue tiles at 5 workgroups per CU -- sit 4 per CU (24 KB: +1 % on slow buffers and on some
# fast ones, -3 % on others), 6 per CU (16 KB), 3 per CU (36 KB: +1-4 % on slow and medium buffers, -4 % on fast ones), the
# 16-residue tile, the 64-residue tile and the 128-residue tile + 8 KB that was the default until late round 3
# (profiles/r03_k1_ab_lean_*.log).
# Round 4: 36 KB won on seven of eight boxes at B=64, N=512 (1.5-3.8 %; profiles/r04_k1_tuner_tables.log) and is what the
# library default (-1: by chain length) now takes from 256 residues on; 20 KB stays a candidate.
_K1_CANDIDATE_PATTERN = (
    {"rows_per_block": 1, "lds_pad_kb": -1, "jt": 0},      # the default: 32-residue tiles, idle LDS by chain length (36 KB = 3 workgroups per CU from N = 256, 20 KB = 5 below)
    {"rows_per_block": 1, "lds_pad_kb": 20, "jt": 32},
    {"rows_per_block": 1, "lds_pad_kb": 24, "jt": 32},
    {"rows_per_block": 1, "lds_pad_kb": 16, "jt": 32},
    {"rows_per_block": 1, "lds_pad_kb": 36, "jt": 32},
    {"rows_per_block": 1, "lds_pad_kb": 0, "jt": 16},
    {"rows_per_block": 1, "lds_pad_kb": 20, "jt": 64},
    {"rows_per_block": 1, "lds_pad_kb": 8, "jt": 128},
)
# flat kernel (any other N >= 16): chunks per workgroup, KB of idle LDS
# (round 3: 32-pair chunks -- 36 KB per short-lived workgroup, 4 workgroups per CU -- run 6.1-6.3 TB/s on every buffer:
# 3-5 % ahead of the 128-pair default on slow buffers, 13 % behind on fast ones; profiles/r03_k1_ab_small_granule.log)
_K1_CANDIDATE_FLAT = (
    {"flat_cpw": 1, "flat_lds_pad_kb": 0, "flat_fl_log2": 0},      # the default: 64-pair chunks
    {"flat_cpw": 1, "flat_lds_pad_kb": 8, "flat_fl_log2": 0},
    {"flat_cpw": 1, "flat_lds_pad_kb": 0, "flat_fl_log2": 7},
    {"flat_cpw": 2, "flat_lds_pad_kb": 0, "flat_fl_log2": 0},
    {"flat_cpw": 1, "flat_lds_pad_kb": 0, "flat_fl_log2": 5},
)


def _cand_label(c) -> str:
    if "rows_per_block" in c:
        pad = "+autoKB" if c["lds_pad_kb"] < 0 else (f"+{c['lds_pad_kb']}KB" if c["lds_pad_kb"] else "")
        return f"{c['rows_per_block']}row" + pad + (f" jt{c['jt']}" if c["jt"] else "")
    return f"{c['flat_cpw']}chunk" + (f"+{c['flat_lds_pad_kb']}KB" if c["flat_lds_pad_kb"] else "") + \
        (f" {1 << c['flat_fl_log2']}pairs" if c.get("flat_fl_log2") else "")


def _autotune_k1(device, args, n_pairs: int, N: int, A: int, force: bool = False) -> None:
    if not force and not _AUTOTUNE_ENV:   # (the hot path does not even get here: see the caller)
        return
    if A != 15 or N < 16 or n_pairs < (1 << 22):
        return
    # which kernel this shape takes decides which knob is tuned (pairwise_distance.hip: flat_eligible)
    pattern = (N % 16 == 0)
    result_key = "rows_per_block" if pattern else "flat_cpw"
    if result_key in _K1_TUNED.get(device, {}):
        return
    if torch.cuda.is_current_stream_capturing():
        return
    if not pattern and (_lib.get_tuning("k1_flat", device) == 0 or _lib.get_tuning("k1_variant", device) != 0):
        return
    with _K1_TUNE_LOCK:
        if result_key in _K1_TUNED.get(device, {}):   # another thread tuned this device while we waited
            return
        lib = _lib.load()
        stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        candidates = _K1_CANDIDATE_PATTERN if pattern else _K1_CANDIDATE_FLAT
        if pattern:
            # candidate 0 (lds_pad_kb = -1, jt = 0) resolves to {36 KB from N = 256, 20 KB below; 32-residue tiles}
            # (pairwise_distance.hip, launch_a15): the explicit candidate that names the same launch is not timed -- a "pick"
            # between two names of one launch would be noise and would make bench.py re-time the default for nothing
            alias = {"rows_per_block": 1, "lds_pad_kb": 36 if N >= 256 else 20, "jt": 32}
            candidates = tuple(c for c in candidates if c != alias)
        # one private configuration struct per candidate: nothing shared is touched while timing
        cfgs = [_lib.k1_config(device, **c) for c in candidates]

        def launch(k):
            K1_DISPATCHES[0] += 1
            _lib.check(lib.ps_pairwise_distance_cfg_f32(*args, ctypes.byref(cfgs[k]), stream),
                       "ps_pairwise_distance_cfg_f32 (autotune)")

        # the first ~70 ms of GPU work after idle run ~2.5 % slow (clock ramp): warm up before timing anything,
        # then time the candidates in interleaved rounds
        t_end = time.perf_counter() + 0.12
        while time.perf_counter() < t_end:
            launch(0)
            torch.cuda.current_stream(device).synchronize()
        # Each candidate's figure is its MEAN over all rounds (5 x 3 launches), not its minimum or median: the small-tile
        # kernels spread 3-4 % from launch to launch with a tail of slow launches (profiles/r03_k1_launch_series.log;
        # a median of 2.60 ms went with a 20-launch mean of 2.72 ms in profiles/r03_final_bench_n1.json), and a caller's
        # throughput is the reciprocal of the mean.
        rounds = [[] for _ in candidates]
        for _ in range(5):
            for k in range(len(candidates)):
                launch(k)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                launch(k)
                launch(k)
                launch(k)
                e1.record()
                e1.synchronize()
                rounds[k].append(e0.elapsed_time(e1) / 3)
        timings = [sum(r) / len(r) for r in rounds]
        best = 0
        for k in range(1, len(candidates)):
            if timings[k] < timings[best] * 0.99:   # prefer the earlier candidate unless the gain is clear (1 %)
                best = k
        for field, value in candidates[best].items():
            _lib.set_tuning("k1_" + field, value, device)
        report = dict(candidates[best])
        report["flat_ms" if not pattern else "ms"] = {_cand_label(c): timings[k] for k, c in enumerate(candidates)}
        _K1_TUNED.setdefault(device, {}).update(report)


def set_exact_sqrt(flag: bool, device=None) -> None:
    """K1 arithmetic on ``device`` (default: the current one): False (default) = hardware square root, exact for
    85 % of inputs and 1 ulp off for the rest; True = correctly rounded square root (slower on fast allocations).
    ``PROTSTRUC_AMD_EXACT_SQRT=1`` in the environment makes True the default of every device."""
    _lib.set_tuning("k1_exact_sqrt", 1 if flag else 0, device)


def get_exact_sqrt(device=None) -> bool:
    return bool(_lib.get_tuning("k1_exact_sqrt", device))


def set_exact_angles(flag: bool, device=None) -> None:
    """K3 / featuriser arithmetic on ``device`` (default: the current one).  False (default): the fast forms -- exact
    where the reference is exact, otherwise within the conditioning gates (3.8e-6 of off-diagonal dihedrals more than
    1e-5 from the reference at unit scale).  True: geometry.dihedral / geometry.angle in the reference's order of
    operations (three cross products, division by |b1|, library atan2 / acos): no entry beyond 1e-5 on well-conditioned
    inputs, on the same per-CU sweep kernels as the fast forms (DESIGN.md section 4 has both modes' times).  ``PROTSTRUC_AMD_EXACT_ANGLES=1`` makes True the default of every device."""
    _lib.set_exact_angles(flag, device)


def get_exact_angles(device=None) -> bool:
    return bool(_lib.get_exact_angles(device))


def autotune_pairwise_distance(xyz: torch.Tensor, atom_mask: Optional[torch.Tensor], out_dist: torch.Tensor,
                               out_mask: torch.Tensor):
    """Time K1's launch configurations on the given buffers now and keep the fastest for this device (results are
    identical for every configuration).  Optional: the default configuration is already the one that was fastest on
    every buffer measured."""
    pairwise_distance(xyz, atom_mask, out_dist=out_dist, out_mask=out_mask, _autotune=True)
    torch.cuda.current_stream(xyz.device).synchronize()
    return k1_autotune_result(xyz.device)


def allocate_fast_outputs(xyz: torch.Tensor, atom_mask: Optional[torch.Tensor] = None, candidates: int = 4):
    """Output buffers for ``pairwise_distance(..., out_dist=, out_mask=)``, chosen as the fastest of ``candidates``
    fresh allocations.

    Why this exists: on MI355X the rate at which K1's store stream is absorbed depends on which physical memory the
    output landed in -- 6.0 to 7.1 TB/s for the same kernel in one process ("fast and slow
    allocations") -- and a caller who keeps its output buffers for many steps inherits that luck for the whole run.
    This helper allocates the (dist, mask) pair ``candidates`` times (all held at once: candidates x 1125 B per
    residue pair; fewer if the device runs out of memory), times nine launches on each (three interleaved rounds), returns the fastest pair and releases the others.
    Returns ``(dist, mask, report)``; results written into the buffers are the same whichever pair is chosen."""
    xyz = _f32c(xyz, "xyz")
    B, N, A = xyz.shape[:3]
    shape = (B, N, N, A, A)
    pairs = []
    with _on(xyz.device):
        for _ in range(max(1, int(candidates))):
            try:
                d_ = torch.empty(shape, dtype=torch.float32, device=xyz.device)
                pairs.append((d_, torch.empty(shape, dtype=torch.bool, device=xyz.device)))
            except torch.cuda.OutOfMemoryError:      # a shape that does not fit `candidates` times: choose among what did fit
                d_ = None
                if not pairs:
                    raise
                break
        pairwise_distance(xyz, atom_mask, out_dist=pairs[0][0], out_mask=pairs[0][1])   # warm-up
        t_end = time.perf_counter() + 0.12
        while time.perf_counter() < t_end:
            pairwise_distance(xyz, atom_mask, out_dist=pairs[0][0], out_mask=pairs[0][1])
            torch.cuda.current_stream(xyz.device).synchronize()
        ms = [0.0] * len(pairs)   # mean over 3 interleaved rounds of 3 launches (the tuner's statistic, for the same reason)
        for _ in range(3):
            for k, (d, m) in enumerate(pairs):
                pairwise_distance(xyz, atom_mask, out_dist=d, out_mask=m)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _r in range(3):
                    pairwise_distance(xyz, atom_mask, out_dist=d, out_mask=m)
                e1.record()
                e1.synchronize()
                ms[k] += e0.elapsed_time(e1) / 9
        best = min(range(len(pairs)), key=lambda k: ms[k])
        d, m = pairs[best]
        del pairs
        torch.cuda.empty_cache()
    return d, m, {"ms_per_candidate": ms, "chosen": best}


def k1_autotune_result(device=None):
    """What the one-time K1 autotune chose on ``device`` (None if it has not run)."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    return _K1_TUNED.get(device)


def pairwise_distance(xyz: torch.Tensor, atom_mask: Optional[torch.Tensor] = None, *,
                      row_begin: int = 0, row_end: Optional[int] = None, compact: bool = False,
                      out_dist: Optional[torch.Tensor] = None, out_mask: Optional[torch.Tensor] = None,
                      want_dist: bool = True, want_mask: bool = True,
                      _autotune: bool = False) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]:
    """K1.  Returns (dist fp32, dist_mask bool) of shape (B, rows, N, A, A).

    rows = N for the default full matrix.  With ``row_begin/row_end`` only those
    residue rows are computed: into a compact (B, row_end-row_begin, N, A, A)
    buffer if ``compact`` else into rows [row_begin,row_end) of a full-size
    buffer (``out_dist`` / ``out_mask`` may supply that buffer, e.g. the
    destination of an all-gather)."""
    xyz = _f32c(xyz, "xyz")
    B, N, A = xyz.shape[:3]
    row_end = N if row_end is None else row_end
    if not (0 <= row_begin <= row_end <= N):
        raise ValueError(f"row range [{row_begin},{row_end}) outside [0,{N})")
    out_rows, origin = (row_end - row_begin, row_begin) if compact else (N, 0)
    shape = (B, out_rows, N, A, A)
    mask_u8 = _u8c(atom_mask, "atom_mask")
    _same_device(xyz, atom_mask=mask_u8)
    # caller-supplied outputs are dereferenced by a kernel launched on xyz.device: a CPU tensor or a tensor of another
    # GPU would be a wild device write, so device, shape, dtype and layout are all checked before anything is launched
    for name, t, dt in (("out_dist", out_dist if want_dist else None, torch.float32),
                        ("out_mask", out_mask if want_mask else None, torch.bool)):
        if t is not None and (not isinstance(t, torch.Tensor) or t.device != xyz.device or tuple(t.shape) != shape
                              or t.dtype != dt or not t.is_contiguous()):
            raise ValueError(f"{name} must be a contiguous {str(dt).replace('torch.', '')} tensor of shape {shape} "
                             f"on {xyz.device}")
    with _on(xyz.device):
        dist = dmask = None
        if want_dist:
            dist = out_dist if out_dist is not None else torch.empty(shape, dtype=torch.float32, device=xyz.device)
        if want_mask:
            dmask = out_mask if out_mask is not None else torch.empty(shape, dtype=torch.bool, device=xyz.device)
        args = (_ptr(xyz), _ptr(mask_u8), _ptr(dist), _ptr(dmask), B, N, A, row_begin, row_end, out_rows, origin)
        if _autotune or _AUTOTUNE_ENV:
            _autotune_k1(xyz.device, args, B * (row_end - row_begin) * N, N, A, force=_autotune)
        cfg_ref = _lib.k1_config_ref(xyz.device.index)   # this device's settings, snapshotted for this launch
        rc = 0
        if not (B == 0 or N == 0 or row_begin == row_end):   # empty input: nothing to launch (an empty tensor has no device pointer)
            K1_DISPATCHES[0] += 1
            rc = _K1_LAUNCH(*args, cfg_ref, _stream(xyz))
    if rc:
        _lib.check(rc, "ps_pairwise_distance_cfg_f32")
    return dist, dmask


def backbone_dihedrals(xyz: torch.Tensor, chain_idx: torch.Tensor, residue_mask: torch.Tensor, *,
                       want_dihedrals: bool = True, want_mask: bool = True, want_nterm: bool = True,
                       want_cterm: bool = True):
    """K2.  Returns (dihedrals (B,N,3) fp32, dihedral_mask (B,N,3) bool, nterm (B,N) bool, cterm (B,N) bool);
    an output that is not wanted is neither allocated nor written and comes back as None."""
    xyz = _f32c(xyz, "xyz")
    B, N, A = xyz.shape[:3]
    chain = _f32c(chain_idx, "chain_idx")
    rmask = _u8c(residue_mask, "residue_mask")
    dev = xyz.device
    if not (want_dihedrals or want_mask or want_nterm or want_cterm):
        raise ValueError("at least one output must be requested")
    with _on(dev):
        dih = torch.empty(B, N, 3, dtype=torch.float32, device=dev) if want_dihedrals else None
        dmask = torch.empty(B, N, 3, dtype=torch.bool, device=dev) if want_mask else None
        nterm = torch.empty(B, N, dtype=torch.bool, device=dev) if want_nterm else None
        cterm = torch.empty(B, N, dtype=torch.bool, device=dev) if want_cterm else None
        rc = 0
        if not (B == 0 or N == 0):   # empty input: nothing to launch (an empty tensor has no device pointer)
            rc = _lib.load().ps_backbone_dihedrals_f32(
                _ptr(xyz), _ptr(chain), _ptr(rmask), _ptr(dih), _ptr(dmask), _ptr(nterm), _ptr(cterm), B, N, A,
                _stream(xyz))
    _lib.check(rc, "ps_backbone_dihedrals_f32")
    return dih, dmask, nterm, cterm


def pairwise_angles(xyz: torch.Tensor, slots_i: Sequence[int], slots_j: Sequence[int], n_points: int, *,
                    row_begin: int = 0, row_end: Optional[int] = None, compact: bool = False,
                    out: Optional[torch.Tensor] = None, _one_column: bool = False) -> torch.Tensor:
    """K3.  n_points = 4: dihedral, 3: planar angle, over points (slots_i of residue i ++ slots_j of residue j).
    Only residue rows [row_begin, row_end) are computed, with K1's row addressing: into rows [row_begin, row_end) of
    a full-size (B, N, N) buffer (``out`` may supply it, e.g. the destination of an all-gather) or, with ``compact``,
    into a (B, row_end - row_begin, N) buffer.  ``_one_column`` (tests): the device's arithmetic (``set_exact_angles``)
    through the simple one-column kernel at any shape (bit 1 of ``exact_angles`` of the C ABI, diagnostic)."""
    xyz = _f32c(xyz, "xyz")
    B, N, A = xyz.shape[:3]
    slots = [int(s) for s in slots_i] + [int(s) for s in slots_j]
    src = [0] * len(slots_i) + [1] * len(slots_j)
    if len(slots) < n_points:
        raise IndexError(f"need {n_points} atoms in total, got {len(slots)}")  # the reference indexes past the end
    slots, src = slots[:n_points], src[:n_points]
    row_end = N if row_end is None else row_end
    if not (0 <= row_begin <= row_end <= N):
        raise ValueError(f"row range [{row_begin},{row_end}) outside [0,{N})")
    out_rows, origin = (row_end - row_begin, row_begin) if compact else (N, 0)
    arr = ctypes.c_int * n_points
    with _on(xyz.device):
        if out is None:
            out = torch.empty(B, out_rows, N, dtype=torch.float32, device=xyz.device)
        else:
            _check_out(out, (B, out_rows, N), "out", xyz.device)
        rc = 0
        if not (B == 0 or N == 0 or row_begin == row_end):   # empty input: nothing to launch (an empty tensor has no device pointer)
            rc = _lib.load().ps_pairwise_angles_f32(
                _ptr(xyz), _ptr(out), B, N, A, n_points, arr(*src), arr(*slots), row_begin, row_end, out_rows, origin,
                _lib.get_exact_angles(xyz.device) | (2 if _one_column else 0), _stream(xyz))
    _lib.check(rc, "ps_pairwise_angles_f32")
    return out


def inter_residue_geometry(xyz: torch.Tensor, atom_mask: Optional[torch.Tensor] = None, *, _one_column: bool = False):
    """Fused featuriser: dict of six (B,N,N) fp32 planes and three (B,N,N) bool planes.  The distance planes use the
    device's K1 square-root mode (``set_exact_sqrt``), so they equal the slices of ``pairwise_distance`` bit for bit."""
    xyz = _f32c(xyz, "xyz")
    B, N, A = xyz.shape[:3]
    if A < 5:
        raise IndexError("inter_residue_geometry needs the N, CA, C, O, CB atom slots")

    m = _u8c(atom_mask, "atom_mask")
    dev = xyz.device
    fkeys = ["d_ca", "d_cb", "d_no", "omega", "theta", "phi"]
    mkeys = ["d_ca_mask", "d_cb_mask", "d_no_mask"]
    with _on(dev):
        # nine planes in two allocations, every plane on a 16-byte boundary (plane stride padded): the kernel's flat mask
        # stores then share one 16-byte grid and decode each group once for the three mask planes (any N)
        plane = B * N * N
        fstride, kstride = (plane + 3) & ~3, (plane + 15) & ~15
        f = torch.empty(6, fstride, dtype=torch.float32, device=dev)
        k = torch.empty(3, kstride, dtype=torch.bool, device=dev)
        rc = 0
        if not (B == 0 or N == 0):   # empty input: nothing to launch (an empty tensor has no device pointer)
            fp, kp = f.data_ptr(), k.data_ptr()                       # plane addresses by arithmetic, not by 9 views
            rc = _lib.load().ps_inter_residue_geometry_f32(_ptr(xyz), _ptr(m), *[fp + 4 * fstride * i for i in range(6)],
                                                           *[kp + kstride * i for i in range(3)], B, N, A,
                                                           _lib.get_tuning("k1_exact_sqrt", dev),
                                                           _lib.get_exact_angles(dev) | (2 if _one_column else 0), _stream(xyz))
    _lib.check(rc, "ps_inter_residue_geometry_f32")
    out = {key: f[i, :plane].view(B, N, N) for i, key in enumerate(fkeys)}
    out.update({key: k[i, :plane].view(B, N, N) for i, key in enumerate(mkeys)})
    return out


def pointwise(mode: int, a: torch.Tensor, b: torch.Tensor, c: torch.Tensor, d: Optional[torch.Tensor] = None):
    """angle (mode 0) / dihedral (1) / gram_schmidt (2) over broadcast (*,3) point tensors."""
    pts = [a, b, c] + ([d] if d is not None else [])
    pts = torch.broadcast_tensors(*[_f32c(p, "points") for p in pts])
    shape = pts[0].shape[:-1]
    if pts[0].shape[-1] != 3:
        raise ValueError("points must have a trailing axis of size 3")
    flat = [p.reshape(-1, 3).contiguous() for p in pts]
    n = flat[0].shape[0]
    dev = flat[0].device
    with _on(dev):
        out = torch.empty((n, 9) if mode == 2 else (n,), dtype=torch.float32, device=dev)
        rc = 0
        if not (n == 0):   # empty input: nothing to launch (an empty tensor has no device pointer)
            rc = _lib.load().ps_pointwise_f32(mode, _ptr(flat[0]), _ptr(flat[1]), _ptr(flat[2]),
                                              _ptr(flat[3]) if mode == 1 else None, _ptr(out), n, _stream(flat[0]))
    _lib.check(rc, "ps_pointwise_f32")
    return out.reshape(*shape, 3, 3) if mode == 2 else out.reshape(shape)


def frames(xyz: torch.Tensor, a1: int, a2: int, a3: int, t_atom: int = 1, *, want_rot: bool = True,
           want_trans: bool = True):
    """K4.  Returns (rot (B,N,3,3) or None, trans (B,N,3) or None)."""
    xyz = _f32c(xyz, "xyz")
    B, N, A = xyz.shape[:3]
    dev = xyz.device
    with _on(dev):
        rot = torch.empty(B, N, 3, 3, dtype=torch.float32, device=dev) if want_rot else None
        trans = torch.empty(B, N, 3, dtype=torch.float32, device=dev) if want_trans else None
        rc = 0
        if not (B == 0 or N == 0):   # empty input: nothing to launch (an empty tensor has no device pointer)
            rc = _lib.load().ps_frames_f32(_ptr(xyz), _ptr(rot), _ptr(trans), B, N, A, int(a1), int(a2), int(a3),
                                           int(t_atom), _stream(xyz))
    _lib.check(rc, "ps_frames_f32")
    return rot, trans


def diffuse_(xyz: torch.Tensor, beta: torch.Tensor, rng_state: Optional[torch.Tensor] = None,
             noise: Optional[torch.Tensor] = None) -> torch.Tensor:
    """K5, in place on a contiguous fp32 ``xyz``.  ``rng_state``: int64 device tensor of RNG_STATE_WORDS
    words, [0] = seed, [1] = draw offset, the rest zero."""
    _require_device(xyz, "xyz")
    if xyz.dtype != torch.float32 or not xyz.is_contiguous():
        raise ValueError("diffuse_ needs a contiguous float32 xyz (it is updated in place)")
    B = xyz.shape[0]
    nps = xyz[0].numel() if B else 0
    beta = _f32c(beta, "beta")
    if beta.shape != (B,):
        raise ValueError(f"beta must have shape ({B},), got {tuple(beta.shape)}")
    if noise is not None:
        noise = _f32c(noise, "noise")
        if noise.shape != xyz.shape:
            raise ValueError("noise must have the shape of xyz")
    elif rng_state is None:
        raise ValueError("either rng_state or noise is required")
    _check_rng_state(rng_state, xyz.device)
    _same_device(xyz, beta=beta, noise=noise)
    with _on(xyz.device):
        rc = 0
        if not (xyz.numel() == 0):   # empty input: nothing to launch (an empty tensor has no device pointer)
            rc = _lib.load().ps_diffuse_f32(_ptr(xyz), _ptr(beta), B, nps, _ptr(rng_state), _ptr(noise), _stream(xyz))
    _lib.check(rc, "ps_diffuse_f32")
    return xyz


def diffuse_frames_(xyz: torch.Tensor, beta: torch.Tensor, a1: int, a2: int, a3: int, t_atom: int = 1,
                    rng_state: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None,
                    out_rot: Optional[torch.Tensor] = None, out_trans: Optional[torch.Tensor] = None):
    """Fused K5 + K4: diffuse ``xyz`` in place and return the frames of the new coordinates."""
    _require_device(xyz, "xyz")
    if xyz.dtype != torch.float32 or not xyz.is_contiguous():
        raise ValueError("diffuse_frames_ needs a contiguous float32 xyz (it is updated in place)")
    B, N, A = xyz.shape[:3]
    beta = _f32c(beta, "beta")
    if beta.shape != (B,):
        raise ValueError(f"beta must have shape ({B},), got {tuple(beta.shape)}")
    if noise is not None:
        noise = _f32c(noise, "noise")
        if noise.shape != xyz.shape:
            raise ValueError("noise must have the shape of xyz")
    elif rng_state is None:
        raise ValueError("either rng_state or noise is required")
    _check_rng_state(rng_state, xyz.device)
    _same_device(xyz, beta=beta, noise=noise)
    for slot in (a1, a2, a3, t_atom):
        if not 0 <= int(slot) < A:
            raise ValueError(f"atom slot {slot} outside [0, {A})")
    dev = xyz.device
    _check_out(out_rot, (B, N, 3, 3), "out_rot", dev)
    _check_out(out_trans, (B, N, 3), "out_trans", dev)
    with _on(dev):
        rot = out_rot if out_rot is not None else torch.empty(B, N, 3, 3, dtype=torch.float32, device=dev)
        trans = out_trans if out_trans is not None else torch.empty(B, N, 3, dtype=torch.float32, device=dev)
        rc = 0
        if not (xyz.numel() == 0):   # empty input: nothing to launch (an empty tensor has no device pointer)
            rc = _lib.load().ps_diffuse_frames_f32(_ptr(xyz), _ptr(beta), B, N, A, _ptr(rng_state), _ptr(noise), _ptr(rot),
                                                   _ptr(trans), int(a1), int(a2), int(a3), int(t_atom), _stream(xyz))
    _lib.check(rc, "ps_diffuse_frames_f32")
    return rot, trans


def diffusion_trajectory_(xyz: torch.Tensor, betas: torch.Tensor, a1: int, a2: int, a3: int, t_atom: int,
                          rng_state: torch.Tensor, want_rot: bool = True, want_trans: bool = True,
                          want_xyz: bool = False, *, out_rot: Optional[torch.Tensor] = None,
                          out_trans: Optional[torch.Tensor] = None, out_xyz: Optional[torch.Tensor] = None):
    """K55: T diffusion steps in one launch, coordinates resident in LDS.  ``betas``: (T, B).
    Returns (rot (T,B,N,3,3) | None, trans (T,B,N,3) | None, xyz_traj (T,B,N,A,3) | None); ``xyz`` ends as step T.
    ``out_rot`` / ``out_trans`` / ``out_xyz`` supply caller-owned output buffers (and imply the matching ``want_``)."""
    _require_device(xyz, "xyz")
    if xyz.dtype != torch.float32 or not xyz.is_contiguous():
        raise ValueError("diffusion_trajectory_ needs a contiguous float32 xyz (it is updated in place)")
    B, N, A = xyz.shape[:3]
    betas = _f32c(betas, "betas")
    if betas.ndim != 2 or betas.shape[1] != B:
        raise ValueError(f"betas must have shape (T, {B}), got {tuple(betas.shape)}")
    T = betas.shape[0]
    if rng_state is None:
        raise ValueError(f"rng_state must be an int64 tensor of {RNG_STATE_WORDS} words")
    _check_rng_state(rng_state, xyz.device)
    for slot in (a1, a2, a3, t_atom):
        if not 0 <= int(slot) < A:
            raise ValueError(f"atom slot {slot} outside [0, {A})")
    dev = xyz.device
    _same_device(xyz, betas=betas)
    _check_out(out_rot, (T, B, N, 3, 3), "out_rot", dev)
    _check_out(out_trans, (T, B, N, 3), "out_trans", dev)
    _check_out(out_xyz, (T, B, N, A, 3), "out_xyz", dev)
    with _on(dev):
        rot = out_rot if out_rot is not None else (
            torch.empty(T, B, N, 3, 3, dtype=torch.float32, device=dev) if want_rot else None)
        trans = out_trans if out_trans is not None else (
            torch.empty(T, B, N, 3, dtype=torch.float32, device=dev) if want_trans else None)
        traj = out_xyz if out_xyz is not None else (
            torch.empty(T, B, N, A, 3, dtype=torch.float32, device=dev) if want_xyz else None)
        rc = 0
        if not (xyz.numel() == 0 or T == 0):   # empty input: nothing to launch (an empty tensor has no device pointer)
            rc = _lib.load().ps_diffusion_trajectory_f32(_ptr(xyz), _ptr(betas), T, B, N, A, _ptr(rng_state), _ptr(rot),
                                                         _ptr(trans), _ptr(traj), int(a1), int(a2), int(a3), int(t_atom),
                                                         _stream(xyz))
    _lib.check(rc, "ps_diffusion_trajectory_f32")
    return rot, trans, traj


def standardize_(xyz: torch.Tensor, atom_mask: Optional[torch.Tensor]):
    """K6, in place.  Returns (mu (B,3), std (B,3))."""
    _require_device(xyz, "xyz")
    if xyz.dtype != torch.float32 or not xyz.is_contiguous():
        raise ValueError("standardize_ needs a contiguous float32 xyz (it is updated in place)")
    B, N, A = xyz.shape[:3]
    m = _u8c(atom_mask, "atom_mask")
    dev = xyz.device
    with _on(dev):
        # a structure without atoms has 0 / 0 statistics in the reference: NaN, not uninitialised memory
        alloc = torch.empty if N * A > 0 else (lambda *a, **k: torch.full(a, float("nan"), **k))
        mu = alloc(B, 3, dtype=torch.float32, device=dev)
        std = alloc(B, 3, dtype=torch.float32, device=dev)
        rc = 0
        if not (xyz.numel() == 0):   # empty input: nothing to launch (an empty tensor has no device pointer)
            rc = _lib.load().ps_standardize_f32(_ptr(xyz), _ptr(m), _ptr(mu), _ptr(std), B, N, A, _stream(xyz))
    _lib.check(rc, "ps_standardize_f32")
    return mu, std


def affine_(xyz: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor) -> torch.Tensor:
    """xyz[b] <- xyz[b] * scale[b] + shift[b] per axis, in place (unstandardize)."""
    _require_device(xyz, "xyz")
    if xyz.dtype != torch.float32 or not xyz.is_contiguous():
        raise ValueError("affine_ needs a contiguous float32 xyz (it is updated in place)")
    B = xyz.shape[0]
    n_atoms = xyz[0].numel() // 3 if B else 0
    scale = _f32c(scale, "scale")
    shift = _f32c(shift, "shift")
    with _on(xyz.device):
        rc = 0
        if not (xyz.numel() == 0):   # empty input: nothing to launch (an empty tensor has no device pointer)
            rc = _lib.load().ps_affine_f32(_ptr(xyz), _ptr(scale), _ptr(shift), B, n_atoms, _stream(xyz))
    _lib.check(rc, "ps_affine_f32")
    return xyz


def rigid(xyz: torch.Tensor, R: Optional[torch.Tensor] = None, t: Optional[torch.Tensor] = None, *,
          transpose: bool = False, inplace: bool = False) -> torch.Tensor:
    """x' = R x + t (or R^T x + t).  R: (3,3) | (B,3,3) | (B,N,3,3);  t: (3,) | (B,3) | (B,N,3) | (B,1,3) | (B,N,A,3)."""
    _require_device(xyz, "xyz")
    if xyz.dtype != torch.float32 or not xyz.is_contiguous():
        raise ValueError("rigid needs a contiguous float32 xyz")
    B, N, A = xyz.shape[:3]
    r_mode = t_mode = 0
    if R is not None:
        R = _f32c(R, "rotation")
        r_mode = {2: 1, 3: 2, 4: 3}.get(R.ndim, -1)
        ok = R.shape[-2:] == (3, 3) and (r_mode == 1 or (R.shape[0] == B and (r_mode == 2 or R.shape[1] == N)))
        if r_mode < 0 or not ok:
            raise ValueError(f"rotation must be (3,3), ({B},3,3) or ({B},{N},3,3), got {tuple(R.shape)}")
    if t is not None:
        t = _f32c(t, "translation")
        if t.shape == (3,):
            t_mode = 1
        elif t.shape == (B, 3) or t.shape == (B, 1, 3):
            t_mode = 2
        elif t.shape == (B, N, 3):
            t_mode = 3
        elif t.shape == (B, N, A, 3):
            t_mode = 4
        elif t.shape == (1, 3):
            t, t_mode = t.reshape(3), 1
        else:
            raise ValueError(f"translation shape {tuple(t.shape)} does not broadcast against xyz {tuple(xyz.shape)}")
        t = t.contiguous()
    with _on(xyz.device):
        out = xyz if inplace else torch.empty_like(xyz)
        rc = 0
        if not (xyz.numel() == 0):   # empty input: nothing to launch (an empty tensor has no device pointer)
            rc = _lib.load().ps_rigid_f32(_ptr(xyz), _ptr(out), _ptr(R), r_mode, int(transpose), _ptr(t), t_mode, B, N, A,
                                          _stream(xyz))
    _lib.check(rc, "ps_rigid_f32")
    return out


def center_of_mass(xyz: torch.Tensor, atom: int = 1) -> torch.Tensor:
    """(B,3) nanmean over residues of one atom slot."""
    xyz = _f32c(xyz, "xyz")
    B, N, A = xyz.shape[:3]
    with _on(xyz.device):
        # no residues: the reference's nanmean over nothing is NaN
        com = (torch.empty if xyz.numel() else (lambda *a, **k: torch.full(a, float("nan"), **k)))(
            B, 3, dtype=torch.float32, device=xyz.device)
        rc = 0
        if not (xyz.numel() == 0):   # empty input: nothing to launch (an empty tensor has no device pointer)
            rc = _lib.load().ps_center_of_mass_f32(_ptr(xyz), _ptr(com), B, N, A, int(atom), _stream(xyz))
    _lib.check(rc, "ps_center_of_mass_f32")
    return com


def frames_to_backbone(rot: torch.Tensor, trans: torch.Tensor, ideal: torch.Tensor, n_slots: int) -> torch.Tensor:
    """(B,N,n_slots,3): rot @ ideal[a] + trans for the first len(ideal) slots, zeros after."""
    rot = _f32c(rot, "orientations")
    trans = _f32c(trans, "translations")
    B, N = rot.shape[:2]
    if rot.shape != (B, N, 3, 3) or trans.shape != (B, N, 3):
        raise ValueError("orientations must be (B,N,3,3) and translations (B,N,3)")
    ideal = _f32c(ideal.to(rot.device), "ideal")
    with _on(rot.device):
        xyz = torch.empty(B, N, n_slots, 3, dtype=torch.float32, device=rot.device)
        rc = 0
        if not (xyz.numel() == 0):   # empty input: nothing to launch (an empty tensor has no device pointer)
            rc = _lib.load().ps_frames_to_backbone_f32(_ptr(rot), _ptr(trans), _ptr(ideal), ideal.shape[0], _ptr(xyz), B, N,
                                                       n_slots, _stream(rot))
    _lib.check(rc, "ps_frames_to_backbone_f32")
    return xyz


def kabsch(src: torch.Tensor, dst: torch.Tensor, atom_mask: torch.Tensor):
    """Per-structure optimal (R (B,3,3), t (B,3)) taking ``src`` (B,N,A,3) onto ``dst`` ((B|1),N,A,3) over masked atoms."""
    src = _f32c(src, "source xyz")
    dst = _f32c(dst.to(src.device), "target xyz")
    B = src.shape[0]
    n_atoms = src[0].numel() // 3
    if dst[0].numel() // 3 != n_atoms or dst.shape[0] not in (1, B):
        raise ValueError("source and target must have the same number of atoms per structure")
    m = _u8c(atom_mask.to(src.device), "atom_mask").reshape(-1, n_atoms)
    if m.shape[0] not in (1, B):
        raise ValueError("atom_mask must have the batch size of the source (or 1)")
    dev = src.device
    with _on(dev):
        R = torch.empty(B, 3, 3, dtype=torch.float32, device=dev)
        t = torch.empty(B, 3, dtype=torch.float32, device=dev)
        rc = _lib.load().ps_kabsch_f32(_ptr(src), _ptr(dst), _ptr(m), _ptr(R), _ptr(t), B, n_atoms,
                                       int(dst.shape[0] == 1 and B > 1), int(m.shape[0] == 1 and B > 1), _stream(src))
    _lib.check(rc, "ps_kabsch_f32")
    return R, t


def min_dist_to_points(xyz_one: torch.Tensor, query: torch.Tensor, atom: int = 1) -> torch.Tensor:
    """(N,) distance from atom slot ``atom`` of each residue of one structure (N,A,3) to its nearest query point."""
    xyz_one = _f32c(xyz_one, "xyz")
    query = _f32c(query.to(xyz_one.device), "query_xyz").reshape(-1, 3)
    N, A = xyz_one.shape[:2]
    with _on(xyz_one.device):
        out = torch.empty(N, dtype=torch.float32, device=xyz_one.device)
        rc = _lib.load().ps_min_dist_to_points_f32(_ptr(xyz_one), _ptr(query), _ptr(out), N, A, int(atom),
                                                   query.shape[0], _stream(xyz_one))
    _lib.check(rc, "ps_min_dist_to_points_f32")
    return out
