#!/usr/bin/python3
"""Same-process A/B of two BUILDS of the library (e.g. the product .so against one compiled with an extra -D):
all are loaded with ctypes, and ps_pairwise_distance_f32 of each is timed in interleaved rounds on the same
buffers.  Usage: python3 tools/k1_ab_libs.py libA.so libB.so [libC.so ...] [A:N ...]   (default shapes: 37:128 25:128
14:256); every argument ending in .so is a library, the others are shapes; the first library is the bit reference.
K1_CFGS="jt=32,lds_pad_kb=24;jt=64" in the environment times those ps_k1_config settings (through
ps_pairwise_distance_cfg_f32) next to the default, on K1_NBUF (default 1) output-buffer pairs per shape (K1_HOLD=1 keeps every pair allocated, so that the
pairs are distinct allocations rather than one block handed back by the caching allocator)."""
import ctypes
import os
import sys

import torch

paths = [a for a in sys.argv[1:] if a.endswith(".so")]
libs = [ctypes.CDLL(os.path.abspath(p)) for p in paths]
names = [os.path.basename(p).replace("libprotstruc_hip", "product").replace("lib_", "").replace(".so", "") for p in paths]
shapes = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:] if not a.endswith(".so")] or [(37, 128), (25, 128), (14, 256)]
vp, i32 = ctypes.c_void_p, ctypes.c_int
FIELDS = ("struct_size", "exact_sqrt", "variant", "flat", "rows_per_block", "lds_pad_kb", "flat_cpw", "flat_lds_pad_kb", "jt",
          "xcd_remap", "store_nt", "flat_fl_log2", "rowphase", "experiment")


class Cfg(ctypes.Structure):
    _fields_ = [(n, i32) for n in FIELDS]


for lib in libs:
    lib.ps_pairwise_distance_cfg_f32.restype = i32
    lib.ps_pairwise_distance_cfg_f32.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, ctypes.POINTER(Cfg), vp]
held = []
specs = [""] + [c for c in os.environ.get("K1_CFGS", "").split(";") if c]
nbuf = int(os.environ.get("K1_NBUF", "1"))


def make_cfg(lib, spec):
    c = Cfg()
    lib.ps_k1_config_default(ctypes.byref(c))
    for kv in filter(None, spec.split(",")):
        k, v = kv.split("=")
        setattr(c, k, int(v))
    return c
g = torch.Generator().manual_seed(0)
for A, N, *rest in shapes:       # "A:N" or "A:N:B"
    B = rest[0] if rest else max(1, int(8e9 / (N * N * A * A * 5)))
    xyz = torch.randn(B, N, A, 3, generator=g).cuda()
    mask = (torch.rand(B, N, A, generator=g) < 0.9).cuda()
    st = torch.cuda.current_stream().cuda_stream
    nb = B * N * N * A * A * 5
    for kb in range(nbuf):
        d = torch.empty(B, N, N, A, A, device="cuda")
        m = torch.empty(B, N, N, A, A, dtype=torch.bool, device="cuda")
        if os.environ.get("K1_PRINT_PTRS"):
            print(f"buf{kb}: dist at {d.data_ptr():#x} (mod 1 GiB {d.data_ptr() % (1 << 30):#x})  mask at {m.data_ptr():#x}", flush=True)
        for spec in specs:
            cfgs = [make_cfg(lib, spec) for lib in libs]

            def run(k):
                rc = libs[k].ps_pairwise_distance_cfg_f32(xyz.data_ptr(), mask.data_ptr(), d.data_ptr(), m.data_ptr(), B, N, A, 0, N, N, 0,
                                                          ctypes.byref(cfgs[k]), st)
                assert rc == 0, rc
            outs, ok = [], []
            for k in range(len(libs)):
                d.fill_(float("nan")); m.fill_(False)
                try:
                    run(k)
                except AssertionError:          # this build does not know the configuration
                    continue
                torch.cuda.synchronize(); outs.append((d.clone(), m.clone())); ok.append(k)
            same = all(torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) for o in outs[1:])
            del outs
            tot = [float("nan")] * len(libs)
            for k in ok:
                tot[k] = 0.0
            R = 5
            for rnd in range(R):
                for k in ok:
                    run(k); run(k)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(5): run(k)
                    e1.record(); torch.cuda.synchronize()
                    tot[k] += e0.elapsed_time(e1) / 5
            print(f"A={A:3d} N={N:4d} B={B:4d} buf{kb} [{spec or 'default':24s}] " + "  ".join(f"{n} {nb / (t / R) / 1e9:5.2f}" for n, t in zip(names, tot)) +
                  f"  TB/s (mean of {R} rounds)   same bits: {same}", flush=True)
        if os.environ.get("K1_HOLD"):      # keep the pair alive: the next one is then a NEW allocation, not this block handed back
            held.append((d, m))
        del d, m
    del xyz, mask
