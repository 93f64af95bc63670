#!/usr/bin/python3
"""Same-process A/B of two BUILDS of the library (e.g. the product .so against one compiled with an extra -D):
all are loaded with ctypes, and ps_pairwise_distance_f32 of each is timed in interleaved rounds on the same
buffers.  Usage: python3 tools/k1_ab_libs.py libA.so libB.so [libC.so ...] [A:N ...]   (default shapes: 37:128 25:128
14:256); every argument ending in .so is a library, the others are shapes; the first library is the bit reference."""
import ctypes
import os
import sys

import torch

paths = [a for a in sys.argv[1:] if a.endswith(".so")]
libs = [ctypes.CDLL(os.path.abspath(p)) for p in paths]
names = [os.path.basename(p).replace("libprotstruc_hip", "product").replace("lib_", "").replace(".so", "") for p in paths]
shapes = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:] if not a.endswith(".so")] or [(37, 128), (25, 128), (14, 256)]
vp, i32 = ctypes.c_void_p, ctypes.c_int
for lib in libs:
    lib.ps_pairwise_distance_f32.restype = i32
    lib.ps_pairwise_distance_f32.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
g = torch.Generator().manual_seed(0)
for A, N in shapes:
    B = max(1, int(8e9 / (N * N * A * A * 5)))
    xyz = torch.randn(B, N, A, 3, generator=g).cuda()
    mask = (torch.rand(B, N, A, generator=g) < 0.9).cuda()
    d = torch.empty(B, N, N, A, A, device="cuda")
    m = torch.empty(B, N, N, A, A, dtype=torch.bool, device="cuda")
    st = torch.cuda.current_stream().cuda_stream

    def run(lib):
        rc = lib.ps_pairwise_distance_f32(xyz.data_ptr(), mask.data_ptr(), d.data_ptr(), m.data_ptr(), B, N, A, 0, N, N, 0, st)
        assert rc == 0, rc
    outs = []
    for lib in libs:
        run(lib); torch.cuda.synchronize(); outs.append((d.clone(), m.clone()))
    same = all(torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) for o in outs[1:])
    best = [float("inf")] * len(libs)
    for rnd in range(4):
        for k, lib in enumerate(libs):
            run(lib); run(lib)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): run(lib)
            e1.record(); torch.cuda.synchronize()
            best[k] = min(best[k], e0.elapsed_time(e1) / 5)
    nb = B * N * N * A * A * 5
    print(f"A={A:3d} N={N:4d} B={B:4d}  " + "  ".join(f"{n} {nb / b / 1e9:5.2f}" for n, b in zip(names, best)) +
          f"  TB/s   same bits: {same}", flush=True)
    del xyz, mask, d, m, outs
