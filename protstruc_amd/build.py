"""Build libprotstruc_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m protstruc_amd.build [--force]

The library is built IN-TREE (protstruc_amd/lib/) so that it travels with the
repository snapshot to the GPU box; nothing is JIT-compiled at import time.
-ffp-contract=off is part of the numerical contract (see csrc/ps_common.hpp).
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libprotstruc_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not is_stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    tmp = f"{LIB_PATH}.{os.getpid()}.tmp"  # build aside and rename: concurrent builders (one per rank) cannot corrupt the .so
    cmd = [HIPCC] + FLAGS + ["-o", tmp] + sources()
    if verbose:
        print("[protstruc_amd.build]", " ".join(cmd), flush=True)
    try:
        subprocess.run(cmd, check=True)
        os.replace(tmp, LIB_PATH)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB_PATH)
