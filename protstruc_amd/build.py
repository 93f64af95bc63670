"""Build libprotstruc_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m protstruc_amd.build [--force]

The library is built IN-TREE (protstruc_amd/lib/) so that it travels with the
repository snapshot to the GPU box; nothing is JIT-compiled at import time.
-ffp-contract=off is part of the numerical contract (see csrc/ps_common.hpp).
"""
import glob
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libprotstruc_hip.so")
# tools/ only: the same sources with -DPS_EXPERIMENTS (store-only and other timing modes that can write wrong
# values).  Never loaded by the package unless PROTSTRUC_AMD_LIB points at it explicitly.
EXPERIMENTS_LIB_PATH = os.path.join(LIB_DIR, "libprotstruc_hip_experiments.so")
# the multi-GPU exchange step lives in its own library so that libprotstruc_hip.so has no RCCL dependency
RCCL_SRC_DIR = os.path.join(HERE, "csrc_rccl")
RCCL_LIB_PATH = os.path.join(LIB_DIR, "libprotstruc_rccl.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ROCM_LIB = os.environ.get("ROCM_LIB", "/opt/rocm/lib")
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _deps():
    return sources() + sorted(glob.glob(os.path.join(CSRC, "*.hpp"))) + \
        [os.path.join(HERE, "..", "include", "protstruc_hip.h")]


def source_hash(experiments=False):
    """Digest of everything the library is compiled from (file names, contents, flags)."""
    h = hashlib.sha256((" ".join(FLAGS) + (" -DPS_EXPERIMENTS" if experiments else "")).encode())
    for d in _deps():
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def is_stale(path=None):
    """True when ``path`` is missing or was built from other sources than the ones in the tree now.  Compared by
    content digest (recorded next to the library at build time), not by mtime: a snapshot copied to another
    machine keeps its contents but not necessarily its timestamps."""
    path = path or LIB_PATH
    if not os.path.exists(path):
        return True
    try:
        with open(path + ".srchash") as f:
            recorded = f.read().strip()
    except OSError:
        return True
    return recorded != source_hash(experiments=(os.path.abspath(path) == os.path.abspath(EXPERIMENTS_LIB_PATH)))


def build(force=False, verbose=True, experiments=False):
    out = EXPERIMENTS_LIB_PATH if experiments else LIB_PATH
    if not force and not is_stale(out):
        return out
    os.makedirs(LIB_DIR, exist_ok=True)
    tmp = f"{out}.{os.getpid()}.tmp"  # build aside and rename: concurrent builders (one per rank) cannot corrupt the .so
    cmd = [HIPCC] + FLAGS + (["-DPS_EXPERIMENTS"] if experiments else []) + ["-o", tmp] + sources()
    if verbose:
        print("[protstruc_amd.build]", " ".join(cmd), flush=True)
    try:
        digest = source_hash(experiments)
        subprocess.run(cmd, check=True)
        os.replace(tmp, out)
        with open(out + ".srchash", "w") as f:
            f.write(digest + "\n")
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return out


def rccl_sources():
    return sorted(glob.glob(os.path.join(RCCL_SRC_DIR, "*.cpp")))


def _rccl_hash():
    h = hashlib.sha256(b"rccl-lib-v1")
    for d in rccl_sources() + [os.path.join(HERE, "..", "include", "protstruc_rccl.h")]:
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def rccl_is_stale():
    if not os.path.exists(RCCL_LIB_PATH):
        return True
    try:
        with open(RCCL_LIB_PATH + ".srchash") as f:
            return f.read().strip() != _rccl_hash()
    except OSError:
        return True


def build_rccl(force=False, verbose=True):
    """libprotstruc_rccl.so: host-only C++ over RCCL (ps_allgather_rows, include/protstruc_rccl.h).  Linked against
    librccl.so.1 by SONAME: inside a PyTorch process that is the RCCL PyTorch already loaded, so one copy serves both."""
    if not force and not rccl_is_stale():
        return RCCL_LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    tmp = f"{RCCL_LIB_PATH}.{os.getpid()}.tmp"
    cmd = [HIPCC, "-O2", "-std=c++17", "-fPIC", "-shared", "-o", tmp] + rccl_sources() + \
        ["-L" + ROCM_LIB, "-lrccl", "-Wl,-rpath," + ROCM_LIB]
    if verbose:
        print("[protstruc_amd.build]", " ".join(cmd), flush=True)
    try:
        digest = _rccl_hash()
        subprocess.run(cmd, check=True)
        os.replace(tmp, RCCL_LIB_PATH)
        with open(RCCL_LIB_PATH + ".srchash", "w") as f:
            f.write(digest + "\n")
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return RCCL_LIB_PATH


EXAMPLE_SRC = os.path.join(HERE, "..", "examples", "c_abi_demo.c")
EXAMPLE_BIN = os.path.join(HERE, "..", "build", "c_abi_demo")


def build_c_example(verbose=True):
    """examples/c_abi_demo.c with plain gcc -std=c99: proves that include/protstruc_hip.h is a C header and that the
    boundary needs neither C++ nor PyTorch on the caller's side.  The binary is a GPU test (tests/test_c_abi_from_c.py)."""
    os.makedirs(os.path.dirname(EXAMPLE_BIN), exist_ok=True)
    rocm_inc = os.path.join(os.path.dirname(ROCM_LIB.rstrip("/")), "include")
    cmd = ["gcc", "-std=c99", "-O1", "-ffp-contract=off", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__",
           "-I" + rocm_inc, "-I" + os.path.join(HERE, "..", "include"), EXAMPLE_SRC,
           "-L" + LIB_DIR, "-lprotstruc_hip", "-L" + ROCM_LIB, "-lamdhip64", "-lm",
           "-Wl,-rpath,$ORIGIN/../protstruc_amd/lib", "-Wl,-rpath," + ROCM_LIB, "-o", EXAMPLE_BIN]
    if verbose:
        print("[protstruc_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return EXAMPLE_BIN


if __name__ == "__main__":
    force = "--force" in sys.argv
    print(build(force=force, experiments="--experiments" in sys.argv))
    print(build_rccl(force=force))
    print(build_c_example())
