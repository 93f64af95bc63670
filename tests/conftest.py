import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# ---- multi-rank rehearsals ---------------------------------------------------------------------------------------
# tools/rehearse_rowshard.py runs the row-sharded path with the real HIP kernels on 2 ranks (gloo, both on the one GPU
# of the test box) and on 1 rank through RCCL.  A process that has initialised the GPU must not start other programs
# on this pool, so the launchers are started HERE, at session start, before this pytest process has touched the GPU
# (torch.cuda.device_count() does not initialise it); tests/test_gpu_multirank.py waits for their verdicts.
REHEARSALS = {}


def pytest_sessionstart(session):
    import subprocess
    import tempfile

    markexpr = (session.config.option.markexpr or "").strip()
    if "not gpu" in markexpr or os.environ.get("PS_NO_REHEARSAL"):
        return
    try:
        if torch.cuda.device_count() < 1:
            return
    except Exception:  # noqa: BLE001
        return
    outdir = tempfile.mkdtemp(prefix="ps_rehearse_")
    script = os.path.join(ROOT, "tools", "rehearse_rowshard.py")
    for name, extra in (("gloo_world2", ["--world", "2", "--backend", "gloo"]),
                        ("rccl_world1", ["--world", "1", "--backend", "nccl", "--quick"])):
        out = os.path.join(outdir, name + ".json")
        log = open(os.path.join(outdir, name + ".log"), "w")
        proc = subprocess.Popen([sys.executable, script, "--out", out] + extra, stdout=log, stderr=subprocess.STDOUT)
        REHEARSALS[name] = {"proc": proc, "out": out, "log": log.name}
    # the plain-C consumer of the C ABI (examples/c_abi_demo.c): compiled with gcc and run as its own process
    log = open(os.path.join(outdir, "c_abi_demo.log"), "w")
    code = ("import sys, subprocess; sys.path.insert(0, %r); from protstruc_amd import build; "
            "build.build(verbose=False); exe = build.build_c_example(verbose=False); "
            "sys.exit(subprocess.run([exe]).returncode)" % ROOT)
    proc = subprocess.Popen([sys.executable, "-c", code], stdout=log, stderr=subprocess.STDOUT)
    REHEARSALS["c_abi_demo"] = {"proc": proc, "out": None, "log": log.name}


def pytest_sessionfinish(session, exitstatus):
    for r in REHEARSALS.values():
        if r["proc"].poll() is None:
            r["proc"].kill()      # exactly the launcher we started


def load_golden(name):
    """Fixture written by tools/make_golden.py: inputs and the reference's outputs."""
    with np.load(os.path.join(GOLDEN_DIR, name + ".npz")) as z:
        return {k: torch.from_numpy(z[k]) for k in z.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden


def has_gpu():
    return torch.cuda.is_available()
