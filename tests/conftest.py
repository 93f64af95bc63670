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


# ---- multi-process rehearsals -------------------------------------------------------------------------------------
# tools/gpu_session_rehearsals.py runs, as separate processes: the row-sharded path with the real HIP kernels on 2 ranks
# (gloo, both on the one GPU of the test box) and on 1 rank through RCCL, the plain-C consumer of the C ABI, and -- once
# those have ended, so that at most five processes use the card at any time -- ``python3 bench.py --gpus 2 --backend
# gloo ...`` exactly as the driver invokes the bench for N > 1.  A process that has initialised the GPU must not start
# other programs on this pool, so the orchestrator is started HERE, at session start, before this pytest process has
# touched the GPU (torch.cuda.device_count() does not initialise it); the tests wait for the verdict files.
REHEARSALS = {}
_ORCHESTRATOR = None


def pytest_sessionstart(session):
    global _ORCHESTRATOR
    import subprocess
    import tempfile

    markexpr = (session.config.option.markexpr or "").strip()
    if "not gpu" in markexpr or os.environ.get("PS_NO_REHEARSAL"):
        return
    # A run that cannot select a rehearsal test (a -k expression without negation that names none of them, or explicit
    # test files other than theirs) does not pay for the child processes.
    tokens = ("multirank", "rehears", "bench_gpus", "c_abi", "c99", "consumer", "rank", "gloo", "rccl", "rowshard", "world")
    kw = (session.config.option.keyword or "").strip().lower()
    if kw and "not " not in kw and not any(t in kw for t in tokens):
        return
    files = [a for a in session.config.args if a.endswith(".py") or ".py::" in a]
    if files and not any(("multirank" in f or "c_abi" in f) for f in files):
        return
    try:
        if torch.cuda.device_count() < 1:
            return
    except Exception:  # noqa: BLE001
        return
    outdir = tempfile.mkdtemp(prefix="ps_rehearse_")
    script = os.path.join(ROOT, "tools", "gpu_session_rehearsals.py")
    log = open(os.path.join(outdir, "orchestrator.log"), "w")
    _ORCHESTRATOR = subprocess.Popen([sys.executable, script, "--outdir", outdir], stdout=log, stderr=subprocess.STDOUT)
    for name in ("gloo_world2", "rccl_world1", "c_abi_demo", "bench_gpus2"):
        REHEARSALS[name] = {"exit": os.path.join(outdir, name + ".exit"), "out": os.path.join(outdir, name + ".json"),
                            "log": os.path.join(outdir, name + ".log"), "stdout": os.path.join(outdir, name + ".stdout")}


def wait_rehearsal(name, timeout):
    """(exit code, log text) of a rehearsal, waiting up to ``timeout`` seconds for it to end."""
    import time

    if name not in REHEARSALS:
        pytest.fail("the rehearsals were not started (conftest.pytest_sessionstart found no GPU?)")
    r = REHEARSALS[name]
    t_end = time.time() + timeout
    while not os.path.exists(r["exit"]):
        if time.time() > t_end or (_ORCHESTRATOR is not None and _ORCHESTRATOR.poll() is not None
                                   and not os.path.exists(r["exit"])):
            log = open(r["log"]).read()[-3000:] if os.path.exists(r["log"]) else "(no log)"
            pytest.fail(f"rehearsal {name} did not finish in {timeout} s; log:\n{log}")
        time.sleep(0.5)
    with open(r["exit"]) as f:
        code = int(f.read().strip())
    return code, open(r["log"]).read()


def pytest_sessionfinish(session, exitstatus):
    if _ORCHESTRATOR is not None and _ORCHESTRATOR.poll() is None:
        _ORCHESTRATOR.terminate()      # exactly the process we started (its children end with their own timeouts)


def load_golden(name):
    """Fixture written by tools/make_golden.py: inputs and the reference's outputs."""
    with np.load(os.path.join(GOLDEN_DIR, name + ".npz")) as z:
        return {k: torch.from_numpy(z[k]) for k in z.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden


def has_gpu():
    return torch.cuda.is_available()
