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
# touched the GPU (torch.cuda.device_count() does not initialise it) -- in pytest_collection_finish, once the selected
# items are known; the tests wait for the verdict files.
REHEARSALS = {}
_ORCHESTRATOR = None


_WHY_NOT_STARTED = "pytest_collection_finish has not run"


def pytest_collection_finish(session):
    """Start the rehearsal children if (and only if) a test that waits for them was SELECTED -- decided from the collected
    items, after -m / -k / file arguments have been applied, not guessed from the command line.  Collection imports the
    test modules but does not touch the GPU (no test module calls torch.cuda at import time), so this is still early
    enough for the rule above."""
    global _ORCHESTRATOR, _WHY_NOT_STARTED
    import subprocess
    import tempfile

    if _ORCHESTRATOR is not None:
        return
    if os.environ.get("PS_NO_REHEARSAL"):
        _WHY_NOT_STARTED = "PS_NO_REHEARSAL is set"
        return
    wanted = [it for it in session.items
              if os.path.basename(str(it.fspath)) in ("test_gpu_multirank.py", "test_c_abi_from_c.py")]
    if not wanted:
        _WHY_NOT_STARTED = "no selected test waits for a rehearsal"
        return
    try:
        if torch.cuda.device_count() < 1:
            _WHY_NOT_STARTED = "torch.cuda.device_count() == 0 (no GPU visible)"
            return
    except Exception as exc:  # noqa: BLE001
        _WHY_NOT_STARTED = f"torch.cuda.device_count() raised {type(exc).__name__}: {exc}"
        return
    outdir = tempfile.mkdtemp(prefix="ps_rehearse_")
    script = os.path.join(ROOT, "tools", "gpu_session_rehearsals.py")
    log = open(os.path.join(outdir, "orchestrator.log"), "w")
    # its own session / process group: pytest_sessionfinish can end the orchestrator AND everything it started (the
    # rehearsal launchers and their ranks) by signalling that group -- exactly the processes this session created
    _ORCHESTRATOR = subprocess.Popen([sys.executable, script, "--outdir", outdir], stdout=log, stderr=subprocess.STDOUT,
                                     start_new_session=True)
    for name in ("gloo_world2", "rccl_world1", "c_abi_demo", "bench_gpus2"):
        REHEARSALS[name] = {"exit": os.path.join(outdir, name + ".exit"), "out": os.path.join(outdir, name + ".json"),
                            "log": os.path.join(outdir, name + ".log"), "stdout": os.path.join(outdir, name + ".stdout")}


def wait_rehearsal(name, timeout):
    """(exit code, log text) of a rehearsal, waiting up to ``timeout`` seconds for it to end."""
    import time

    if name not in REHEARSALS:
        pytest.fail(f"the rehearsals were not started: {_WHY_NOT_STARTED}")
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
    """End whatever the orchestrator still runs (a failed or interrupted session leaves its children behind otherwise,
    on the GPU, until their own 300-570 s timeouts): signal the process GROUP the orchestrator leads -- it was started
    with start_new_session=True, so the group is exactly the orchestrator, the rehearsal launchers and their ranks."""
    import signal
    import time

    if _ORCHESTRATOR is None:
        return
    try:
        pgid = os.getpgid(_ORCHESTRATOR.pid)
    except ProcessLookupError:
        return
    if pgid == os.getpgid(0):          # never our own group (cannot happen with start_new_session=True)
        return
    for sig, wait in ((signal.SIGTERM, 3.0), (signal.SIGKILL, 0.0)):
        try:
            os.killpg(pgid, sig)
        except ProcessLookupError:
            break
        t_end = time.time() + wait
        while time.time() < t_end and _ORCHESTRATOR.poll() is None:
            time.sleep(0.1)
    try:
        _ORCHESTRATOR.wait(timeout=5)
    except Exception:  # noqa: BLE001
        pass


def load_golden(name):
    """Fixture written by tools/make_golden.py: inputs and the reference's outputs."""
    with np.load(os.path.join(GOLDEN_DIR, name + ".npz")) as z:
        return {k: torch.from_numpy(z[k]) for k in z.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden


def has_gpu():
    return torch.cuda.is_available()
