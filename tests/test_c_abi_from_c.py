"""The drop-in boundary from plain C: examples/c_abi_demo.c is compiled with `gcc -std=c99 -Wall -Werror` against
include/protstruc_hip.h (CPU test: the header is a C header and every symbol links) and run on the GPU box as its own
process, where it checks distances (1 ulp with the hardware square root, bit-exact with exact_sqrt), the pair mask,
frames and the argument-error convention without Python or PyTorch on the caller's side."""
import os

import pytest

from tests import conftest


def test_c99_consumer_compiles_and_links():
    from protstruc_amd import build
    build.build(force=False, verbose=False)
    exe = build.build_c_example(verbose=False)
    assert os.path.exists(exe) and os.access(exe, os.X_OK)


@pytest.mark.gpu
def test_c99_consumer_runs_on_the_gpu():
    code, log = conftest.wait_rehearsal("c_abi_demo", 330)
    assert code == 0 and "c_abi_demo ok" in log, log[-2000:]
    assert "K1 plan for B=2 N=21 A=15: k1_pairdist_a15_flat<64> (flat)" in log   # ps_k1_plan_f32 called from plain C
    assert "exact_sqrt=1" in log and "(0 one ulp off)" in log
    assert "hipGraph: 10 captured steps replayed twice, draw counter 20" in log
