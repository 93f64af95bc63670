"""bench.py's post-run verification (`check_outputs`) must catch the failures it exists for: a wrong distance, a wrong
mask byte, an unwritten (NaN) region, a store-only style run.  Device-independent torch code, so it runs on the CPU
against the oracle's outputs."""
import importlib.util
import os

import torch

from oracle import protstruc_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_check_outputs_accepts_correct_and_rejects_corrupted():
    bench = _bench()
    xyz, mask = bench.synth(3, b=3, n=12)
    d, m = O.pairwise_distance_matrix(xyz, mask)
    assert bench.check_outputs(xyz, mask, d.clone(), m.clone(), n_blocks=256) == []
    # one wrong distance inside a sampled block (all 3*12*12 blocks are sampled with 256 draws, almost surely)
    bad = d.clone()
    bad[:, :, :, 7, 3] += 1e-3
    assert any("sampled blocks" in f for f in bench.check_outputs(xyz, mask, bad, m.clone(), n_blocks=256))
    # one flipped mask byte: caught by the exact per-structure checksum even if the block is not sampled
    badm = m.clone()
    badm[2, 5, 6, 1, 1] = ~badm[2, 5, 6, 1, 1]
    assert any("checksum" in f for f in bench.check_outputs(xyz, mask, d.clone(), badm, n_blocks=1))
    # an unwritten region
    hole = d.clone()
    hole[:, 4:6] = float("nan")
    assert bench.check_outputs(xyz, mask, hole, m.clone(), n_blocks=256) != []
    # asymmetric structure (rows written, columns stale)
    asym = d.clone()
    asym[:, 3, 9] *= 1.0001
    fails = bench.check_outputs(xyz, mask, asym, m.clone(), n_blocks=256)
    assert fails != []


def test_host_cpu_info_fields():
    info = _bench().host_cpu_info()
    assert set(info) == {"host_cpu", "host_logical_cpus", "host_physical_cores", "host_sockets"}
    assert info["host_logical_cpus"] >= 1
