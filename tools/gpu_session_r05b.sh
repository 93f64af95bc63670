# Round 5 GPU session (bash tools/gpu_session_r05b.sh [outdir-name] [pytest -k expression]): GPU tests (default: the K3 /
# featuriser ones incl. the dispatch-arm tables against the oracle); output kept short, logs under gpurun_out/.
set -o pipefail
O=gpurun_out/${1:-r05b}
K=${2:-k3 or inter_residue or featuris}
mkdir -p $O
timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "$K" > $O/pytest_k3.log 2>&1; rc=$?; tail -15 $O/pytest_k3.log | cut -c1-300
exit $rc
