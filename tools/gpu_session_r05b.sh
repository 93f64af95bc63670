# Round 5, second GPU session (bash tools/gpu_session_r05b.sh [outdir-name]): the K3 / featuriser GPU tests incl. the dispatch-arm
# tables against the oracle, K3's error statistics in both modes, and bench.py with other_configs.
set -o pipefail
O=gpurun_out/${1:-r05b}
mkdir -p $O
timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "k3 or inter_residue or featuris" > $O/pytest_k3.log 2>&1; rc=$?; tail -5 $O/pytest_k3.log
timeout -k 10 200 python3 tools/k3_error_stats.py > $O/k3_error_stats.log 2>&1; cat $O/k3_error_stats.log
timeout -k 10 500 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"; tail -3 $O/bench_n1.err
python3 - $O <<'PY'
import json, sys
d = json.loads(open(sys.argv[1] + "/bench_n1.json").read().strip().splitlines()[-1])
print("frac", d["roofline"]["frac"], "value", d["value"])
print(json.dumps(d.get("other_configs"), indent=1)[:6000])
PY
exit $rc
