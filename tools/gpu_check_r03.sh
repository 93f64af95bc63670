# Full GPU test-suite, then two default bench runs; prints one summary line per bench (run on the GPU box:
#   bash tools/gpu_check_r03.sh [outdir-name]).
set -o pipefail
O=gpurun_out/${1:-check}
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
for k in 1 2; do
  timeout -k 10 240 python bench.py --no-cpu-baseline > $O/bench_$k.json 2> $O/bench_$k.err || exit 1
done
python3 - $O <<'PY'
import json, sys
for k in (1, 2):
    d = json.loads(open(f"{sys.argv[1]}/bench_{k}.json").read().strip().splitlines()[-1]); r = d["roofline"]
    t = d["config"]["k1_autotune"]
    print(round(r["frac"], 4), r["kernel"], "min/max", [round(x, 3) for x in r["kernel_ms_min_max_this_rank"]],
          "fill", round(r["buffer_fill_GBps"]), {k2: round(v, 3) for k2, v in t["ms"].items()},
          "lottery", [round(x, 3) for x in r["allocation_lottery"]["frac_of_hbm_peak_per_candidate"]])
PY
