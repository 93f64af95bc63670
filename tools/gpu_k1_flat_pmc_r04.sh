# HBM traffic and kernel trace of K1's flat short-chain kernels (tools/profile_workload.py k1f), round 4
set -o pipefail
O=gpurun_out/${1:-r04k1f}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o k1f -- python3 tools/profile_workload.py k1f 6 > $O/t.log 2>&1; echo "trace rc=$?"
python3 tools/summarize_rocprof.py stats $O/t $O/k1f_trace_stats.csv
timeout -k 10 240 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $O/w -o k1f -- python3 tools/profile_workload.py k1f 4 > $O/w.log 2>&1; echo "w rc=$?"
timeout -k 10 240 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/f -o k1f -- python3 tools/profile_workload.py k1f 4 > $O/f.log 2>&1; echo "f rc=$?"
python3 tools/summarize_rocprof.py pmc $O/w $O/k1f_write_size.json
python3 tools/summarize_rocprof.py pmc $O/f $O/k1f_fetch_size.json
rm -rf $O/t $O/w $O/f
grep k1f $O/t.log; grep "k1_pairdist" $O/k1f_trace_stats.csv
python3 - "$O" <<'P'
import json, sys
for f in ("k1f_write_size.json", "k1f_fetch_size.json"):
    for k, e in json.load(open(f"{sys.argv[1]}/{f}")).items():
        if "k1_pairdist" in k: print(f, k, e["per_dispatch_mean"], round(e["mean_ns_under_pmc"]))
P
