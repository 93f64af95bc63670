# K5 / K6 / K4 at config 5's shape under the rocprofv3 kernel trace (run on the GPU box: bash tools/gpu_k5_trace_r04.sh [outdir] [reps])
set -o pipefail
O=gpurun_out/${1:-r04k5}
R=${2:-20}
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k5_trace -o k5 -- python3 tools/profile_workload.py k5 $R > $O/k5_trace.log 2>&1; echo "k5 trace rc=$?"
python3 tools/summarize_rocprof.py stats $O/k5_trace $O/k5_trace_stats.csv
python3 - "$O" <<'PY'
import csv, glob, sys
o = sys.argv[1]
rows = []
for f in glob.glob(o + "/k5_trace/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = {}
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:30]
    seq.setdefault(name, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in seq.items():
    if k[:1] == "k":
        print(k, " ".join(f"{x:.1f}" for x in v))
PY
rm -rf $O/k5_trace
cat $O/k5_trace_stats.csv | cut -c1-110
# where do K5 / K6's bytes come from and go to?  L2 hits / misses and the L2 <-> fabric requests (one TCC pass)
timeout -k 10 200 rocprofv3 --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum -d $O/k5_tcc -o p -- python3 tools/profile_workload.py k5 10 > $O/k5_tcc.log 2>&1; echo "k5 tcc rc=$?"
python3 tools/summarize_rocprof.py pmc $O/k5_tcc $O/k5_pmc_tcc.json && rm -rf $O/k5_tcc
timeout -k 10 200 rocprofv3 --output-format csv --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_sum -d $O/k5_tcc2 -o p -- python3 tools/profile_workload.py k5 10 > $O/k5_tcc2.log 2>&1; echo "k5 tcc2 rc=$?"
python3 tools/summarize_rocprof.py pmc $O/k5_tcc2 $O/k5_pmc_tcc2.json && rm -rf $O/k5_tcc2
