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
    name = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("void ", "").split("(")[0][:30]
    seq.setdefault(name, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in seq.items():
    if k.startswith("k"):
        print(k, " ".join(f"{x:.1f}" for x in v))
PY
rm -rf $O/k5_trace
cat $O/k5_trace_stats.csv | cut -c1-110
