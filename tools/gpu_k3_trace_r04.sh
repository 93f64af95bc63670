# K3 at config 3 under the rocprofv3 kernel trace (run on the GPU box: bash tools/gpu_k3_trace_r04.sh [outdir-name] [reps])
set -o pipefail
O=gpurun_out/${1:-r04k3}
R=${2:-20}
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k3_trace -o k3 -- python3 tools/profile_workload.py k3 $R > $O/k3_trace.log 2>&1; echo "k3 trace rc=$?"
python3 tools/summarize_rocprof.py stats $O/k3_trace $O/k3_trace_stats.csv
# the dispatch timeline (start, duration, gap to the previous dispatch) of the K3 kernels
python3 - "$O" <<'PY'
import csv, glob, sys
o = sys.argv[1]
rows = []
for f in glob.glob(o + "/k3_trace/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"]) if rows else 0
prev_end = None
with open(o + "/k3_timeline.csv", "w") as fh:
    fh.write("kernel,start_us,duration_us,gap_us\n")
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40].replace(",", ";")
        fh.write(f"{name},{(s - t0) / 1e3:.1f},{(e - s) / 1e3:.1f},{((s - prev_end) / 1e3 if prev_end else 0):.1f}\n")
        prev_end = e
PY
rm -rf $O/k3_trace
cat $O/k3_trace_stats.csv
