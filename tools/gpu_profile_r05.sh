# Round-5 measurement set (run on the GPU box: bash tools/gpu_profile_r05.sh [outdir-name]):
#   * bench.py plain (the line as the driver sees it, with other_configs);
#   * bench.py --no-cpu-baseline under rocprofv3 --kernel-trace --stats TWICE in one lease (two processes): each trace is cut
#     at the K1 dispatch ordinals the line reports -- the timed region AND the allocation lottery's kept pair -- so profiles/
#     holds a kernel-trace mean for a fast-class pair next to whatever the default allocation drew;
#   * the WRITE_SIZE / FETCH_SIZE counter passes (separate runs, as the guide prescribes) -> profiles/k1_traffic.json;
#   * K3 / featuriser at config 3 in both arithmetic modes under the kernel trace and one SQ counter pass -> k3_valu_bound.json.
set -o pipefail
O=gpurun_out/${1:-r05fin}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
timeout -k 10 400 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
for k in 1 2; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace$k -o b -- python3 bench.py --no-cpu-baseline --no-other-configs > $O/bench_traced_$k.json 2> $O/bench_traced_$k.err; echo "trace $k rc=$?"
  python3 tools/summarize_rocprof.py ranges $O/trace$k $O/bench_kernel_trace_ranges_$k.json $O/bench_traced_$k.json
done
python3 tools/summarize_rocprof.py stats $O/trace1 $O/bench_kernel_stats.csv
timeout -k 10 240 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $O/pmcw -o b -- python3 bench.py --no-cpu-baseline --no-other-configs --no-lottery --steps 3 > $O/bench_pmcw.json 2> $O/bench_pmcw.err; echo "pmcw rc=$?"
timeout -k 10 240 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/pmcf -o b -- python3 bench.py --no-cpu-baseline --no-other-configs --no-lottery --steps 3 > $O/bench_pmcf.json 2> $O/bench_pmcf.err; echo "pmcf rc=$?"
python3 tools/summarize_rocprof.py pmc $O/pmcw $O/bench_pmc_w.json 0
python3 tools/summarize_rocprof.py pmc $O/pmcf $O/bench_pmc_f.json 0
python3 tools/make_k1_traffic.py $O/bench_pmc_w.json $O/bench_pmc_f.json $O/k1_traffic.json 5 > /dev/null; echo "traffic rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k3_trace_train -o k3 -- python3 tools/profile_workload.py k3 20 > $O/k3_trace_train.log 2>&1; echo "k3 train trace rc=$?"
python3 tools/summarize_rocprof.py stats $O/k3_trace_train $O/k3_trace_stats_back_to_back.csv
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k3_trace -o k3 -- python3 tools/profile_workload.py k3p 20 > $O/k3_trace.log 2>&1; echo "k3 host-paced trace rc=$?"
timeout -k 10 200 rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CU_CYCLES SQ_WAVES -d $O/k3_pmc -o k3 -- python3 tools/profile_workload.py k3 5 > $O/k3_pmc.log 2>&1; echo "k3 pmc rc=$?"
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
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48].replace(",", ";")
        fh.write(f"{name},{(s - t0) / 1e3:.1f},{(e - s) / 1e3:.1f},{((s - prev_end) / 1e3 if prev_end else 0):.1f}\n")
        prev_end = e
PY
python3 tools/summarize_rocprof.py pmc $O/k3_pmc $O/k3_pmc.json
python3 tools/make_k3_valu_bound.py $O/k3_pmc.json $O/k3_valu_bound.json 5 > /dev/null; echo "valu bound rc=$?"
rm -rf $O/trace1 $O/trace2 $O/pmcw $O/pmcf $O/k3_trace $O/k3_trace_train $O/k3_pmc
tail -c 400 $O/bench_n1.json; echo; cat $O/bench_kernel_trace_ranges_1.json $O/bench_kernel_trace_ranges_2.json; cat $O/k3_trace_stats.csv
