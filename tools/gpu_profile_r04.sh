# Round-4 measurement set (run on the GPU box: bash tools/gpu_profile_r04.sh [outdir-name]):
#   the headline bench plain, under rocprofv3 --kernel-trace --stats (with --no-lottery, so that the LAST `steps` K1
#   dispatches of the trace are the timed region), and under the WRITE_SIZE / FETCH_SIZE counter passes
#   (separate runs, as the guide prescribes) -> profiles/k1_traffic.json; K3 at config 3 under the kernel trace and one SQ
#   counter pass (VALU instructions per pair).
set -o pipefail
O=gpurun_out/${1:-r04fin}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
timeout -k 10 240 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o b -- python3 bench.py --no-cpu-baseline --no-lottery > $O/bench_traced.json 2> $O/bench_traced.err; echo "trace rc=$?"
python3 tools/summarize_rocprof.py stats $O/trace $O/bench_kernel_stats.csv
# the kernel the timed launches took (the tuner may have picked another tile length than the default <128>)
K=$(python3 -c "import json,sys; print(json.load(open(sys.argv[1]))['roofline']['kernel'].rstrip('>'))" $O/bench_traced.json)
python3 tools/summarize_rocprof.py trace $O/trace $O/bench_kernel_trace_summary.json "$K" 20
timeout -k 10 240 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $O/pmcw -o b -- python3 bench.py --no-cpu-baseline --steps 3 > $O/bench_pmcw.json 2> $O/bench_pmcw.err; echo "pmcw rc=$?"
timeout -k 10 240 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/pmcf -o b -- python3 bench.py --no-cpu-baseline --steps 3 > $O/bench_pmcf.json 2> $O/bench_pmcf.err; echo "pmcf rc=$?"
python3 tools/summarize_rocprof.py pmc $O/pmcw $O/bench_pmc_w.json 0
python3 tools/summarize_rocprof.py pmc $O/pmcf $O/bench_pmc_f.json 0
python3 tools/make_k1_traffic.py $O/bench_pmc_w.json $O/bench_pmc_f.json $O/k1_traffic.json 3 > /dev/null; echo "traffic rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k3_trace -o k3 -- python3 tools/profile_workload.py k3 10 > $O/k3_trace.log 2>&1; echo "k3 trace rc=$?"
timeout -k 10 200 rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CU_CYCLES SQ_WAVES -d $O/k3_pmc -o k3 -- python3 tools/profile_workload.py k3 5 > $O/k3_pmc.log 2>&1; echo "k3 pmc rc=$?"
python3 tools/summarize_rocprof.py stats $O/k3_trace $O/k3_trace_stats.csv
python3 tools/summarize_rocprof.py pmc $O/k3_pmc $O/k3_pmc.json
rm -rf $O/trace $O/pmcw $O/pmcf $O/k3_trace $O/k3_pmc
tail -c 600 $O/bench_n1.json; echo; cat $O/bench_kernel_trace_summary.json
