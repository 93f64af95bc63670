# Kernel trace of the headline bench only (the first half of tools/gpu_profile_r03.sh): bash tools/gpu_trace_only_r03.sh [outdir]
set -o pipefail
O=gpurun_out/${1:-trace}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o b -- python3 bench.py --no-cpu-baseline --no-lottery > $O/bench_traced.json 2> $O/bench_traced.err; echo "trace rc=$?"
python3 tools/summarize_rocprof.py stats $O/trace $O/bench_kernel_stats.csv
K=$(python3 -c "import json,sys; print(json.load(open(sys.argv[1]))['roofline']['kernel'].rstrip('>'))" $O/bench_traced.json)
python3 tools/summarize_rocprof.py trace $O/trace $O/bench_kernel_trace_summary.json "$K" 20
rm -rf $O/trace
python3 -c "import json,sys; r=json.load(open(sys.argv[1]))['roofline']; print('bench HIP events: kernel_ms', r['kernel_ms'], 'frac', r['frac'], r['kernel'])" $O/bench_traced.json
cat $O/bench_kernel_trace_summary.json
