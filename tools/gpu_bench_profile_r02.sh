# Round-2 measurement of the headline bench: plain run, rocprofv3 kernel trace of the same command, and the
# WRITE_SIZE / FETCH_SIZE counter passes (separate runs, csv output).  Run on the GPU box: bash tools/gpu_bench_profile_r02.sh
set -o pipefail
O=gpurun_out/r02g
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
timeout -k 10 200 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o b -- python3 bench.py --no-cpu-baseline > $O/bench_traced.json 2> $O/bench_traced.err; echo "trace rc=$?"
python3 tools/summarize_rocprof.py stats $O/trace $O/bench_kernel_stats.csv
python3 tools/summarize_rocprof.py trace $O/trace $O/bench_kernel_trace_summary.json k1_pairdist_a15_pat 20
timeout -k 10 200 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $O/pmcw -o b -- python3 bench.py --no-cpu-baseline --steps 3 > $O/bench_pmcw.json 2> $O/bench_pmcw.err; echo "pmcw rc=$?"
timeout -k 10 200 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/pmcf -o b -- python3 bench.py --no-cpu-baseline --steps 3 > $O/bench_pmcf.json 2> $O/bench_pmcf.err; echo "pmcf rc=$?"
python3 tools/summarize_rocprof.py pmc $O/pmcw $O/bench_pmc_w.json 0
python3 tools/summarize_rocprof.py pmc $O/pmcf $O/bench_pmc_f.json 0
rm -rf $O/trace $O/pmcw $O/pmcf
tail -c 700 $O/bench_n1.json; echo; cat $O/bench_kernel_trace_summary.json; grep -A12 k1_pairdist_a15_pat $O/bench_pmc_w.json | head -20; grep -A12 k1_pairdist_a15_pat $O/bench_pmc_f.json | head -20
