set -o pipefail
mkdir -p gpurun_out/r02d
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
# K3: kernel trace, then SQ counters (separate passes)
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02d/k3_trace -o k3 -- python3 tools/profile_workload.py k3 10 > gpurun_out/r02d/k3_trace.log 2>&1; echo "k3 trace rc=$?"
timeout -k 10 200 rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d gpurun_out/r02d/k3_pmc -o k3 -- python3 tools/profile_workload.py k3 5 > gpurun_out/r02d/k3_pmc.log 2>&1; echo "k3 pmc rc=$?"
timeout -k 10 200 rocprofv3 --output-format csv --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD -d gpurun_out/r02d/k3_pmc2 -o k3 -- python3 tools/profile_workload.py k3 5 > gpurun_out/r02d/k3_pmc2.log 2>&1; echo "k3 pmc2 rc=$?"
# K1 atom14 / atom37: fixed-A kernel and any-A kernel
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02d/k1a_trace -o k1a -- python3 tools/profile_workload.py k1a 10 > gpurun_out/r02d/k1a_trace.log 2>&1; echo "k1a trace rc=$?"
timeout -k 10 200 rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d gpurun_out/r02d/k1a_pmc -o k1a -- python3 tools/profile_workload.py k1a 5 > gpurun_out/r02d/k1a_pmc.log 2>&1; echo "k1a pmc rc=$?"
timeout -k 10 200 rocprofv3 --output-format csv --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD -d gpurun_out/r02d/k1a_pmc2 -o k1a -- python3 tools/profile_workload.py k1a 5 > gpurun_out/r02d/k1a_pmc2.log 2>&1; echo "k1a pmc2 rc=$?"
timeout -k 10 200 rocprofv3 --output-format csv --pmc WRITE_SIZE -d gpurun_out/r02d/k1a_pmcw -o k1a -- python3 tools/profile_workload.py k1a 5 > gpurun_out/r02d/k1a_pmcw.log 2>&1; echo "k1a pmcw rc=$?"
for d in k3_trace k1a_trace; do python3 tools/summarize_rocprof.py stats gpurun_out/r02d/$d gpurun_out/r02d/${d}_stats.csv; done
for d in k3_pmc k3_pmc2 k1a_pmc k1a_pmc2 k1a_pmcw; do python3 tools/summarize_rocprof.py pmc gpurun_out/r02d/$d gpurun_out/r02d/${d}.json; done
# keep only the condensed files (raw traces are large)
rm -rf gpurun_out/r02d/k3_trace gpurun_out/r02d/k1a_trace gpurun_out/r02d/k3_pmc gpurun_out/r02d/k3_pmc2 gpurun_out/r02d/k1a_pmc gpurun_out/r02d/k1a_pmc2 gpurun_out/r02d/k1a_pmcw
timeout -k 10 400 tools/microbench/alloc_bench > gpurun_out/r02d/alloc_bench.log 2>&1; echo "alloc rc=$?"; cat gpurun_out/r02d/alloc_bench.log
ls -la gpurun_out/r02d
