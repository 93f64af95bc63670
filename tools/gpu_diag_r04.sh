# Round-4 diagnosis (run on the GPU box: bash tools/gpu_diag_r04.sh [outdir-name]): where do K3's and K5 / K6's
# cycles go?  Counter passes only (--pmc alone, as the guide prescribes), one kernel trace for the clock reference.
#   K3 at config 3 (B=128, N=512): wave cycles split into active / waiting-on-memory / issue-stalled, VMEM / SMEM
#   instruction cycles, TA FIFO back-pressure, L2 -> fabric write requests, effective clock (GRBM_GUI_ACTIVE).
#   K5 / K6 at config 5 (B=256, N=384): the same split.
set -o pipefail
O=gpurun_out/${1:-r04a}
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
run_pmc() {   # name workload reps counters...
  local name=$1 wl=$2 reps=$3; shift 3
  timeout -k 10 200 rocprofv3 --output-format csv --pmc "$@" -d $O/$name -o p -- python3 tools/profile_workload.py $wl $reps > $O/$name.log 2>&1
  echo "$name rc=$?"
  python3 tools/summarize_rocprof.py pmc $O/$name $O/$name.json && rm -rf $O/$name
}
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k3_trace -o k3 -- python3 tools/profile_workload.py k3 10 > $O/k3_trace.log 2>&1; echo "k3 trace rc=$?"
python3 tools/summarize_rocprof.py stats $O/k3_trace $O/k3_trace_stats.csv && rm -rf $O/k3_trace
run_pmc k3_pmc_waves k3 5 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES &&
run_pmc k3_pmc_mem k3 5 SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SMEM SQ_INSTS_SMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_ACTIVE_INST_SCA &&
run_pmc k3_pmc_issue k3 5 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES &&
run_pmc k3_pmc_tcc k3 5 TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_WRITE_sum GRBM_GUI_ACTIVE GRBM_COUNT &&
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k5_trace -o k5 -- python3 tools/profile_workload.py k5 20 > $O/k5_trace.log 2>&1; echo "k5 trace rc=$?"
python3 tools/summarize_rocprof.py stats $O/k5_trace $O/k5_trace_stats.csv && rm -rf $O/k5_trace
run_pmc k5_pmc_waves k5 10 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES &&
run_pmc k5_pmc_issue k5 10 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_LDS SQ_LEVEL_WAVES GRBM_GUI_ACTIVE
ls -la $O
