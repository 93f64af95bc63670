# Counter passes on K1's row-phase kernel at the small atom counts (tools/profile_workload.py k1s), round 3
set -o pipefail
O=gpurun_out/${1:-r03d}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
timeout -k 10 240 rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_WR SQ_INSTS_LDS -d $O/a -o k1s -- python3 tools/profile_workload.py k1s 6 > $O/a.log 2>&1; echo "a rc=$?"
timeout -k 10 240 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $O/w -o k1s -- python3 tools/profile_workload.py k1s 6 > $O/w.log 2>&1; echo "w rc=$?"
timeout -k 10 240 rocprofv3 --output-format csv --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d $O/q -o k1s -- python3 tools/profile_workload.py k1s 6 > $O/q.log 2>&1; echo "q rc=$?"
python3 tools/summarize_rocprof.py pmc $O/a $O/k1s_pmc.json
python3 tools/summarize_rocprof.py pmc $O/w $O/k1s_pmcw.json
python3 tools/summarize_rocprof.py pmc $O/q $O/k1s_pmcq.json
rm -rf $O/a $O/w $O/q
