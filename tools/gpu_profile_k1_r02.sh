# SQ counter passes on the headline K1 launch (B=64, N=512, A=15; default configuration), round 2
set -o pipefail
O=gpurun_out/r02x
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
timeout -k 10 200 rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/a -o k1 -- python3 tools/profile_workload.py k1 10 > $O/a.log 2>&1; echo "a rc=$?"
timeout -k 10 200 rocprofv3 --output-format csv --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD -d $O/b -o k1 -- python3 tools/profile_workload.py k1 10 > $O/b.log 2>&1; echo "b rc=$?"
python3 tools/summarize_rocprof.py pmc $O/a $O/k1_pmc.json
python3 tools/summarize_rocprof.py pmc $O/b $O/k1_pmc2.json
rm -rf $O/a $O/b
cat $O/k1_pmc.json $O/k1_pmc2.json
