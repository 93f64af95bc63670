# Write-request counters of K1's row-phase kernel at the unaligned shapes, per plane (tools/profile_workload.py k1u), round 4
set -o pipefail
O=gpurun_out/${1:-r04x}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
timeout -k 10 240 rocprofv3 --output-format csv --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d $O/q -o k1u -- python3 tools/profile_workload.py k1u 2 > $O/q.log 2>&1; echo "q rc=$?"
timeout -k 10 240 rocprofv3 --output-format csv --pmc SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES -d $O/a -o k1u -- python3 tools/profile_workload.py k1u 2 > $O/a.log 2>&1; echo "a rc=$?"
python3 tools/summarize_rocprof.py pmcseq $O/q $O/k1u_write_requests.json rowphase
python3 tools/summarize_rocprof.py pmcseq $O/a $O/k1u_sq.json rowphase
rm -rf $O/a $O/q
python3 - "$O" <<'P'
import json, sys
O = sys.argv[1]
for f in ("k1u_write_requests.json", "k1u_sq.json"):
    for e in json.load(open(f"{O}/{f}")):
        print(f, {k: (round(v) if isinstance(v, float) else v) for k, v in e.items()})
P
