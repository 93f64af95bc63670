# Round 5: where the flat K3 kernel's time goes -- SQ / LDS / TA counter passes at N = 64, 99, 48 with the sweep at N = 512 beside it
set -o pipefail
O=gpurun_out/${1:-r05flatpmc}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
timeout -k 10 200 rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CU_CYCLES SQ_WAVES -d $O/p1 -o k -- python3 tools/profile_workload.py k3flat 5 > $O/p1.log 2>&1; echo "pass 1 rc=$?"
timeout -k 10 200 rocprofv3 --output-format csv --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS -d $O/p2 -o k -- python3 tools/profile_workload.py k3flat 5 > $O/p2.log 2>&1; echo "pass 2 rc=$?"
timeout -k 10 200 rocprofv3 --output-format csv --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d $O/p3 -o k -- python3 tools/profile_workload.py k3flat 5 > $O/p3.log 2>&1; echo "pass 3 rc=$?"
for k in 1 2 3; do python3 tools/summarize_rocprof.py pmc $O/p$k $O/pmc_$k.json 2 || true; done
rm -rf $O/p1 $O/p2 $O/p3
python3 - $O <<'PY'
import json, sys
for k in (1, 2, 3):
    try:
        d = json.load(open(f"{sys.argv[1]}/pmc_{k}.json"))
    except Exception as e:
        print("pass", k, "unreadable", e); continue
    for name, v in d.items():
        if not name.startswith("k3_"): continue
        print(name[:70], "ns", int(v["mean_ns_under_pmc"]), {c: int(x) for c, x in v["per_dispatch_mean"].items()})
PY
