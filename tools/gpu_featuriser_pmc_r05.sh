# Round 5: why the featuriser stops at ~3.4 TB/s below ~128 residues -- counter passes at N = 64, 48, 160 (tiles), 128, 512 (sweep)
set -o pipefail
O=gpurun_out/${1:-r05fpmc}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
timeout -k 10 200 rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CU_CYCLES SQ_WAVES -d $O/p1 -o k -- python3 tools/profile_workload.py featshort 5 > $O/p1.log 2>&1; echo "pass 1 rc=$?"
timeout -k 10 200 rocprofv3 --output-format csv --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum -d $O/p2 -o k -- python3 tools/profile_workload.py featshort 5 > $O/p2.log 2>&1; echo "pass 2 rc=$?"
timeout -k 10 200 rocprofv3 --output-format csv --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_WR TA_BUSY_avr -d $O/p3 -o k -- python3 tools/profile_workload.py featshort 5 > $O/p3.log 2>&1; echo "pass 3 rc=$?"
timeout -k 10 200 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $O/p4 -o k -- python3 tools/profile_workload.py featshort 5 > $O/p4.log 2>&1; echo "pass 4 rc=$?"
timeout -k 10 200 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/p5 -o k -- python3 tools/profile_workload.py featshort 5 > $O/p5.log 2>&1; echo "pass 5 rc=$?"
for k in 1 2 3 4 5; do python3 tools/summarize_rocprof.py pmc $O/p$k $O/pmc_$k.json 2 || true; done
rm -rf $O/p1 $O/p2 $O/p3 $O/p4 $O/p5
python3 - $O <<'PY'
import json, sys
for k in (1, 2, 3, 4, 5):
    try:
        d = json.load(open(f"{sys.argv[1]}/pmc_{k}.json"))
    except Exception as e:
        print("pass", k, "unreadable", e); continue
    for name, v in d.items():
        if not name.startswith("k3_"): continue
        print(f"pass {k}  {name[:62]:62s} us {v['mean_ns_under_pmc']/1e3:6.1f}  " + "  ".join(f"{c}={int(x)}" for c, x in v["per_dispatch_mean"].items()))
PY
