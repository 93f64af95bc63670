# Round 5, final source: counters on the tile kernels (k3_flat, k3_featurise_tiles; four-column tiles at N = 64 / 160) with the
# sweeps beside them -- instructions, issue activity, and the bytes written against the algorithmic bytes
set -o pipefail
O=gpurun_out/${1:-r05tpmc}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
for w in k3flat featshort; do
timeout -k 10 200 rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE -d $O/${w}_1 -o k -- python3 tools/profile_workload.py $w 5 > $O/${w}_1.log 2>&1; echo "$w pass 1 rc=$?"
timeout -k 10 200 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $O/${w}_2 -o k -- python3 tools/profile_workload.py $w 5 > $O/${w}_2.log 2>&1; echo "$w pass 2 rc=$?"
timeout -k 10 200 rocprofv3 --output-format csv --pmc SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/${w}_3 -o k -- python3 tools/profile_workload.py $w 5 > $O/${w}_3.log 2>&1; echo "$w pass 3 rc=$?"
for k in 1 2 3; do python3 tools/summarize_rocprof.py pmc $O/${w}_$k $O/pmc_${w}_$k.json 2 || true; rm -rf $O/${w}_$k; done
done
python3 - $O <<'PY'
import json, sys
for w in ("k3flat", "featshort"):
    for k in (1, 2, 3):
        try:
            d = json.load(open(f"{sys.argv[1]}/pmc_{w}_{k}.json"))
        except Exception as e:
            print(w, "pass", k, "unreadable", e); continue
        for name, v in d.items():
            if not name.startswith("k3_"): continue
            print(f"{w:9s} pass {k}  {name[:58]:58s} us {v['mean_ns_under_pmc']/1e3:6.1f}  " + "  ".join(f"{c}={int(x)}" for c, x in v["per_dispatch_mean"].items()))
PY
