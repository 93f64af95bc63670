# Round 5: the shader clock while the K3 kernels run -- GRBM_GUI_ACTIVE (cycles the GPU is busy) per dispatch over the dispatch's duration
set -o pipefail
O=gpurun_out/${1:-r05clk}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
for w in featshort k3; do
timeout -k 10 200 rocprofv3 --output-format csv --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY -d $O/$w -o k -- python3 tools/profile_workload.py $w 5 > $O/$w.log 2>&1; echo "$w rc=$?"
python3 tools/summarize_rocprof.py pmc $O/$w $O/pmc_$w.json 2 || true
rm -rf $O/$w
done
python3 - $O <<'PY'
import json, sys
for w in ("featshort", "k3"):
    d = json.load(open(f"{sys.argv[1]}/pmc_{w}.json"))
    for name, v in d.items():
        if not name.startswith("k3_"): continue
        c = v["per_dispatch_mean"]; us = v["mean_ns_under_pmc"] / 1e3
        print(f"{w:9s} {name[:60]:60s} us {us:6.1f}  GUI_ACTIVE/us = {c.get('GRBM_GUI_ACTIVE', 0)/us:7.1f} MHz(?)  " + "  ".join(f"{k}={int(x)}" for k, x in c.items()))
PY
