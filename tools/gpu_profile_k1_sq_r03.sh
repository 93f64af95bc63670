# SQ counter pass of the headline bench (VALU instructions per 16-byte slot of every pattern-kernel template the tuner tries):
#   bash tools/gpu_profile_k1_sq_r03.sh [outdir-name]     (run on the GPU box)
set -o pipefail
O=gpurun_out/${1:-k1sq}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
timeout -k 10 240 rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES -d $O/pmc -o b -- python3 bench.py --no-cpu-baseline --no-lottery --steps 3 > $O/bench.json 2> $O/bench.err; echo "pmc rc=$?"
python3 tools/summarize_rocprof.py pmc $O/pmc $O/k1_sq_pmc.json 0
rm -rf $O/pmc
python3 - $O/k1_sq_pmc.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
slots = 64 * 512 * 512 * 225 / 4 * 1.25          # 16-byte slots per launch: distance plane + mask plane
for k, v in sorted(d.items()):
    if "k1_pairdist" in k:
        m = v["per_dispatch_mean"]
        print(k, "dispatches", v["dispatches_used"], "VALU wave-instr per slot %.1f" % (m["SQ_INSTS_VALU"] * 64 / slots),
              "active/insts %.2f" % (m["SQ_ACTIVE_INST_VALU"] / m["SQ_INSTS_VALU"]), "vgpr", v["vgpr"], "ns", round(v["mean_ns_under_pmc"]))
PY
