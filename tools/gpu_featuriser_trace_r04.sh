# The fused featuriser against the padded length, from the kernel trace (no host pacing in the numbers): round 4
#   bash tools/gpu_featuriser_trace_r04.sh [outdir-name] [N ...]
set -o pipefail
O=gpurun_out/${1:-r04feat}; shift
NS="${*:-512 511 510 500 496 480 400 384 383 256 255 252 200 160 129 128 101 100}"
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o f -- python3 tools/k3_featuriser_shapes.py 10 $NS > $O/shapes_under_trace.log 2>&1; echo "trace rc=$?"
python3 - "$O" $NS <<'P'
import csv, glob, sys
O, NS = sys.argv[1], [int(v) for v in sys.argv[2:]]
rows = []
for f in glob.glob(O + "/trace/**/*kernel_trace.csv", recursive=True):
    rows += [r for r in csv.DictReader(open(f)) if "k3_featurise" in r["Kernel_Name"] or "k3_inter_residue_geometry" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = 13                                  # 3 warm-ups + 10 timed launches per length
out = open(O + "/featuriser_trace.log", "w")
for k, N in enumerate(NS):
    grp = rows[k * per:(k + 1) * per][3:]
    if not grp: break
    us = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in grp)
    B = max(1, round(2 ** 25 / (N * N)))
    name = grp[0]["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    line = f"N={N:4d} B={B:5d}  mean {sum(us) / len(us):7.1f}  min {us[0]:7.1f}  max {us[-1]:7.1f} us   {B * N * N / (sum(us) / len(us)) / 1e3:6.1f} G pairs/s   {name}"
    print(line); out.write(line + "\n")
P
rm -rf $O/trace
