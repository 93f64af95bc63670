# Does a kernel's duration grow over a run of launches because the clock drops?  GRBM_GUI_ACTIVE (cycles summed over the 8
# XCDs) and duration per dispatch of the K3 workload (run on the GPU box: bash tools/gpu_clock_probe_r04.sh [outdir] [reps])
set -o pipefail
O=gpurun_out/${1:-r04clk}
R=${2:-30}
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
timeout -k 10 200 rocprofv3 --output-format csv --pmc GRBM_GUI_ACTIVE -d $O/pmc -o p -- python3 tools/profile_workload.py k3 $R > $O/pmc.log 2>&1; echo "pmc rc=$?"
python3 - "$O" <<'PY'
import csv, glob, sys
o = sys.argv[1]
rows = []
for f in glob.glob(o + "/pmc/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
with open(o + "/clock_per_dispatch.csv", "w") as fh:
    fh.write("dispatch,kernel,duration_us,gui_active_cycles_per_xcd,GHz\n")
    for r in rows:
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
            continue
        ns = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        cyc = float(r["Counter_Value"]) / 8
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:36].replace(",", ";")
        fh.write(f"{r['Dispatch_Id']},{name},{ns / 1e3:.1f},{cyc:.0f},{cyc / ns:.3f}\n")
PY
rm -rf $O/pmc
grep "featurise\|inter_residue" $O/clock_per_dispatch.csv | awk -F, '{printf "%s us %s GHz | ", $3, $5}'; echo
grep "sweep<4; 12" $O/clock_per_dispatch.csv | awk -F, '{printf "%s us %s GHz | ", $3, $5}'; echo
