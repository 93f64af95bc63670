# Round 5: K3 across chain lengths (bash tools/gpu_k3_flat_r05.sh [outdir]): K3 GPU tests; the dispatcher's pick against the one-column
# kernel at 2^25 pairs per launch, both arithmetic modes (-> profiles/r05_k3_shapes.log).  The flat kernel's three lane maps side by
# (history: the lane maps 0 and 1 and their PS_K3_FLAT_MAP / PS_K3_ROWMAJOR_MIN switches were removed from the source after this run)
# side (-DPS_K3_AB build, PS_K3_FLAT_MAP = 0: elements 64 apart, 1: a lane per column at 57..64, 2: 2 x 2 tiles): profiles/r05_k3_flat_lane_maps.log
set -o pipefail
O=gpurun_out/${1:-r05flat}
mkdir -p $O
timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "k3" > $O/pytest_k3.log 2>&1; rc=$?; tail -4 $O/pytest_k3.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 tools/k3_shapes.py 20 512 384 300 256 220 200 192 180 160 150 140 130 128 120 110 100 99 80 65 64 57 56 48 40 33 32 24 16 > $O/k3_shapes.log 2>&1; grep -v amdgpu $O/k3_shapes.log
PS_K3_FAITHFUL=1 timeout -k 10 300 python3 tools/k3_shapes.py 20 512 256 200 160 140 128 99 64 48 33 32 16 > $O/k3_shapes_faithful.log 2>&1; grep -v amdgpu $O/k3_shapes_faithful.log
