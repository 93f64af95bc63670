# Round 5: K3 across chain lengths (bash tools/gpu_k3_flat_r05.sh [outdir]): the dispatcher's pick against the one-column kernel at
# 2^25 pairs per launch, both arithmetic modes (-> profiles/r05_k3_shapes.log).
set -o pipefail
O=gpurun_out/${1:-r05flat}
mkdir -p $O
timeout -k 10 300 python3 tools/k3_shapes.py 20 512 384 300 256 200 192 180 160 150 140 130 128 110 100 99 80 65 64 57 56 48 40 33 32 24 16 > $O/k3_shapes.log 2>&1; grep -v amdgpu $O/k3_shapes.log
PS_K3_FAITHFUL=1 timeout -k 10 300 python3 tools/k3_shapes.py 20 512 256 160 140 128 99 64 48 33 32 16 > $O/k3_shapes_faithful.log 2>&1; grep -v amdgpu $O/k3_shapes_faithful.log
