# Round 5: where the four-column tiles replace the sweeps (-DPS_K3_AB build: PS_K3_FLAT_MAX_N=700 PS_K3_FLAT_UTIL=1000 = tiles
# everywhere they fit, PS_K3_FLAT_MAX_N=100 = sweeps from 100 residues on), then the product's own picks on the final source
set -o pipefail
O=gpurun_out/${1:-r05fvs}
mkdir -p $O
AB=$PWD/protstruc_amd/lib/libprotstruc_hip_ab.so
L="128 184 188 192 196 256 288 300 320 352 384 400 448 480 500 512"
for r in 1 2; do
echo "== tiles everywhere, pass $r"; PROTSTRUC_AMD_LIB=$AB PS_K3_FLAT_MAX_N=700 PS_K3_FLAT_UTIL=1000 timeout -k 10 300 python3 tools/k3_shapes.py 20 $L 2>&1 | grep "N=" | tee -a $O/tiles.log
echo "== sweeps, pass $r"; PROTSTRUC_AMD_LIB=$AB PS_K3_FLAT_MAX_N=100 timeout -k 10 300 python3 tools/k3_shapes.py 20 $L 2>&1 | grep "N=" | tee -a $O/sweeps.log
done
echo "== faithful: tiles everywhere"; PS_K3_FAITHFUL=1 PROTSTRUC_AMD_LIB=$AB PS_K3_FLAT_MAX_N=700 PS_K3_FLAT_UTIL=1000 timeout -k 10 300 python3 tools/k3_shapes.py 20 192 320 448 512 2>&1 | grep "N=" | tee $O/tiles_faithful.log
echo "== faithful: sweeps"; PS_K3_FAITHFUL=1 PROTSTRUC_AMD_LIB=$AB PS_K3_FLAT_MAX_N=100 timeout -k 10 300 python3 tools/k3_shapes.py 20 192 320 448 512 2>&1 | grep "N=" | tee $O/sweeps_faithful.log
echo "== the product's picks (final source)"
timeout -k 10 300 python3 tools/k3_shapes.py 20 512 500 480 448 400 384 352 320 301 300 288 256 220 200 192 180 160 150 140 130 128 120 110 100 99 80 65 64 57 56 48 40 36 33 32 24 16 2>&1 | grep -v amdgpu | tee $O/k3_shapes.log
echo "== the product's picks, faithful"
PS_K3_FAITHFUL=1 timeout -k 10 300 python3 tools/k3_shapes.py 20 512 448 320 256 200 192 160 140 128 99 64 48 33 32 16 2>&1 | grep -v amdgpu | tee $O/k3_shapes_faithful.log
