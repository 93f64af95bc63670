# Round 5: k3_flat with tiles of two row pairs x FOUR columns where a row of the tile is one aligned 16-byte store (N % 4 == 0; two halves sharing the index arithmetic and the row points)
# stores) against tiles of two columns (git HEAD before the change, same flags: libprotstruc_hip_old.so), same box:
# identity tests on the new build, then chain lengths at 2^25 pairs, alternating
set -o pipefail
O=gpurun_out/${1:-r05flat4}
mkdir -p $O
AB=$PWD/protstruc_amd/lib/libprotstruc_hip_ab.so
OLD=$PWD/protstruc_amd/lib/libprotstruc_hip_old.so
PROTSTRUC_AMD_LIB=$AB PS_K3_FUZZ_TRIALS=20000 timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "k3" --deselect tests/test_gpu_parity.py::test_k3_every_dispatch_arm_vs_oracle > $O/pytest_k3.log 2>&1; rc=$?; tail -4 $O/pytest_k3.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
L="220 200 160 140 100 99 80 66 65 64 48 40 36 33"
for rep in 1 2; do
echo "== new (2 x 4 tiles), pass $rep"; PROTSTRUC_AMD_LIB=$AB timeout -k 10 300 python3 tools/k3_shapes.py 20 $L 2>&1 | grep -v amdgpu | tee $O/k3_shapes_new_$rep.log
echo "== before (2 x 2 tiles), pass $rep"; PROTSTRUC_AMD_LIB=$OLD timeout -k 10 300 python3 tools/k3_shapes.py 20 $L 2>&1 | grep -v amdgpu | tee $O/k3_shapes_old_$rep.log
done
echo "== faithful, new"; PS_K3_FAITHFUL=1 PROTSTRUC_AMD_LIB=$AB timeout -k 10 300 python3 tools/k3_shapes.py 20 300 200 140 99 64 48 2>&1 | grep -v amdgpu | tee $O/k3_shapes_faithful_new.log
echo "== faithful, before"; PS_K3_FAITHFUL=1 PROTSTRUC_AMD_LIB=$OLD timeout -k 10 300 python3 tools/k3_shapes.py 20 300 200 140 99 64 48 2>&1 | grep -v amdgpu | tee $O/k3_shapes_faithful_old.log
