# Round 5: the featuriser's tile kernel with four-column tiles (16-byte float rows, 4-byte mask rows) against two-column tiles
# (PS_K3F_TILES_WIDE=0), the -DPS_K3_AB build of one source, same box, alternating passes; identity tests first
set -o pipefail
O=gpurun_out/${1:-r05fwide}
mkdir -p $O
AB=$PWD/protstruc_amd/lib/libprotstruc_hip_ab.so
PROTSTRUC_AMD_LIB=$AB PS_FEAT_FUZZ_TRIALS=600 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "inter_residue or featuris" --deselect tests/test_gpu_parity.py::test_featuriser_every_dispatch_arm_vs_oracle > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
L="320 288 200 192 176 160 144 100 96 80 64 60 52 48 44 40 24 16 8"
for rep in 1 2; do
echo "== four-column tiles, pass $rep"; PROTSTRUC_AMD_LIB=$AB timeout -k 10 300 python3 tools/k3_featuriser_shapes.py 20 $L 2>&1 | grep "N=" | tee $O/feat_wide_$rep.log
echo "== two-column tiles (PS_K3F_TILES_WIDE=0), pass $rep"; PS_K3F_TILES_WIDE=0 PROTSTRUC_AMD_LIB=$AB timeout -k 10 300 python3 tools/k3_featuriser_shapes.py 20 $L 2>&1 | grep "N=" | tee $O/feat_narrow_$rep.log
done
