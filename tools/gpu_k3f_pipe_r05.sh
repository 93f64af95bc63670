# Round 5: the featuriser's tile kernel as one 512-thread workgroup per CU with double-buffered staging (the -DPS_K3_AB build)
# against the product library of the same tree (two 256-thread workgroups per CU): identity tests, chain lengths, wave stamps
set -o pipefail
O=gpurun_out/${1:-r05fpipe}
mkdir -p $O
AB=$PWD/protstruc_amd/lib/libprotstruc_hip_ab.so
PROTSTRUC_AMD_LIB=$AB PS_FEAT_FUZZ_TRIALS=150 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "inter_residue or featuris" --deselect tests/test_gpu_parity.py::test_featuriser_every_dispatch_arm_vs_oracle > $O/pytest.log 2>&1; rc=$?; tail -4 $O/pytest.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
L="200 160 129 100 96 80 64 48 40 33 24 16 8"
echo "== AB build (pipelined, 1 x 512)"; PROTSTRUC_AMD_LIB=$AB timeout -k 10 300 python3 tools/k3_featuriser_shapes.py 20 $L 2>&1 | grep "N=" | tee $O/feat_ab.log
echo "== product library"; timeout -k 10 300 python3 tools/k3_featuriser_shapes.py 20 $L 2>&1 | grep "N=" | tee $O/feat_product.log
for P in 2 3 6 8; do echo "== AB, PS_K3F_TILES_PASSES=$P"; PS_K3F_TILES_PASSES=$P PROTSTRUC_AMD_LIB=$AB timeout -k 10 300 python3 tools/k3_featuriser_shapes.py 20 160 96 64 48 33 2>&1 | grep "N=" | tee $O/feat_ab_p$P.log; done
echo "#### stamps"
PROTSTRUC_AMD_LIB=$AB timeout -k 10 300 python3 tools/k3f_stamps.py 64 160 2>&1 | grep -v amdgpu.ids | tee $O/stamps.log
