# Round 5: the featuriser's tile kernel against the sweep / one-column kernels (-DPS_K3_AB build: PS_K3F_TILES_MIN / PS_K3F_TILES_MAX),
# identity tests first (sentinels, any alignment, fuzz), then chain lengths at 2^25 pairs, same box.
set -o pipefail
O=gpurun_out/${1:-r05ftiles}
mkdir -p $O
AB=$PWD/protstruc_amd/lib/libprotstruc_hip_ab.so
PROTSTRUC_AMD_LIB=$AB PS_K3F_TILES_MIN=4 PS_K3F_TILES_MAX=700 PS_FEAT_FUZZ_TRIALS=150 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "inter_residue or featuris" --deselect tests/test_gpu_parity.py::test_featuriser_every_dispatch_arm_vs_oracle > $O/pytest.log 2>&1; rc=$?; tail -4 $O/pytest.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
L="512 500 384 256 255 200 160 129 128 101 100 99 80 64 48 40 33 24 16"
echo "== tiles everywhere"; PROTSTRUC_AMD_LIB=$AB PS_K3F_TILES_MIN=4 PS_K3F_TILES_MAX=700 timeout -k 10 300 python3 tools/k3_featuriser_shapes.py 20 $L 2>&1 | grep "N=" | tee $O/feat_tiles.log
echo "== product (sweep / one-column)"; PROTSTRUC_AMD_LIB=$AB timeout -k 10 300 python3 tools/k3_featuriser_shapes.py 20 $L 2>&1 | grep "N=" | tee $O/feat_default.log
