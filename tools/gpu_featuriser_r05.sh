# Round 5: the featuriser with several structures per staging pass (bash tools/gpu_featuriser_r05.sh [outdir]): its GPU tests,
# then chain lengths at 2^25 pairs -- the product, one structure per pass (PS_K3F_KS_MAX=1 in the -DPS_K3_AB build: round 4's
# behaviour), and the sweep below 64 residues (PS_K3F_MIN_N).
set -o pipefail
O=gpurun_out/${1:-r05feat}
mkdir -p $O
timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "inter_residue or featuris" > $O/pytest_feat.log 2>&1; rc=$?; tail -4 $O/pytest_feat.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
L="512 500 384 256 255 200 160 129 128 101 100 99 80 64"
echo "== product"; timeout -k 10 300 python3 tools/k3_featuriser_shapes.py 20 $L 48 33 2>&1 | grep "N=" | tee $O/feat_shapes.log
AB=$PWD/protstruc_amd/lib/libprotstruc_hip_ab.so
echo "== one structure per pass (round 4)"; PROTSTRUC_AMD_LIB=$AB PS_K3F_KS_MAX=1 timeout -k 10 300 python3 tools/k3_featuriser_shapes.py 20 $L 2>&1 | grep "N=" | tee $O/feat_shapes_ks1.log
echo "== sweep from 32 residues"; PROTSTRUC_AMD_LIB=$AB PS_K3F_MIN_N=32 timeout -k 10 300 python3 tools/k3_featuriser_shapes.py 20 63 56 48 40 33 32 2>&1 | grep "N=" | tee $O/feat_shapes_min32.log
echo "== one-column below 64 (same box)"; PROTSTRUC_AMD_LIB=$AB timeout -k 10 300 python3 tools/k3_featuriser_shapes.py 20 63 56 48 40 33 32 2>&1 | grep "N=" | tee $O/feat_shapes_onecol.log
