# Round 5: the tile featuriser with its store-offset selects computed once per tile (the -DPS_K3_AB build of the new source)
# against the same source before the change (git HEAD, same flags: libprotstruc_hip_old.so): identity tests, then chain lengths, same box
set -o pipefail
O=gpurun_out/${1:-r05fhoist}
mkdir -p $O
AB=$PWD/protstruc_amd/lib/libprotstruc_hip_ab.so
PROTSTRUC_AMD_LIB=$AB PS_FEAT_FUZZ_TRIALS=300 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "inter_residue or featuris" > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
L="200 160 129 100 96 80 64 63 48 40 33 16"
for rep in 1 2; do
echo "== new (AB build), pass $rep"; PROTSTRUC_AMD_LIB=$AB timeout -k 10 300 python3 tools/k3_featuriser_shapes.py 20 $L 2>&1 | grep "N=" | tee $O/feat_new_$rep.log
echo "== before (HEAD source, same flags), pass $rep"; PROTSTRUC_AMD_LIB=$PWD/protstruc_amd/lib/libprotstruc_hip_old.so timeout -k 10 300 python3 tools/k3_featuriser_shapes.py 20 $L 2>&1 | grep "N=" | tee $O/feat_old_$rep.log
done
