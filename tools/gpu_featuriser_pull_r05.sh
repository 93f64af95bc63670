# Round 5, verdict item 7 (one bounded attempt): the featuriser at rows that are not whole 64-byte segments, a wave pulling 1 / 2 / 4
# consecutive tasks at a time (-DPS_K3_AB build, PS_K3F_PULL), same box; then the featuriser tests with the product library.
AB=$PWD/protstruc_amd/lib/libprotstruc_hip_ab.so
for pull in 1 2 4 1 2; do
  echo "== PS_K3F_PULL=$pull"
  PROTSTRUC_AMD_LIB=$AB PS_K3F_PULL=$pull timeout -k 10 200 python3 tools/k3_featuriser_shapes.py 30 500 511 255 383 512 2>&1 | grep "N="
done
