# Round 5: the featuriser's tile kernel with more workgroups than resident slots (the later of a CU's two workgroups runs at
# half speed until the earlier one leaves: smaller workgroups shorten the tail) -- the -DPS_K3_AB build, PS_K3F_TILES_OVER
set -o pipefail
O=gpurun_out/${1:-r05fover}
mkdir -p $O
AB=$PWD/protstruc_amd/lib/libprotstruc_hip_ab.so
for V in 1 2 3 4 8; do echo "== PS_K3F_TILES_OVER=$V"; PS_K3F_TILES_OVER=$V PROTSTRUC_AMD_LIB=$AB timeout -k 10 300 python3 tools/k3_featuriser_shapes.py 20 200 160 96 64 48 33 16 2>&1 | grep "N=" | tee $O/feat_over$V.log; done
echo "#### stamps, OVER=4"
PS_K3F_TILES_OVER=4 PROTSTRUC_AMD_LIB=$AB timeout -k 10 300 python3 tools/k3f_stamps.py 64 2>&1 | grep -v amdgpu.ids | tee $O/stamps4.log
