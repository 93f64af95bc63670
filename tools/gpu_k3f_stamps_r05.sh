# Round 5: wave time stamps of the featuriser's tile kernel (-DPS_K3_AB build): two 256-thread workgroups per CU (product)
# against one 512-thread workgroup per CU
set -o pipefail
O=gpurun_out/${1:-r05fst}
mkdir -p $O
AB=$PWD/protstruc_amd/lib/libprotstruc_hip_ab.so
echo "#### product: 2 workgroups x 256 threads per CU"
K3F_STAMPS_DUMP=$O PROTSTRUC_AMD_LIB=$AB timeout -k 10 300 python3 tools/k3f_stamps.py 64 48 160 2>&1 | grep -v amdgpu.ids | tee $O/stamps.log
echo "#### 1 workgroup x 512 threads per CU"
PS_K3F_TILES_WGS=1 PS_K3F_TILES_THREADS=512 PROTSTRUC_AMD_LIB=$AB timeout -k 10 300 python3 tools/k3f_stamps.py 64 48 160 2>&1 | grep -v amdgpu.ids | tee $O/stamps_1x512.log
echo "#### 2 workgroups x 512 threads per CU"
PS_K3F_TILES_WGS=2 PS_K3F_TILES_THREADS=512 PROTSTRUC_AMD_LIB=$AB timeout -k 10 300 python3 tools/k3f_stamps.py 64 48 160 2>&1 | grep -v amdgpu.ids | tee $O/stamps_2x512.log
