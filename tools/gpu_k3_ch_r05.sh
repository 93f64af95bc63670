# Round 5: rows per pulled task of the K3 sweep at config 3 and two other lengths (-DPS_K3_AB build, PS_K3_CH), same box.
AB=$PWD/protstruc_amd/lib/libprotstruc_hip_ab.so
for ch in 0 2 4 8; do
  echo "== PS_K3_CH=$ch (0: the product's rule)"
  PROTSTRUC_AMD_LIB=$AB PS_K3_CH=$ch timeout -k 10 200 python3 tools/k3_modes_time.py 40 2>&1 | grep "mode"
  PROTSTRUC_AMD_LIB=$AB PS_K3_CH=$ch timeout -k 10 200 python3 tools/k3_shapes.py 20 256 128 2>&1 | grep "N="
done
PROTSTRUC_AMD_LIB=$AB PS_K3_CH=2 timeout -k 10 120 python3 tools/k3_stamps.py 2>&1 | grep -v amdgpu.ids | head -10
