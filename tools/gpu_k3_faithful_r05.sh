# Round 5, first GPU session of the faithful sweep kernels (run on the GPU box: bash tools/gpu_k3_faithful_r05.sh [outdir-name]):
# the library-identity sweep, the K3 / featuriser GPU tests, then both arithmetic modes timed at config 3
# (four columns per lane at 512 threads against two at 1024 for the faithful kernels: PS_K3_NC=2 in the -DPS_K3_AB build).
set -o pipefail
O=gpurun_out/${1:-r05a}
mkdir -p $O
timeout -k 10 120 tools/microbench/libm_identity > $O/libm_identity.log 2>&1; echo "libm_identity rc=$?"; cat $O/libm_identity.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "k3 or inter_residue or featuris" > $O/pytest_k3.log 2>&1; rc=$?; tail -5 $O/pytest_k3.log
timeout -k 10 200 python3 tools/k3_modes_time.py 30 > $O/k3_modes.log 2>&1; cat $O/k3_modes.log
if [ -f protstruc_amd/lib/libprotstruc_hip_ab.so ]; then
  PROTSTRUC_AMD_LIB=$PWD/protstruc_amd/lib/libprotstruc_hip_ab.so PS_K3_NC=2 timeout -k 10 200 python3 tools/k3_modes_time.py 30 > $O/k3_modes_nc2.log 2>&1; cat $O/k3_modes_nc2.log
fi
exit $rc
