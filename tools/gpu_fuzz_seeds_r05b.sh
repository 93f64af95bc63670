# Round 5, final build (featuriser tile kernel, one-map k3_flat): the differential fuzzes on four more seeds
set -o pipefail
O=gpurun_out/${1:-r05fuzzb}
mkdir -p $O
for seed in 31 32 33 34; do
  PS_K3_FUZZ_SEED=$seed PS_K3_FUZZ_TRIALS=20000 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "test_k3_differential_fuzz" 2>&1 | tail -1 | sed "s/^/k3 fuzz seed $seed, 20000 trials: /" | tee -a $O/fuzz.log
  PS_FEAT_FUZZ_SEED=$seed PS_FEAT_FUZZ_TRIALS=2000 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "test_inter_residue_geometry_differential_fuzz" 2>&1 | tail -1 | sed "s/^/featuriser fuzz seed $seed, 2000 trials: /" | tee -a $O/fuzz.log
done
