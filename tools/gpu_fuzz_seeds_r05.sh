# Round 5: the K3 and featuriser differential fuzzes on other seeds (the flat kernel, the faithful sweeps and the multi-structure
# staging are new): bash tools/gpu_fuzz_seeds_r05.sh [outdir]
set -o pipefail
O=gpurun_out/${1:-r05fuzz}
mkdir -p $O
for seed in 11 12 13; do
  PS_K3_FUZZ_SEED=$seed PS_K3_FUZZ_TRIALS=4000 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "test_k3_differential_fuzz" 2>&1 | tail -1 | sed "s/^/k3 fuzz seed $seed, 4000 trials: /" | tee -a $O/fuzz.log
  PS_FEAT_FUZZ_SEED=$seed PS_FEAT_FUZZ_TRIALS=500 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "test_inter_residue_geometry_differential_fuzz" 2>&1 | tail -1 | sed "s/^/featuriser fuzz seed $seed, 500 trials: /" | tee -a $O/fuzz.log
done
