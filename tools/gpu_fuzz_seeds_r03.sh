# One-off long differential fuzz of K1 on other seeds (the committed test runs 3000 trials of one seed):
#   bash tools/gpu_fuzz_seeds_r03.sh [trials per seed]      (run on the GPU box; ~15 s per 20 000 trials)
T=${1:-20000}
mkdir -p gpurun_out
for seed in 11 23 37 41 59 73; do
  PS_FUZZ_SEED=$seed PS_FUZZ_TRIALS=$T timeout -k 10 300 python -m pytest tests/test_gpu_k1_fuzz.py -x -q -m gpu > gpurun_out/fuzz_seed_$seed.log 2>&1 || { echo "seed $seed FAILED"; tail -20 gpurun_out/fuzz_seed_$seed.log; exit 1; }
  echo "seed $seed: $(tail -1 gpurun_out/fuzz_seed_$seed.log)"
done
