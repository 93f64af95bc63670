for rep in 1 2; do
for pad in 20 36 28; do echo "== lds_pad_kb=$pad"; timeout -k 10 200 python tools/k1_n_sweep.py lds_pad_kb=$pad 512 256 128 64 32 1024 2048 496 2>&1 | grep "^A="; done
done
