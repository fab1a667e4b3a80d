# On the GPU box: build the ablation variants of lstm_fwd_pair_kernel and time the postnet recurrence under each.
bash scripts/build_lstm_abl.sh > /dev/null 2>&1
echo "full kernel:"; python scripts/lstm_bench.py 2>/dev/null | grep "T=612 ndir=1 bf16"
for v in 1 2 4 8 15; do
  echo "LSTM_ABL=$v (1 = no stores, 2 = no transcendentals, 4 = no LDS publication / barrier, 8 = 1/8 of the dot products):"
  KANTTS_LIB=$PWD/kan-tts_amd/variants/libkantts_lstmabl$v.so python scripts/lstm_bench.py 2>/dev/null | grep "T=612 ndir=1 bf16"
done
