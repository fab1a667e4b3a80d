# SAM-BERT step (captured, batch 32, bf16 mode) with and without one environment switch, three interleaved pairs.
# Usage (GPU box, repo root): bash scripts/step_ab.sh KANTTS_TN_TILE=64129      (prints ms per step: switch set / unset)
F="--no-hifigan --no-fp32 --no-roofline --no-forward-only --no-cpu-baseline --no-inference --steps 40 --warmup 5"
for i in 1 2 3; do
  for on in 1 0; do
    if [ $on = 1 ]; then e="$1"; else e="KANTTS_AB_UNUSED=1"; fi
    echo -n "$e  "
    env "$e" python bench.py $F 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
  done
done
