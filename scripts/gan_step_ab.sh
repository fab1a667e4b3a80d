# GAN step (hipGraph replay, batch 32, bf16 mode) with and without one environment switch, three interleaved pairs.
# Usage (GPU box, repo root): bash scripts/gan_step_ab.sh KANTTS_CCONV_WGRAD_NO_XCD_MAP=1
for i in 1 2 3; do
  for on in 1 0; do
    if [ $on = 1 ]; then e="$1"; else e="KANTTS_AB_UNUSED=1"; fi
    echo -n "$e  "
    env "$e" python scripts/hifigan_bench.py 32 10 bf16 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('graph step ms', d.get('gan_step_graph_ms'), 'generator forward ms', d.get('generator_forward_ms'))"
  done
done
