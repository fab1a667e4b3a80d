# GAN step (hipGraph replay, batch 32) with and without the XCD-aware workgroup mapping of cconv_wgrad_kernel, interleaved
for i in 1 2 3; do
  for v in 1 0; do
    if [ $v = 1 ]; then export KANTTS_CCONV_WGRAD_NO_XCD_MAP=1; else unset KANTTS_CCONV_WGRAD_NO_XCD_MAP; fi
    echo -n "3-D grid=$v  "
    python scripts/hifigan_bench.py 32 10 bf16 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('graph step ms', d.get('gan_step_graph_ms'), 'eager', d.get('gan_step_ms'))"
  done
done
