#!/bin/bash
# Round 4, visit G: per-wave radix-4 mel-STFT kernel (tests, A/B), bench legs with graph-replayed upsampling timing
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_melspec.py tests/test_dsp_reference_fixture.py tests/test_audio_processor.py tests/test_multiband.py tests/test_independent_pins.py tests/test_hifigan.py -m gpu -q -x -k "mel or dsp or stft or multi or gan_step_losses" 2>&1 | tail -3 | tee gpurun_out/r4g_mel_tests.log
for v in "X=new" "KANTTS_MELSPEC_V1=1" "X=new2"; do
  echo "$v" | tee -a gpurun_out/r4g_mel_bench.log
  env $v timeout 200 python scripts/mel_bench.py 2>&1 | grep n_fft | tee -a gpurun_out/r4g_mel_bench.log
done
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference --no-fp32 > gpurun_out/r4g_bench.json 2> gpurun_out/r4g_bench.err
grep "^\[bench" gpurun_out/r4g_bench.err | tail -5
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4g_bench.json').read().strip().splitlines()[-1])
print('sambert', d['ms_per_step'], 'fwd', d['roofline'].get('forward_ms'))
h=d['hifigan']
print('gan', h.get('gan_step_ms'), 'G fwd', h.get('generator_forward_ms'))
print('up', h['upsampling']['ms'], h['upsampling']['frac'], h['upsampling']['stage_us'])
print('dual', h['upsampling_dual_path']['ms'], h['upsampling_dual_path']['frac'], h['upsampling_dual_path']['stage_us'])
m=d['melspec']; print('mel', m['forward_ms'], m['roofline_saturating']['forward_ms'], m['roofline_saturating']['frac'], m['parity_error'])
PY
