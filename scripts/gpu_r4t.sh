#!/bin/bash
# Round 4, visit T: frame-major mel output: parity on the device, PMC traffic of the register kernel, timing, GAN step
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_melspec.py tests/test_multiband.py tests/test_dsp_reference_fixture.py tests/test_hifigan.py tests/test_trainer.py tests/test_voc_dataset.py -m gpu -q -x -k "mel or mrstft or multispec or dsp or gan or voc" 2>&1 | tail -3 | tee gpurun_out/r4t_tests.log
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/r4t_melpmc_$c -o pmc -- python $R/scripts/mel_pmc.py > $R/gpurun_out/r4t_melpmc_$c.log 2>&1
  f=$(find $R/gpurun_out/r4t_melpmc_$c -name "*counter_collection.csv" | head -n 1)
  [ -n "$f" ] && python $R/scripts/pmc_summary.py "$f" melspec | tee -a $R/gpurun_out/r4t_mel_pmc.txt
  rm -rf $R/gpurun_out/r4t_melpmc_$c
done
cd $R
timeout 600 python scripts/mel_bench.py 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r4t_mel_bench.log
for v in "X=1" "X=2"; do
  env $v timeout 300 python scripts/hifigan_bench.py 32 3 bf16 2> gpurun_out/r4t_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'gan step graph %.2f ms  eager %.2f ms  G fwd %.3f ms' % (d.get('gan_step_graph_ms',-1), d['gan_step_ms'], d['generator_forward_ms']))" | tee -a gpurun_out/r4t_gan.log
done
