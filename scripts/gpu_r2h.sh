#!/bin/bash
# round-2 visit H: profile after the LSTM prefetch rings / fused residual gradient / conv_c1 embeddings / BM=128 tiles;
# A/B: generator forward with and without the streaming upsampling path, conv_wgrad atomics budgets
mkdir -p gpurun_out
timeout 300 python scripts/bgemm_bench.py > gpurun_out/r2h_bgemm.log 2>&1; grep -E "fwd   6528|dgrad|wgrad|fwd   2048|fwd   19584" gpurun_out/r2h_bgemm.log | head -24
for v in "" 1; do
KANTTS_NO_UPSTREAM=$v python - <<'PY'
import os, sys, torch
sys.path.insert(0, "kan-tts_amd")
import kantts._hip as hip
from kantts.models.hifigan.hifigan import Generator
hip.set_precision("bf16")
torch.manual_seed(0)
G = Generator().cuda()
x = torch.randn(32, 80, 32, device="cuda")
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
with torch.no_grad():
    print("NO_UPSTREAM=%r generator forward (no grad) ms:" % os.environ.get("KANTTS_NO_UPSTREAM"), t(lambda: G(x)))
y = G(x)
def fb():
    G.zero_grad(); G(x).square().mean().backward()
print("  forward+backward ms:", t(fb, 5))
PY
done
for cap in 6 12 24; do
  KANTTS_WGRAD_ATOMICS=$cap timeout 200 python scripts/conv_shape_bench.py 32 > gpurun_out/r2h_conv_shapes_cap$cap.log 2>&1
  echo "cap=$cap: $(grep 'conv launches total' gpurun_out/r2h_conv_shapes_cap$cap.log)"
done
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2h_prof -o sam -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-hifigan --no-cpu-baseline --no-fp32 --no-inference > $GRAFT_REPO_ROOT/gpurun_out/r2h_rocprof.log 2>&1 )
f=$(find gpurun_out/r2h_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -70 "$f" > gpurun_out/r2h_sambert_kernel_stats_top.csv && cut -c1-140 gpurun_out/r2h_sambert_kernel_stats_top.csv | head -28
rm -rf gpurun_out/r2h_prof
grep -o '"ms_per_step": [0-9.]*' gpurun_out/r2h_rocprof.log | head -2
