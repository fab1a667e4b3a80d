#!/bin/bash
# round-3 visit P: what each launch family costs on the SAM-BERT step's critical path (entry points replaced by no-ops; timing only)
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_sambert.py tests/test_gpu_bf16_ops.py -m gpu -x -q 2>&1 | tail -n 5
A="--steps 20 --warmup 5 --no-hifigan --no-cpu-baseline --no-fp32 --no-inference --no-roofline"
run() {
  timeout 300 python scripts/ablate_bench.py "$1" $A > gpurun_out/r3p_tmp.log 2>&1
  python - "$1" <<'PY' | tee -a gpurun_out/r3p_ablation.log
import json, sys
line = [l for l in open("gpurun_out/r3p_tmp.log") if l.startswith("{")]
if not line:
    print("%-60s FAILED: %s" % (sys.argv[1] or "(nothing ablated)", open("gpurun_out/r3p_tmp.log").read()[-300:].replace("\n", " | ")))
else:
    d = json.loads(line[-1])
    print("%-60s step %.3f ms   forward %.3f ms   (%s)" % (sys.argv[1] or "(nothing ablated)", d["ms_per_step"], (d["roofline"] or {}).get("forward_ms", -1), d["config"]["launch"]))
PY
}
rm -f gpurun_out/r3p_ablation.log
run ""
run "kantts_ln128_fwd,kantts_ln128_bwd"
run "kantts_attn_fwd,kantts_attn_bwd"
run "kantts_lstm_fwd,kantts_lstm_bwd"
run "kantts_dropout2_add,kantts_relu_gate_bf16"
run "kantts_fsmn_dwconv_fwd,kantts_fsmn_dwconv_bwd"
run "kantts_ffn_pair"
run "kantts_bgemm_tn,kantts_bgemm_tn_grouped"
run "kantts_bgemm_nt"
run "kantts_adam_step"
