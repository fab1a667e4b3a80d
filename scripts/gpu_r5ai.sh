#!/bin/bash
# Finer critical-path ablation of the embedding / length-regulator family (scripts/ablate_bench.py: timing only).
T=${1:-r5ai}
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
ARGS="--no-hifigan --no-inference --no-cpu-baseline --no-fp32 --no-roofline --steps 30"
run() {
  timeout 200 python scripts/ablate_bench.py "$2" $ARGS > gpurun_out/${T}_abl.json 2> gpurun_out/${T}_abl.err
  python - "$1" $T <<'PY'
import json, sys
try:
    for l in open("gpurun_out/%s_abl.json" % sys.argv[2]):
        if l.startswith("{"): d = json.loads(l)
    print("%-28s ms_per_step %.3f forward_ms %.3f" % (sys.argv[1], d["ms_per_step"], d["roofline"].get("forward_ms") or -1))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
run baseline ""
run teacher_plan kantts_teacher_plan
run lr_index kantts_lr_index
run lr_gather kantts_lr_gather_fwd,kantts_lr_gather_bwd
run embed_fwd kantts_embed_sum_fwd
run embed_bwd kantts_embed_sum_bwd
run conv_c1 kantts_conv_c1_launch
run dropout2 kantts_dropout2_add
run fir kantts_fsmn_dwconv_fwd,kantts_fsmn_dwconv_bwd,kantts_fsmn_dwconv_bwd_ws
run baseline2 ""
