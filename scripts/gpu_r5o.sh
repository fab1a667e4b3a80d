#!/bin/bash
# Round 5: critical-path cost of kernel families of the captured SAM-BERT step (entry points replaced by no-ops; timing
# only, scripts/ablate_bench.py) -- the profiler serialises concurrent graph branches, this does not.
T=${1:-r5o}
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
ARGS="--no-hifigan --no-inference --no-cpu-baseline --no-fp32 --no-roofline --steps 30"
run() {
  timeout 200 python scripts/ablate_bench.py "$2" $ARGS > gpurun_out/${T}_abl.json 2> gpurun_out/${T}_abl.err
  python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/r5o_abl.json").read().strip().splitlines()[-1])
    print("%-28s ms_per_step %.3f forward_ms %.3f" % (sys.argv[1], d["ms_per_step"], d["roofline"].get("forward_ms") or -1))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
run baseline ""
run lstm kantts_lstm_fwd,kantts_lstm_bwd
run fsmn kantts_fsmn_dwconv_fwd,kantts_fsmn_dwconv_bwd,kantts_fsmn_dwconv_bwd_ws,kantts_dropout2_add
run wgrad_tn kantts_bgemm_tn,kantts_bgemm_tn_grouped,kantts_rows_sum_many
run bgemm_nt kantts_bgemm_nt
run lnbwd kantts_bgemm_nt_lnbwd
run pnca_fwd kantts_pnca_block_fwd
run pnca_bwd kantts_pnca_block_bwd
run pnca_attn_bwd kantts_pnca_attn_bwd
run enc_attn kantts_attn_fwd,kantts_attn_bwd
run ffn_pair kantts_ffn_pair
run ln128 kantts_ln128_fwd,kantts_ln128_bwd,kantts_ln128_bwd_rows
run embed_lr kantts_embed_sum_fwd,kantts_embed_sum_bwd,kantts_lr_gather_fwd,kantts_lr_gather_bwd,kantts_lr_index,kantts_teacher_plan
run loss_opt kantts_masked_l1_many,kantts_scale_many,kantts_sumsq_det,kantts_adam_step
run images kantts_cast_f32_bf16,kantts_tapmajor_bf16,kantts_fragmajor_bf16
run gemm_fp32 kantts_gemm_seg_launch
run conv_c1 kantts_conv_c1_launch
run baseline2 ""
