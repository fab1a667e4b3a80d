"""The kernel SOURCES on the CPU (no GPU needed): kan-tts_amd/csrc/*.hip compiled for the host against tests/hipemu (every
thread a fibre, every wave-level operation -- shuffles, DPP, ballots, MFMA, the transposing LDS read, direct-to-LDS loads --
a rendezvous of the wave's 64 fibres) and loaded by the product's own ctypes loader in place of libkantts_hip.so.

Nothing new is asserted here: the cases below ARE the suite's emulated-ABI tests and the op-level GPU tests, re-collected
with ``conftest._emulate`` / ``run_both`` pointed at that library.  Where the original compares the product's host logic
(under the numpy model of the C ABI) with the oracle or a reference-recorded golden, this module compares what the kernel
source computes with the same oracle / golden; where the original compares GPU and numpy model per entry point, this one
compares kernel source and numpy model.  What it cannot see: anything about timing, and hardware behaviour the stand-in
does not model (tests/hipemu/README.md)."""
import functools
import importlib
import os

import pytest

import util

pytestmark = pytest.mark.skipif(not os.path.exists(os.environ.get("HIPEMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")),
                                reason="the host build of the kernel sources needs the ROCm clang")
FULL = os.environ.get("KANTTS_HOSTSIM_FULL", "") not in ("", "0")
full_only = pytest.mark.skipif(not FULL, reason="minutes of CPU time: set KANTTS_HOSTSIM_FULL=1")

# ---- emulated-ABI tests re-collected here (their fixtures come along)
from test_host_logic_emulated import (  # noqa: F401
    test_tiny_sambert_forward_backward_matches_oracle,
    test_dropout_backward_consistent_with_forward_mask,
    test_dropout2_add_and_fsmn_encoder_training_path,
    test_row_mask_handed_to_the_consuming_layernorm_changes_no_gradient,
    test_arena_adam_matches_torch_adam,
    test_arena_adam_with_lr_and_step_count_in_device_memory,
    test_free_running_inference_matches_oracle,
    test_layernorm_backward_in_the_consumers_input_gradient_launch,
    test_relu_gate_of_a_hidden_gradient_in_its_producers_epilogue,
)
from test_ops_sweep import (  # noqa: F401
    test_fused_linear_modes,
    test_attention_ragged_lengths_and_bands,
    test_attention_longer_than_a_workgroup,
    test_lstm_uni_and_bidirectional_with_lengths,
    test_fsmn_memory_length_regulator_embedding_and_masked_l1,
    test_elementwise_losses_weight_norm_sin_add,
    test_fused_sambert_loss_equals_the_two_criteria,
    test_mean_of_branch_outputs_in_one_launch,
    test_gan_criteria_in_one_launch_each,
    test_element_losses_beyond_one_launch_and_empty_terms,
)
from test_bf16_path_emulated import (  # noqa: F401
    bf16_mode,
    test_linear_modes_bf16_emulated,
    test_conv_mode_and_ffn_bf16_emulated,
    test_tiny_sambert_bf16_mode_emulated_close_to_oracle,
    test_deferred_grouped_weight_gradients_equal_immediate_ones,
    test_ffn_pair_equals_the_two_launch_form,
    test_shared_input_linears_equal_separate_ones,
)
from test_conv_sweep import test_conv_cl_random_configurations, test_conv_transpose_cl_random_configurations  # noqa: F401
from test_cconv import (  # noqa: F401
    bf16_all_sizes,
    test_cconv_path_random_configurations_emulated,
    test_cconv_transposed_random_configurations_emulated,
    test_image_handover_is_bit_identical_emulated,
    test_fused_residual_stack_equals_the_chain_emulated,
)
from test_mas import (  # noqa: F401
    test_mas_dp_emulation_is_bit_exact_vs_reference,
    test_align_attention_host_logic,
    test_sambert_mas_host_logic_matches_reference_fixture,
)
from test_melspec import (  # noqa: F401
    test_melspec_host_logic_emulated,
    test_melspec_backward_emulated,
    test_dsp_melspectrogram_emulated,
    _register_form_cases,
    _wide_filterbank_case,
    _several_pairs_per_wave_case,
    test_mel_layouts_agree_emulated,
)


def test_register_resident_fft_form_kernel_source(emulated_cabi):
    """melspec_reg_kernel (one wave per frame, frame in registers, three wave-private LDS exchanges) compiled for the host:
    against the oracle and against the radix-2 kernel, n_fft 1024 and 2048 (tests/test_melspec.py)."""
    _register_form_cases("cpu")


def test_register_form_with_a_filterbank_beyond_its_chunk_table_kernel_source(emulated_cabi):
    """More than 256 chunks of 8 bins: melspec_reg_kernel's per-channel loop over the global weights (a branch no shipped
    configuration reaches; CPU only until it has been on a device)."""
    _wide_filterbank_case("cpu")


def test_register_form_walks_several_pairs_of_frames_per_wave_kernel_source(emulated_cabi):
    """The persistent-grid loop of melspec_reg_kernel (prefetch beside the filterbank, reuse of the exchange cells): found
    untested by the line-coverage run of this file (scripts/kernel_coverage.py) -- the benchmark's 67 584-frame launch is
    the only place the device takes it."""
    _several_pairs_per_wave_case("cpu")


from test_hifigan import (  # noqa: F401
    test_conv_variants_emulated_match_torch,
    test_conv_win_emulated_matches_torch,
    test_multiscale_discriminator_with_average_pooling_matches_the_reference_fixture,
    test_generator_with_relu_activation_matches_the_reference_fixture,
    test_weight_norm_table_images_equal_the_per_layer_kernel_emulated,
    test_weight_norm_table_backward_equals_per_layer_emulated,
    test_one_channel_weight_gradient_with_a_persistent_grid_emulated,
    test_one_channel_layer_hands_over_its_bf16_image_emulated,
    test_one_output_channel_convolution_emulated,
)
from test_hifigan_nsf import test_nsf_generator_host_logic_matches_reference_fixture  # noqa: F401
from test_sambert_se import test_sambert_se_host_logic_matches_reference_fixture  # noqa: F401
from test_multiband import test_pqmf_emulated, test_multispec_emulated  # noqa: F401
from test_edge_cases import test_sambert_single_and_one_token_utterances_emulated  # noqa: F401
from test_device_batching import (  # noqa: F401
    test_device_voc_batches_equal_host_collate_emulated,
    test_device_am_batches_equal_host_collate_emulated,
)


@pytest.fixture(autouse=True)
def kernel_source_instead_of_the_numpy_model(monkeypatch):
    """Everything that would install oracle/cabi_numpy.EmulatedLib (the ``emulated_cabi`` fixture, ``util.emulation()``)
    installs the host build of the kernel sources instead."""
    import conftest

    monkeypatch.setattr(conftest, "_emulate", util.install_kernel_source)
    nan_fill = os.environ.get("KANTTS_HOSTSIM_NANFILL", "") not in ("", "0")
    if nan_fill:  # torch.empty() then returns NaN / max-int: an output cell no kernel writes, or an input read before it is
        import torch  # written, shows up in the comparisons (the audit described in tests/hipemu/README.md)

        old = (torch.are_deterministic_algorithms_enabled(), torch.utils.deterministic.fill_uninitialized_memory)
        torch.use_deterministic_algorithms(True, warn_only=True)
        torch.utils.deterministic.fill_uninitialized_memory = True
    yield
    if nan_fill:
        torch.use_deterministic_algorithms(old[0], warn_only=True)
        torch.utils.deterministic.fill_uninitialized_memory = old[1]


@pytest.mark.parametrize("B,seed", [(1, 77), pytest.param(3, 5, marks=full_only)])
def test_fused_decode_step_on_the_kernel_source(B, seed):
    importlib.import_module("test_decode_graph").test_fused_decode_step_matches_python_loop_emulated(B, seed)


# ---- training steps on the kernel source
@full_only
def test_six_training_steps_retrace_the_reference_loss_curve_on_the_kernel_source():
    """tests/golden/sambert_curve.pt (recorded from the reference's trainer): the six losses to 2e-4, as in the GPU test
    (parameter checksums not compared, for the reason given at test_trainer.test_sambert_loss_curve_matches_reference_gpu)."""
    import test_trainer

    with util.kernel_source_on_cpu():
        test_trainer._curve("cpu", param_tol=None)


@full_only
@pytest.mark.parametrize("name", ["test_hifigan_host_logic_emulated", "test_gan_step_losses_emulated"])
def test_hifigan_generator_discriminators_and_gan_step_on_the_kernel_source(name):
    getattr(importlib.import_module("test_hifigan"), name)()


@pytest.mark.parametrize("key", ["mrstft", "mrstft_subband"])
def test_multi_resolution_stft_loss_and_its_gradient_on_the_kernel_source(key):
    """tests/golden/multiband.pt (reference values); gradient tolerance of the device test (two fp32 FFTs)."""
    with util.kernel_source_on_cpu():
        importlib.import_module("test_multiband")._mrstft("cpu", key, grad_tol=5e-3)


@full_only
def test_vocoder_edge_cases_on_the_kernel_source():
    importlib.import_module("test_edge_cases").test_vocoder_minimal_and_odd_lengths_emulated()


# ---- the op-level GPU tests with the kernel source in the place of the device (second leg: the numpy model of the ABI)
def _op_test(module, name, precision="fp32", **kw):
    import kantts._hip as hip

    m = importlib.import_module(module)
    old_run, old_prec = m.run_both, hip.get_precision()
    m.run_both = functools.partial(util.run_both, device="hostsim")
    hip.set_precision(precision)
    try:
        getattr(m, name)(**kw)
    finally:
        m.run_both = old_run
        hip.set_precision(old_prec)


@pytest.mark.parametrize("prec,tol", [("fp32", 2e-5), ("ref", 2e-5), ("bf16", 3e-2)])
@pytest.mark.parametrize("M,N,K", [(70, 50, 45), (256, 384, 128), (33, 1, 256), (130, 240, 80)])
def test_op_linear_fwd_bwd(prec, tol, M, N, K):
    _op_test("test_gpu_ops", "test_linear_fwd_bwd", prec=prec, tol=tol, M=M, N=N, K=K)


@pytest.mark.parametrize("name,kw", [
    ("test_layernorm", dict(M=37, C=128)), ("test_layernorm", dict(M=2048, C=512)), ("test_layernorm", dict(M=5, C=80)),
    ("test_self_attention", dict(drop=0.0)), ("test_self_attention", dict(drop=0.1)),
    ("test_pnca_attention", dict(bw=0)), ("test_pnca_attention", dict(bw=3)), ("test_pnca_attention", dict(bw=50)),
    ("test_attention_long_sequence_fallback_kernels", {}),
    ("test_lstm_uni_bi_and_concat", {}), ("test_lstm_bf16_recurrence_close_to_fp32", {}),
    ("test_fsmn_memory", dict(C=128, K=41, lp=20, T=70, B=3)), ("test_fsmn_memory", dict(C=256, K=41, lp=37, T=70, B=3)),
    ("test_fsmn_memory", dict(C=80, K=41, lp=40, T=300, B=3)), ("test_fsmn_memory", dict(C=64, K=41, lp=20, T=520, B=8)),
    ("test_dropout2_add_kernel", {}),
], ids=lambda v: v if isinstance(v, str) else "-".join("%s" % x for x in v.values()))
def test_op_fp32_mode(name, kw):
    _op_test("test_gpu_ops", name, **kw)


@pytest.mark.parametrize("name,kw", [
    ("test_linear_plain", dict(M=6528, K=128, N=384, x_bf16=True, out_bf16=False)),
    ("test_linear_plain", dict(M=1000, K=160, N=256, x_bf16=False, out_bf16=True)),
    ("test_linear_plain", dict(M=77, K=80, N=240, x_bf16=False, out_bf16=False)),
    ("test_linear_plain", dict(M=33, K=8, N=8, x_bf16=False, out_bf16=False)),
    ("test_linear_epilogues_and_modes", {}),
    ("test_conv_mode_and_fused_ffn", dict(k1=3)), ("test_conv_mode_and_fused_ffn", dict(k1=1)),
    pytest.param("test_ffn_pair_kernel_forward_and_backward_forms", dict(M=6528, T=204, F=1024, KT=1),
                 marks=full_only),  # the benchmarked block
    ("test_ffn_pair_kernel_forward_and_backward_forms", dict(M=111, T=37, F=1024, KT=1)),
    ("test_ffn_pair_kernel_forward_and_backward_forms", dict(M=95, T=19, F=1024, KT=3)),
    ("test_ffn_pair_kernel_forward_and_backward_forms", dict(M=32, T=32, F=1024, KT=5)),
    ("test_layer_norm128", dict(M=6528)), ("test_layer_norm128", dict(M=100)), ("test_layer_norm128", dict(M=5)),
    ("test_layernorm_backward_as_the_epilogue_of_the_input_gradient", dict(M=6528, a_f32=False, with_res=True, with_rows=True)),
    ("test_layernorm_backward_as_the_epilogue_of_the_input_gradient", dict(M=100, a_f32=True, with_res=True, with_rows=False)),
    ("test_layernorm_backward_as_the_epilogue_of_the_input_gradient", dict(M=37, a_f32=False, with_res=False, with_rows=True)),
    ("test_layernorm_backward_as_the_epilogue_of_the_input_gradient", dict(M=5, a_f32=True, with_res=False, with_rows=False)),
], ids=lambda v: v if isinstance(v, str) else "-".join("%s" % x for x in v.values()))
def test_op_bf16_mode(name, kw):
    _op_test("test_gpu_bf16_ops", name, precision="bf16", **kw)


# ---- entry points no emulated-ABI test reaches with host tensors: the GPU tests' arithmetic, on the kernel source
@pytest.mark.parametrize("Cin,Cout,s,T", [(64, 32, 2, 70), (128, 64, 2, 37), (256, 128, 8, 19)])
def test_upsampling_stream_kernel_and_sin_add_image(Cin, Cout, s, T):
    """kantts_sinadd_lrelu_fwd + kantts_upsample_stream (narrow layers) / the 2-tap polyphase cconv form (wide layers)
    against torch's conv_transpose1d on the same bf16-rounded operands (tests/test_hifigan.py::
    test_upsample_streaming_kernels_gpu with host tensors)."""
    import torch
    import torch.nn.functional as F

    import kantts._hip as hip
    from kantts._hip import ops
    from util import assert_close, rel_l2

    with util.kernel_source_on_cpu():
        hip.set_precision("bf16")
        try:
            g = torch.Generator().manual_seed(Cin + T)
            B = 2
            x = torch.randn(B, T, Cin, generator=g)
            w = torch.randn(Cin, Cout, 2 * s, generator=g) * (1.0 / (2 * Cin) ** 0.5)
            b = torch.randn(Cout, generator=g)
            res = torch.randn(B, T * s, Cout, generator=g)
            h, act = ops.sin_add(x, act_slope=0.1)
            assert_close(h, torch.sin(x) + x, 1e-6, what="sin_add")
            a = F.leaky_relu(torch.sin(x) + x, 0.1).to(torch.bfloat16)
            assert float((act.float() - a.float()).abs().max()) <= 2e-2
            wq = w.to(torch.bfloat16).float()
            ref = F.conv_transpose1d(act.float().transpose(1, 2), wq, b, stride=s)[:, :, :T * s].transpose(1, 2)
            y = ops.upsample_forward(act, w, b, s, res=res)
            assert y is not None and y.dtype == torch.float32
            assert rel_l2(y, ref + res) <= 2e-5
            if Cin <= 128:
                y2 = ops.upsample_forward(act, w, b, s, out_bf16=True)
                assert y2.dtype == torch.bfloat16 and rel_l2(y2.float(), ref) <= 4e-3
                rb = res.to(torch.bfloat16)  # bf16 residual added before the bf16 store (the all-bf16 inference chain)
                y4 = ops.upsample_forward(act, w, b, s, res=rb, out_bf16=True)
                assert y4.dtype == torch.bfloat16 and rel_l2(y4.float(), ref + rb.float()) <= 4e-3
                hb = h.to(torch.bfloat16)
                a3 = F.leaky_relu(hb.float(), 0.1).to(torch.bfloat16).float()
                ref3 = F.conv_transpose1d(a3.transpose(1, 2), wq, b, stride=s)[:, :, :T * s].transpose(1, 2)
                y3 = ops.upsample_forward(hb, w, b, s, out_bf16=True, in_slope=0.1)
                assert rel_l2(y3.float(), ref3) <= 4e-3
        finally:
            hip.set_precision("fp32")


def test_sum_of_squares_both_forms_and_the_legacy_layernorm_backward_entry():
    import torch

    from kantts._hip import ops

    with util.kernel_source_on_cpu() as lib:
        assert lib.kantts_abi_version() == 1 and lib.kantts_target_arch() == b"gfx950"
        x = torch.randn(100003, generator=torch.Generator().manual_seed(0))
        want = float((x.double() ** 2).sum())
        out = torch.zeros(1)
        ops.sumsq_into(x, out)
        assert abs(float(out) - want) <= 1e-5 * want
        out2, ws = torch.zeros(1), torch.zeros(2048)
        ops.sumsq_into(x, out2, workspace=ws)
        assert abs(float(out2) - want) <= 1e-5 * want
        # kantts_ln128_bwd (kept for ABI stability) = kantts_ln128_bwd_rows without a row mask
        M = 37
        g = torch.Generator().manual_seed(1)
        xx, dy, gam = torch.randn(M, 128, generator=g), torch.randn(M, 128, generator=g), torch.rand(128, generator=g) + 0.5
        mean = xx.mean(-1).contiguous()
        rstd = (xx.var(-1, unbiased=False) + 1e-6).rsqrt().contiguous()
        outs = []
        for entry in ("kantts_ln128_bwd", "kantts_ln128_bwd_rows"):
            dx, dg, db = torch.empty_like(xx), torch.zeros(128), torch.zeros(128)
            args = [dy.data_ptr(), 0, xx.data_ptr(), gam.data_ptr(), mean.data_ptr(), rstd.data_ptr(), None, dx.data_ptr(),
                    dg.data_ptr(), db.data_ptr()]
            tail = [None, M, None] if entry.endswith("_rows") else [M, None]
            assert getattr(lib, entry)(*(args + tail)) == 0
            outs.append((dx, dg, db))
        for a, b in zip(*outs):
            assert torch.equal(a, b)
        xh = (xx - mean[:, None]) * rstd[:, None]
        gy = dy * gam
        want_dx = (gy - gy.mean(-1, keepdim=True) - xh * (gy * xh).mean(-1, keepdim=True)) * rstd[:, None]
        assert float((outs[0][0] - want_dx).abs().max()) <= 1e-5
        assert float((outs[0][1] - (dy * xh).sum(0)).abs().max()) <= 1e-4 and float((outs[0][2] - dy.sum(0)).abs().max()) <= 1e-4


def test_arena_images_built_by_the_image_kernels():
    """kantts_cast_f32_bf16 / kantts_tapmajor_bf16 / kantts_fragmajor_bf16 (ParamArena(bf16_shadow=True)): the images equal
    the ones built from the parameters with tensor operations and follow an update of the master."""
    import torch
    import torch.nn as nn

    from kantts._hip.ops_bf16 import ffn_frag_weights, frag_major
    from kantts.models.sambert import PositionwiseConvFeedForward
    from kantts.train.optim import ParamArena

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.a = PositionwiseConvFeedForward(128, 1024, (1, 1))
            self.b = PositionwiseConvFeedForward(128, 1024, (3, 1))
            self.c = PositionwiseConvFeedForward(64, 256, (3, 1))

        def forward(self, x):
            return x

    import kantts._hip as hip

    with util.kernel_source_on_cpu():
        hip.set_precision("bf16")  # the forward pre-hook refreshes the images in bf16 mode only
        torch.manual_seed(3)
        net = Net()
        arena = ParamArena(net, bf16_shadow=True)

        def check():
            for blk, kt in ((net.a, 1), (net.b, 3)):
                w1, w2 = blk.w_1.weight, blk.w_2.weight
                f1, f2, t2, t1 = ffn_frag_weights(w1, w2)
                assert torch.equal(f1, frag_major(w1.detach().permute(2, 0, 1).reshape(kt * 1024, 128)))
                assert torch.equal(f2, frag_major(w2.detach().reshape(128, 1024)))
                assert torch.equal(t2, frag_major(w2.detach().reshape(128, 1024).t()))
                assert torch.equal(t1, frag_major(w1.detach().permute(2, 1, 0).reshape(kt * 128, 1024)))
            for p in net.parameters():
                assert torch.equal(p._kantts_bf16, p.detach().to(torch.bfloat16))
                if p.dim() == 3 and p.shape[2] > 1:
                    assert torch.equal(p._kantts_bf16_tap, p.detach().permute(2, 0, 1).to(torch.bfloat16))

        check()
        with torch.no_grad():
            arena.flat.mul_(1.5).add_(0.01)
        try:
            net(torch.zeros(1))  # forward pre-hook: refresh
            check()
        finally:
            hip.set_precision("fp32")


# ---- the benchmarked architecture (not the tiny one) on the kernel source, at a batch the CPU finishes in a minute
_FULL_CONFIG_BOUNDS = {  # the bounds of tests/test_bench_config_parity.py (device, B = 32) ...
    "fp32": dict(mel_mean=1e-5, mel_max=5e-4, loss=1e-4, grad_global=2e-3, grad_worst=2e-2),
    # ... except the bf16 gradient as a whole: 24 tokens instead of 2048 average the operand rounding less (2.5 % measured
    # here, with and without the LayerNorm-backward epilogues, against 0.6 % at B = 32 on the device)
    "bf16": dict(mel_mean=3e-3, mel_max=2.5e-2, loss=5e-4, grad_global=5e-2, grad_worst=0.2),
}


@pytest.mark.parametrize("mode", ["fp32", "bf16", "bf16-no_ln_bwd_epilogue", "bf16-no_relu_gate_epilogue"])
def test_full_sambert_configuration_matches_oracle_on_the_kernel_source(mode, monkeypatch):
    """BASELINE config 2's model (sambert_16k.yaml zhcn: 8 + 12 blocks of width 128 / 1024, FSMN + LSTM postnet) forward,
    losses and every parameter gradient against oracle/torch_oracle.py -- tests/test_bench_config_parity.py with host
    tensors at B = 2 x T_in = 12 and the same bounds; the last cases without the LayerNorm-backward epilogue of the QKV
    input gradient (on by default) and without the ReLU-gate hand-over (both on by default)."""
    import torch

    import kantts._hip as hip
    import torch_oracle as O
    from kantts._hip import ops_bf16
    from kantts.models.sambert.kantts_sambert import KanTtsSAMBERT
    from kantts.train.loss import MelReconLoss, ProsodyReconLoss

    prec = mode.split("-")[0]
    monkeypatch.setitem(ops_bf16.LNBWD, "on", "no_ln_bwd_epilogue" not in mode)
    monkeypatch.setitem(ops_bf16.RELUGATE, "on", "no_relu_gate_epilogue" not in mode)
    cfg = O.sambert_config(tiny=False)
    cfg = {k: (0.0 if "dropout" in k else v) for k, v in cfg.items()}
    torch.manual_seed(0)
    m = KanTtsSAMBERT(dict(cfg))
    m.eval()
    P = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in m.state_dict().items()}
    batch = O.synthetic_sambert_batch(B=2, T_in=12, seed=1234, min_len=6, dur_hi=6)
    out = O.sambert_forward(P, cfg, **batch)
    L = O.sambert_losses(out, batch["input_lengths"], batch["output_lengths"], batch["mel_targets"])
    L["total"].backward()
    bnd = _FULL_CONFIG_BOUNDS[prec]
    hip.set_precision(prec)
    try:
        with util.kernel_source_on_cpu():
            res = m(**batch)
            mel_, mel = MelReconLoss()(batch["output_lengths"], batch["mel_targets"], res["dec_outputs"], res["postnet_outputs"])
            d, p, e = ProsodyReconLoss()(batch["input_lengths"], res["duration_targets"], res["pitch_targets"],
                                         res["energy_targets"], res["log_duration_predictions"], res["pitch_predictions"],
                                         res["energy_predictions"])
            total = mel_ + mel + d + p + e
            total.backward()
    finally:
        hip.set_precision("fp32")
    assert torch.equal(res["LR_length_rounded"], out["LR_length_rounded"])
    assert res["x_band_width"] == out["x_band_width"] and res["h_band_width"] == out["h_band_width"]
    for k in ("dec_outputs", "postnet_outputs"):
        dd = (res[k].detach() - out[k].detach()).abs()
        assert float(dd.mean()) <= bnd["mel_mean"] and float(dd.max()) <= bnd["mel_max"], (k, float(dd.mean()), float(dd.max()))
    assert abs(float(total.detach()) - float(L["total"].detach())) <= bnd["loss"]
    num = den = worst = 0.0
    wname = ""
    for n, prm in m.named_parameters():
        if prm.requires_grad and P[n].grad is not None:
            assert prm.grad is not None, n
            e2 = float((prm.grad.double() - P[n].grad.double()).pow(2).sum())
            r2 = float(P[n].grad.double().pow(2).sum())
            num, den = num + e2, den + r2
            if (e2 / (r2 + 1e-60)) ** 0.5 > worst:
                worst, wname = (e2 / (r2 + 1e-60)) ** 0.5, n
    assert (num / den) ** 0.5 <= bnd["grad_global"] and worst <= bnd["grad_worst"], ((num / den) ** 0.5, worst, wname)


@pytest.mark.parametrize("tile", [64128, 64064, 4128064, 3064064, 4064128])
def test_cconv_round4_tiles_on_the_kernel_source(tile):
    """The tile / ring-depth variants of cconv_kernel added for the upsampling stages (64 x 128, 64 x 64, four-stage
    128 x 64; args->tile = stages * 1000000 + tile code) against the numpy model of the entry point: a 2-tap polyphase
    contraction (the transposed-convolution form) and a 7-tap convolution with ragged channel counts."""
    import torch

    import kantts._hip as hip

    g = torch.Generator().manual_seed(tile)
    # (B, Tsrc, Tdst, Cin, Cout, K, in_add, in_kstep)
    for si, (B, Ts, Td, Cin, Cout, K, add, kstep) in enumerate([(3, 37, 37, 64, 256, 2, 0, -1), (2, 50, 50, 80, 136, 7, -6, 1)]):
        x = torch.randn(B, Ts, Cin, generator=g).to(torch.bfloat16)
        w = (torch.randn(K, Cout, Cin, generator=g) / (K * Cin) ** 0.5).to(torch.bfloat16)
        bias = torch.randn(Cout, generator=g)
        outs = []
        for src in ("kernel", "model"):
            o32 = torch.full((B, Td, Cout), float("nan"))
            obf = torch.zeros((B, Td, Cout), dtype=torch.bfloat16)
            ctx = util.kernel_source_on_cpu() if src == "kernel" else util.emulation()
            import conftest

            with ctx if src == "kernel" else _numpy_model():
                assert hip.cconv(x, w, out=o32, out_bf=obf, B=B, Tsrc=Ts, Tdst=Td, groups=1, CR=Cin, NG=Cout, K=K, in_mul=1,
                                 in_add=add, in_kstep=kstep, in_div=1, phases=1, bias=bias, out_leaky=0.1,
                                 tile=tile if src == "kernel" else 0)
            outs.append((o32, obf.float()))
        (a32, abf), (c32, cbf) = outs
        assert not torch.isnan(a32).any(), (tile, si)
        assert float((a32 - c32).abs().max()) <= 2e-5 * max(1.0, float(c32.abs().max())), (tile, si)
        assert float((abf - cbf).abs().max()) <= 1e-2 * max(1.0, float(cbf.abs().max())), (tile, si)


import contextlib  # noqa: E402


@contextlib.contextmanager
def _numpy_model():
    """oracle/cabi_numpy.EmulatedLib, whatever the autouse fixture of this file installed for conftest._emulate."""
    import cabi_numpy
    import kantts._hip as hip
    import kantts._hip.ops as ops
    import kantts._hip.ops_bf16 as ops_bf16
    import kantts.utils.audio_torch as audio_torch

    emu = cabi_numpy.EmulatedLib()

    def ptr(t, dtype=None):
        if t is None:
            return None
        if dtype is not None and t.dtype != dtype:
            raise TypeError("expected %s, got %s" % (dtype, t.dtype))
        return t.data_ptr()

    p = util._Patch()
    try:
        for mod in (hip, ops, ops_bf16, audio_torch):
            p.setattr(mod, "lib", lambda: emu)
            p.setattr(mod, "ptr", ptr)
            p.setattr(mod, "stream", lambda: None)
        yield emu
    finally:
        p.undo()


@pytest.mark.parametrize("tile", ["64128", "128256", "64129", "128257"])
def test_weight_gradient_contraction_with_both_output_tiles(tile, bf16_mode, monkeypatch):
    """bgemm_tn_kernel<.., BN, BK>: the 64 x 128 tile of round 2 and the 128 x 256 tile of round 4 (KANTTS_TN_TILE forces
    one), each under the XCD-aware workgroup mapping of round 6 (1-D grid, padded last round of groups) and under the 3-D
    grid it replaced (code + 1), on the linear-layer cases (ragged channel counts, fp32 / bf16 operands, taps, dropout on
    the A operand) and on the deferred / grouped launches."""
    monkeypatch.setenv("KANTTS_TN_TILE", tile)
    import test_bf16_path_emulated as T

    T.test_linear_modes_bf16_emulated(bf16_mode)
    T.test_deferred_grouped_weight_gradients_equal_immediate_ones(bf16_mode)


@pytest.mark.parametrize("tile,slices", [("64128", 13), ("64128", 9), ("128256", 16), ("64129", 13)])
def test_weight_gradient_contraction_in_xcd_order_on_the_kernel_source(tile, slices, monkeypatch):
    """bgemm_tn_kernel on the XCD-aware 1-D grid of round 6: 13 token slices = 13 groups of output tiles dealt to 8 XCDs (a
    padded second round), 16 = two full rounds, 9 = below 3/4 of the places of its last round -> the launcher keeps the 3-D
    grid, and "64129" = the 3-D grid forced; ragged channel counts, three taps, against float64."""
    import torch

    import kantts._hip as hip

    monkeypatch.setenv("KANTTS_TN_TILE", tile)
    monkeypatch.setenv("KANTTS_TN_SLICES", str(slices))
    g = torch.Generator().manual_seed(slices)
    T, B, N, K, taps = 300, 4, 72, 136, 3
    M = B * T
    a = torch.randn(M, N, generator=g)
    b = torch.randn(M, K, generator=g).to(torch.bfloat16)
    c0 = torch.randn(N, K * taps, generator=g)
    c, db = c0.clone(), torch.zeros(N)
    with util.kernel_source_on_cpu():
        hip._launch_tuning[0] = None
        try:
            assert hip.bgemm_tn(a, N, b, K, M, N, K, c, K * taps, taps, c_ts=1, T=T, ntaps=taps, shift0=-1, shift_step=1, db=db)
        finally:
            monkeypatch.delenv("KANTTS_TN_TILE")
            monkeypatch.delenv("KANTTS_TN_SLICES")
            hip.apply_launch_tuning()
    ab = a.to(torch.bfloat16).double().view(B, T, N)
    bb = b.double().view(B, T, K)
    ref = c0.double().clone().view(N, K, taps)
    for tap in range(taps):
        sh = tap - 1
        shifted = torch.zeros_like(bb)
        if sh < 0:
            shifted[:, -sh:] = bb[:, :T + sh]
        elif sh > 0:
            shifted[:, :T - sh] = bb[:, sh:]
        else:
            shifted = bb
        ref[:, :, tap] += torch.einsum("btn,btk->nk", ab, shifted)
    assert float((c.double().view(N, K, taps) - ref).abs().max()) <= 2e-3 * M ** 0.5
    assert float((db.double() - ab.sum((0, 1))).abs().max()) <= 2e-3 * M ** 0.5


@pytest.mark.parametrize("To,Ti", [(90, 64), (75, 65), (70, 128), (60, 200), (33, 256), (40, 300)])
def test_monotonic_alignment_search_every_kernel_variant_on_the_kernel_source(To, Ti):
    """mas_wave_kernel with 1 / 2 / 4 ballot words per row, the exact 64-column boundaries, and mas_block_kernel (more than
    256 symbols) -- bit-exact against the numpy model of the ABI, ties and -inf scores included (the device twin:
    tests/test_mas.py::test_mas_dp_gpu_every_kernel_variant_vs_oracle)."""
    import torch

    from kantts._hip import ops

    g = torch.Generator().manual_seed(To * 1000 + Ti)
    B = 3
    attn = torch.softmax(torch.randn(B, 1, To, Ti, generator=g) * 2, dim=3)
    attn[1] = torch.round(attn[1] * 64) / 64  # coarse grid: ties and zeros (-inf scores)
    in_lens = torch.tensor([Ti, max(1, Ti - 7), max(1, Ti // 2)])
    out_lens = torch.tensor([To, To - 5, max(1, To // 3)])
    with util.kernel_source_on_cpu():
        hard = ops.mas_width1(attn, in_lens, out_lens)
    with _numpy_model():
        ref = ops.mas_width1(attn, in_lens, out_lens)
    assert torch.equal(hard, ref)
    if To >= 2 * Ti:
        assert torch.equal(hard.sum((1, 2, 3)).long(), out_lens)


@pytest.mark.parametrize("CR,NG,T,G,mul", [(16, 48, 200, 2, 1), (16, 48, 70, 1, 1), (32, 32, 200, 1, 2), (24, 32, 70, 4, 1),
                                           (64, 64, 200, 2, 1), (48, 64, 70, 1, 1), (64, 32, 200, 1, 1), (48, 24, 70, 2, 2),
                                           # 2 x 35 row tiles of 256 (x 2 groups): many workgroups, long sequences
                                           (32, 32, 8900, 1, 1), (16, 48, 4400, 2, 1)])
def test_cconv_window_form_every_variant_on_the_kernel_source(CR, NG, T, G, mul):
    """cconv_narrow_kernel<256 | 128 rows, 64 | 32 outputs per group, 32 | 64 reduction channels>: the grouped / narrow
    layers of the discriminators (at most 64 input channels per group, no fold, no upsampling), every instantiation the
    launcher picks from (rows >= 192 -> 256-row tiles), strides 1 and 2, against the numpy model of the ABI."""
    import torch

    import kantts._hip as hip

    g = torch.Generator().manual_seed(CR * 1000 + NG + T)
    B, K = 2, 5
    Cin, Cout, Ts = G * CR, G * NG, T * mul + 3
    x = torch.randn(B, Ts, Cin, generator=g).to(torch.bfloat16)
    w = (torch.randn(K, Cout, CR, generator=g) / (K * CR) ** 0.5).to(torch.bfloat16)
    bias = torch.randn(Cout, generator=g)
    outs = []
    for src in ("kernel", "model"):
        o32 = torch.full((B, T, Cout), float("nan"))
        obf = torch.zeros((B, T, Cout), dtype=torch.bfloat16)
        with util.kernel_source_on_cpu() if src == "kernel" else _numpy_model():
            assert hip.cconv(x, w, out=o32, out_bf=obf, B=B, Tsrc=Ts, Tdst=T, groups=G, CR=CR, NG=NG, K=K, in_mul=mul,
                             in_add=-2, in_kstep=1, in_div=1, phases=1, bias=bias, out_leaky=0.1, bf_leaky=0.2, tile=0)
        outs.append((o32, obf.float()))
    (a32, abf), (c32, cbf) = outs
    assert not torch.isnan(a32).any()
    assert float((a32 - c32).abs().max()) <= 2e-5 * max(1.0, float(c32.abs().max()))
    assert float((abf - cbf).abs().max()) <= 1e-2 * max(1.0, float(cbf.abs().max()))


def test_one_channel_layer_scalar_kernels_on_the_kernel_source(monkeypatch):
    """conv_c1_wgrad_kernel / conv_c1_dgrad_kernel without the matrix cores: what a 1 -> 256 layer takes (no shipped
    configuration has one) and what KANTTS_C1_NO_MFMA=1 forces for the A/B runs -- same outputs and gradients as torch."""
    import test_hifigan as T

    from util import assert_close, rel_l2

    def run():
        for case in ((2, 300, 1, 256, 7, 1, 1, 3, 1, None, 0.1, False), (2, 333, 1, 32, 5, 3, 1, 2, 1, None, 0.1, False)):
            y, ref, gy, gr = T._win_case(case, "cpu")
            assert_close(y, ref, 5e-5, what=str(case))
            for a, c in zip(gy, gr):
                assert rel_l2(a, c) < 2e-4, (case, rel_l2(a, c))

    with util.kernel_source_on_cpu():
        run()


@pytest.mark.parametrize("slices", [1, 3, 8, 13])
def test_cconv_weight_gradient_token_slices_on_the_kernel_source(slices):
    """cconv_wgrad_kernel / cconv_wgrad_taps_kernel with the token range cut into slices: partial tiles into the workspace +
    cconv_wgrad_reduce_kernel, or atomics when no workspace is handed over (one slice: read-modify-write) -- 2.6 % of the
    GAN step on the device, device twin tests/test_cconv.py::test_cconv_wgrad_gpu_vs_emulated."""
    import torch

    import kantts._hip as hip

    g = torch.Generator().manual_seed(slices + 11)
    # (B, Tsrc, Tdst, inner, Cin, Cout, groups, K, stride, dil, pad, up)
    shapes = [(2, 70, 70, 1, 64, 64, 1, 3, 1, 1, 1, 1), (3, 61, 21, 5, 64, 256, 1, 5, 3, 1, 2, 1),
              (2, 90, 90, 1, 72, 136, 1, 3, 1, 1, 1, 1), (2, 120, 60, 1, 128, 256, 2, 9, 2, 1, 4, 1),
              (2, 50, 200, 1, 64, 64, 1, 7, 1, 1, 6, 4), (1, 9, 9, 1, 8, 8, 1, 1, 1, 1, 0, 1),
              # enough 64-token steps for >= 8 slices: the XCD-aware 1-D grid of cconv_wgrad_kernel (round 6), with a padded
              # last round of slices at 13 (13 of 16 places taken; below 3/4 the launcher keeps the 3-D grid)
              (2, 400, 400, 1, 136, 72, 1, 3, 1, 1, 1, 1), (2, 380, 380, 1, 256, 256, 2, 3, 1, 2, 2, 1)]
    for si, (B, Ts, Td, P, Cin, Cout, G, K, stride, dil, pad, up) in enumerate(shapes):
        CR, NG = Cin // G, Cout // G
        x = torch.randn(B, Ts, P, Cin, generator=g).to(torch.bfloat16)
        dy = torch.randn(B, Td, P, Cout, generator=g).to(torch.bfloat16)
        base = torch.randn(K, Cout, CR, generator=g)  # the call accumulates
        outs = []
        for src in ("kernel", "model"):
            dw, db = base.clone(), torch.ones(Cout)
            with util.kernel_source_on_cpu() if src == "kernel" else _numpy_model():
                assert hip.cconv_wgrad(x, dy, dw, db, B=B, Tsrc=Ts, Tdst=Td, groups=G, CR=CR, NG=NG, K=K, stride=stride,
                                       dil=dil, pad=pad, inner=P, up=up, slices=slices if src == "kernel" else 0)
            outs.append((dw, db))
        (adw, adb), (cdw, cdb) = outs
        scale = (B * Td * P) ** 0.5
        assert float((adw - cdw).abs().max()) <= 2e-5 * scale, (slices, si)
        assert float((adb - cdb).abs().max()) <= 2e-5 * scale, (slices, si)


@pytest.mark.parametrize("tile,CR,NG,G,T", [(0, 32, 32, 2, 200), (0, 128, 136, 1, 37), (64064, 32, 128, 1, 2112)])
def test_cconv_fp32_gate_epilogue_and_the_tile_remap_on_the_kernel_source(tile, CR, NG, G, T):
    """(i) the epilogue that gates by an fp32 pre-activation (out_gate fp32 + residual), window form and tiled form;
    (ii) 66 tiles of 64 x 64: the blockIdx -> tile remap that keeps the n tiles of a row tile behind one XCD's L2 (taken from
    64 tiles on; 66 is not a multiple of 8) must still cover every tile exactly once."""
    import torch

    import kantts._hip as hip

    g = torch.Generator().manual_seed(T + CR)
    B, K = 1 if T > 1000 else 2, 1 if T > 1000 else 3
    Cin, Cout = G * CR, G * NG
    x = torch.randn(B, T, Cin, generator=g).to(torch.bfloat16)
    w = (torch.randn(K, Cout, CR, generator=g) / (K * CR) ** 0.5).to(torch.bfloat16)
    res, gate = torch.randn(B, T, Cout, generator=g), torch.randn(B, T, Cout, generator=g)
    outs = []
    for src in ("kernel", "model"):
        o32 = torch.full((B, T, Cout), float("nan"))
        obf = torch.zeros((B, T, Cout), dtype=torch.bfloat16)
        with util.kernel_source_on_cpu() if src == "kernel" else _numpy_model():
            assert hip.cconv(x, w, out=o32, out_bf=obf, B=B, Tsrc=T, Tdst=T, groups=G, CR=CR, NG=NG, K=K, in_mul=1,
                             in_add=-(K // 2), in_kstep=1, in_div=1, phases=1, res=res, out_gate=gate, out_gate_slope=0.3,
                             tile=tile if src == "kernel" else 0)
        outs.append((o32, obf.float()))
    (a32, abf), (c32, cbf) = outs
    assert not torch.isnan(a32).any()
    assert float((a32 - c32).abs().max()) <= 2e-5 * max(1.0, float(c32.abs().max()))
    assert float((abf - cbf).abs().max()) <= 1e-2 * max(1.0, float(cbf.abs().max()))


@pytest.mark.parametrize("precision", [0, 1])
def test_lstm_abi_forms_the_host_never_uses_on_the_kernel_source(precision):
    """kantts_lstm_fwd / _bwd with reverse_first = 1 (a lone right-to-left direction) and without the recurrent bias: declared
    in include/kantts_hip.h, never passed by kantts._hip.ops.  Kernel source against the numpy model of the ABI, and the
    reversed single direction against direction 1 of a two-direction call with the same weights."""
    import torch

    import kantts._hip as hip

    g = torch.Generator().manual_seed(41 + precision)
    B, T, H = 3, 11, 128
    G = 4 * H
    lens = torch.tensor([T, 4, 1], dtype=torch.int32)
    gx2 = torch.randn(B, T, 2 * G, generator=g) * 0.5
    whh2 = torch.randn(2, G, H, generator=g) * 0.08
    dout2 = torch.randn(B, T, 2 * H, generator=g)

    def run(gx, whh, bhh, dout, ndir, rev):
        gx, whh, dout = gx.contiguous(), whh.contiguous(), dout.contiguous()
        out = torch.full((B, T, ndir * H), float("nan"))
        gates, cst = torch.zeros(ndir, B, T, G), torch.zeros(ndir, B, T, H)
        dg = torch.zeros(ndir, B, T, G)
        L = hip.lib()
        p = hip.ptr
        assert L.kantts_lstm_fwd(p(gx), p(whh), p(bhh), p(lens), p(out), p(gates), p(cst), B, T, H, ndir, rev, precision,
                                 hip.stream()) == 0
        assert L.kantts_lstm_bwd(p(dout), p(whh), p(lens), p(gates), p(cst), p(dg), B, T, H, ndir, rev, precision,
                                 hip.stream()) == 0
        return out, dg

    res = {}
    for src in ("kernel", "model"):
        with util.kernel_source_on_cpu() if src == "kernel" else _numpy_model():
            both = run(gx2, whh2, None, dout2, 2, 0)
            lone = run(gx2[..., G:], whh2[1:], None, dout2[..., H:], 1, 1)
        res[src] = (both, lone)
        # the lone reversed direction IS direction 1 of the pair (same arithmetic, same order)
        assert torch.equal(lone[0], both[0][..., H:]), src
        valid = (torch.arange(T)[None, :] < lens[:, None])[None, :, :, None]
        assert torch.equal(lone[1] * valid, both[1][1:] * valid), src
    tol = 2e-5 if precision == 0 else 2e-2
    for a, c in zip(res["kernel"][1], res["model"][1]):
        valid = (torch.arange(T)[None, :] < lens[:, None])
        a = a * (valid[..., None] if a.dim() == 3 else valid[None, :, :, None])
        c = c * (valid[..., None] if c.dim() == 3 else valid[None, :, :, None])
        assert float((a - c).abs().max()) <= tol * max(1.0, float(c.abs().max()))


@pytest.mark.parametrize("no_cconv", [False, True])
def test_convolutions_in_bf16_mode_on_the_kernel_source(no_cconv, monkeypatch):
    """tests/test_hifigan.py::test_conv_win_gpu_matches_torch[bf16] with host tensors: bf16 operands, fp32 accumulation,
    device tolerances.  With the channels-last bf16 kernels switched off (KANTTS_NO_CCONV=1) the LDS-window kernels take
    their bf16-operand instantiations (conv_win_kernel<true, ...>, conv_wgrad_kernel<true, ...>) -- what a shape those
    kernels decline falls back to."""
    import kantts._hip as hip
    import test_hifigan as T
    from util import rel_l2

    if no_cconv:
        monkeypatch.setenv("KANTTS_NO_CCONV", "1")
    with util.kernel_source_on_cpu():
        prev = hip.get_precision()
        hip.set_precision("bf16")
        try:
            for case in T._WIN_CASES[2:-1]:
                y, ref, gy, gr = T._win_case(case, "cpu")
                assert float((y - ref).abs().max()) <= 4e-2 * max(1.0, float(ref.abs().max())), case
                for a, c in zip(gy, gr):
                    assert rel_l2(a, c) < 6e-2, case
        finally:
            hip.set_precision(prev)


def test_fast_gemm_with_64_row_tiles_in_a_fresh_process():
    """gemm_fast_kernel<.., 64, ..>: chosen on the device from 512 workgroups on; KANTTS_GEMM_BM forces it, but the
    launcher reads the switch once per process -- the linear-layer cases of this file again, in a child process."""
    import subprocess
    import sys

    env = dict(os.environ, KANTTS_GEMM_BM="64")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-p", "no:cacheprovider", "-k",
                        "test_op_linear_fwd_bwd or test_fused_linear_modes or linear_epilogues"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-2000:] + r.stderr[-1000:]
