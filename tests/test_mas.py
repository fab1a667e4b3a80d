"""MAS alignment path: DP kernel, alignment attention, model wiring, losses, prior, batches.

Pins:
  * tests/golden/mas_dp.pt        -- b_mas of the reference (alignment.py:63-71, plain-Python semantics of its numba code)
                                     on random / all-ties / zero-holed maps: the 0/1 path must be BIT-EXACT;
  * tests/golden/sambert_tiny_mas.pt -- KanTtsSAMBERT(MAS=True) forward + the six losses + backward of the reference.
CPU tests drive the host logic through the emulated C ABI; `-m gpu` tests run the HIP kernels."""
import os

import numpy as np
import pytest
import torch

from util import GOLDEN, assert_close, rel_l2, run_both


def _fix(name):
    return torch.load(os.path.join(GOLDEN, name + ".pt"), weights_only=False)


def _conv_attention_torch(q, k, prior, lens):
    """fp32 torch restatement of ConvAttention.forward after the projections (attention.py:103-125), channels last."""
    d = -0.0005 * ((q[:, :, None, :] - k[:, None, :, :]) ** 2).sum(-1)
    d = d[:, None]
    if prior is not None:
        d = torch.log_softmax(d, dim=3) + torch.log(prior[:, None] + 1e-8)
    logprob = d
    mask = torch.arange(k.shape[1], device=k.device)[None, :] >= lens[:, None]
    soft = torch.softmax(d.masked_fill(mask[:, None, None, :], -float("inf")), dim=3)
    return soft, logprob


def _attn_inputs(B=3, T1=37, T2=11, C=80, seed=5):
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(B, T1, C, generator=g).requires_grad_(True)
    k = torch.randn(B, T2, C, generator=g).requires_grad_(True)
    prior = torch.rand(B, T1, T2, generator=g)
    prior[:, :, -1] = 0.0  # padded prior columns are exactly 0 -> log(1e-8)
    lens = torch.tensor([T2, T2 - 3, 1][:B], dtype=torch.int32)
    return q, k, prior, lens


# ---------------------------------------------------------------------------------------------------------- CPU
def test_mas_dp_emulation_is_bit_exact_vs_reference(emulated_cabi):
    from kantts.models.sambert.alignment import b_mas

    fix = _fix("mas_dp")
    hard = b_mas(fix["attn"], fix["in_lens"], fix["out_lens"])
    assert torch.equal(hard, fix["hard"])
    # every valid mel frame maps to exactly one phoneme, monotonically, ending on the last one
    for b in range(hard.shape[0]):
        To, Ti = int(fix["out_lens"][b]), int(fix["in_lens"][b])
        path = hard[b, 0, :To, :Ti]
        if To >= Ti:  # a monotone path with unit steps covers every phoneme only if there are enough frames
            assert torch.all(path.sum(1) >= 1)
        assert path[To - 1, Ti - 1] == 1 and path[0, 0] == 1
        assert float(hard[b].sum()) == float(path.sum())


@pytest.mark.parametrize("with_prior", [True, False])
def test_align_attention_host_logic(emulated_cabi, with_prior):
    from kantts._hip import ops

    q, k, prior, lens = _attn_inputs()
    pr = prior if with_prior else None
    soft, logprob = ops.align_attention(q, k, pr, lens)
    rs, rl = _conv_attention_torch(q, k, pr, lens.long())
    assert_close(soft.detach(), rs.detach(), 1e-6, what="soft")
    assert_close(logprob.detach(), rl.detach(), 2e-5, what="logprob")
    g = torch.Generator().manual_seed(1)
    c1, c2 = torch.randn(soft.shape, generator=g), torch.randn(soft.shape, generator=g)
    got = torch.autograd.grad((soft * c1).sum() + (logprob * c2).sum(), [q, k])
    ref = torch.autograd.grad((rs * c1).sum() + (rl * c2).sum(), [q, k])
    for a, b, n in zip(got, ref, "qk"):
        assert rel_l2(a, b) < 1e-5, n


def test_beta_binomial_prior_matches_scipy():
    from scipy.stats import betabinom

    from kantts.datasets.batching import beta_binomial_prior_distribution

    for P, M in [(7, 20), (23, 77), (1, 3)]:
        ref = np.array([betabinom(P, i, M + 1 - i).pmf(np.arange(P)) for i in range(1, M + 1)])
        got = beta_binomial_prior_distribution(P, M)
        assert got.dtype == torch.float64 and tuple(got.shape) == (M, P)
        assert np.abs(got.numpy() - ref).max() < 1e-12


def test_mas_collate_layout():
    from kantts.datasets.batching import am_collate, beta_binomial_prior_distribution

    rng = np.random.RandomState(0)
    items = []
    for n_sym, n_mel in [(5, 17), (8, 31)]:
        ling = [rng.randint(0, 5, n_sym) for _ in range(6)]
        items.append((ling, rng.randn(n_mel, 80).astype(np.float32), None, rng.randn(n_mel), rng.randn(n_mel),
                      beta_binomial_prior_distribution(n_sym, n_mel), None, None))
    out = am_collate(items, r=3, pad_ids=[0] * 6)
    assert out["durations"] is None
    assert tuple(out["mel_targets"].shape) == (2, 33, 80)
    assert tuple(out["pitch_contours"].shape) == (2, 33) and tuple(out["attn_priors"].shape) == (2, 33, 8)
    assert out["attn_priors"].dtype == torch.float32
    assert torch.all(out["attn_priors"][0, 17:] == 0) and torch.all(out["attn_priors"][0, :, 5:] == 0)
    assert torch.allclose(out["attn_priors"][1, :31].double(), items[1][5], atol=1e-7)
    assert out["valid_input_lengths"].tolist() == [4, 7]


def _mas_model_run(fix, device):
    from kantts.models.sambert.kantts_sambert import KanTtsSAMBERT
    from kantts.train.loss import (AttentionBinarizationLoss, AttentionCTCLoss, MelReconLoss, ProsodyReconLoss)

    torch.manual_seed(fix["seed_w"])
    m = KanTtsSAMBERT(dict(fix["cfg"]))
    for k, (shape, s, a) in fix["weight_checksums"].items():
        v = m.state_dict()[k]
        assert tuple(v.shape) == shape and abs(float(v.double().sum()) - s) <= 1e-6 * max(1.0, a), k
    m = m.to(device).eval()
    b = {k: (v.clone().to(device) if torch.is_tensor(v) else v) for k, v in fix["batch"].items()}
    res = m(**b)
    mel_, mel = MelReconLoss()(b["output_lengths"], b["mel_targets"], res["dec_outputs"], res["postnet_outputs"])
    d, p, e = ProsodyReconLoss()(b["input_lengths"], res["duration_targets"], res["pitch_targets"], res["energy_targets"],
                                 res["log_duration_predictions"], res["pitch_predictions"], res["energy_predictions"])
    ctc = AttentionCTCLoss()(res["attn_logprob"], b["input_lengths"], b["output_lengths"])
    kl = AttentionBinarizationLoss()(fix["epoch"], res["attn_hard"], res["attn_soft"])
    total = mel_ + mel + d + p + e + ctc + kl
    total.backward()
    losses = dict(mel_loss_=mel_, mel_loss=mel, dur_loss=d, pitch_loss=p, energy_loss=e, attn_ctc_loss=ctc,
                  attn_kl_loss=kl, total=total)
    return m, res, {k: float(v.detach()) for k, v in losses.items()}


def _check_mas_model(fix, m, res, losses, tol):
    out = fix["outputs"]
    # alignment / index tensors: bit-exact
    assert torch.equal(res["attn_hard"].cpu(), out["attn_hard"])
    assert torch.equal(res["duration_targets"].cpu(), out["duration_targets"])
    assert torch.equal(res["LR_length_rounded"].cpu(), out["LR_length_rounded"])
    assert res["x_band_width"] == fix["x_band_width"]
    assert_close(res["attn_soft"].detach().cpu(), out["attn_soft"], 1e-5, what="attn_soft")
    assert_close(res["attn_logprob"].detach().cpu(), out["attn_logprob"], 1e-4, what="attn_logprob")
    for k in ["pitch_targets", "energy_targets", "dec_outputs", "postnet_outputs", "log_duration_predictions"]:
        dlt = (res[k].detach().cpu() - out[k]).abs()
        assert float(dlt.mean()) <= tol and float(dlt.max()) <= 20 * tol, (k, float(dlt.mean()), float(dlt.max()))
    for k, v in fix["losses"].items():
        assert abs(losses[k] - v) <= 1e-4 * max(1.0, abs(v)), (k, losses[k], v)
    named = dict(m.named_parameters())
    for k, g in fix["grads"].items():
        assert rel_l2(named[k].grad.cpu(), g) <= 1e-3, k
    for k, (s, nrm) in fix["grad_summaries"].items():
        assert named[k].grad is not None, k
        assert abs(float(named[k].grad.double().norm()) - nrm) <= 2e-3 * nrm + 1e-7, k


def test_sambert_mas_host_logic_matches_reference_fixture(emulated_cabi):
    fix = _fix("sambert_tiny_mas")
    m, res, losses = _mas_model_run(fix, "cpu")
    _check_mas_model(fix, m, res, losses, tol=1e-5)


def test_mas_state_dict_keys_and_reference_call_style(emulated_cabi):
    """ConvAttention keeps the reference's parameter names and its (B, C, T) forward contract."""
    from kantts.models.sambert.attention import ConvAttention

    torch.manual_seed(3)
    att = ConvAttention(n_mel_channels=80, n_text_channels=32, n_att_channels=80)
    keys = set(att.state_dict().keys())
    for want in ["key_proj.0.conv.weight", "key_proj.2.conv.bias", "query_proj.0.conv.weight", "query_proj.2.conv.weight",
                 "query_proj.4.conv.bias", "attn_proj.weight", "attn_proj.bias"]:
        assert want in keys, want
    q, k = torch.randn(2, 80, 9), torch.randn(2, 32, 5)
    mask = torch.tensor([[False] * 5, [False, False, False, True, True]])
    soft, logprob = att(q, k, mask=mask, attn_prior=None)
    assert tuple(soft.shape) == (2, 1, 9, 5) and tuple(logprob.shape) == (2, 1, 9, 5)
    assert torch.all(soft[1, :, :, 3:] == 0) and torch.allclose(soft.sum(3), torch.ones(2, 1, 9), atol=1e-6)


# ---------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_mas_dp_gpu_bit_exact_vs_reference_fixture():
    from kantts.models.sambert.alignment import b_mas

    fix = _fix("mas_dp")
    hard = b_mas(fix["attn"].cuda(), fix["in_lens"].cuda(), fix["out_lens"].cuda())
    assert torch.equal(hard.cpu(), fix["hard"])
    # numpy in / numpy out, as the reference's caller (kantts_sambert.py:759-764) uses it
    hard_np = b_mas(fix["attn"].numpy(), fix["in_lens"].numpy(), fix["out_lens"].numpy(), width=1)
    assert isinstance(hard_np, np.ndarray) and np.array_equal(hard_np, fix["hard"].numpy())


@pytest.mark.gpu
def test_mas_dp_gpu_training_shape_vs_oracle():
    """B=32, 612 frames x 64 phonemes (BASELINE training shape): HIP path == emulated-ABI oracle, bit for bit, and the
    size-independent properties of a monotone alignment hold."""
    from kantts._hip import ops
    from util import emulation

    g = torch.Generator().manual_seed(11)
    B, To, Ti = 32, 612, 64
    attn = torch.softmax(torch.randn(B, 1, To, Ti, generator=g) * 2 +
                         -0.05 * (torch.arange(To)[:, None] * Ti / To - torch.arange(Ti)[None, :]) ** 2, dim=3)
    in_lens = torch.randint(20, Ti + 1, (B,), generator=g)
    out_lens = torch.randint(300, To + 1, (B,), generator=g)
    hard = ops.mas_width1(attn.cuda(), in_lens.cuda(), out_lens.cuda()).cpu()
    with emulation():
        ref = ops.mas_width1(attn, in_lens, out_lens)
    assert torch.equal(hard, ref)
    dur = hard.sum(2)[:, 0]
    assert torch.equal(dur.sum(1).long(), out_lens)                      # every frame assigned exactly once
    idx = hard[:, 0].argmax(2)                                           # phoneme index per frame
    for b in range(B):
        p = idx[b, :out_lens[b]]
        step = p[1:] - p[:-1]
        assert int(p[0]) == 0 and int(p[-1]) == int(in_lens[b]) - 1 and bool(((step == 0) | (step == 1)).all())


@pytest.mark.gpu
@pytest.mark.parametrize("To,Ti", [(90, 64), (75, 65), (130, 128), (60, 200), (33, 256), (2100, 330), (9000, 40)])
def test_mas_dp_gpu_every_kernel_variant_vs_oracle(To, Ti):
    """1 / 2 / 4 ballot words per row, the exact 64-column boundaries, and the row-parallel fallback (more than 256
    symbols, or more than 64 KB of masks) -- all bit-exact against the emulated-ABI oracle, ties included."""
    from kantts._hip import ops
    from util import emulation

    g = torch.Generator().manual_seed(To * 1000 + Ti)
    B = 3
    attn = torch.softmax(torch.randn(B, 1, To, Ti, generator=g) * 2, dim=3)
    attn[1] = torch.round(attn[1] * 64) / 64  # coarse grid: ties and zeros (-inf scores)
    in_lens = torch.tensor([Ti, max(1, Ti - 7), max(1, Ti // 2)])
    out_lens = torch.tensor([To, To - 5, max(1, To // 3)])
    hard = ops.mas_width1(attn.cuda(), in_lens.cuda(), out_lens.cuda()).cpu()
    with emulation():
        ref = ops.mas_width1(attn, in_lens, out_lens)
    assert torch.equal(hard, ref)
    if To >= 2 * Ti:  # enough frames for a monotone path from (0, 0): every frame assigned exactly once
        assert torch.equal(hard.sum((1, 2, 3)).long(), out_lens)


@pytest.mark.gpu
@pytest.mark.parametrize("with_prior", [True, False])
def test_align_attention_gpu_vs_oracle(with_prior):
    from kantts._hip import ops

    q, k, prior, lens = _attn_inputs(B=3, T1=61, T2=300, C=80, seed=8)

    def fn(q_, k_, p_, l_):
        return ops.align_attention(q_, k_, p_ if with_prior else None, l_)

    go, gg, co, cg = run_both(fn, q, k, prior, torch.tensor([300, 257, 1], dtype=torch.int32))
    assert_close(go[0], co[0], 1e-6, what="soft")
    assert_close(go[1], co[1], 5e-5, what="logprob")
    for a, b, n in zip(gg, cg, "qk"):
        assert rel_l2(a, b) < 1e-5, n
    rs, rl = _conv_attention_torch(q.detach(), k.detach(), prior if with_prior else None, torch.tensor([300, 257, 1]))
    assert_close(go[0], rs, 1e-6, what="soft vs torch")


@pytest.mark.gpu
def test_sambert_mas_gpu_matches_reference_fixture():
    import kantts._hip as hip

    hip.set_precision("fp32")
    fix = _fix("sambert_tiny_mas")
    m, res, losses = _mas_model_run(fix, "cuda")
    _check_mas_model(fix, m, res, losses, tol=1e-5)


@pytest.mark.gpu
def test_mas_training_steps_run_and_reduce_alignment_loss():
    """Sambert_Trainer with MAS: True -- eager step with the two attention losses, no host round trip for the DP."""
    import kantts._hip as hip
    from kantts.train.loss import (AttentionBinarizationLoss, AttentionCTCLoss, MelReconLoss, ProsodyReconLoss)
    from kantts.models.sambert.kantts_sambert import KanTtsSAMBERT

    hip.set_precision("fp32")
    fix = _fix("sambert_tiny_mas")
    torch.manual_seed(0)
    m = KanTtsSAMBERT(dict(fix["cfg"])).cuda().train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    b = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in fix["batch"].items()}
    ctc_hist = []
    for _ in range(8):
        res = m(**{k: (v.clone() if torch.is_tensor(v) else v) for k, v in b.items()})
        mel_, mel = MelReconLoss()(b["output_lengths"], b["mel_targets"], res["dec_outputs"], res["postnet_outputs"])
        d, p, e = ProsodyReconLoss()(b["input_lengths"], res["duration_targets"], res["pitch_targets"],
                                     res["energy_targets"], res["log_duration_predictions"], res["pitch_predictions"],
                                     res["energy_predictions"])
        ctc = AttentionCTCLoss()(res["attn_logprob"], b["input_lengths"], b["output_lengths"])
        kl = AttentionBinarizationLoss()(50, res["attn_hard"], res["attn_soft"])
        total = mel_ + mel + d + p + e + ctc + kl
        opt.zero_grad()
        total.backward()
        opt.step()
        assert torch.isfinite(total)
        assert torch.equal(res["duration_targets"].sum(1).long().cpu(),
                           torch.full((3,), b["mel_targets"].shape[1], dtype=torch.long))
        ctc_hist.append(float(ctc.detach()))
    assert ctc_hist[-1] < ctc_hist[0]


# ---------------------------------------------------------------------------------------------- trainer / CLI (MAS: True)
def _mas_cli(tmp_path):
    """kantts.bin.train_sambert with the Model/Loss sections of configs/sambert_16k_MAS.yaml on synthetic MAS batches."""
    from kantts.bin.train_sambert import train as train_am
    from kantts.utils.synthetic import sambert_16k_config

    cfg = sambert_16k_config(tiny=True)
    cfg["MAS"] = True
    am = {"model_type": "sambert", "Model": {"KanTtsSAMBERT": {
        "params": cfg,
        "optimizer": {"type": "Adam", "params": {"lr": 0.001, "betas": [0.9, 0.98], "eps": 1e-9, "weight_decay": 0.0}},
        "scheduler": {"type": "NoamLR", "params": {"warmup_steps": 4000}}}},
        "Loss": {"MelReconLoss": {"enable": True, "params": {"loss_type": "mae"}},
                 "ProsodyReconLoss": {"enable": True, "params": {"loss_type": "mae"}},
                 "AttentionCTCLoss": {"enable": True},
                 "AttentionBinarizationLoss": {"enable": True, "params": {"start_epoch": 0, "warmup_epoch": 100}}},
        "grad_norm": 1.0, "batch_size": 2, "log_interval_steps": 1, "save_interval_steps": 100, "train_max_steps": 100}
    tr = train_am(am, [], str(tmp_path / "am_mas"), synthetic=2)
    assert tr.steps == 3
    assert tr.with_MAS and "AttentionCTCLoss" in tr.criterion and "AttentionBinarizationLoss" in tr.criterion
    sd = tr.model["KanTtsSAMBERT"].state_dict()
    assert "align_attention.key_proj.0.conv.weight" in sd
    return tr


def test_mas_training_cli_emulated(tmp_path):
    from util import emulation

    with emulation():
        _mas_cli(tmp_path)


@pytest.mark.gpu
def test_mas_training_cli_gpu(tmp_path):
    _mas_cli(tmp_path)


@pytest.mark.gpu
def test_captured_mas_step_equals_eager_mas_step_gpu(tmp_path):
    """Sambert_Trainer(graph=True) on a sambert_16k_MAS-shaped config -- the whole MAS step (alignment
    learner, device-side Monotonic Alignment Search, CTC + binarisation terms, backward, clip + Adam) replayed from a
    hipGraph -- against the eager MAS trainer: same seeded weights and batches, dropout off (eval-mode modules, dropout
    probabilities 0), four steps over two batch shapes; weights after the steps agree."""
    import kantts._hip as hip
    from kantts.models import model_builder
    from kantts.train.loss import criterion_builder
    from kantts.train.trainer import Sambert_Trainer
    from kantts.utils.synthetic import SAMBERT_VOCAB, sambert_16k_config, sambert_mas_batch, to_collate_format

    hip.set_precision("fp32")
    cfg = sambert_16k_config(tiny=True)
    cfg["MAS"] = True
    for k in list(cfg):
        if "dropout" in k:
            cfg[k] = 0.0
    for k, v in SAMBERT_VOCAB.items():
        cfg.setdefault(k, v)
    config = {"model_type": "sambert", "Model": {"KanTtsSAMBERT": {
        "params": cfg,
        "optimizer": {"type": "Adam", "params": {"lr": 0.001, "betas": [0.9, 0.98], "eps": 1e-9, "weight_decay": 0.0}},
        "scheduler": {"type": "NoamLR", "params": {"warmup_steps": 4000}}}},
        "Loss": {"MelReconLoss": {"enable": True, "params": {"loss_type": "mae"}},
                 "ProsodyReconLoss": {"enable": True, "params": {"loss_type": "mae"}},
                 "AttentionCTCLoss": {"enable": True},
                 "AttentionBinarizationLoss": {"enable": True, "params": {"start_epoch": 0, "warmup_epoch": 4}}},
        "grad_norm": 1.0, "batch_size": 2, "log_interval_steps": 1}
    batches = [to_collate_format(sambert_mas_batch(B=2, seed=1234 + s)) for s in range(2)]

    def run(graph):
        torch.manual_seed(0)
        model, opt, sch = model_builder(config, device="cuda")
        model["KanTtsSAMBERT"].eval()
        crit = criterion_builder(config, "cuda")
        tr = Sambert_Trainer(config, model, opt, sch, crit, torch.device("cuda"), None, batches, None, max_steps=10 ** 6,
                             save_dir=str(tmp_path / ("g" if graph else "e")), save_interval=10 ** 6, valid_interval=10 ** 6,
                             log_interval=1, grad_clip=1.0, graph=graph)
        tr.set_model_state = lambda state="train": None
        tr.epoch = 2  # a binarisation warm-up ratio of 0.5: the device-side ratio of the captured step must carry it
        losses = []
        for b in (batches[0], batches[1], batches[0], batches[1]):
            losses.append(float(tr.train_step(b)))
            tr.steps += 1
        flat = tr.optimizer["KanTtsSAMBERT"].arena.flat.detach().cpu().clone()
        return losses, flat, tr

    le, fe, _ = run(False)
    lg, fg, trg = run(True)
    assert trg.with_MAS and len(trg._graphs) >= 1
    for a, b in zip(lg, le):
        assert abs(a - b) <= 2e-4 * max(1.0, abs(b)), (lg, le)
    assert float((fg - fe).norm() / fe.norm()) < 1e-4
