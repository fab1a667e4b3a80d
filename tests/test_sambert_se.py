"""Speaker-embedding input (SE: True): KanTtsSAMBERT takes the speaker stream as per-token 192-d vectors
instead of ids (reference kantts_sambert.py:717-720, :928).  Pinned to a forward/backward of the reference
(tests/golden/sambert_tiny_se.pt, oracle/make_golden.py::sambert_se_case)."""
import os

import numpy as np
import pytest
import torch

from util import GOLDEN


def _run(device):
    from kantts.models.sambert.kantts_sambert import KanTtsSAMBERT
    from kantts.train.loss import MelReconLoss, ProsodyReconLoss

    fix = torch.load(os.path.join(GOLDEN, "sambert_tiny_se.pt"), weights_only=False)
    torch.manual_seed(fix["seed_w"])
    m = KanTtsSAMBERT(dict(fix["cfg"]))
    assert sorted(m.state_dict().keys()) == fix["state_keys"]  # no spk_tokenizer.* in SE checkpoints
    for k, (shape, s, a) in fix["weight_checksums"].items():
        v = m.state_dict()[k]
        assert tuple(v.shape) == shape and abs(float(v.double().sum()) - s) <= 1e-6 * max(1.0, a), k
    m = m.to(device).eval()
    b = {k: v.to(device) for k, v in fix["batch"].items()}
    res = m(**b)
    mel_, mel = MelReconLoss()(b["output_lengths"], b["mel_targets"], res["dec_outputs"], res["postnet_outputs"])
    d, p, e = ProsodyReconLoss()(b["input_lengths"], res["duration_targets"], res["pitch_targets"], res["energy_targets"],
                                 res["log_duration_predictions"], res["pitch_predictions"], res["energy_predictions"])
    total = mel_ + mel + d + p + e
    total.backward()
    assert torch.equal(res["LR_length_rounded"].cpu(), fix["outputs"]["LR_length_rounded"])
    assert res["x_band_width"] == fix["x_band_width"]
    for k, ref in fix["outputs"].items():
        if ref.is_floating_point():
            dlt = (res[k].detach().cpu() - ref).abs()
            assert float(dlt.mean()) <= 1e-5 and float(dlt.max()) <= 2e-4, (k, float(dlt.mean()), float(dlt.max()))
    assert abs(float(total.detach()) - fix["losses"]["total"]) <= 1e-4
    named = dict(m.named_parameters())
    for k, (s, nrm) in fix["grad_summaries"].items():
        assert abs(float(named[k].grad.double().norm()) - nrm) <= 2e-3 * nrm + 1e-7, k


def test_sambert_se_host_logic_matches_reference_fixture(emulated_cabi):
    _run("cpu")


@pytest.mark.gpu
def test_sambert_se_gpu_matches_reference_fixture():
    import kantts._hip as hip

    hip.set_precision("fp32")
    _run("cuda")


def test_se_collate_repeats_the_utterance_embedding():
    from kantts.datasets.batching import am_collate

    rng = np.random.RandomState(1)
    items = []
    for n_sym, n_mel in [(4, 9), (6, 14)]:
        ling = [rng.randint(0, 5, n_sym) for _ in range(6)]
        items.append((ling, rng.randn(n_mel, 80).astype(np.float32), rng.randint(1, 4, n_sym - 1), rng.randn(n_sym - 1),
                      rng.randn(n_sym - 1), None, None, rng.randn(1, 192).astype(np.float32)))
    out = am_collate(items, r=3, pad_ids=[0] * 6, se=True)
    spk = out["input_speakers"]
    assert tuple(spk.shape) == (2, 6, 192) and spk.dtype == torch.float32
    assert torch.equal(spk[0, :4], torch.from_numpy(items[0][7]).repeat(4, 1)) and torch.all(spk[0, 4:] == 0)
    assert torch.equal(spk[1], torch.from_numpy(items[1][7]).repeat(6, 1))
