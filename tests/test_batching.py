"""Batch assembly against fixtures produced by the reference's own collate functions
(oracle/make_golden.py::collate_case): integer tensors bit-exact, float tensors exact (pure copies)."""
import os

import numpy as np
import torch

from util import GOLDEN


def test_am_and_voc_collate_match_reference_fixture():
    from kantts.datasets.batching import am_collate, voc_collate

    fix = torch.load(os.path.join(GOLDEN, "collate.pt"), weights_only=False)
    got = am_collate(fix["items"], fix["r"], fix["pad_ids"])
    ref = fix["am"]
    assert set(got) == set(ref)
    for k, v in ref.items():
        if v is None:
            assert got[k] is None, k
        else:
            assert got[k].dtype == v.dtype and got[k].shape == v.shape, k
            assert torch.equal(got[k], v), k
    # the r-padding frames sit on the slot after the last symbol and every row sums to the padded mel length
    assert torch.all(got["durations"].sum(1) == got["mel_targets"].shape[1])
    rng = np.random.RandomState(0)
    np.random.seed(fix["voc_seed"])
    wav, mel = voc_collate(fix["vitems"], 200, 1600, rng=np.random)
    assert torch.equal(wav, fix["voc"][0]) and torch.equal(mel, fix["voc"][1])
    assert wav.shape == (3, 1, 1600) and mel.shape == (3, 80, 8)


def test_collated_batch_trains_one_step_emulated(tmp_path):
    """The collate output is exactly what Sambert_Trainer.train_step consumes."""
    import torch_oracle as O
    from kantts.datasets.batching import am_collate
    from kantts.models import model_builder
    from kantts.train.loss import MelReconLoss, ProsodyReconLoss
    from kantts.train.trainer import Sambert_Trainer
    from util import emulation

    fix = torch.load(os.path.join(GOLDEN, "collate.pt"), weights_only=False)
    batch = am_collate(fix["items"], fix["r"], fix["pad_ids"])
    cfg = O.sambert_config(tiny=True)
    config = {"model_type": "sambert", "Model": {"KanTtsSAMBERT": {
        "params": cfg,
        "optimizer": {"type": "Adam", "params": {"lr": 0.001, "betas": [0.9, 0.98], "eps": 1e-9, "weight_decay": 0.0}},
        "scheduler": {"type": "NoamLR", "params": {"warmup_steps": 4000}}}}, "grad_norm": 1.0, "batch_size": 4}
    with emulation():
        torch.manual_seed(0)
        model, opt, sch = model_builder(config, device="cpu")
        tr = Sambert_Trainer(config, model, opt, sch, {"MelReconLoss": MelReconLoss(), "ProsodyReconLoss": ProsodyReconLoss()},
                             torch.device("cpu"), None, [batch], None, save_dir=str(tmp_path), grad_clip=1.0)
        loss = tr.train_step(batch)
        assert torch.isfinite(loss)
