"""kantts.bin inference entry points (reference kantts/bin/infer_sambert.py, infer_hifigan.py) driven end to end on
the emulated C ABI (CPU) and on the GPU: checkpoint / config discovery, file outputs, shapes."""
import os

import numpy as np
import pytest
import torch
import yaml

import torch_oracle as O
from util import emulation


class _FakeLingUnit:
    """Stands in for the text front-end (out of scope): symbols are already integer streams."""

    def __init__(self, cfg):
        self.cfg = cfg

    def using_byte(self):
        return False

    def get_unit_size(self):
        return {k: self.cfg[k] for k in O.SAMBERT_VOCAB}

    def encode_symbol_sequence(self, seq):
        n = len(seq.split()) + 1  # + the trailing "~"
        g = np.random.default_rng(n)
        return [g.integers(0, 5, n), g.integers(0, 5, n), g.integers(0, 5, n), g.integers(0, 5, n),
                g.integers(0, 5, n), np.zeros(n, dtype=np.int64)]


def _run(tmp_path, device):
    from kantts.bin.infer_hifigan import hifigan_infer
    from kantts.bin.infer_sambert import am_infer
    from kantts.models.hifigan.hifigan import Generator
    from kantts.models.sambert.kantts_sambert import KanTtsSAMBERT

    cfg = O.sambert_config(tiny=True)
    params = {k: v for k, v in cfg.items() if k not in O.SAMBERT_VOCAB}
    am_dir = tmp_path / "am" / "ckpt"
    am_dir.mkdir(parents=True)
    config = {"model_type": "sambert", "Model": {"KanTtsSAMBERT": {
        "params": params,
        "optimizer": {"type": "Adam", "params": {"lr": 0.001, "betas": [0.9, 0.98], "eps": 1e-9, "weight_decay": 0.0}},
        "scheduler": {"type": "NoamLR", "params": {"warmup_steps": 4000}}}}, "grad_norm": 1.0, "batch_size": 2}
    (tmp_path / "am" / "config.yaml").write_text(yaml.dump(config))
    torch.manual_seed(0)
    m = KanTtsSAMBERT(dict(cfg))
    with torch.no_grad():
        m.variance_adaptor.duration_predictor.fc.bias.fill_(1.2)
    torch.save({"model": m.state_dict()}, am_dir / "checkpoint_1.pth")
    sent = tmp_path / "sentences.txt"
    sent.write_text("utt_a\ta b c d e f\nutt_b\tg h i j\n")
    am_infer(str(sent), str(am_dir / "checkpoint_1.pth"), str(tmp_path / "out"), ling_unit=_FakeLingUnit(cfg))
    mel = np.load(tmp_path / "out" / "feat" / "utt_a_mel.npy")
    dur = np.loadtxt(tmp_path / "out" / "feat" / "utt_a_dur.txt")
    assert mel.ndim == 2 and mel.shape[1] == 80 and mel.shape[0] == int(dur.sum()) and np.isfinite(mel).all()
    assert os.path.exists(tmp_path / "out" / "feat" / "utt_b_energy.txt")
    # vocoder: tiny generator checkpoint in the reference's layout
    voc_dir = tmp_path / "voc" / "ckpt"
    voc_dir.mkdir(parents=True)
    gparams = {"channels": 32}
    (tmp_path / "voc" / "config.yaml").write_text(yaml.dump(
        {"Model": {"Generator": {"params": gparams}}, "audio_config": {"sampling_rate": 16000}}))
    torch.manual_seed(1)
    torch.save({"model": {"generator": Generator(**gparams).state_dict()}}, voc_dir / "checkpoint_1.pth")
    np.save(tmp_path / "utt_a.npy", mel[:6].astype(np.float32))
    rtf = hifigan_infer(str(tmp_path / "utt_a.npy"), str(voc_dir / "checkpoint_1.pth"), str(tmp_path / "wav"))
    from scipy.io import wavfile

    sr, wav = wavfile.read(tmp_path / "wav" / "utt_a_gen.wav")
    assert sr == 16000 and wav.dtype == np.int16 and wav.shape[0] == 6 * 256 and rtf > 0


def test_inference_entry_points_emulated(tmp_path):
    with emulation():
        _run(tmp_path, "cpu")


@pytest.mark.gpu
def test_inference_entry_points_gpu(tmp_path):
    import kantts._hip as hip

    hip.set_precision("fp32")
    _run(tmp_path, "cuda")


def test_synthetic_workload_matches_the_oracles_generator():
    """bench.py draws its inputs from kantts.utils.synthetic (product side); the CPU baseline and the parity tests
    use the oracle's generator: both must produce the same tensors and config."""
    from kantts.utils import synthetic

    assert synthetic.sambert_16k_config() == O.sambert_config(tiny=False)
    assert synthetic.sambert_16k_config(tiny=True) == O.sambert_config(tiny=True)
    a, b = synthetic.sambert_batch(B=5, T_in=40, seed=9), O.synthetic_sambert_batch(B=5, T_in=40, seed=9)
    assert a.keys() == b.keys()
    for k in a:
        assert torch.equal(a[k], b[k]), k


def _train_clis(tmp_path):
    from kantts.bin.train_hifigan import train as train_voc
    from kantts.bin.train_sambert import train as train_am
    from kantts.utils.synthetic import sambert_16k_config

    cfg = sambert_16k_config(tiny=True)
    am = {"model_type": "sambert", "Model": {"KanTtsSAMBERT": {
        "params": cfg,
        "optimizer": {"type": "Adam", "params": {"lr": 0.001, "betas": [0.9, 0.98], "eps": 1e-9, "weight_decay": 0.0}},
        "scheduler": {"type": "NoamLR", "params": {"warmup_steps": 4000}}}},
        "grad_norm": 1.0, "batch_size": 2, "log_interval_steps": 2, "save_interval_steps": 2, "train_max_steps": 100}
    tr = train_am(am, [], str(tmp_path / "am"), synthetic=3)
    assert tr.steps == 4 and os.path.exists(tmp_path / "am" / "ckpt" / "checkpoint_2.pth")
    assert os.path.exists(tmp_path / "am" / "config.yaml")
    tr2 = train_am(am, [], str(tmp_path / "am2"), resume_path=str(tmp_path / "am" / "ckpt" / "checkpoint_2.pth"),
                   synthetic=1)
    assert tr2.steps == 3  # resumed at step 2, one more batch
    opt = {"type": "Adam", "params": {"lr": 2e-4, "betas": [0.5, 0.9], "weight_decay": 0.0}}
    sch = {"type": "MultiStepLR", "params": {"gamma": 0.5, "milestones": [200000]}}
    voc = {"model_type": "hifigan", "Model": {
        "Generator": {"params": {"channels": 32}, "optimizer": opt, "scheduler": sch},
        "MultiPeriodDiscriminator": {"params": {"periods": [2, 3]}, "optimizer": opt, "scheduler": sch}},
        "Loss": {"generator_adv_loss": {"enable": True, "params": {}, "weights": 1.0},
                 "discriminator_adv_loss": {"enable": True, "params": {}, "weights": 1.0},
                 "mel_loss": {"enable": True, "params": {}, "weights": 45.0},
                 "feat_match_loss": {"enable": True, "params": {}, "weights": 2.0}},
        "generator_grad_norm": -1, "discriminator_grad_norm": -1, "discriminator_train_start_steps": 0,
        "generator_train_start_steps": 0, "batch_size": 2, "batch_max_steps": 1024, "log_interval_steps": 1,
        "save_interval_steps": 2, "audio_config": {"hop_length": 256, "sampling_rate": 16000}}
    tv = train_voc(voc, [], str(tmp_path / "voc"), synthetic=2)
    assert tv.steps == 3 and os.path.exists(tmp_path / "voc" / "ckpt" / "checkpoint_2.pth")


def test_training_entry_points_emulated(tmp_path):
    with emulation():
        _train_clis(tmp_path)


def test_infer_sambert_with_a_speaker_embedding_file_emulated(tmp_path):
    """``infer_sambert --se_file`` (reference kantts/bin/infer_sambert.py:99-106,177-178): an SE checkpoint takes the
    utterance's speaker embedding from a .npy file, repeated over the symbols; the file changes the output; without it
    the entry point refuses instead of feeding speaker ids to an embedding-input model."""
    from kantts.bin.infer_sambert import am_infer
    from kantts.models.sambert.kantts_sambert import KanTtsSAMBERT
    from util import GOLDEN

    cfg = dict(torch.load(os.path.join(GOLDEN, "sambert_tiny_se.pt"), weights_only=False)["cfg"])
    assert cfg.get("SE")
    params = {k: v for k, v in cfg.items() if k not in O.SAMBERT_VOCAB}
    am_dir = tmp_path / "am" / "ckpt"
    am_dir.mkdir(parents=True)
    config = {"model_type": "sambert", "Model": {"KanTtsSAMBERT": {
        "params": params,
        "optimizer": {"type": "Adam", "params": {"lr": 0.001, "betas": [0.9, 0.98], "eps": 1e-9, "weight_decay": 0.0}},
        "scheduler": {"type": "NoamLR", "params": {"warmup_steps": 4000}}}}, "grad_norm": 1.0, "batch_size": 2}
    (tmp_path / "am" / "config.yaml").write_text(yaml.dump(config))
    torch.manual_seed(0)
    m = KanTtsSAMBERT(dict(cfg))
    with torch.no_grad():
        m.variance_adaptor.duration_predictor.fc.bias.fill_(1.2)
    torch.save({"model": m.state_dict()}, am_dir / "checkpoint_1.pth")
    sent = tmp_path / "sentences.txt"
    sent.write_text("utt_a\ta b c d e f\n")
    dim = int(cfg.get("speaker_units", 192))
    mels = []
    with emulation():
        for seed in (1, 2):
            se = np.random.default_rng(seed).standard_normal((1, dim)).astype(np.float32)
            np.save(tmp_path / ("se%d.npy" % seed), se)
            out = tmp_path / ("out%d" % seed)
            am_infer(str(sent), str(am_dir / "checkpoint_1.pth"), str(out), se_file=str(tmp_path / ("se%d.npy" % seed)),
                     ling_unit=_FakeLingUnit(cfg))
            mels.append(np.load(out / "feat" / "utt_a_mel.npy"))
        with pytest.raises(ValueError, match="se_file"):
            am_infer(str(sent), str(am_dir / "checkpoint_1.pth"), str(tmp_path / "out3"), ling_unit=_FakeLingUnit(cfg))
    assert all(x.ndim == 2 and x.shape[1] == 80 and np.isfinite(x).all() for x in mels)
    n = min(len(mels[0]), len(mels[1]))
    assert n > 0 and np.abs(mels[0][:n] - mels[1][:n]).max() > 1e-4  # the embedding reaches the decoder


def _reference_denorm_f0():
    """The reference's own ``denorm_f0`` (kantts/bin/infer_sambert.py:26-56), loaded from its source file when the checkout
    is present (the build container); None on a box without it."""
    import ast

    path = "/root/reference/kantts/bin/infer_sambert.py"
    if not os.path.exists(path):
        return None
    tree = ast.parse(open(path).read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "denorm_f0"]
    ns = {"np": np}
    exec(compile(ast.Module(body=fn, type_ignores=[]), path, "exec"), ns)
    return ns["denorm_f0"]


def test_nsf_f0_denormalisation_follows_the_reference():
    """``denorm_f0`` of the NSF acoustic models (reference :26-56): both normalisations, against the reference's function when
    its checkout is here and against the definition (hand values) everywhere."""
    from kantts.bin.infer_sambert import denorm_f0

    g = np.random.default_rng(3)
    mel = g.standard_normal((50, 82)).astype(np.float32)
    mel[:, -1] = g.uniform(0, 1, 50)
    mel[:, -2] = g.uniform(-0.2, 1.0, 50)
    mvn = np.array([[210.0], [45.0]], dtype=np.float32)  # mvn.npy: mean row, std row
    got_ms = denorm_f0(mel.copy(), scale=float(mvn[1, 0]), offset=float(mvn[0, 0]))
    got_gl = denorm_f0(mel.copy(), scale=730.0 - 30.0, offset=30.0)
    assert np.array_equal(got_ms[:, :80], mel[:, :80]) and set(np.unique(got_ms[:, -1])) <= {0.0, 1.0}
    assert np.allclose(got_ms[:, -2], np.maximum(mel[:, -2] * 45.0 + 210.0, 30.0), atol=1e-4)
    assert np.allclose(got_gl[:, -2], np.maximum(mel[:, -2] * 700.0 + 30.0, 30.0), atol=1e-4)
    assert np.array_equal(got_gl[:, -1], (mel[:, -1] >= 0.6).astype(np.float32))
    ref = _reference_denorm_f0()
    if ref is not None:
        assert np.allclose(got_ms, ref(mel.copy(), norm_type="mean_std", f0_feature=mvn), atol=1e-4)
        assert np.allclose(got_gl, ref(mel.copy(), norm_type="global", f0_feature=[730.0, 30.0]), atol=1e-4)


def test_infer_sambert_accepts_an_nsf_acoustic_model_emulated(tmp_path):
    """``infer_sambert`` on an NSF checkpoint (reference :180-193, 219-220; sambert_se_nsf_global_16k.yaml: num_mels 82,
    nsf_norm_type global): the two channels behind the mel bins leave the entry point as f0 in Hz (>= 30) and a 0 / 1
    voicing flag."""
    from kantts.bin.infer_sambert import am_infer
    from kantts.models.sambert.kantts_sambert import KanTtsSAMBERT

    cfg = O.sambert_config(tiny=True)
    cfg.update(num_mels=82, NSF=True, nsf_norm_type="global", nsf_f0_global_minimum=30.0, nsf_f0_global_maximum=730.0)
    params = {k: v for k, v in cfg.items() if k not in O.SAMBERT_VOCAB}
    am_dir = tmp_path / "am" / "ckpt"
    am_dir.mkdir(parents=True)
    config = {"model_type": "sambert", "Model": {"KanTtsSAMBERT": {
        "params": params,
        "optimizer": {"type": "Adam", "params": {"lr": 0.001, "betas": [0.9, 0.98], "eps": 1e-9, "weight_decay": 0.0}},
        "scheduler": {"type": "NoamLR", "params": {"warmup_steps": 4000}}}}, "grad_norm": 1.0, "batch_size": 2}
    (tmp_path / "am" / "config.yaml").write_text(yaml.dump(config))
    torch.manual_seed(0)
    m = KanTtsSAMBERT(dict(cfg))
    with torch.no_grad():
        m.variance_adaptor.duration_predictor.fc.bias.fill_(1.2)
    torch.save({"model": m.state_dict()}, am_dir / "checkpoint_1.pth")
    sent = tmp_path / "sentences.txt"
    sent.write_text("utt_a\ta b c d e f\n")
    with emulation():
        am_infer(str(sent), str(am_dir / "checkpoint_1.pth"), str(tmp_path / "out"), ling_unit=_FakeLingUnit(cfg))
    mel = np.load(tmp_path / "out" / "feat" / "utt_a_mel.npy")
    assert mel.ndim == 2 and mel.shape[1] == 82 and np.isfinite(mel).all()
    assert (mel[:, -2] >= 30.0).all() and set(np.unique(mel[:, -1])) <= {0.0, 1.0}
