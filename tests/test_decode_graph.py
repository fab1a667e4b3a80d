"""Free-running decode with the step index in device memory (kantts/models/sambert/decode_graph.py, csrc/decode.hip):
the fused step function ("fused": eager, "graph": replayed from a captured hipGraph) must reproduce the per-op Python
loop of round 1 ("loop"), which is pinned to the reference's own inference run (tests/golden/sambert_tiny_infer.pt)."""
import pytest
import torch

import torch_oracle as O
from util import assert_close, emulation


def _run(device, B, seed, mode, cfg_over=None):
    from kantts.models.sambert.kantts_sambert import KanTtsSAMBERT

    cfg = O.sambert_config(tiny=True)
    cfg.update(cfg_over or {})
    torch.manual_seed(0)
    m = KanTtsSAMBERT(dict(cfg))
    with torch.no_grad():
        m.variance_adaptor.duration_predictor.fc.bias.fill_(1.5)
    m = m.to(device).eval()
    m.mel_decoder.decode_mode = mode
    batch = O.synthetic_sambert_batch(B=B, T_in=12, seed=seed, min_len=6, dur_hi=6)
    args = {k: batch[k].to(device) for k in ("inputs_ling", "inputs_emotion", "inputs_speaker", "input_lengths")}
    with torch.no_grad():
        res = m(**args)
        res2 = m(**args)  # a second utterance batch of the same shape re-uses the cached decoder state
    for k in ("dec_outputs", "postnet_outputs"):
        assert torch.equal(res[k], res2[k]), k
    return {k: res[k].detach().cpu() for k in ("dec_outputs", "postnet_outputs", "LR_length_rounded")}


@pytest.mark.parametrize("B,seed", [(1, 77), (3, 5)])
def test_fused_decode_step_matches_python_loop_emulated(B, seed):
    with emulation():
        ref = _run("cpu", B, seed, "loop")
        got = _run("cpu", B, seed, "fused")
    assert torch.equal(got["LR_length_rounded"], ref["LR_length_rounded"])
    assert_close(got["dec_outputs"], ref["dec_outputs"], 2e-6, what="dec")
    assert_close(got["postnet_outputs"], ref["postnet_outputs"], 2e-6, what="postnet")


@pytest.mark.gpu
@pytest.mark.parametrize("B,seed", [(1, 77), (3, 5)])
def test_graph_decode_matches_python_loop_gpu(B, seed):
    import kantts._hip as hip

    hip.set_precision("fp32")
    ref = _run("cuda", B, seed, "loop")
    for mode in ("fused", "graph"):
        got = _run("cuda", B, seed, mode)
        assert torch.equal(got["LR_length_rounded"], ref["LR_length_rounded"])
        assert_close(got["dec_outputs"], ref["dec_outputs"], 1e-5, what=mode + " dec")
        assert_close(got["postnet_outputs"], ref["postnet_outputs"], 1e-5, what=mode + " postnet")
