"""Where a batch-1 utterance of the inference leg (bench.py inference_leg, decoder mode "graph") spends its wall time:
every stage bracketed by device synchronisations (so host issue time and device time of a stage add up; the sum is a
little above the un-instrumented time).  Usage (GPU box): python scripts/infer_breakdown.py [n_utterances]"""
import os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "kan-tts_amd"))
import torch
import bench
import kantts._hip as hip
from kantts.models.hifigan.hifigan import Generator
from kantts.models.sambert.kantts_sambert import KanTtsSAMBERT
from kantts.utils.synthetic import inference_utterances

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
MODE = sys.argv[2] if len(sys.argv) > 2 else "graph"  # "kernel": each autoregressive loop as one launch
from kantts.utils import synthetic
cfg = synthetic.sambert_16k_config()
hip.set_precision("bf16")
dev = "cuda"
torch.manual_seed(0)
am = KanTtsSAMBERT(dict(cfg))
with torch.no_grad():
    am.variance_adaptor.duration_predictor.fc.bias.fill_(1.5)
am = am.to(dev).eval()
voc = Generator().to(dev).eval()
voc.remove_weight_norm()
lens, ling, emo, spk = inference_utterances(128)
order = torch.argsort(lens, descending=True)
am.mel_decoder.decode_mode = MODE
am.variance_adaptor.duration_predictor.ar_kernel = None if MODE == "kernel" else False
T = collections.OrderedDict()


def wrap(obj, name, label):
    f = getattr(obj, name)

    def g(*a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = f(*a, **k)
        torch.cuda.synchronize()
        T[label] = T.get(label, 0.0) + time.perf_counter() - t0
        return r
    setattr(obj, name, g)


va = am.variance_adaptor
wrap(am.text_encoder, "forward", "text encoder")
wrap(va.duration_predictor, "infer", "  duration AR loop")
wrap(va.pitch_predictor, "forward", "  pitch predictor")
wrap(va.energy_predictor, "forward", "  energy predictor")
wrap(va, "forward", "variance adaptor (total)")
wrap(am.mel_decoder, "forward", "mel decoder loop")
wrap(am.mel_postnet, "forward", "postnet")
wrap(voc, "forward", "vocoder")


def synth(idx):
    ln = lens[idx]
    Tm = int(ln.max())
    args = dict(inputs_ling=ling[idx, :Tm].to(dev), inputs_emotion=emo[idx, :Tm].to(dev),
                inputs_speaker=spk[idx, :Tm].to(dev), input_lengths=ln.to(dev))
    with torch.no_grad():
        res = am(**args)
        mel = res["postnet_outputs"].transpose(1, 2).contiguous()
        wav = voc(mel)
    nfr = res["LR_length_rounded"].clamp(max=mel.shape[2])
    return int(nfr.sum())


sel = [order[i:i + 1] for i in range(0, 128, max(1, 128 // n))][:n]
synth(sel[0])
T.clear()
torch.cuda.synchronize()
t0 = time.perf_counter()
frames = sum(synth(i) for i in sel)
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print("decoder mode %s: %d utterances, %d frames, %.2f ms per utterance (instrumented)" % (MODE, len(sel), frames, 1e3 * tot / len(sel)))
for k, v in T.items():
    print("%-32s %7.2f ms per utterance  %5.1f %%" % (k, 1e3 * v / len(sel), 100 * v / tot))
print("%-32s %7.2f ms per utterance" % ("(outside the stages)", 1e3 * (tot - sum(v for k, v in T.items() if not k.startswith("  "))) / len(sel)))
