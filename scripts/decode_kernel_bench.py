"""Time of the one-launch decoder loop (csrc/ar_infer.hip) per step, against the number of decoder layers: the weights of a
step are 0.42 MB + 0.69 MB per layer of bf16, the L2 of an XCD is 4 MB -- does the per-layer time jump where the step's
weights stop fitting?  Usage (GPU box): python scripts/decode_kernel_bench.py [B] [L]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("kan-tts_amd", "oracle", ""):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
import torch_oracle as O
import kantts._hip as hip
from kantts.models.sambert.kantts_sambert import KanTtsSAMBERT
from kantts.models.utils import get_mask_from_lengths

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
L = int(sys.argv[2]) if len(sys.argv) > 2 else 96
hip.set_precision("bf16")
PROF = "ARPROF" in os.environ.get("KANTTS_LIB", "")  # scripts/build_arprof.sh: per-phase wall-clock ticks (100 MHz)
PHASES = ["prenet + entry projection", "LayerNorm 0", "QKV product", "attention scores + softmax", "attention values",
          "output projection", "LayerNorm 1", "feed-forward 1", "feed-forward 2", "final LayerNorm + output product"]
for layers in ((12,) if PROF else (1, 2, 4, 5, 6, 8, 12)):
    cfg = O.sambert_config(tiny=True)
    cfg["decoder_num_layers"] = layers
    torch.manual_seed(0)
    m = KanTtsSAMBERT(dict(cfg)).cuda().eval()
    dec = m.mel_decoder
    dec.decode_mode = "kernel"
    d_mem = dec.mel_dec.pnca[0].pnca_attn.d_mem
    memory = 0.7 * torch.randn(B, L, d_mem, device="cuda")
    lens = torch.full((B,), L, device="cuda")
    bw = torch.full((B,), 6, device="cuda", dtype=torch.int32)
    mask = get_mask_from_lengths(lens, L)

    def run():
        with torch.no_grad():
            return dec(memory, 6, 6, mask=mask, bw_dev=bw)[0]

    run()
    torch.cuda.synchronize()
    if PROF:
        import ctypes
        buf = (ctypes.c_ulonglong * 16)()
        hip.lib().kantts_ar_profile_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
        hip.lib().kantts_ar_profile_read(None, 1)
        run()
        torch.cuda.synchronize()
        hip.lib().kantts_ar_profile_read(buf, 1)
        tot = sum(buf[:10])
        for i, nme in enumerate(PHASES):
            per = L * (layers if 1 <= i <= 8 else 1)
            print("  %-36s %8.2f us per step  (%.2f us per occurrence)  %5.1f %%" % (
                nme, buf[i] / 100.0 / L, buf[i] / 100.0 / per, 100.0 * buf[i] / tot))
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("B %d  L %d  layers %2d  %.2f ms per run  %.1f us per step  %.1f us per step and layer" % (
        B, L, layers, 1e3 * dt, 1e6 * dt / L, 1e6 * dt / L / layers), flush=True)
