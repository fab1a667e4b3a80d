"""Upsampling chain (four transposed convolutions of the V1 generator, bf16 storage) -- per-stage and chain time."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kan-tts_amd"))
import torch
import kantts._hip as hip
from kantts._hip import ops
from kantts.models.hifigan.hifigan import Generator
from kantts.models.hifigan.layers import effective_weight

hip.set_precision("bf16")
torch.manual_seed(0)
G = Generator().cuda()
B, frames = 32, 32


def ev(fn, n=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


with torch.no_grad():
    acts, ws, T, C, belems = [], [], frames, 512, 0
    for i, s_ in enumerate((8, 8, 2, 2)):
        layer = G.transpose_upsamples[i][1]
        acts.append(torch.randn(B, T, C, device="cuda").to(torch.bfloat16))
        w_ = effective_weight(layer.deconv).detach().contiguous()
        ws.append((w_, layer.deconv.bias.detach(), s_, ops.upsample_weights(w_, s_, layer.deconv.bias)))
        belems += B * T * C + B * T * s_ * (C // 2) + C * (C // 2) * 2 * s_
        T, C = T * s_, C // 2

    def up_b(i):
        w, b, s_, prep = ws[i]
        return ops.upsample_forward(acts[i], w, b, s_, out_bf16=True, in_slope=0.1 if acts[i].shape[2] <= 128 else 1.0, prepared=prep)

    def graph_us(fn, reps=8, n=20):
        """device time per call from a replayed hipGraph of ``reps`` back-to-back calls (eager issue is host-bound)"""
        s_ = torch.cuda.Stream()
        s_.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s_):
            fn()
        torch.cuda.current_stream().wait_stream(s_)
        torch.cuda.synchronize()
        g_ = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_, capture_error_mode="thread_local"):
            for _ in range(reps):
                fn()
        return ev(g_.replay, n) / reps

    if len(sys.argv) > 1 and sys.argv[1] == "graph":
        st = [round(graph_us(lambda i=i: up_b(i)), 1) for i in range(4)]
        chain = graph_us(lambda: [up_b(i) for i in range(4)], reps=4)
    else:
        st = [round(ev(lambda i=i: up_b(i)), 1) for i in range(4)]
        chain = ev(lambda: [up_b(i) for i in range(4)])
    print("stage_us", st, "chain_us %.1f" % chain, "GB/s %.0f" % (belems * 2 / chain / 1e3), "frac %.3f" % (belems * 2 / chain / 1e3 / 8000))
