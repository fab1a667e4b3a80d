"""Upsampling stages 0 / 1 (the wide transposed convolutions: 2-tap polyphase contraction on cconv_kernel) over tile shapes
and ring depths, and stages 2 / 3 (upsample_stream_kernel) over workgroups per CU.  Usage (GPU box):
    python scripts/up_tile_sweep.py            # stages 0-1 sweep
    KANTTS_UPSTREAM_WG_PER_CU=1 python scripts/up_tile_sweep.py narrow"""
import os
import sys

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


def ev(fn, n=40):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


with torch.no_grad():
    acts, ws, T, C = [], [], frames, 512
    for i, s_ in enumerate((8, 8, 2, 2)):
        layer = G.transpose_upsamples[i][1]
        acts.append(torch.randn(B, T, C, device="cuda").to(torch.bfloat16))
        w_ = effective_weight(layer.deconv).detach().contiguous()
        ws.append((w_, layer.deconv.bias.detach(), s_, ops.upsample_weights(w_, s_, layer.deconv.bias)))
        T, C = T * s_, C // 2
    if len(sys.argv) > 1 and sys.argv[1] == "narrow":
        for i in (2, 3):
            w, b, s_, prep = ws[i]
            t = ev(lambda: ops.upsample_forward(acts[i], w, b, s_, out_bf16=True, in_slope=0.1, prepared=prep))
            print("stage %d  WG_PER_CU=%s  %.1f us" % (i, os.environ.get("KANTTS_UPSTREAM_WG_PER_CU", "default"), t))
        sys.exit(0)
    for i in (0, 1):
        w, b, s_, (wl, _) = ws[i]
        act = acts[i]
        Bq, Tq, Cin = act.shape
        Cout = w.shape[1]
        brep = b.repeat(s_)
        ref = None
        tiles = (0, 128128, 3128128, 4128128, 128064, 3128064, 4128064, 64128, 3064128, 4064128, 64064, 3064064, 4064064,
                 256064, 3256064, 256032)
        if len(sys.argv) > 1 and sys.argv[1] == "ab":  # the launcher's own choice against the round-3 choice, same box
            tiles = (0, 128064, 4064064, 0, 128064, 4064064) if i == 0 else (0, 128128, 0, 128128)
        for tile in tiles:
            out = torch.empty((Bq, Tq * s_, Cout), device="cuda", dtype=torch.bfloat16)

            def run():
                assert hip.cconv(act, wl, out=None, out_bf=out, B=Bq, Tsrc=Tq, Tdst=Tq, groups=1, CR=Cin, NG=s_ * Cout, K=2,
                                 in_mul=1, in_add=0, in_kstep=-1, in_div=1, phases=1, bias=brep, tile=tile)

            try:
                t = ev(run)
            except Exception as exc:  # a tile the launcher refuses
                print("stage %d tile %8d: %s" % (i, tile, str(exc)[:80]))
                continue
            if ref is None:
                ref = out.float().clone()
            err = float((out.float() - ref).abs().max())
            print("stage %d  tile %8d  %6.1f us   max |diff to default| %.3g" % (i, tile, t, err))
