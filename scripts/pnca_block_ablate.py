"""Per-launch time of the two fused PNCA block kernels at the benchmark shape (M = 6528), called through the C ABI and
replayed from a hipGraph (20 launches per replay).  With the ablation build (scripts/build_pbdbg.sh, KANTTS_LIB=...,
KANTTS_PB_DBG=mask) this prices the phases of the kernels; with the product library it is the plain timing."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "kan-tts_amd")]
import torch  # noqa: E402

import kantts._hip as hip  # noqa: E402
import kantts._hip.ops_bf16 as ob  # noqa: E402

dev = "cuda"
B, L = 32, 204
M = B * L
g = torch.Generator().manual_seed(0)
f32 = dict(device=dev, dtype=torch.float32)
bf = dict(device=dev, dtype=torch.bfloat16)


def rnd(*s, scale=1.0):
    return (torch.randn(*s, generator=g) * scale).to(dev)


x, xn = rnd(M, 128), rnd(M, 128).to(torch.bfloat16)
hkv = rnd(M, 3072)
lens = torch.tensor([204 - 3 * i for i in range(B)], dtype=torch.int32, device=dev)
rows = (torch.arange(L, device=dev)[None, :] >= lens[:, None]).reshape(M)
wqkv, wfx, wfh = ob.frag_major(rnd(384, 128, scale=.1)), ob.frag_major(rnd(128, 128, scale=.1)), ob.frag_major(rnd(128, 128, scale=.1))
w1, w2 = ob.frag_major(rnd(1024, 128, scale=.1)), ob.frag_major(rnd(128, 1024, scale=.05))
wt2, wt1 = ob.frag_major(rnd(1024, 128, scale=.05)), ob.frag_major(rnd(128, 1024, scale=.1))
vec = lambda n: rnd(n, scale=.1)  # noqa: E731
qkv = torch.empty(M, 384, **f32)
ox, oh, y1, out, g1, dox, doh = (torch.empty(M, 128, **f32) for _ in range(7))
lsx, lsh = torch.empty(B, 8, L, **f32), torch.empty(B, 8, L, **f32)
xn1, xn2 = torch.empty(M, 128, **bf), torch.empty(M, 128, **bf)
mean1, rstd1, mean2, rstd2 = (torch.empty(M, **f32) for _ in range(4))
hid, dz = torch.empty(M, 1024, **bf), torch.empty(M, 1024, **bf)
dg, db = torch.zeros(128, **f32), torch.zeros(128, **f32)
bq, bx, bh, b1, b2, g1v, be1, g2v, be2 = vec(384), vec(128), vec(128), vec(1024), vec(128), vec(128) + 1, vec(128), vec(128) + 1, vec(128)
dy = rnd(M, 128)
hk = hkv[:, 256:512]


def fwd():
    hip.pnca_block_fwd(x, xn, hk, 3072, B, L, lens=lens, bw_dev=None, bw_x=5, bw_h=5, rowmask=rows, wqkv=wqkv, bqkv=bq,
                       wfcx=wfx, wfch=wfh, bfcx=bx, bfch=bh, ln1=(g1v, be1, 1e-6), w1=w1, w2=w2, bias1=b1, bias2=b2, att_p=0.1,
                       fc_p=0.1, drop1_p=0.1, drop2_p=0.1, seeds=(1, 2, 3, 4, 5), qkv=qkv, ox=ox, oh=oh, lse_x=lsx, lse_h=lsh, y1=y1,
                       xn1=xn1, mean1=mean1, rstd1=rstd1, hid=hid, out=out, ln2=(g2v, be2, 1e-6, xn2, mean2, rstd2))


def bwd():
    hip.pnca_block_bwd(dy, hid, y1, mean1, rstd1, g1v, rows, wt2, wt1, wfx, wfh, alpha1=1 / 0.9, drop2_p=0.1, drop2_seed=5, fc_p=0.1,
                       fc_seed=3, dz=dz, g1=g1, d_ox=dox, d_oh=doh)


def timed(fn, n=20, reps=10):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            for _ in range(n):
                fn()
        gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            gr.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * reps) * 1e3


print("KANTTS_PB_DBG=%s  forward %.1f us  backward %.1f us per launch (back to back, graph replay)"
      % (os.environ.get("KANTTS_PB_DBG", "0"), timed(fwd), timed(bwd)))
