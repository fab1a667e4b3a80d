"""Times the MAS path kernels at the BASELINE training shape (B=32, 612 mel frames, 64 symbols, C=80)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kan-tts_amd"))
from kantts._hip import ops  # noqa: E402


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


g = torch.Generator().manual_seed(0)
B, T1, T2, C = 32, 612, 64, 80
q = torch.randn(B, T1, C, generator=g).cuda().requires_grad_(True)
k = torch.randn(B, T2, C, generator=g).cuda().requires_grad_(True)
prior = torch.rand(B, T1, T2, generator=g).cuda()
il = torch.randint(32, T2 + 1, (B,), generator=g).cuda()
ol = torch.randint(300, T1 + 1, (B,), generator=g).cuda()
il32 = il.to(torch.int32)
soft, logprob = ops.align_attention(q, k, prior, il32)
print("align_attention fwd  %.1f us" % timeit(lambda: ops.align_attention(q, k, prior, il32)))
c = torch.randn_like(soft)


def fb():
    s, l = ops.align_attention(q, k, prior, il32)
    torch.autograd.grad((s * c).sum() + (l * c).sum(), [q, k])


print("align_attention f+b  %.1f us (incl. the two torch reductions of the test loss)" % timeit(fb))
print("mas_width1 (device)  %.1f us" % timeit(lambda: ops.mas_width1(soft, il, ol)))


def host_roundtrip():
    a = soft.detach().cpu().numpy()
    i, o = il.cpu().numpy(), ol.cpu().numpy()
    return torch.from_numpy(a).cuda(), i, o


print("reference-style D2H + H2D of the map alone (no DP)  %.1f us" % timeit(host_roundtrip, n=10))
