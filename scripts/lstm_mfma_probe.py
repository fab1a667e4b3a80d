"""VERDICT round 4, item 5 measured: the LSTM recurrence (H = 128, forward, as training runs it: gates and cell states saved)
on v_mfma_f32_16x16x32_bf16 with sixteen sequences per workgroup (scripts/lstm_mfma_probe.hip, a prototype outside the
product) against the product's kantts_lstm_fwd (a workgroup per sequence, v_dot2).  Both get the same random weights and
input projections; outputs are compared, then both are timed.  Usage (GPU box): python scripts/lstm_mfma_probe.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kan-tts_amd"))
import torch
import kantts._hip as hip
from kantts._hip import check, lib, ptr, stream

probe = ctypes.CDLL(os.path.join(ROOT, "kan-tts_amd", "variants", "liblstm_mfma_probe.so"))
probe.lstm_mfma_probe_fwd.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
H = 128


def timed(fn, n=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for B, T in ((32, 612), (32, 64), (128, 612), (512, 612)):
    g = torch.Generator().manual_seed(B + T)
    whh = (0.08 * torch.randn(4 * H, H, generator=g)).cuda()
    gx = (0.7 * torch.randn(B, T, 4 * H, generator=g)).cuda()
    # ---- the product: one workgroup per sequence
    out = torch.empty(B, T, H, device="cuda")
    gates = torch.empty(1, B, T, 4 * H, device="cuda")
    cs = torch.empty(1, B, T, H, device="cuda")

    def product():
        check(lib().kantts_lstm_fwd(ptr(gx), ptr(whh), None, None, ptr(out), ptr(gates), ptr(cs), B, T, H, 1, 0, 1, stream()),
              "lstm_fwd")

    # ---- the prototype: sixteen sequences per workgroup, operands in the layout its lanes touch
    groups = B // 16
    wp = whh.view(4, H, H).permute(1, 0, 2).reshape(4 * H, H)  # row 4 c + gate
    wfrag = wp.view(32, 16, 4, 4, 8).permute(0, 2, 3, 1, 4).reshape(-1).to(torch.bfloat16).contiguous()
    gxp = gx.view(groups, 16, T, 4, 32, 4).permute(2, 0, 4, 5, 1, 3).contiguous()  # (T, groups, tile, kg, li, gate)
    outp = torch.empty(T, groups, 8, 4, 16, 4, device="cuda")
    gatesp = torch.empty(T, groups, 32, 64, 4, device="cuda")
    csp = torch.empty_like(outp)

    def proto():
        rc = probe.lstm_mfma_probe_fwd(wfrag.data_ptr(), gxp.data_ptr(), outp.data_ptr(), gatesp.data_ptr(), csp.data_ptr(),
                                       T, groups, stream())
        assert rc == 0, rc

    product()
    proto()
    torch.cuda.synchronize()
    # outp (T, groups, wave, kg, li, j): cell = 16 wave + 4 j + kg
    got = outp.permute(1, 4, 0, 2, 5, 3).reshape(B, T, H)
    err = float((got - out).abs().max())
    t_prod, t_proto = timed(product), timed(proto)
    print("B %4d  T %4d   product (1 sequence / workgroup, %3d workgroups) %7.3f ms = %.3f us per step   "
          "MFMA prototype (16 / workgroup, %2d workgroups) %7.3f ms = %.3f us per step   max |h difference| %.2e"
          % (B, T, B, t_prod, 1e3 * t_prod / T, groups, t_proto, 1e3 * t_proto / T, err), flush=True)
