"""Monotonic alignment search -- drop-in for kantts/models/sambert/alignment.py of the reference.

The reference runs a numba DP on the host (alignment.py:32-71, called from kantts_sambert.py:752-764 after
``attn.data.cpu().numpy()``).  Here the DP is one HIP kernel (csrc/mas.hip) and the map never leaves HBM.
``b_mas`` keeps the reference's name, argument order and (B, 1, T_mel, T_text) layout; it takes device tensors
(numpy inputs are uploaded, and a numpy array is returned for them, so reference-style callers keep working).
There is no CPU implementation in the product: without the HIP library the call raises.
"""
import numpy as np
import torch

from kantts._hip import ops


def b_mas(b_attn_map, in_lens, out_lens, width=1):
    assert width == 1
    as_numpy = isinstance(b_attn_map, np.ndarray)
    if as_numpy:
        dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
        if dev is None:
            raise RuntimeError("b_mas runs on the GPU (libkantts_hip); no CPU fallback")
        b_attn_map = torch.from_numpy(np.ascontiguousarray(b_attn_map, dtype=np.float32)).to(dev)
        in_lens = torch.as_tensor(np.asarray(in_lens)).to(dev)
        out_lens = torch.as_tensor(np.asarray(out_lens)).to(dev)
    out = ops.mas_width1(b_attn_map, in_lens, out_lens)
    return out.cpu().numpy() if as_numpy else out


def mas_width1(attn_map):
    """One utterance, (T_mel, T_text) -> hard map (reference alignment.py:32-60)."""
    t = attn_map if torch.is_tensor(attn_map) else None
    if t is None:
        return b_mas(np.asarray(attn_map)[None, None], np.array([attn_map.shape[1]]), np.array([attn_map.shape[0]]))[0, 0]
    lens_i = torch.tensor([t.shape[1]], device=t.device)
    lens_o = torch.tensor([t.shape[0]], device=t.device)
    return ops.mas_width1(t[None, None], lens_i, lens_o)[0, 0]
