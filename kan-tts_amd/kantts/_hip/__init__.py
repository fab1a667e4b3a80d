"""ctypes binding of libkantts_hip.so (the C ABI declared in include/kantts_hip.h).

This is the only place the product talks to native code.  There is NO CPU fallback: every wrapper
raises if the library is missing or a tensor is not on a HIP device, so a GPU test can never pass
on a silent eager path.
"""
import ctypes
import os
from ctypes import (POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_longlong,
                    c_uint64, c_void_p)

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# KANTTS_LIB: experiment builds of the SAME sources (scripts/build_variants.sh); never a different implementation
LIB_PATH = os.environ.get("KANTTS_LIB") or os.path.normpath(os.path.join(_HERE, "..", "..", "libkantts_hip.so"))
_lib = None

GEMM_MAX_SEG = 4
PREC_FP32, PREC_BF16, PREC_REF = 0, 1, 2


class GemmSeg(Structure):
    _fields_ = [
        ("a", c_void_p), ("a_gate", c_void_p), ("b", c_void_p),
        ("a_is", c_int64), ("a_ks", c_int64), ("b_js", c_int64), ("b_ks", c_int64), ("b_tap", c_int64),
        ("klen", c_int32), ("ntaps", c_int32),
        ("a_tok_axis", c_int32), ("a_shift0", c_int32), ("a_shift_step", c_int32),
        ("b_tok_axis", c_int32), ("b_shift0", c_int32), ("b_shift_step", c_int32),
        ("a_drop_p", c_float), ("a_drop_seed", c_uint64),
        ("a_inner", c_int32), ("a_Tq", c_int32), ("a_Tsrc", c_int32), ("a_mul", c_int32), ("a_div", c_int32),
        ("a_up", c_int32),
        ("b_inner", c_int32), ("b_Tq", c_int32), ("b_Tsrc", c_int32), ("b_mul", c_int32), ("b_div", c_int32),
        ("b_up", c_int32),
        ("a_slope", c_float), ("a_act", c_int32), ("b_slope", c_float), ("b_act", c_int32),
        ("a_gate_slope", c_float), ("a_mode", c_int32), ("b_mode", c_int32),
    ]


class GemmArgs(Structure):
    _fields_ = [
        ("seg", GemmSeg * GEMM_MAX_SEG),
        ("nseg", c_int32), ("M", c_int32), ("N", c_int32), ("T", c_int32),
        ("c", c_void_p), ("c_is", c_int64), ("c_js", c_int64),
        ("bias", c_void_p), ("bias2", c_void_p), ("res", c_void_p), ("r_is", c_int64), ("r_js", c_int64),
        ("rowmask", c_void_p), ("kmask", c_void_p), ("a_rowsum", c_void_p),
        ("alpha", c_float), ("relu", c_int32), ("accumulate", c_int32), ("splitk", c_int32),
        ("precision", c_int32), ("drop_p", c_float), ("drop_seed", c_uint64), ("seed_dev", c_void_p),
        ("groups", c_int32), ("a_gs", c_int64), ("b_gs", c_int64), ("c_gs", c_int64), ("bias_gs", c_int64),
        ("r_gs", c_int64), ("out_slope", c_float), ("out_act", c_int32), ("gate", c_void_p), ("gate_slope", c_float),
        ("z_taps", c_int32), ("c_tap", c_int64),
    ]


class ConvArgs(Structure):
    """kantts_conv_args (include/kantts_hip.h); ``in_`` is the C field ``in``."""
    _fields_ = [
        ("in_", c_void_p), ("in_gate", c_void_p), ("w", c_void_p), ("out", c_void_p), ("bias", c_void_p),
        ("res", c_void_p), ("out_gate", c_void_p),
        ("B", c_int32), ("Tsrc", c_int32), ("Tdst", c_int32), ("Cin_tot", c_int32), ("Ntot", c_int32),
        ("CR", c_int32), ("NG", c_int32), ("groups", c_int32), ("K", c_int32),
        ("in_mul", c_int32), ("in_add", c_int32), ("in_kstep", c_int32), ("in_div", c_int32), ("phases", c_int32), ("inner", c_int32), ("up", c_int32),
        ("in_slope", c_float), ("in_act", c_int32), ("in_gate_slope", c_float),
        ("out_slope", c_float), ("out_act", c_int32), ("out_gate_slope", c_float),
        ("precision", c_int32),
    ]


class ConvWArgs(Structure):
    """kantts_convw_args (include/kantts_hip.h)."""
    _fields_ = [
        ("x", c_void_p), ("dy", c_void_p), ("dy_gate", c_void_p), ("dw", c_void_p), ("db", c_void_p),
        ("B", c_int32), ("Tsrc", c_int32), ("Tdst", c_int32), ("Cin_tot", c_int32), ("Ntot", c_int32),
        ("CR", c_int32), ("NG", c_int32), ("groups", c_int32), ("K", c_int32),
        ("stride", c_int32), ("dil", c_int32), ("pad", c_int32), ("inner", c_int32), ("up", c_int32),
        ("x_slope", c_float), ("x_act", c_int32), ("dy_gate_slope", c_float), ("precision", c_int32),
    ]


class ConvC1Args(Structure):
    """kantts_conv_c1_args (include/kantts_hip.h)."""
    _fields_ = [
        ("x", c_void_p), ("dx", c_void_p), ("y", c_void_p), ("gate", c_void_p), ("w", c_void_p), ("bias", c_void_p),
        ("dw", c_void_p), ("db", c_void_p),
        ("B", c_int32), ("Tsrc", c_int32), ("Tdst", c_int32), ("Cout", c_int32), ("K", c_int32), ("stride", c_int32),
        ("dil", c_int32), ("pad", c_int32), ("inner", c_int32),
        ("out_slope", c_float), ("out_act", c_int32), ("gate_slope", c_float), ("y_bf16", c_void_p),
    ]


class ConvN1Args(Structure):
    """kantts_conv_n1_args (include/kantts_hip.h)."""
    _fields_ = [
        ("x", c_void_p), ("dx", c_void_p), ("y", c_void_p), ("w", c_void_p), ("bias", c_void_p), ("dw", c_void_p),
        ("db", c_void_p),
        ("B", c_int32), ("Tsrc", c_int32), ("Tdst", c_int32), ("Cin", c_int32), ("K", c_int32), ("stride", c_int32),
        ("dil", c_int32), ("pad", c_int32), ("inner", c_int32), ("w_ks", c_int32), ("w_cs", c_int32),
        ("in_slope", c_float), ("in_act", c_int32),
    ]


class CConvArgs(Structure):
    """kantts_cconv_args (include/kantts_hip.h); ``in_`` is the C field ``in``."""
    _fields_ = [
        ("in_", c_void_p), ("w", c_void_p), ("bias", c_void_p), ("res", c_void_p), ("out_gate", c_void_p),
        ("out", c_void_p), ("out_bf", c_void_p),
        ("B", c_int32), ("Tsrc", c_int32), ("Tdst", c_int32), ("Cin_tot", c_int32), ("Ntot", c_int32),
        ("CR", c_int32), ("NG", c_int32), ("groups", c_int32), ("K", c_int32),
        ("in_mul", c_int32), ("in_add", c_int32), ("in_kstep", c_int32), ("in_div", c_int32), ("phases", c_int32),
        ("inner", c_int32), ("up", c_int32),
        ("out_slope", c_float), ("out_act", c_int32), ("out_gate_slope", c_float), ("out_gate_bf16", c_int32),
        ("bf_slope", c_float), ("bf_act", c_int32), ("tile", c_int32), ("res_after_gate", c_int32),
    ]


class CConvWArgs(Structure):
    """kantts_cconvw_args (include/kantts_hip.h)."""
    _fields_ = [
        ("x", c_void_p), ("dy", c_void_p), ("dw", c_void_p), ("db", c_void_p),
        ("B", c_int32), ("Tsrc", c_int32), ("Tdst", c_int32), ("Cin_tot", c_int32), ("Ntot", c_int32),
        ("CR", c_int32), ("NG", c_int32), ("groups", c_int32), ("K", c_int32),
        ("stride", c_int32), ("dil", c_int32), ("pad", c_int32), ("inner", c_int32), ("up", c_int32),
        ("slices", c_int32), ("workspace", c_void_p), ("ws_floats", c_longlong),
    ]


BGEMM_MAX_SEG = 12


class BGemmSeg(Structure):
    _fields_ = [("a", c_void_p), ("b", c_void_p), ("lda", c_int64), ("ldb", c_int64), ("klen", c_int32),
                ("a_shift", c_int32)]


class BGemmArgs(Structure):
    """kantts_bgemm_args (include/kantts_hip.h)."""
    _fields_ = [
        ("seg", BGemmSeg * BGEMM_MAX_SEG),
        ("nseg", c_int32), ("M", c_int32), ("N", c_int32), ("T", c_int32),
        ("a_f32", c_int32), ("b_kn", c_int32),
        ("c", c_void_p), ("ldc", c_int64), ("c_bf16", c_int32), ("relu", c_int32),
        ("bias", c_void_p), ("bias2", c_void_p), ("alpha", c_float), ("drop_p", c_float),
        ("drop_seed", c_uint64), ("seed_dev", c_void_p),
        ("res", c_void_p), ("ldr", c_int64),
        ("gate", c_void_p), ("ldg", c_int64), ("gate_bf16", c_int32), ("a_drop_p", c_float),
        ("a_drop_seed", c_uint64), ("a_drop_ld", c_int64),
        ("rowmask", c_void_p),
        ("ln_gamma", c_void_p), ("ln_beta", c_void_p), ("ln_out", c_void_p), ("ln_out_bf16", c_int32), ("ln_eps", c_float),
        ("ln_mean", c_void_p), ("ln_rstd", c_void_p),
    ]


class LnBwdArgs(Structure):
    """kantts_lnbwd_args (include/kantts_hip.h)."""
    _fields_ = [
        ("x", c_void_p), ("gamma", c_void_p), ("mean", c_void_p), ("rstd", c_void_p), ("dres", c_void_p),
        ("zero_rows", c_void_p), ("dx", c_void_p), ("dgamma_accum", c_void_p), ("dbeta_accum", c_void_p),
        ("part_rows", c_void_p),
    ]


LOSS_MAX_TERMS = 5


class LossTerm(Structure):
    """kantts_loss_term (include/kantts_hip.h)."""
    _fields_ = [("pred", c_void_p), ("target", c_void_p), ("lens", c_void_p), ("grad", c_void_p),
                ("B", c_int32), ("T", c_int32), ("C", c_int32), ("target_log1p", c_int32)]


ELOSS_MAX_TERMS = 64


class ElossTerm(Structure):
    """kantts_eloss_term (include/kantts_hip.h)."""
    _fields_ = [("a", c_void_p), ("b", c_void_p), ("grad", c_void_p), ("n", ctypes.c_longlong), ("target", ctypes.c_float),
                ("scale", ctypes.c_float), ("mode", c_int32), ("out", c_int32)]


WN_BWD_MAX = 64


class WnBwdArgs(Structure):
    """kantts_wn_bwd_args (include/kantts_hip.h)."""
    _fields_ = [("dw", c_void_p * WN_BWD_MAX), ("desc", c_int32 * WN_BWD_MAX), ("tile0", c_int32 * (WN_BWD_MAX + 1)),
                ("nl", c_int32)]


class BGemmTnArgs(Structure):
    """kantts_bgemm_tn_args (include/kantts_hip.h)."""
    _fields_ = [
        ("a", c_void_p), ("b", c_void_p), ("lda", c_int64), ("ldb", c_int64),
        ("M", c_int32), ("N", c_int32), ("K", c_int32), ("T", c_int32),
        ("a_f32", c_int32), ("b_f32", c_int32),
        ("ntaps", c_int32), ("shift0", c_int32), ("shift_step", c_int32), ("slices", c_int32),
        ("c", c_void_p), ("c_ns", c_int64), ("c_ks", c_int64), ("c_ts", c_int64),
        ("db", c_void_p), ("alpha", c_float), ("a_drop_p", c_float),
        ("a_drop_seed", c_uint64), ("seed_dev", c_void_p),
    ]


class FfnArgs(Structure):
    """kantts_ffn_args (include/kantts_hip.h)."""
    _fields_ = [
        ("x", c_void_p), ("ldx", c_int64), ("x_f32", c_int32),
        ("M", c_int32), ("T", c_int32), ("K1", c_int32), ("F", c_int32), ("N", c_int32), ("KT", c_int32), ("pad", c_int32),
        ("w1", c_void_p), ("w2", c_void_p), ("bias1", c_void_p), ("bias2", c_void_p),
        ("relu", c_int32), ("alpha1", c_float), ("drop1_p", c_float), ("drop2_p", c_float), ("xdrop_p", c_float),
        ("drop1_seed", c_uint64), ("drop2_seed", c_uint64), ("xdrop_seed", c_uint64), ("seed_dev", c_void_p),
        ("gate", c_void_p), ("rowmask1", c_void_p), ("rowmask2", c_void_p), ("xrowmask", c_void_p),
        ("t_out", c_void_p), ("res", c_void_p), ("ldr", c_int64), ("y", c_void_p), ("ldy", c_int64), ("y_bf16", c_int32),
        ("KT2", c_int32), ("s2_first", c_int32), ("s2_step", c_int32),
        ("ln_gamma", c_void_p), ("ln_beta", c_void_p), ("ln_out", c_void_p), ("ln_out_bf16", c_int32), ("ln_eps", c_float),
        ("ln_mean", c_void_p), ("ln_rstd", c_void_p),
    ]


class PncaBlockArgs(Structure):
    """kantts_pnca_block_args (include/kantts_hip.h)."""
    _fields_ = [
        ("x", c_void_p), ("xn", c_void_p), ("hkv", c_void_p), ("ldh", c_int64),
        ("B", c_int32), ("L", c_int32), ("H", c_int32), ("C", c_int32), ("F", c_int32),
        ("lens", c_void_p), ("bw_dev", c_void_p), ("bw_x", c_int32), ("bw_h", c_int32), ("rowmask", c_void_p),
        ("wqkv", c_void_p), ("bqkv", c_void_p), ("wfcx", c_void_p), ("wfch", c_void_p), ("bfcx", c_void_p), ("bfch", c_void_p),
        ("ln1_gamma", c_void_p), ("ln1_beta", c_void_p), ("ln1_eps", c_float),
        ("w1", c_void_p), ("w2", c_void_p), ("bias1", c_void_p), ("bias2", c_void_p),
        ("att_p", c_float), ("fc_p", c_float), ("drop1_p", c_float), ("drop2_p", c_float),
        ("seed_x", c_uint64), ("seed_h", c_uint64), ("fc_seed", c_uint64), ("drop1_seed", c_uint64), ("drop2_seed", c_uint64),
        ("seed_dev", c_void_p),
        ("qkv", c_void_p), ("ox", c_void_p), ("oh", c_void_p), ("lse_x", c_void_p), ("lse_h", c_void_p), ("y1", c_void_p),
        ("xn1", c_void_p), ("mean1", c_void_p), ("rstd1", c_void_p), ("hid", c_void_p), ("out", c_void_p),
        ("ln2_gamma", c_void_p), ("ln2_beta", c_void_p), ("ln2_out", c_void_p), ("ln2_out_bf16", c_int32), ("ln2_eps", c_float),
        ("ln2_mean", c_void_p), ("ln2_rstd", c_void_p),
    ]


class PncaBlockBwdArgs(Structure):
    """kantts_pnca_block_bwd_args (include/kantts_hip.h)."""
    _fields_ = [
        ("dy", c_void_p), ("hid", c_void_p), ("y1", c_void_p), ("mean1", c_void_p), ("rstd1", c_void_p), ("ln1_gamma", c_void_p),
        ("rowmask", c_void_p), ("M", c_int32), ("C", c_int32), ("F", c_int32),
        ("wt2", c_void_p), ("wt1", c_void_p), ("wfcxT", c_void_p), ("wfchT", c_void_p),
        ("alpha1", c_float), ("drop2_p", c_float), ("fc_p", c_float), ("drop2_seed", c_uint64), ("fc_seed", c_uint64),
        ("seed_dev", c_void_p),
        ("dz", c_void_p), ("g1", c_void_p), ("d_ox", c_void_p), ("d_oh", c_void_p), ("ws", c_void_p), ("ws_floats", c_int64),
    ]


class PncaAttnBwdArgs(Structure):
    """kantts_pnca_attn_bwd_args (include/kantts_hip.h)."""
    _fields_ = [
        ("qkv", c_void_p), ("hkv", c_void_p), ("ldh", c_int64), ("ox", c_void_p), ("oh", c_void_p), ("d_ox", c_void_p),
        ("d_oh", c_void_p), ("lse_x", c_void_p), ("lse_h", c_void_p),
        ("B", c_int32), ("L", c_int32), ("H", c_int32), ("C", c_int32),
        ("lens", c_void_p), ("bw_dev", c_void_p), ("bw_x", c_int32), ("bw_h", c_int32), ("att_p", c_float),
        ("seed_x", c_uint64), ("seed_h", c_uint64), ("seed_dev", c_void_p),
        ("wqkvT", c_void_p), ("x", c_void_p), ("mean0", c_void_p), ("rstd0", c_void_p), ("ln0_gamma", c_void_p),
        ("dres", c_void_p), ("zero_rows", c_void_p),
        ("dqkv", c_void_p), ("dhkv", c_void_p), ("lddh", c_int64), ("dx", c_void_p), ("ws", c_void_p), ("ws_floats", c_int64),
    ]


class PlanArgs(Structure):
    """kantts_plan_args (include/kantts_hip.h)."""
    _fields_ = [("in_lens", c_void_p), ("out_lens", c_void_p), ("dur", c_void_p), ("mel", c_void_p), ("pos", c_void_p),
                ("inv_ts", c_void_p),
                ("B", c_int32), ("N", c_int32), ("T_mel", c_int32), ("Tp", c_int32), ("max_len", c_int32), ("r", c_int32),
                ("d_mel", c_int32), ("depth", c_int32),
                ("in_l64", c_void_p), ("in_l32", c_void_p), ("in_mask", c_void_p), ("out_l64", c_void_p), ("out_l32", c_void_p),
                ("out_mask", c_void_p), ("lfr_l64", c_void_p), ("lfr_l32", c_void_p), ("lfr_mask", c_void_p),
                ("valid", c_void_p), ("pos_enc", c_void_p), ("prev", c_void_p), ("bw_val", c_void_p), ("bw_dev", c_void_p),
                ("dec_input", c_void_p)]


class DecodeArgs(Structure):
    """kantts_decode_args (include/kantts_hip.h)."""
    _fields_ = [("w", c_void_p), ("f", c_void_p), ("memory", c_void_p), ("hkv", c_void_p), ("xkv", c_void_p),
                ("out", c_void_p), ("lens", c_void_p), ("bw_seq", c_void_p),
                ("B", c_int32), ("L", c_int32), ("d_mem", c_int32), ("d_mel", c_int32), ("d_out", c_int32),
                ("n_layer", c_int32), ("bw", c_int32), ("in_scale", c_float), ("eps", c_float)]


class DurArArgs(Structure):
    """kantts_durar_args (include/kantts_hip.h)."""
    _fields_ = [("w", c_void_p), ("f", c_void_p), ("gc", c_void_p), ("out", c_void_p), ("lens", c_void_p),
                ("B", c_int32), ("T", c_int32)]


class EncAttnArgs(Structure):
    """kantts_enc_attn_args (include/kantts_hip.h)."""
    _fields_ = [("x", c_void_p), ("xn", c_void_p), ("lens", c_void_p), ("rowmask", c_void_p), ("wqkv", c_void_p),
                ("bqkv", c_void_p), ("wfc", c_void_p), ("bfc", c_void_p), ("ln1_gamma", c_void_p), ("ln1_beta", c_void_p),
                ("ln1_eps", c_float), ("att_p", c_float), ("fc_p", c_float), ("att_seed", c_uint64), ("fc_seed", c_uint64),
                ("seed_dev", c_void_p), ("qkv", c_void_p), ("o", c_void_p), ("lse", c_void_p), ("y1", c_void_p),
                ("xn1", c_void_p), ("xn1_bf16", c_int32), ("mean1", c_void_p), ("rstd1", c_void_p), ("B", c_int32),
                ("L", c_int32)]


class CtcArgs(Structure):
    """kantts_ctc_args (include/kantts_hip.h)."""
    _fields_ = [("logits", c_void_p), ("in_lens", c_void_p), ("out_lens", c_void_p), ("ws", c_void_p), ("loss", c_void_p),
                ("grad", c_void_p), ("B", c_int32), ("T1", c_int32), ("T2", c_int32), ("blank", c_float),
                ("grad_scale", c_float)]


ROWSUM_MAX = 32


class RowSumArgs(Structure):
    """kantts_rowsum_args (include/kantts_hip.h)."""
    _fields_ = [("n", c_int32), ("cols", c_int32), ("split", c_int32), ("rows", c_int32 * ROWSUM_MAX),
                ("src", c_void_p * ROWSUM_MAX), ("dst0", c_void_p * ROWSUM_MAX), ("dst1", c_void_p * ROWSUM_MAX)]


class FragMajorDesc(Structure):
    """kantts_fragmajor_desc (include/kantts_hip.h)."""
    _fields_ = [("src_off", c_int64), ("dst_off", c_int64), ("sr", c_int64), ("sk", c_int64), ("R", c_int32),
                ("K", c_int32)]


class TapMajorDesc(Structure):
    _fields_ = [("src_off", c_int64), ("dst_off", c_int64), ("N", c_int32), ("Cin", c_int32), ("KT", c_int32),
                ("pad_", c_int32)]


def available() -> bool:
    return os.path.exists(LIB_PATH)


def lib():
    """Load (once) and return the C-ABI library; fails loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libkantts_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C kan-tts_amd/csrc`. There is no CPU fallback." % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        L.kantts_abi_version.restype = c_int
        L.kantts_target_arch.restype = c_char_p
        L.kantts_gemm_seg_launch.argtypes = [POINTER(GemmArgs), c_void_p]
        i, f, p, ll, u64 = c_int, c_float, c_void_p, c_longlong, c_uint64
        L.kantts_layernorm_fwd.argtypes = [p, p, p, p, p, p, i, i, f, p]
        L.kantts_layernorm_bwd.argtypes = [p, p, p, p, p, p, p, p, i, i, p]
        L.kantts_attn_fwd.argtypes = [p, p, p, i, i, i, p, i, p, p, p, p, i, i, i, i, i, i, f, u64, p, p]
        L.kantts_attn_bwd.argtypes = [p, p, p, i, i, i, p, i, p, i, p, p, p, p, p, i, i, i, i, p, p, i, i, i, i, i,
                                      i, f, u64, p, p]
        L.kantts_pnca_attn_fwd.argtypes = [p, p, i, p, p, p, p, p, p, i, i, i, i, i, i, f, u64, u64, p, p]
        L.kantts_pnca_attn_bwd.argtypes = [p, p, i, p, p, p, p, p, p, p, p, p, p, p, i, i, i, i, i, i, f, u64, u64, p, p]
        L.kantts_lstm_fwd.argtypes = [p, p, p, p, p, p, p, i, i, i, i, i, i, p]
        L.kantts_attn_decode.argtypes = [p, p, p, i, i, i, p, i, p, p, i, i, i, i, i, i, i, p]
        L.kantts_lstm_cell.argtypes = [p, p, p, p, i, i, p]
        L.kantts_mas_width1.argtypes = [p, p, p, p, p, i, i, i, p]
        L.kantts_align_attn_fwd.argtypes = [p, p, p, p, p, p, i, i, i, i, p]
        L.kantts_align_attn_bwd.argtypes = [p, p, p, p, p, p, p, p, p, p, i, i, i, i, p]
        L.kantts_lstm_bwd.argtypes = [p, p, p, p, p, p, i, i, i, i, i, i, p]
        L.kantts_embed_sum_fwd.argtypes = [POINTER(c_void_p), i, p, p, p, p, i, i, i, f, p]
        L.kantts_embed_sum_bwd.argtypes = [POINTER(c_void_p), i, p, p, i, i, f, p]
        L.kantts_lr_index.argtypes = [p, p, p, p, p, p, i, i, i, p]
        L.kantts_lr_gather_fwd.argtypes = [p, p, p, p, i, i, i, i, i, i, p]
        L.kantts_lr_gather_bwd.argtypes = [p, p, p, p, i, i, i, i, i, i, i, p]
        L.kantts_fsmn_dwconv_fwd.argtypes = [p, p, p, p, p, i, i, i, i, i, p]
        L.kantts_fsmn_dwconv_bwd.argtypes = [p, p, p, p, p, p, p, ll, i, i, i, i, i, p]
        L.kantts_fsmn_dwconv_bwd_ws.argtypes = [i, i, i, i]
        L.kantts_fsmn_dwconv_bwd_ws.restype = ll
        L.kantts_masked_l1.argtypes = [p, p, p, p, p, i, i, i, p]
        L.kantts_sumsq.argtypes = [p, p, ll, p]
        L.kantts_elem_loss.argtypes = [p, p, f, i, f, p, p, ll, p]
        L.kantts_adam_step.argtypes = [p, p, p, p, ll, f, f, f, f, f, f, f, p, f, p, p]
        L.kantts_melspec_fwd.argtypes = [p, i, i, i, i, i, i, p, p, f, p, p, p, p, i, f, p, p, p]
        L.kantts_melspec_norm_fwd.argtypes = [p, i, i, i, i, i, i, p, p, f, p, p, p, p, i, f, f, f, f, i, p, p, p]
        L.kantts_melspec_bwd.argtypes = [p, p, i, i, i, i, i, i, p, p, f, p, p, p, p, i, f, p, p]
        L.kantts_melspec_norm_fwd_fm.argtypes = [p, i, i, i, i, i, i, p, p, f, p, p, p, p, i, f, f, f, f, i, i, p, p, p]
        L.kantts_melspec_bwd_fm.argtypes = [p, p, i, i, i, i, i, i, p, p, f, p, p, p, p, i, f, i, p, p]
        L.kantts_weight_norm_fwd.argtypes = [p, p, p, i, i, p]
        L.kantts_weight_norm_bwd.argtypes = [p, p, p, p, p, i, i, p]
        L.kantts_weight_norm_strided_fwd.argtypes = [p, p, p, i, i, i, ll, ll, ll, p]
        L.kantts_weight_norm_strided_bwd.argtypes = [p, p, p, p, p, i, i, i, ll, ll, ll, p]
        L.kantts_sinadd_fwd.argtypes = [p, p, ll, p]
        L.kantts_sinadd_bwd.argtypes = [p, p, p, ll, p]
        L.kantts_conv_win_launch.argtypes = [POINTER(ConvArgs), c_void_p]
        L.kantts_conv_wgrad_launch.argtypes = [POINTER(ConvWArgs), c_void_p]
        L.kantts_conv_c1_launch.argtypes = [POINTER(ConvC1Args), c_int, c_void_p]
        L.kantts_pnca_decode_step.argtypes = [p, i, p, p, p, p, p, p, i, i, i, i, i, p, i, p]
        L.kantts_step_rows.argtypes = [p, p, i, i, ll, ll, ll, ll, i, p, p]
        L.kantts_step_rowmask.argtypes = [p, p, i, i, p, p]
        L.kantts_upsample_stream.argtypes = [p, p, p, p, p, i, i, i, i, i, f, i, p]
        L.kantts_sinadd_lrelu_fwd.argtypes = [p, p, p, f, ll, p]
        L.kantts_dropout2_add.argtypes = [p, p, p, ll, f, c_uint64, f, c_uint64, p, p]
        L.kantts_sumsq_det.argtypes = [p, p, p, ll, ll, p]
        L.kantts_stft_mag_bwd.argtypes = [p, p, i, i, i, i, i, i, p, p, f, p, p]
        L.kantts_bgemm_nt.argtypes = [POINTER(BGemmArgs), c_void_p]
        L.kantts_ffn_pair.argtypes = [POINTER(FfnArgs), c_void_p]
        L.kantts_pnca_block_fwd.argtypes = [POINTER(PncaBlockArgs), c_void_p]
        L.kantts_pnca_block_bwd.argtypes = [POINTER(PncaBlockBwdArgs), c_void_p]
        L.kantts_rows_sum_many.argtypes = [POINTER(RowSumArgs), c_void_p]
        L.kantts_teacher_plan.argtypes = [POINTER(PlanArgs), c_void_p]
        L.kantts_pnca_attn_qkv_bwd.argtypes = [POINTER(PncaAttnBwdArgs), c_void_p]
        L.kantts_copy_roof.argtypes = [c_void_p, ctypes.c_longlong, c_void_p, ctypes.c_longlong, c_void_p]
        L.kantts_launch_tuning.argtypes = [c_int, c_int, c_int]
        L.kantts_pnca_decode_run.argtypes = [POINTER(DecodeArgs), c_void_p]
        L.kantts_dur_ar_run.argtypes = [POINTER(DurArArgs), c_void_p]
        L.kantts_dur_ar_run_f32.argtypes = [POINTER(DurArArgs), c_void_p]
        L.kantts_enc_attn_fwd.argtypes = [POINTER(EncAttnArgs), c_void_p]
        L.kantts_ctc_attn.argtypes = [POINTER(CtcArgs), c_void_p]
        L.kantts_ctc_attn_workspace.argtypes = [c_int, c_int, c_int]
        L.kantts_ctc_attn_workspace.restype = c_longlong
        L.kantts_pnca_decode_blob_sizes.argtypes = [c_int, c_int, c_int, c_int, POINTER(ctypes.c_longlong),
                                                    POINTER(ctypes.c_longlong)]
        L.kantts_pnca_block_bwd_ws_floats.argtypes = [c_int]
        L.kantts_pnca_block_bwd_ws_floats.restype = ctypes.c_longlong
        L.kantts_fragmajor_bf16.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]
        L.kantts_bgemm_tn.argtypes = [POINTER(BGemmTnArgs), c_void_p]
        L.kantts_bgemm_tn_grouped.argtypes = [POINTER(BGemmTnArgs), c_int, p, p, p, p, p, c_void_p]
        L.kantts_cast_f32_bf16.argtypes = [p, p, ll, p]
        L.kantts_tapmajor_bf16.argtypes = [p, p, p, i, i, p]
        L.kantts_relu_gate_bf16.argtypes = [p, i, p, i, p, f, ll, p]
        L.kantts_ln128_fwd.argtypes = [p, p, p, p, i, p, p, i, f, p]
        L.kantts_ln128_bwd.argtypes = [p, i, p, p, p, p, p, p, p, p, i, p]
        L.kantts_bgemm_nt_lnbwd.argtypes = [POINTER(BGemmArgs), POINTER(LnBwdArgs), c_void_p]
        L.kantts_ln128_bwd_rows.argtypes = [p, i, p, p, p, p, p, p, p, p, p, i, p]
        L.kantts_cconv_launch.argtypes = [POINTER(CConvArgs), c_void_p]
        L.kantts_cconv_wgrad_launch.argtypes = [POINTER(CConvWArgs), c_void_p]
        L.kantts_cconv_wgrad_ws_floats.argtypes = [POINTER(CConvWArgs)]
        L.kantts_cconv_wgrad_ws_floats.restype = ll
        L.kantts_act_cast_bf16.argtypes = [p, p, i, p, i, f, ll, p]
        L.kantts_ragged_rows_f32.argtypes = [p, p, p, p, p, p, i, i, i, i, p]
        L.kantts_weight_norm_tap_images.argtypes = [p, p, p, p, p, i, i, i, i, p]
        L.kantts_weight_norm_table.argtypes = [p, p, p, p, p, i, i, p]
        L.kantts_weight_norm_table_bwd.argtypes = [p, p, p, POINTER(WnBwdArgs), p]
        L.kantts_masked_l1_many.argtypes = [POINTER(LossTerm), i, p, p]
        L.kantts_scale_many.argtypes = [POINTER(c_void_p), POINTER(ll), i, p, p]
        L.kantts_conv_n1_launch.argtypes = [POINTER(ConvN1Args), i, p]
        L.kantts_elem_loss_many.argtypes = [POINTER(ElossTerm), i, p, p]
        L.kantts_mean_many.argtypes = [POINTER(c_void_p), i, f, p, p, f, ll, p]
        L.kantts_scale_to_many.argtypes = [p, f, POINTER(c_void_p), i, ll, p]
        L.kantts_ragged_rows_i64.argtypes = [p, p, p, p, p, p, i, i, i, i, p]
        _lib = L
    return _lib


EXPORTED_SYMBOLS = [
    "kantts_abi_version", "kantts_target_arch", "kantts_gemm_seg_launch", "kantts_layernorm_fwd",
    "kantts_layernorm_bwd", "kantts_attn_fwd", "kantts_attn_bwd", "kantts_pnca_attn_fwd", "kantts_pnca_attn_bwd", "kantts_lstm_fwd", "kantts_lstm_bwd",
    "kantts_embed_sum_fwd", "kantts_embed_sum_bwd", "kantts_lr_index", "kantts_lr_gather_fwd",
    "kantts_lr_gather_bwd", "kantts_fsmn_dwconv_fwd", "kantts_fsmn_dwconv_bwd", "kantts_fsmn_dwconv_bwd_ws", "kantts_masked_l1",
    "kantts_sumsq", "kantts_elem_loss", "kantts_adam_step", "kantts_melspec_fwd", "kantts_melspec_norm_fwd", "kantts_melspec_bwd", "kantts_melspec_norm_fwd_fm", "kantts_melspec_bwd_fm", "kantts_weight_norm_fwd", "kantts_weight_norm_bwd", "kantts_weight_norm_strided_fwd", "kantts_weight_norm_strided_bwd",
    "kantts_sinadd_fwd", "kantts_sinadd_bwd", "kantts_conv_win_launch", "kantts_conv_wgrad_launch", "kantts_conv_c1_launch", "kantts_attn_decode",
    "kantts_lstm_cell", "kantts_mas_width1", "kantts_align_attn_fwd", "kantts_align_attn_bwd",
    "kantts_bgemm_nt", "kantts_ffn_pair", "kantts_fragmajor_bf16", "kantts_bgemm_tn", "kantts_cast_f32_bf16", "kantts_tapmajor_bf16", "kantts_relu_gate_bf16",
    "kantts_ln128_fwd", "kantts_ln128_bwd", "kantts_ln128_bwd_rows", "kantts_bgemm_nt_lnbwd", "kantts_stft_mag_bwd", "kantts_bgemm_tn_grouped", "kantts_sumsq_det",
    "kantts_pnca_decode_step", "kantts_step_rows", "kantts_step_rowmask", "kantts_upsample_stream",
    "kantts_sinadd_lrelu_fwd", "kantts_dropout2_add",
    "kantts_cconv_launch", "kantts_cconv_wgrad_launch", "kantts_cconv_wgrad_ws_floats", "kantts_act_cast_bf16",
    "kantts_ragged_rows_f32", "kantts_ragged_rows_i64", "kantts_weight_norm_tap_images",
    "kantts_weight_norm_table", "kantts_weight_norm_table_bwd", "kantts_masked_l1_many", "kantts_scale_many",
    "kantts_mean_many", "kantts_scale_to_many", "kantts_elem_loss_many", "kantts_conv_n1_launch",
    "kantts_pnca_block_fwd", "kantts_pnca_block_bwd", "kantts_pnca_block_bwd_ws_floats", "kantts_rows_sum_many",
    "kantts_melspec_tuning", "kantts_teacher_plan", "kantts_copy_roof", "kantts_pnca_attn_qkv_bwd",
    "kantts_pnca_decode_run", "kantts_pnca_decode_blob_sizes", "kantts_dur_ar_run", "kantts_dur_ar_run_f32", "kantts_ctc_attn", "kantts_ctc_attn_workspace", "kantts_enc_attn_fwd",
    "kantts_launch_tuning",
]


_launch_tuning = [None]


def apply_launch_tuning():
    """KANTTS_TN_TILE / KANTTS_TN_SLICES / KANTTS_C1_WGRAD_WGS (sweep and test switches of the weight-gradient launches) are
    read HERE, in the host layer, and handed to the library through kantts_launch_tuning when they change: the C ABI
    consults no environment for them.  Called by the wrappers of the launches they shape."""
    L = lib()
    env = os.environ
    want = (id(L), int(env.get("KANTTS_TN_TILE", "0") or 0), int(env.get("KANTTS_TN_SLICES", "0") or 0),
            int(env.get("KANTTS_C1_WGRAD_WGS", "0") or 0))
    if want != _launch_tuning[0]:
        rc = L.kantts_launch_tuning(want[1], max(want[2], 0), max(want[3], 0))
        if rc != 0:
            raise RuntimeError("libkantts_hip: launch_tuning failed with code %d" % rc)
        _launch_tuning[0] = want


def check(rc, what):
    if rc != 0:
        raise RuntimeError("libkantts_hip: %s failed with code %d" % (what, rc))


def stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t, dtype=None):
    """Device pointer of a tensor (None -> NULL); refuses host tensors -- no CPU fallback."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("kantts HIP op called with a CPU tensor; the product path has no CPU fallback")
    if dtype is not None and t.dtype != dtype:
        raise TypeError("expected %s, got %s" % (dtype, t.dtype))
    return t.data_ptr()


# ----------------------------------------------------------------------------------------------
# global numerics mode of the contraction kernels: "fp32" (parity path) or "bf16" (throughput)
_precision = {"gemm": PREC_BF16 if os.environ.get("KANTTS_PRECISION", "fp32") == "bf16" else PREC_FP32}
if os.environ.get("KANTTS_GEMM_IMPL", "") == "ref":
    _precision["gemm"] = PREC_REF


def set_precision(mode: str):
    _precision["gemm"] = {"fp32": PREC_FP32, "bf16": PREC_BF16, "ref": PREC_REF}[mode]


def get_precision() -> str:
    return {PREC_FP32: "fp32", PREC_BF16: "bf16", PREC_REF: "ref"}[_precision["gemm"]]


class precision_scope:
    """``with precision_scope("fp32"): ...`` -- the contraction mode for the ops issued inside the block (the per-module
    precision override: in bf16 mode the token-level front of INFERENCE -- text encoder, variance adaptor, duration loop --
    runs fp32 so that the index tensors derived from it are bit-exact; KanTtsSAMBERT.forward).  ``None`` = leave as is."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.before = _precision["gemm"]
        if self.mode is not None:
            set_precision(self.mode)
        return self

    def __exit__(self, *exc):
        _precision["gemm"] = self.before
        return False


def make_seg(a, a_is, a_ks, b, b_js, b_ks, klen, ntaps=1, b_tap=0, a_tok_axis=0, a_shift0=0, a_shift_step=0,
             b_tok_axis=0, b_shift0=0, b_shift_step=0, a_gate=None, a_drop_p=0.0, a_drop_seed=0, a_map=None,
             b_map=None, a_leaky=None, b_leaky=None, a_gate_slope=0.0):
    """a / b / a_gate are (tensor, element_offset) pairs or tensors.  a_map / b_map: dict(inner, Tq, Tsrc,
    mul, div, up) extended token maps; a_leaky / b_leaky: LeakyReLU slope applied on load."""

    def addr(x):
        if x is None:
            return None
        if isinstance(x, tuple):
            return ptr(x[0], torch.float32) + 4 * int(x[1])
        return ptr(x, torch.float32)

    s = GemmSeg()
    s.a, s.a_gate, s.b = addr(a), addr(a_gate), addr(b)
    s.a_is, s.a_ks, s.b_js, s.b_ks, s.b_tap = int(a_is), int(a_ks), int(b_js), int(b_ks), int(b_tap)
    s.klen, s.ntaps = int(klen), int(ntaps)
    s.a_tok_axis, s.a_shift0, s.a_shift_step = int(a_tok_axis), int(a_shift0), int(a_shift_step)
    s.b_tok_axis, s.b_shift0, s.b_shift_step = int(b_tok_axis), int(b_shift0), int(b_shift_step)
    s.a_drop_p, s.a_drop_seed = float(a_drop_p), int(a_drop_seed)
    if a_map:
        s.a_inner, s.a_Tq, s.a_Tsrc = int(a_map.get("inner", 0)), int(a_map.get("Tq", 0)), int(a_map.get("Tsrc", 0))
        s.a_mul, s.a_div, s.a_up = int(a_map.get("mul", 0)), int(a_map.get("div", 0)), int(a_map.get("up", 0))
    if b_map:
        s.b_inner, s.b_Tq, s.b_Tsrc = int(b_map.get("inner", 0)), int(b_map.get("Tq", 0)), int(b_map.get("Tsrc", 0))
        s.b_mul, s.b_div, s.b_up = int(b_map.get("mul", 0)), int(b_map.get("div", 0)), int(b_map.get("up", 0))
    if a_leaky is not None:
        s.a_act, s.a_slope = 1, float(a_leaky)
    if b_leaky is not None:
        s.b_act, s.b_slope = 1, float(b_leaky)
    s.a_gate_slope = float(a_gate_slope)
    return s


def _staging_mode(base, row_stride, k_stride, klen, rows, tok_axis, group_stride, gate=None):
    """Pick the operand staging layout of csrc/gemm.hip (0/1 scalar, 2 float4 along k, 3 float4 along rows)."""
    aligned = (base % 16 == 0) and (gate is None or gate % 16 == 0) and (group_stride % 4 == 0)
    if k_stride == 1 and aligned and klen % 4 == 0 and row_stride % 4 == 0 and tok_axis != 2:
        return 2
    if row_stride == 1 and aligned and rows % 4 == 0 and k_stride % 4 == 0 and tok_axis != 1:
        return 3
    if row_stride == 1 and k_stride != 1:
        return 1
    return 0


_gemm_log = {} if os.environ.get("KANTTS_GEMM_LOG") else None


def gemm(segs, M, N, c, c_is, c_js, bias=None, bias2=None, res=None, r_is=0, r_js=0, rowmask=None, kmask=None,
         a_rowsum=None, alpha=1.0, relu=False, accumulate=False, splitk=1, T=0, drop_p=0.0, drop_seed=0,
         precision=None, c_off=0, groups=1, a_gs=0, b_gs=0, c_gs=0, bias_gs=0, r_gs=0, out_leaky=None, gate=None,
         gate_slope=0.0, res_off=0, z_taps=0, c_tap=0):
    g = GemmArgs()
    assert 1 <= len(segs) <= GEMM_MAX_SEG
    for k, s in enumerate(segs):
        s.a_mode = _staging_mode(s.a or 0, s.a_is, s.a_ks, s.klen, M, s.a_tok_axis, a_gs, s.a_gate)
        s.b_mode = _staging_mode((s.b or 0) + 0, s.b_js, s.b_ks, s.klen, N, 1 if False else (2 if s.b_tok_axis == 2 else 0),
                                 b_gs)
        if s.b_mode == 2 and (s.b_tap % 4 != 0 and s.ntaps > 1):
            s.b_mode = 0
        g.seg[k] = s
    g.nseg, g.M, g.N, g.T = len(segs), int(M), int(N), int(T)
    if _gemm_log is not None:
        s0 = segs[0]
        slow = any(s.a_mode < 2 or s.b_mode < 2 or s.a_inner > 1 or s.a_mul > 1 or s.a_div > 1 or s.a_up > 1 or
                   s.b_inner > 1 or s.b_mul > 1 or s.b_up > 1 for s in segs)
        key = ("generic" if slow else "fast", int(M), int(N), s0.klen, s0.ntaps, int(groups), int(splitk), int(z_taps),
               s0.a_mode, s0.b_mode)
        _gemm_log[key] = _gemm_log.get(key, 0) + 1
    g.c = ptr(c, torch.float32) + 4 * int(c_off)
    g.c_is, g.c_js = int(c_is), int(c_js)
    g.bias, g.bias2 = ptr(bias, torch.float32), ptr(bias2, torch.float32)
    g.res, g.r_is, g.r_js = ptr(res, torch.float32), int(r_is), int(r_js)
    if res is not None and res_off:
        g.res = g.res + 4 * int(res_off)
    g.groups, g.a_gs, g.b_gs, g.c_gs, g.bias_gs, g.r_gs = int(groups), int(a_gs), int(b_gs), int(c_gs), int(bias_gs), int(r_gs)
    if out_leaky is not None:
        g.out_act, g.out_slope = 1, float(out_leaky)
    if gate is not None:
        g.gate = ptr(gate, torch.float32) + 4 * int(c_off)
        g.gate_slope = float(gate_slope)
    g.z_taps, g.c_tap = int(z_taps), int(c_tap)
    g.rowmask = ptr(rowmask)
    g.kmask = ptr(kmask)
    g.a_rowsum = ptr(a_rowsum, torch.float32)
    g.alpha, g.relu, g.accumulate, g.splitk = float(alpha), int(bool(relu)), int(bool(accumulate)), int(splitk)
    g.precision = _precision["gemm"] if precision is None else precision
    g.drop_p, g.drop_seed = float(drop_p), int(drop_seed)
    g.seed_dev = rng_ptr(c.device) if (drop_p > 0 or any(s.a_drop_p > 0 for s in segs)) else None
    if _profile is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib().kantts_gemm_seg_launch(ctypes.byref(g), stream()), "gemm_seg")
        e1.record()
        _profile.append((e0, e1, 2.0 * M * N * max(1, groups) * sum(s.klen * s.ntaps for s in segs)))
        return
    check(lib().kantts_gemm_seg_launch(ctypes.byref(g), stream()), "gemm_seg")


E_UNSUPPORTED = -2


# ----------------------------------------------------------------------------------------------
# bf16-operand contractions (csrc/gemm_bf16.hip)
def _esz(t):
    return 2 if t.dtype == torch.bfloat16 else 4


def _addr(x):
    """tensor or (tensor, element offset) -> device address"""
    if isinstance(x, tuple):
        return ptr(x[0]) + _esz(x[0]) * int(x[1])
    return ptr(x)


def bgemm_nt(segs, M, N, c, ldc, *, T=0, b_kn=False, bias=None, bias2=None, alpha=1.0, relu=False, drop_p=0.0,
             drop_seed=0, res=None, ldr=0, gate=None, ldg=0, rowmask=None, a_drop_p=0.0, a_drop_seed=0, a_drop_ld=0,
             ln=None, lnb=None, c_bf16=None):
    """segs: list of (a, lda, b, ldb, klen, a_shift) with a / b tensors or (tensor, element offset).  All A operands share
    one dtype (fp32 or bf16); B operands are bf16.  Returns False when the library declines the shape (caller falls
    back to the segmented GEMM).  ``ln`` = (gamma, beta, eps, out (M,128) bf16 / fp32, mean (M), rstd (M)): LayerNorm of
    the output rows in the epilogue (N == 128, fp32 output).
    ``lnb`` = (x, gamma, mean, rstd, dres | None, zero_rows | None, dx, dgamma, dbeta): the epilogue is the BACKWARD of a
    LayerNorm(128) whose output gradient is this contraction's result (kantts_bgemm_nt_lnbwd); ``c`` may then be None (the
    result itself is not stored) with ``c_bf16`` saying whether it is rounded to bf16 first, as the two-launch form would."""
    g = BGemmArgs()
    assert 1 <= len(segs) <= BGEMM_MAX_SEG
    a0 = segs[0][0][0] if isinstance(segs[0][0], tuple) else segs[0][0]
    for k, (a, lda, b, ldb, klen, a_shift) in enumerate(segs):
        at = a[0] if isinstance(a, tuple) else a
        bt = b[0] if isinstance(b, tuple) else b
        if at.dtype != a0.dtype or bt.dtype != torch.bfloat16:
            raise TypeError("bgemm_nt: mixed A dtypes or a non-bf16 B operand")
        s = g.seg[k]
        s.a, s.b, s.lda, s.ldb, s.klen, s.a_shift = _addr(a), _addr(b), int(lda), int(ldb), int(klen), int(a_shift)
    g.nseg, g.M, g.N, g.T = len(segs), int(M), int(N), int(T)
    g.a_f32, g.b_kn = int(a0.dtype == torch.float32), int(bool(b_kn))
    if c is None:
        assert lnb is not None and c_bf16 is not None
        g.c, g.ldc, g.c_bf16 = None, int(ldc), int(bool(c_bf16))
    else:
        g.c, g.ldc, g.c_bf16 = _addr(c), int(ldc), int((c[0] if isinstance(c, tuple) else c).dtype == torch.bfloat16)
    g.relu = int(bool(relu))
    g.bias, g.bias2 = ptr(bias, torch.float32), ptr(bias2, torch.float32)
    g.alpha, g.drop_p, g.drop_seed = float(alpha), float(drop_p), int(drop_seed)
    g.res, g.ldr = ptr(res, torch.float32), int(ldr)
    if gate is not None:
        g.gate, g.ldg, g.gate_bf16 = ptr(gate), int(ldg), int(gate.dtype == torch.bfloat16)
    g.a_drop_p, g.a_drop_seed, g.a_drop_ld = float(a_drop_p), int(a_drop_seed), int(a_drop_ld)
    g.rowmask = ptr(rowmask)
    if ln is not None:
        gamma, beta, eps, ln_out, ln_mean, ln_rstd = ln
        g.ln_gamma, g.ln_beta, g.ln_eps = ptr(gamma, torch.float32), ptr(beta, torch.float32), float(eps)
        g.ln_out, g.ln_out_bf16 = ptr(ln_out), int(ln_out.dtype == torch.bfloat16)
        g.ln_mean, g.ln_rstd = ptr(ln_mean, torch.float32), ptr(ln_rstd, torch.float32)
    dev = a0.device
    g.seed_dev = rng_ptr(dev) if (drop_p > 0 or a_drop_p > 0) else None
    if _profile is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if lnb is not None:
        x, gamma, mean, rstd, dres, zero_rows, dx, dgamma, dbeta = lnb[:9]
        l = LnBwdArgs()
        # optional tenth element: (ceil(M / 32), 256) partial rows of [dgamma | dbeta] instead of atomics into them
        l.part_rows = ptr(lnb[9], torch.float32) if len(lnb) > 9 else None
        l.x, l.gamma, l.mean, l.rstd = (ptr(t, torch.float32) for t in (x, gamma, mean, rstd))
        l.dres, l.zero_rows = ptr(dres, torch.float32), ptr(zero_rows, torch.uint8)
        l.dx, l.dgamma_accum, l.dbeta_accum = (ptr(t, torch.float32) for t in (dx, dgamma, dbeta))
        rc = lib().kantts_bgemm_nt_lnbwd(ctypes.byref(g), ctypes.byref(l), stream())
    else:
        rc = lib().kantts_bgemm_nt(ctypes.byref(g), stream())
    if rc == E_UNSUPPORTED:
        return False
    check(rc, "bgemm_nt")
    if _profile is not None:
        e1.record()
        flops = 2.0 * M * N * sum(sg[4] for sg in segs)
        _profile.append((e0, e1, flops))
        # algorithmic bytes of the launch: every A segment, the weight slices, the result (and what the epilogue reads /
        # writes beside it), each touched once at its storage dtype
        ea = 4 if a0.dtype == torch.float32 else 2
        nb = sum(M * sg[4] * ea + N * sg[4] * 2 for sg in segs)
        if c is not None:
            nb += M * N * (2 if g.c_bf16 else 4)
        if res is not None:
            nb += M * N * 4
        if gate is not None:
            nb += M * N * (2 if gate.dtype == torch.bfloat16 else 4)
        if ln is not None:
            nb += M * N * (2 if ln[3].dtype == torch.bfloat16 else 4)
        if lnb is not None:
            nb += M * 128 * 4 * (3 if lnb[4] is not None else 2)  # x in, dx out (+ the residual branch's gradient in)
        _profile_families.append(("bgemm_nt", e0, e1, flops, float(nb)))
    return True


def ffn_pair(x, w1, w2, y, *, M, T, F, KT=1, pad=0, bias1=None, bias2=None, relu=False, alpha1=1.0, drop1_p=0.0,
             drop1_seed=0, drop2_p=0.0, drop2_seed=0, xdrop_p=0.0, xdrop_seed=0, gate=None, rowmask1=None, rowmask2=None,
             xrowmask=None, t_out=None, res=None, KT2=1, s2_first=0, s2_step=0, ln=None):
    """Both contractions of a feed-forward block in one launch (csrc/ffn_pair.hip; kantts_ffn_pair in the header).
    x (M, 128) bf16 / fp32; w1 / w2: FRAGMENT-MAJOR bf16 images (ops_bf16.frag_major) of the (KT*F, 128) and (128, F)
    weight matrices; y (M, 128) fp32 / bf16; t_out bf16 (M, F).  KT2 = 3 (backward form): phase 2 sums three taps of the
    intermediate, y[m] = sum_t t[m + s2_first + t*s2_step] . w2[t]^T with w2 = three (128, F) images.  Returns False when
    the library declines the shape."""
    g = FfnArgs()
    g.x, g.ldx, g.x_f32 = ptr(x), int(x.shape[-1]), int(x.dtype == torch.float32)
    NY = int(y.shape[-1])
    g.M, g.T, g.K1, g.F, g.N, g.KT, g.pad = int(M), int(T), int(x.shape[-1]), int(F), NY, int(KT), int(pad)
    g.w1, g.w2 = ptr(w1, torch.bfloat16), ptr(w2, torch.bfloat16)
    g.bias1, g.bias2 = ptr(bias1, torch.float32), ptr(bias2, torch.float32)
    g.relu, g.alpha1 = int(bool(relu)), float(alpha1)
    g.drop1_p, g.drop2_p, g.xdrop_p = float(drop1_p), float(drop2_p), float(xdrop_p)
    g.drop1_seed, g.drop2_seed, g.xdrop_seed = int(drop1_seed), int(drop2_seed), int(xdrop_seed)
    g.seed_dev = rng_ptr(x.device) if (drop1_p > 0 or drop2_p > 0 or xdrop_p > 0) else None
    g.gate = ptr(gate, torch.bfloat16)
    g.rowmask1, g.rowmask2, g.xrowmask = ptr(rowmask1), ptr(rowmask2), ptr(xrowmask)
    g.t_out = ptr(t_out, torch.bfloat16)
    g.res, g.ldr = ptr(res, torch.float32), NY
    g.y, g.ldy, g.y_bf16 = ptr(y), NY, int(y.dtype == torch.bfloat16)
    g.KT2, g.s2_first, g.s2_step = int(KT2), int(s2_first), int(s2_step)
    if ln is not None:  # (gamma, beta, eps, out (M,128), mean (M), rstd (M)): LayerNorm of the output rows in the epilogue
        gamma, beta, eps, ln_out, ln_mean, ln_rstd = ln
        g.ln_gamma, g.ln_beta, g.ln_eps = ptr(gamma, torch.float32), ptr(beta, torch.float32), float(eps)
        g.ln_out, g.ln_out_bf16 = ptr(ln_out), int(ln_out.dtype == torch.bfloat16)
        g.ln_mean, g.ln_rstd = ptr(ln_mean, torch.float32), ptr(ln_rstd, torch.float32)
    if _profile is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = lib().kantts_ffn_pair(ctypes.byref(g), stream())
    if rc == E_UNSUPPORTED:
        return False
    check(rc, "ffn_pair")
    if _profile is not None:
        e1.record()
        fl = 2.0 * M * F * (x.shape[-1] * KT + NY * KT2)
        _profile.append((e0, e1, fl))
        # x + hidden tile (written forward / read as the gate and written as dz backward) + residual / output + weights
        nb = M * (x.shape[-1] * x.element_size() + F * 2 * (2 if gate is not None else 1) + NY * 4 * (2 if res is not None else 1)) \
            + 2 * F * (x.shape[-1] * KT + NY * KT2)
        _profile_families.append(("ffn_pair", e0, e1, fl, float(nb)))
    return True


def pnca_block_fwd(x, xn, hkv, ldh, B, L, *, lens, bw_dev, bw_x, bw_h, rowmask, wqkv, bqkv, wfcx, wfch, bfcx, bfch, ln1, w1, w2,
                   bias1, bias2, att_p, fc_p, drop1_p, drop2_p, seeds, qkv, ox, oh, lse_x, lse_h, y1, xn1, mean1, rstd1, hid, out,
                   ln2=None):
    """One PNCA decoder block forward in one launch (csrc/pnca_block.hip; kantts_pnca_block_fwd in the header).  x (M, 128)
    fp32, xn its LayerNorm (bf16), hkv: (M, >= 256) fp32 rows at pitch ``ldh``; w*: fragment-major bf16 images;
    ln1 = (gamma, beta, eps) of the feed-forward's LayerNorm; ln2 = (gamma, beta, eps, out, mean, rstd) of the consumer's (or
    None); seeds = (x band, memory band, output dropout, hidden dropout, feed-forward output dropout).  Returns False when
    the library declines (band width above 16)."""
    g = PncaBlockArgs()
    g.x, g.xn, g.hkv, g.ldh = ptr(x, torch.float32), ptr(xn, torch.bfloat16), ptr(hkv, torch.float32), int(ldh)
    g.B, g.L, g.H, g.C, g.F = int(B), int(L), 8, 128, 1024
    g.lens, g.bw_dev, g.bw_x, g.bw_h, g.rowmask = ptr(lens), ptr(bw_dev), int(bw_x), int(bw_h), ptr(rowmask)
    g.wqkv, g.bqkv = ptr(wqkv, torch.bfloat16), ptr(bqkv, torch.float32)
    g.wfcx, g.wfch = ptr(wfcx, torch.bfloat16), ptr(wfch, torch.bfloat16)
    g.bfcx, g.bfch = ptr(bfcx, torch.float32), ptr(bfch, torch.float32)
    g.ln1_gamma, g.ln1_beta, g.ln1_eps = ptr(ln1[0], torch.float32), ptr(ln1[1], torch.float32), float(ln1[2])
    g.w1, g.w2 = ptr(w1, torch.bfloat16), ptr(w2, torch.bfloat16)
    g.bias1, g.bias2 = ptr(bias1, torch.float32), ptr(bias2, torch.float32)
    g.att_p, g.fc_p, g.drop1_p, g.drop2_p = float(att_p), float(fc_p), float(drop1_p), float(drop2_p)
    g.seed_x, g.seed_h, g.fc_seed, g.drop1_seed, g.drop2_seed = (int(v) for v in seeds)
    g.seed_dev = rng_ptr(x.device) if (att_p > 0 or fc_p > 0 or drop1_p > 0 or drop2_p > 0) else None
    g.qkv, g.ox, g.oh = ptr(qkv, torch.float32), ptr(ox, torch.float32), ptr(oh, torch.float32)
    g.lse_x, g.lse_h, g.y1 = ptr(lse_x, torch.float32), ptr(lse_h, torch.float32), ptr(y1, torch.float32)
    g.xn1, g.mean1, g.rstd1 = ptr(xn1, torch.bfloat16), ptr(mean1, torch.float32), ptr(rstd1, torch.float32)
    g.hid, g.out = ptr(hid, torch.bfloat16), ptr(out, torch.float32)
    if ln2 is not None:
        gamma, beta, eps, ln_out, ln_mean, ln_rstd = ln2
        g.ln2_gamma, g.ln2_beta, g.ln2_eps = ptr(gamma, torch.float32), ptr(beta, torch.float32), float(eps)
        g.ln2_out, g.ln2_out_bf16 = ptr(ln_out), int(ln_out.dtype == torch.bfloat16)
        g.ln2_mean, g.ln2_rstd = ptr(ln_mean, torch.float32), ptr(ln_rstd, torch.float32)
    if _profile is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = lib().kantts_pnca_block_fwd(ctypes.byref(g), stream())
    if rc == E_UNSUPPORTED:
        return False
    check(rc, "pnca_block_fwd")
    if _profile is not None:
        e1.record()
        M = int(B) * int(L)
        # SURVEY 8(d)'s per-block count (dense attention: 8 B L^2 128) and the bytes the launch must move once: x, xn, memory
        # K|V in; qkv, contexts, y1, its normalised rows, hidden tile, output, next normalised rows out; the weights
        flops = 2.0 * M * 128 * 384 + 8.0 * B * L * L * 128 + 4.0 * M * 128 * 128 + 4.0 * M * 128 * 1024
        nbytes = M * (128 * 4 + 128 * 2 + 256 * 4 + 384 * 4 + 2 * 128 * 4 + 128 * 4 + 128 * 2 + 1024 * 2 + 128 * 4 + 128 * 2) + \
            2 * (384 * 128 + 2 * 128 * 128 + 2 * 128 * 1024)
        _profile.append((e0, e1, flops))
        _profile_families.append(("pnca_block_fwd", e0, e1, flops, float(nbytes)))
    return True


def pnca_block_bwd(dy, hid, y1, mean1, rstd1, gamma1, rowmask, wt2, wt1, wfcxT, wfchT, *, alpha1, drop2_p, drop2_seed, fc_p,
                   fc_seed, dz, g1, d_ox, d_oh):
    """The row-local half of a PNCA block's backward in one launch (csrc/pnca_block.hip; kantts_pnca_block_bwd in the
    header): feed-forward pair backward, LayerNorm backward + residual, input gradient of the output projection.  Returns the
    (workgroups, 256) partial rows of [dgamma | dbeta] of the LayerNorm: ``rows_sum_accum`` adds them into the gradients."""
    g = PncaBlockBwdArgs()
    g.dy, g.hid, g.y1 = ptr(dy, torch.float32), ptr(hid, torch.bfloat16), ptr(y1, torch.float32)
    g.mean1, g.rstd1, g.ln1_gamma = ptr(mean1, torch.float32), ptr(rstd1, torch.float32), ptr(gamma1, torch.float32)
    g.rowmask, g.M, g.C, g.F = ptr(rowmask), int(dy.shape[0]), 128, 1024
    g.wt2, g.wt1 = ptr(wt2, torch.bfloat16), ptr(wt1, torch.bfloat16)
    g.wfcxT, g.wfchT = ptr(wfcxT, torch.bfloat16), ptr(wfchT, torch.bfloat16)
    g.alpha1, g.drop2_p, g.fc_p = float(alpha1), float(drop2_p), float(fc_p)
    g.drop2_seed, g.fc_seed = int(drop2_seed), int(fc_seed)
    g.seed_dev = rng_ptr(dy.device) if (drop2_p > 0 or fc_p > 0) else None
    g.dz, g.g1 = ptr(dz, torch.bfloat16), ptr(g1, torch.float32)
    g.d_ox, g.d_oh = ptr(d_ox, torch.float32), ptr(d_oh, torch.float32)
    ws = torch.empty(((int(dy.shape[0]) + 31) // 32, 256), device=dy.device, dtype=torch.float32)
    g.ws, g.ws_floats = ptr(ws, torch.float32), ws.numel()
    if _profile is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(lib().kantts_pnca_block_bwd(ctypes.byref(g), stream()), "pnca_block_bwd")
    if _profile is not None:
        e1.record()
        M = int(dy.shape[0])
        flops = 4.0 * M * 128 * 1024 + 4.0 * M * 128 * 128
        nbytes = M * (128 * 4 * 2 + 1024 * 2 * 2 + 128 * 4 * 2 + 2 * 128 * 4) + 2 * (2 * 128 * 1024 + 2 * 128 * 128)
        _profile.append((e0, e1, flops))
        _profile_families.append(("pnca_block_bwd", e0, e1, flops, float(nbytes)))
    return ws


def teacher_plan(in_lens, out_lens, dur, mel, pos, inv_ts, Tp, max_len, r):
    """kantts_teacher_plan: returns a dict of the tensors it writes (see the header)."""
    B, N = dur.shape
    T_mel, d_mel = int(mel.shape[1]), int(mel.shape[2])
    depth, L = int(inv_ts.numel()), Tp // r
    dev = dur.device
    i64, i32 = dict(device=dev, dtype=torch.int64), dict(device=dev, dtype=torch.int32)
    o = {"in_l64": torch.empty(B, **i64), "in_l32": torch.empty(B, **i32),
         "in_mask": torch.empty((B, N), device=dev, dtype=torch.bool),
         "out_l64": torch.empty(B, **i64), "out_l32": torch.empty(B, **i32),
         "out_mask": torch.empty((B, T_mel), device=dev, dtype=torch.bool),
         "lfr_l64": torch.empty(B, **i64), "lfr_l32": torch.empty(B, **i32),
         "lfr_mask": torch.empty((B, L), device=dev, dtype=torch.bool), "valid": torch.empty(B, **i64),
         "pos_enc": torch.empty((B, Tp, depth), device=dev, dtype=torch.float32),
         "prev": torch.empty((B, N, 1), device=dev, dtype=torch.float32),
         "bw_val": torch.empty((), device=dev, dtype=torch.float32), "bw_dev": torch.empty(1, **i32),
         "dec_input": torch.empty((B, L, d_mel), device=dev, dtype=torch.float32)}
    g = PlanArgs()
    g.in_lens, g.out_lens, g.dur = ptr(in_lens, torch.int64), ptr(out_lens, torch.int64), ptr(dur, torch.int64)
    g.mel, g.pos, g.inv_ts = ptr(mel, torch.float32), ptr(pos, torch.float32), ptr(inv_ts, torch.float32)
    g.B, g.N, g.T_mel, g.Tp, g.max_len, g.r, g.d_mel, g.depth = B, N, T_mel, int(Tp), int(max_len), int(r), d_mel, depth
    for k, t in o.items():
        setattr(g, k, ptr(t))
    check(lib().kantts_teacher_plan(ctypes.byref(g), stream()), "teacher_plan")
    return o


def decode_blob_sizes(d_mel, d_mem, d_out, n_layer):
    """(bf16 elements, fp32 elements) of the two parameter blobs kantts_pnca_decode_run reads; None if unsupported."""
    w, f = ctypes.c_longlong(0), ctypes.c_longlong(0)
    rc = lib().kantts_pnca_decode_blob_sizes(int(d_mel), int(d_mem), int(d_out), int(n_layer), ctypes.byref(w), ctypes.byref(f))
    return None if rc != 0 else (int(w.value), int(f.value))


def pnca_decode_run(w, f, memory, hkv, xkv, out, lens32, bw_seq, bw, d_mel, n_layer, in_scale, eps):
    """Every step of the free-running mel decoder for every sequence in one launch (csrc/ar_infer.hip)."""
    B, L, d_mem = memory.shape
    g = DecodeArgs()
    g.w, g.f = ptr(w, torch.bfloat16), ptr(f, torch.float32)
    g.memory, g.hkv, g.xkv, g.out = (ptr(t, torch.float32) for t in (memory, hkv, xkv, out))
    g.lens, g.bw_seq = ptr(lens32, torch.int32), ptr(bw_seq, torch.int32)
    g.B, g.L, g.d_mem, g.d_mel, g.d_out, g.n_layer, g.bw = int(B), int(L), int(d_mem), int(d_mel), int(out.shape[2]), int(n_layer), int(bw)
    g.in_scale, g.eps = float(in_scale), float(eps)
    for t in (memory, hkv, xkv, out):
        assert t.is_contiguous()
    check(lib().kantts_pnca_decode_run(ctypes.byref(g), stream()), "pnca_decode_run")


def dur_ar_run(w, f, gc, out, lens32):
    """The free-running duration predictor for every sequence in one launch (csrc/ar_infer.hip)."""
    B, T = out.shape
    g = DurArArgs()
    f32 = w.dtype == torch.float32  # k-chunk-major fp32 blob: the fp32 loop (kantts_dur_ar_run_f32)
    g.w, g.f, g.gc, g.out, g.lens = ptr(w, w.dtype if f32 else torch.bfloat16), ptr(f, torch.float32), ptr(gc, torch.float32), ptr(out, torch.float32), ptr(lens32, torch.int32)
    g.B, g.T = int(B), int(T)
    assert gc.is_contiguous() and out.is_contiguous() and gc.shape[-1] == 512
    assert w.numel() == 128 * 128 + 2 * 512 * 256
    if f32:
        check(lib().kantts_dur_ar_run_f32(ctypes.byref(g), stream()), "dur_ar_run_f32")
    else:
        check(lib().kantts_dur_ar_run(ctypes.byref(g), stream()), "dur_ar_run")


def enc_attn_fwd(x, xn, B, L, *, lens, rowmask, wqkv, bqkv, wfc, bfc, ln1, att_p, fc_p, seeds, qkv, o, lse, y1, xn1, mean1, rstd1):
    """The attention sub-layer of an encoder block in one launch (csrc/enc_attn.hip; kantts_enc_attn_fwd in the header).
    Returns False when the library declines the shape (L > 128)."""
    g = EncAttnArgs()
    g.x, g.xn = ptr(x, torch.float32), ptr(xn, torch.bfloat16)
    g.lens, g.rowmask = ptr(lens, torch.int32), ptr(rowmask)
    g.wqkv, g.bqkv, g.wfc, g.bfc = ptr(wqkv, torch.bfloat16), ptr(bqkv, torch.float32), ptr(wfc, torch.bfloat16), ptr(bfc, torch.float32)
    g.ln1_gamma, g.ln1_beta, g.ln1_eps = ptr(ln1[0], torch.float32), ptr(ln1[1], torch.float32), float(ln1[2])
    g.att_p, g.fc_p, g.att_seed, g.fc_seed = float(att_p), float(fc_p), int(seeds[0]), int(seeds[1])
    g.seed_dev = rng_ptr(x.device) if (att_p > 0 or fc_p > 0) else None
    g.qkv, g.o, g.lse, g.y1 = ptr(qkv, torch.float32), ptr(o, torch.float32), ptr(lse, torch.float32), ptr(y1, torch.float32)
    g.xn1 = ptr(xn1)
    g.xn1_bf16 = int(xn1 is not None and xn1.dtype == torch.bfloat16)
    g.mean1, g.rstd1 = ptr(mean1, torch.float32), ptr(rstd1, torch.float32)
    g.B, g.L = int(B), int(L)
    rc = lib().kantts_enc_attn_fwd(ctypes.byref(g), stream())
    if rc == E_UNSUPPORTED:
        return False
    check(rc, "enc_attn_fwd")
    return True


def ctc_attn(logits, in_lens32, out_lens32, blank, grad_scale):
    """AttentionCTCLoss and its gradient in one launch (csrc/ctc.hip; kantts_ctc_attn in the header).  logits (B, T1, T2)
    fp32 contiguous; lengths int32 on the device.  Returns (loss (B,) = nll_b / S_b, grad (B, T1, T2) = grad_scale * d loss_b
    / d logits)."""
    B, T1, T2 = logits.shape
    assert logits.is_contiguous() and logits.dtype == torch.float32
    n = int(lib().kantts_ctc_attn_workspace(int(B), int(T1), int(T2)))
    ws = torch.empty((max(n, 1),), device=logits.device, dtype=torch.float32)
    loss = torch.empty((B,), device=logits.device, dtype=torch.float32)
    grad = torch.empty_like(logits)
    g = CtcArgs()
    g.logits, g.in_lens, g.out_lens = ptr(logits, torch.float32), ptr(in_lens32, torch.int32), ptr(out_lens32, torch.int32)
    g.ws, g.loss, g.grad = ptr(ws, torch.float32), ptr(loss, torch.float32), ptr(grad, torch.float32)
    g.B, g.T1, g.T2, g.blank, g.grad_scale = int(B), int(T1), int(T2), float(blank), float(grad_scale)
    check(lib().kantts_ctc_attn(ctypes.byref(g), stream()), "ctc_attn")
    return loss, grad


def pnca_attn_qkv_bwd(qkv, hkv, ldh, ox, oh, d_ox, d_oh, lse_x, lse_h, B, L, *, lens, bw_dev, bw_x, bw_h, att_p, seed_x, seed_h,
                      wqkvT, x, mean0, rstd0, gamma0, dres, zero_rows, dqkv, dhkv, dx):
    """The cross-row half of a PNCA block's backward in one launch (csrc/pnca_block.hip; kantts_pnca_attn_qkv_bwd in the
    header): both attention bands' backward, the QKV projection's input gradient and the first LayerNorm's backward.
    Returns the partial rows of [dgamma0 | dbeta0] (``rows_sum_accum``), or None when the library declines (band above 16)."""
    g = PncaAttnBwdArgs()
    g.qkv, g.hkv, g.ldh = ptr(qkv, torch.float32), ptr(hkv, torch.float32), int(ldh)
    g.ox, g.oh, g.d_ox, g.d_oh = (ptr(t, torch.float32) for t in (ox, oh, d_ox, d_oh))
    g.lse_x, g.lse_h = ptr(lse_x, torch.float32), ptr(lse_h, torch.float32)
    g.B, g.L, g.H, g.C = int(B), int(L), 8, 128
    g.lens, g.bw_dev, g.bw_x, g.bw_h = ptr(lens), ptr(bw_dev), int(bw_x), int(bw_h)
    g.att_p, g.seed_x, g.seed_h = float(att_p), int(seed_x), int(seed_h)
    g.seed_dev = rng_ptr(qkv.device) if att_p > 0 else None
    g.wqkvT, g.x = ptr(wqkvT, torch.bfloat16), ptr(x, torch.float32)
    g.mean0, g.rstd0, g.ln0_gamma = ptr(mean0, torch.float32), ptr(rstd0, torch.float32), ptr(gamma0, torch.float32)
    g.dres, g.zero_rows = ptr(dres, torch.float32), ptr(zero_rows)
    g.dqkv, g.dhkv, g.lddh, g.dx = ptr(dqkv, torch.float32), ptr(dhkv, torch.float32), int(dhkv.shape[-1]), ptr(dx, torch.float32)
    M = int(B) * int(L)
    ws = torch.empty(((M + 31) // 32, 256), device=qkv.device, dtype=torch.float32)
    g.ws, g.ws_floats = ptr(ws, torch.float32), ws.numel()
    if _profile is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = lib().kantts_pnca_attn_qkv_bwd(ctypes.byref(g), stream())
    if rc == E_UNSUPPORTED:
        return None
    check(rc, "pnca_attn_qkv_bwd")
    if _profile is not None:
        e1.record()
        flops = 2.0 * M * 384 * 128 + 2.0 * 8.0 * B * L * L * 128  # projection input gradient + dense-count attention backward
        nbytes = M * (384 * 4 * 2 + 256 * 4 * 2 + 4 * 128 * 4 + 128 * 4 * 3) + 2 * 384 * 128
        _profile.append((e0, e1, flops))
        _profile_families.append(("pnca_attn_qkv_bwd", e0, e1, flops, float(nbytes)))
    return ws


def rows_sum_many(problems):
    """problems: list of (src (rows, 256), dst0 (128), dst1 (128)): dst0 += column sums of src[:, :128], dst1 += those of
    src[:, 128:], up to ROWSUM_MAX problems per launch (kantts_rows_sum_many)."""
    for s0 in range(0, len(problems), ROWSUM_MAX):
        chunk = problems[s0:s0 + ROWSUM_MAX]
        g = RowSumArgs()
        g.n, g.cols, g.split = len(chunk), 256, 128
        for i, (src, d0, d1) in enumerate(chunk):
            assert src.dim() == 2 and src.shape[1] == 256 and src.is_contiguous()
            g.rows[i] = int(src.shape[0])
            g.src[i], g.dst0[i], g.dst1[i] = _addr(src), _addr(d0), _addr(d1)
        check(lib().kantts_rows_sum_many(ctypes.byref(g), stream()), "rows_sum_many")


def rows_sum_accum(src, dst0, dst1):
    """dst0 (128) += column sums of src[:, :128], dst1 (128) += column sums of src[:, 128:]: the per-workgroup partial rows of
    a LayerNorm's dgamma / dbeta (kantts_pnca_block_bwd, kantts_bgemm_nt_lnbwd).  Parameter gradients: when weight gradients
    are deferred (deferred_tn.enabled: the captured step) the problem is only recorded and every recorded one leaves in ONE
    launch at the next flush, on the weight gradients' side stream -- a fork of the stream per call would put a cross-queue
    dependency (~7 us in a replayed graph) into the backward chain of every block."""
    if deferred_tn.enabled:
        # outputs only through their storages: autograd adopts dst0 / dst1 as p.grad only while nobody else references the
        # tensor objects (see bgemm_tn below)
        deferred_tn.rowsums.append((src, deferred_tn._desc(dst0), deferred_tn._desc(dst1)))
        return
    rows_sum_many([(src, dst0, dst1)])


TN_MAX_GROUP = 16


class _DeferredTN:
    """Weight gradients are leaves of the backward graph: nothing downstream of a layer's backward reads them, only the
    optimizer does.  When enabled (ops.wgrad_overlap.enable -- the captured training step and bench.py), bgemm_tn only
    records the problem; ``flush`` (called where the side stream used to be joined: before the optimizer packs the
    gradients, and before a data-parallel bucket is exchanged) issues all recorded problems grouped by shape, up to
    TN_MAX_GROUP per launch.  A group needs only 1/n of the token split per problem, i.e. n times fewer fp32 atomics (the
    resource that bounds the kernel, csrc/gemm_bf16.hip), and ~150 launches per SAM-BERT step become ~15.  Off by
    default: code that reads ``p.grad`` right after ``backward()`` must see finished gradients."""

    def __init__(self):
        self.enabled = False
        self.groups = {}
        self.copies = []
        self.rowsums = []  # (partial rows, descriptor of dgamma, descriptor of dbeta): rows_sum_accum

    def add(self, key, g, a, b, c, db, seed, keep):
        self.groups.setdefault(key, []).append((g, a, b, c, db, seed, keep))

    @staticmethod
    def _desc(t):
        return (t.untyped_storage(), t.storage_offset(), tuple(t.shape), tuple(t.stride()), t.dtype, t.device)

    @staticmethod
    def _rebuild(d):
        st, off, shape, stride, dtype, dev = d
        return torch.empty(0, dtype=dtype, device=dev).set_(st, off, shape, stride)

    def shared_gradient(self, src):
        """A second gradient tensor with the same values as ``src`` (two biases sharing one gradient).  Immediate mode:
        ``src`` itself (autograd clones it for the second parameter).  Deferred mode: a separate buffer that receives a
        copy of ``src`` after the deferred launches have filled it -- never an alias: gradient tensors are scaled in
        place by clipping.  Only storages are kept, so autograd can still adopt both tensors as p.grad."""
        if not self.enabled:
            return src
        from .ops import gzeros  # (from the accumulator pool: a candidate for the direct-gradient plan of the arena)

        dst = gzeros(tuple(src.shape), src.device) if src.dtype == torch.float32 else torch.empty_like(src)
        self.copies.append((self._desc(dst), self._desc(src)))
        return dst

    def flush(self):
        """[round 4] Spreading the LAST flush of a step (~8 grouped launches of 256-2048 workgroups, 160 us back to back at
        the tail of the captured step) over four streams was measured: 7.12 ms against 7.14 ms on one stream, same box
        (profiles/r04_runP_direct_grads_and_spread_flush_ab.log) -- the launches are bound by their fp32 atomics, not by
        idle CUs.  Removed."""
        if not self.groups and not self.copies and not self.rowsums:
            return
        groups, self.groups = self.groups, {}
        copies, self.copies = self.copies, []
        rowsums, self.rowsums = self.rowsums, []
        self._launch([probs[s:s + TN_MAX_GROUP] for probs in groups.values() for s in range(0, len(probs), TN_MAX_GROUP)])
        if rowsums:
            rows_sum_many([(src, self._rebuild(d0), self._rebuild(d1)) for src, d0, d1 in rowsums])
        if copies:
            torch._foreach_copy_([self._rebuild(d) for d, _ in copies], [self._rebuild(s) for _, s in copies])

    def _launch(self, chunks):
        L = lib()
        for chunk in chunks:
            n = len(chunk)
            arr = lambda k: (c_void_p * n)(*[q[k] for q in chunk])  # noqa: E731
            seeds = (c_uint64 * n)(*[q[5] for q in chunk])
            g = chunk[0][0]
            g.slices = 0
            if _profile is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            apply_launch_tuning()
            check(L.kantts_bgemm_tn_grouped(ctypes.byref(g), n, arr(1), arr(2), arr(3), arr(4), seeds, stream()),
                  "bgemm_tn_grouped")
            if _profile is not None:
                e1.record()
                _profile.append((e0, e1, 2.0 * g.M * g.N * g.K * g.ntaps * n))


deferred_tn = _DeferredTN()


def bgemm_tn(a, lda, b, ldb, M, N, K, c, c_ns, c_ks, *, c_ts=0, T=0, ntaps=1, shift0=0, shift_step=0, db=None, alpha=1.0,
             a_drop_p=0.0, a_drop_seed=0, slices=0):
    """c[n*c_ns + k*c_ks + tap*c_ts] += alpha * sum_m a[m][n] * b[m + shift][k] (+ db[n]); False when declined."""
    g = BGemmTnArgs()
    at = a[0] if isinstance(a, tuple) else a
    bt = b[0] if isinstance(b, tuple) else b
    g.a, g.b, g.lda, g.ldb = _addr(a), _addr(b), int(lda), int(ldb)
    g.M, g.N, g.K, g.T = int(M), int(N), int(K), int(T)
    g.a_f32, g.b_f32 = int(at.dtype == torch.float32), int(bt.dtype == torch.float32)
    g.ntaps, g.shift0, g.shift_step, g.slices = int(ntaps), int(shift0), int(shift_step), int(slices)
    g.c, g.c_ns, g.c_ks, g.c_ts = _addr(c), int(c_ns), int(c_ks), int(c_ts)
    g.db, g.alpha = ptr(db, torch.float32), float(alpha)
    g.a_drop_p, g.a_drop_seed = float(a_drop_p), int(a_drop_seed)
    g.seed_dev = rng_ptr(at.device) if a_drop_p > 0 else None
    if deferred_tn.enabled and M > 0 and not (N % 8 or K % 8 or lda % 8 or ldb % 8):
        key = (int(lda), int(ldb), int(M), int(N), int(K), int(T), g.a_f32, g.b_f32, int(ntaps), int(shift0),
               int(shift_step), int(c_ns), int(c_ks), int(c_ts), float(alpha), float(a_drop_p), str(at.device))
        ct = c[0] if isinstance(c, tuple) else c
        # operands are kept alive as tensors; the OUTPUTS only through their storages: autograd's AccumulateGrad adopts
        # a returned gradient tensor as p.grad only while nobody else references the tensor object, otherwise it clones
        # it (and the clone would be taken before the deferred launch has filled the buffer)
        keep = (at, bt, ct.untyped_storage(), None if db is None else db.untyped_storage())
        deferred_tn.add(key, g, g.a, g.b, g.c, g.db, int(a_drop_seed), keep)
        return True
    if _profile is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    apply_launch_tuning()
    rc = lib().kantts_bgemm_tn(ctypes.byref(g), stream())
    if rc == E_UNSUPPORTED:
        return False
    check(rc, "bgemm_tn")
    if _profile is not None:
        e1.record()
        _profile.append((e0, e1, 2.0 * M * N * K * ntaps))
    return True


def cast_bf16(src, dst=None):
    """fp32 -> bf16 copy (numel % 8 == 0, contiguous)."""
    if dst is None:
        dst = torch.empty(src.shape, device=src.device, dtype=torch.bfloat16)
    check(lib().kantts_cast_f32_bf16(ptr(src, torch.float32), ptr(dst, torch.bfloat16), src.numel(), stream()), "cast_bf16")
    return dst


def conv_win(x, w_tap, out, *, B, Tsrc, Tdst, groups, CR, NG, K, in_mul, in_add, in_kstep, in_div, phases, inner=1, up=1,
             bias=None, res=None, in_gate=None, in_gate_slope=0.0, in_leaky=None, out_leaky=None, out_gate=None,
             out_gate_slope=0.0):
    """Windowed channels-last convolution (csrc/conv_win.hip).  Returns False when the kernel does not
    support the shape (the caller then takes the segmented-GEMM route); raises on any other error."""
    if _precision["gemm"] not in (PREC_FP32, PREC_BF16) or os.environ.get("KANTTS_NO_CONVWIN"):
        return False
    g = ConvArgs()
    g.in_, g.in_gate, g.w, g.out = ptr(x, torch.float32), ptr(in_gate, torch.float32), ptr(w_tap, torch.float32), \
        ptr(out, torch.float32)
    g.bias, g.res, g.out_gate = ptr(bias, torch.float32), ptr(res, torch.float32), ptr(out_gate, torch.float32)
    g.B, g.Tsrc, g.Tdst = int(B), int(Tsrc), int(Tdst)
    g.Cin_tot, g.Ntot, g.CR, g.NG, g.groups, g.K = int(groups * CR), int(groups * NG), int(CR), int(NG), int(groups), int(K)
    g.in_mul, g.in_add, g.in_kstep, g.in_div, g.phases = int(in_mul), int(in_add), int(in_kstep), int(in_div), int(phases)
    g.inner, g.up = int(inner), int(up)
    if in_leaky is not None:
        g.in_act, g.in_slope = 1, float(in_leaky)
    if out_leaky is not None:
        g.out_act, g.out_slope = 1, float(out_leaky)
    g.in_gate_slope, g.out_gate_slope = float(in_gate_slope), float(out_gate_slope)
    g.precision = _precision["gemm"]
    if _profile is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = lib().kantts_conv_win_launch(ctypes.byref(g), stream())
    if rc == E_UNSUPPORTED:
        return False
    check(rc, "conv_win")
    if _profile is not None:
        e1.record()
        _profile.append((e0, e1, 2.0 * B * Tdst * inner * groups * NG * CR * K / max(1, in_div)))
        _profile_tags.append(("conv_win", dict(B=B, Tsrc=Tsrc, Tdst=Tdst, inner=inner, Cin=groups * CR, Cout=groups * NG,
                                               groups=groups, K=K, in_mul=in_mul, in_div=in_div, up=up,
                                               gated=in_gate is not None, res=res is not None),
                              4.0 * (B * Tsrc * inner * groups * CR * (2 if in_gate is not None else 1) +
                                     B * Tdst * inner * groups * NG * (1 + (res is not None) + (out_gate is not None)) +
                                     K * groups * NG * CR)))
    return True


def conv_wgrad(x, dy, dw_tap, db, *, B, Tsrc, Tdst, groups, CR, NG, K, stride, dil, pad, inner=1, up=1, dy_gate=None,
               dy_gate_slope=0.0, x_leaky=None):
    """Accumulate the tap-major weight gradient (K, Ntot, CR) and the bias gradient (csrc/conv_wgrad.hip).
    Returns False when the kernel does not take the shape (caller falls back to the segmented GEMM)."""
    if _precision["gemm"] not in (PREC_FP32, PREC_BF16) or os.environ.get("KANTTS_NO_CONVWIN"):
        return False
    g = ConvWArgs()
    g.x, g.dy, g.dy_gate = ptr(x, torch.float32), ptr(dy, torch.float32), ptr(dy_gate, torch.float32)
    g.dw, g.db = ptr(dw_tap, torch.float32), ptr(db, torch.float32)
    g.B, g.Tsrc, g.Tdst = int(B), int(Tsrc), int(Tdst)
    g.Cin_tot, g.Ntot, g.CR, g.NG, g.groups, g.K = int(groups * CR), int(groups * NG), int(CR), int(NG), int(groups), int(K)
    g.stride, g.dil, g.pad, g.inner, g.up = int(stride), int(dil), int(pad), int(inner), int(up)
    if x_leaky is not None:
        g.x_act, g.x_slope = 1, float(x_leaky)
    g.dy_gate_slope = float(dy_gate_slope)
    g.precision = _precision["gemm"]
    if _profile is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = lib().kantts_conv_wgrad_launch(ctypes.byref(g), stream())
    if rc == E_UNSUPPORTED:
        return False
    check(rc, "conv_wgrad")
    if _profile is not None:
        e1.record()
        _profile.append((e0, e1, 2.0 * B * Tdst * inner * groups * NG * CR * K))
        _profile_tags.append(("conv_wgrad", dict(B=B, Tsrc=Tsrc, Tdst=Tdst, inner=inner, Cin=groups * CR, Cout=groups * NG,
                                                 groups=groups, K=K, stride=stride, dil=dil, up=up,
                                                 gated=dy_gate is not None),
                              4.0 * (B * Tsrc * inner * groups * CR + B * Tdst * inner * groups * NG *
                                     (2 if dy_gate is not None else 1) + K * groups * NG * CR)))
    return True


def ragged_rows(src, row_off, lens, Tmax, *, start=None, pad=None, transpose=False):
    """Padded / cropped batch from a flat (rows, C) device buffer (csrc/batching.hip): out (B, Tmax, C) -- or (B, C, Tmax)
    with ``transpose`` -- where out[b, t] = src[row_off[b] + start[b] + t] for t < lens[b], else ``pad`` (C values or None
    for zeros).  src float32 or int64; row_off int64, start / lens int32 device tensors."""
    B, C = int(row_off.shape[0]), int(src.shape[1])
    out = torch.empty((B, C, Tmax) if transpose else (B, Tmax, C), device=src.device, dtype=src.dtype)
    fn = {torch.float32: "kantts_ragged_rows_f32", torch.int64: "kantts_ragged_rows_i64"}[src.dtype]
    check(getattr(lib(), fn)(ptr(src), ptr(row_off, torch.int64), ptr(start, torch.int32), ptr(lens, torch.int32),
                             ptr(pad, src.dtype), ptr(out), B, int(Tmax), C, int(bool(transpose)), stream()), fn)
    return out


def act_cast_bf16(src, *, act_slope=None, gate=None, gate_slope=0.0, dst=None):
    """bf16 operand image of an fp32 tensor (csrc/cconv.hip): LeakyReLU(src) when ``act_slope`` is given, or
    src * (gate > 0 ? 1 : gate_slope) when ``gate`` (fp32 or bf16, same shape) is given, else a plain cast."""
    if dst is None:
        dst = torch.empty(src.shape, device=src.device, dtype=torch.bfloat16)
    gate_bf = gate is not None and gate.dtype == torch.bfloat16
    check(lib().kantts_act_cast_bf16(ptr(src, torch.float32), ptr(gate), int(gate_bf), ptr(dst, torch.bfloat16),
                                     int(act_slope is not None), float(gate_slope if gate is not None else (act_slope or 0.0)),
                                     src.numel(), stream()), "act_cast_bf16")
    return dst


def cconv(x_bf, w_bf, *, out=None, out_bf=None, B, Tsrc, Tdst, groups, CR, NG, K, in_mul, in_add, in_kstep, in_div, phases,
          inner=1, up=1, bias=None, res=None, out_leaky=None, out_gate=None, out_gate_slope=0.0, bf_leaky=None, tile=0,
          res_after_gate=False):
    """bf16 convolution contraction (csrc/cconv.hip, kantts_cconv_launch): x_bf (B, Tsrc, inner, groups*CR) bf16,
    w_bf (K, groups*NG, CR) bf16; writes ``out`` (fp32) and / or ``out_bf`` (bf16, optionally LeakyReLU'd by
    ``bf_leaky``).  Returns False when the kernel does not take the shape."""
    if os.environ.get("KANTTS_NO_CCONV"):
        return False
    g = CConvArgs()
    g.in_, g.w = ptr(x_bf, torch.bfloat16), ptr(w_bf, torch.bfloat16)
    g.bias, g.res = ptr(bias, torch.float32), ptr(res, torch.float32)
    g.out_gate = ptr(out_gate)
    g.out_gate_bf16 = int(out_gate is not None and out_gate.dtype == torch.bfloat16)
    g.out, g.out_bf = ptr(out, torch.float32), ptr(out_bf, torch.bfloat16)
    g.B, g.Tsrc, g.Tdst = int(B), int(Tsrc), int(Tdst)
    g.Cin_tot, g.Ntot, g.CR, g.NG, g.groups, g.K = int(groups * CR), int(groups * NG), int(CR), int(NG), int(groups), int(K)
    g.in_mul, g.in_add, g.in_kstep, g.in_div, g.phases = int(in_mul), int(in_add), int(in_kstep), int(in_div), int(phases)
    g.inner, g.up = int(inner), int(up)
    if out_leaky is not None:
        g.out_act, g.out_slope = 1, float(out_leaky)
    g.out_gate_slope = float(out_gate_slope)
    if bf_leaky is not None:
        g.bf_act, g.bf_slope = 1, float(bf_leaky)
    g.tile = int(tile)
    g.res_after_gate = int(bool(res_after_gate))
    if _profile is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = lib().kantts_cconv_launch(ctypes.byref(g), stream())
    if rc == E_UNSUPPORTED:
        return False
    check(rc, "cconv")
    if _profile is not None:
        e1.record()
        _profile.append((e0, e1, 2.0 * B * Tdst * inner * groups * NG * CR * K / max(1, in_div)))
        _profile_tags.append(("cconv", dict(B=B, Tsrc=Tsrc, Tdst=Tdst, inner=inner, Cin=groups * CR, Cout=groups * NG,
                                            groups=groups, K=K, in_mul=in_mul, in_div=in_div, up=up,
                                            gated=out_gate is not None, res=res is not None),
                              2.0 * B * Tsrc * inner * groups * CR + 2.0 * K * groups * NG * CR +
                              B * Tdst * inner * groups * NG * (4.0 * (out is not None) + 2.0 * (out_bf is not None) +
                                                                4.0 * (res is not None) +
                                                                (0.0 if out_gate is None else out_gate.element_size()))))
    return True


def cconv_wgrad(x_bf, dy_bf, dw_tap, db, *, B, Tsrc, Tdst, groups, CR, NG, K, stride, dil, pad, inner=1, up=1, slices=0):
    """Weight / bias gradient from bf16 operands (kantts_cconv_wgrad_launch): accumulates into dw_tap (K, groups*NG, CR)
    fp32 and db.  Returns False when the kernel does not take the shape."""
    if os.environ.get("KANTTS_NO_CCONV"):
        return False
    g = CConvWArgs()
    g.x, g.dy = ptr(x_bf, torch.bfloat16), ptr(dy_bf, torch.bfloat16)
    g.dw, g.db = ptr(dw_tap, torch.float32), ptr(db, torch.float32)
    g.B, g.Tsrc, g.Tdst = int(B), int(Tsrc), int(Tdst)
    g.Cin_tot, g.Ntot, g.CR, g.NG, g.groups, g.K = int(groups * CR), int(groups * NG), int(CR), int(NG), int(groups), int(K)
    g.stride, g.dil, g.pad, g.inner, g.up, g.slices = int(stride), int(dil), int(pad), int(inner), int(up), int(slices)
    # partial tiles of the token slices: a scratch buffer from the caching allocator (ordered on the launch stream)
    nws = int(lib().kantts_cconv_wgrad_ws_floats(ctypes.byref(g)))
    ws = None
    if nws > 0:
        ws = torch.empty(nws, device=dw_tap.device, dtype=torch.float32)
        g.workspace, g.ws_floats = ptr(ws), nws
    if _profile is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = lib().kantts_cconv_wgrad_launch(ctypes.byref(g), stream())
    if rc == E_UNSUPPORTED:
        return False
    check(rc, "cconv_wgrad")
    if _profile is not None:
        e1.record()
        _profile.append((e0, e1, 2.0 * B * Tdst * inner * groups * NG * CR * K))
        _profile_tags.append(("cconv_wgrad", dict(B=B, Tsrc=Tsrc, Tdst=Tdst, inner=inner, Cin=groups * CR, Cout=groups * NG,
                                                  groups=groups, K=K, stride=stride, dil=dil, up=up, gated=False),
                              2.0 * (B * Tsrc * inner * groups * CR + B * Tdst * inner * groups * NG) +
                              4.0 * K * groups * NG * CR))
    return True


def conv_c1(mode, *, x=None, dx=None, y=None, gate=None, w=None, bias=None, dw=None, db=None, B, Tsrc, Tdst, Cout, K,
            stride, dil, pad, inner=1, out_leaky=None, gate_slope=0.0, y_bf16=None):
    """Single-input-channel convolution kernels (csrc/conv_c1.hip): mode 0 forward, 1 input gradient, 2 weight /
    bias gradient.  Returns False when the shape is not supported."""
    if os.environ.get("KANTTS_NO_CONVWIN"):
        return False
    g = ConvC1Args()
    g.x, g.dx, g.y, g.gate = ptr(x, torch.float32), ptr(dx, torch.float32), ptr(y, torch.float32), ptr(gate, torch.float32)
    g.w, g.bias, g.dw, g.db = ptr(w, torch.float32), ptr(bias, torch.float32), ptr(dw, torch.float32), ptr(db, torch.float32)
    g.B, g.Tsrc, g.Tdst, g.Cout, g.K = int(B), int(Tsrc), int(Tdst), int(Cout), int(K)
    g.stride, g.dil, g.pad, g.inner = int(stride), int(dil), int(pad), int(inner)
    if out_leaky is not None:
        g.out_act, g.out_slope = 1, float(out_leaky)
    g.gate_slope = float(gate_slope)
    g.y_bf16 = ptr(y_bf16, torch.bfloat16)
    if int(mode) == 2:
        apply_launch_tuning()
    rc = lib().kantts_conv_c1_launch(ctypes.byref(g), int(mode), stream())
    if rc == E_UNSUPPORTED:
        return False
    check(rc, "conv_c1")
    return True


def conv_n1(mode, *, x, y, w, w_ks, w_cs, dx=None, bias=None, dw=None, db=None, B, Tsrc, Tdst, Cin, K, stride, dil, pad,
            inner=1, in_leaky=None):
    """One-output-channel convolution kernels (csrc/conv_n1.hip): mode 0 forward, 1 input gradient, 2 weight / bias
    gradient; ``y`` is the output (mode 0) or its gradient.  Returns False when the shape is not supported."""
    if os.environ.get("KANTTS_NO_CONV_N1"):
        return False
    g = ConvN1Args()
    g.x, g.dx, g.y, g.w = ptr(x, torch.float32), ptr(dx, torch.float32), ptr(y, torch.float32), ptr(w, torch.float32)
    g.bias, g.dw, g.db = ptr(bias, torch.float32), ptr(dw, torch.float32), ptr(db, torch.float32)
    g.B, g.Tsrc, g.Tdst, g.Cin, g.K = int(B), int(Tsrc), int(Tdst), int(Cin), int(K)
    g.stride, g.dil, g.pad, g.inner, g.w_ks, g.w_cs = int(stride), int(dil), int(pad), int(inner), int(w_ks), int(w_cs)
    if in_leaky is not None:
        g.in_act, g.in_slope = 1, float(in_leaky)
    rc = lib().kantts_conv_n1_launch(ctypes.byref(g), int(mode), stream())
    if rc == E_UNSUPPORTED:
        return False
    check(rc, "conv_n1")
    return True


# ----------------------------------------------------------------------------------------------
# device-resident RNG offset: every dropout seed is (host seed + *rng_state).  A training step advances
# it once (ops.advance_rng) -- when the step is captured in a hipGraph the increment is replayed too,
# so each replay draws fresh masks although the host seeds are frozen in the kernel arguments.
_rng_state = {}


def rng_state(device):
    # one offset per physical device: "cuda" and "cuda:0" must name the same tensor (a caller that advanced the offset
    # of "cuda" while the kernels read the one of "cuda:0" would silently freeze its dropout masks)
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    key = str(device)
    t = _rng_state.get(key)
    if t is None:
        t = torch.zeros(1, dtype=torch.int64, device=device)
        _rng_state[key] = t
    return t


def rng_ptr(device):
    return ptr(rng_state(device))


# ----------------------------------------------------------------------------------------------
# bench instrumentation: HIP events (on the launch stream) around every GEMM launch
_profile = None
_profile_tags = []  # (kernel, shape dict, algorithmic bytes) of the conv launches, parallel to their _profile entries
_profile_families = []  # (family, start event, end event, flops, algorithmic bytes) of the launches that account bytes


def profile_begin():
    global _profile
    _profile = []
    del _profile_tags[:]
    del _profile_families[:]


def profile_families():
    """Per kernel family of the launches recorded since profile_begin (call after profile_end, which synchronises):
    launches, summed event time, algorithmic flops and bytes."""
    out = {}
    for name, e0, e1, flops, nbytes in _profile_families:
        o = out.setdefault(name, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
        o["launches"] += 1
        o["ms"] += e0.elapsed_time(e1)
        o["flops"] += flops
        o["bytes"] += nbytes
    return out


def profile_end():
    global _profile
    rec, _profile = _profile, None
    torch.cuda.synchronize()
    ms = sum(e0.elapsed_time(e1) for e0, e1, _ in rec)
    return {"launches": len(rec), "ms": ms, "flops": sum(f for _, _, f in rec)}


def profile_end_by_shape():
    """Per distinct conv launch shape: count, total ms, flops and algorithmic bytes per launch (conv launches only;
    scripts/conv_shape_bench.py prints the table)."""
    global _profile
    rec, _profile = _profile, None
    torch.cuda.synchronize()
    out = {}
    tags = list(_profile_tags)
    ci = 0
    # GEMM launches carry no tag: a conv record is recognised by the flop count of the next unmatched tag
    for e0, e1, f in rec:
        if ci < len(tags) and abs(_tag_flops(tags[ci]) - f) <= 1e-6 * max(1.0, f):
            kern, shape, nbytes = tags[ci]
            ci += 1
            key = kern + " " + " ".join("%s=%s" % kv for kv in shape.items())
            o = out.setdefault(key, dict(n=0, ms=0.0, flops=f, bytes=nbytes))
            o["n"] += 1
            o["ms"] += e0.elapsed_time(e1)
    return out


def _tag_flops(tag):
    kern, s, _ = tag
    f = 2.0 * s["B"] * s["Tdst"] * s["inner"] * s["Cout"] * (s["Cin"] // s["groups"]) * s["K"]
    return f / max(1, s.get("in_div", 1)) if kern in ("conv_win", "cconv") else f
