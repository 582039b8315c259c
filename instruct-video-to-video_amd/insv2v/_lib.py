"""ctypes binding of libinsv2v_hip.so (C ABI declared in include/insv2v_hip.h).

There is NO fallback: if the library is missing or a kernel rejects its arguments the
product path raises.  ``import torch`` must precede loading so the library binds to the
HIP runtime PyTorch already loaded (same libamdhip64.so.7 soname => one runtime, shared
streams and device pointers).
"""
import ctypes as C
import os

import torch  # noqa: F401  (must be imported first, see module docstring)

ABI_VERSION = 12
_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("INSV2V_LIB", os.path.join(_HERE, "libinsv2v_hip.so"))  # override: A/B builds only

c_i32, c_i64, c_f32, c_p = C.c_int32, C.c_int64, C.c_float, C.c_void_p


class GemmDesc(C.Structure):
    _fields_ = [("a", c_p), ("a2", c_p), ("w", c_p), ("c", c_p), ("bias", c_p), ("row_bias", c_p), ("residual", c_p),
                ("row_stats", c_p), ("col_sum", c_p),
                ("lda", c_i64), ("lda2", c_i64), ("ldw", c_i64), ("ldc", c_i64), ("ldr", c_i64), ("ld_rb", c_i64),
                ("a_bs", c_i64), ("w_bs", c_i64), ("c_bs", c_i64), ("r_bs", c_i64),
                ("workspace", c_p), ("workspace_bytes", c_i64),
                ("M", c_i32), ("N", c_i32), ("K", c_i32), ("k_split", c_i32), ("rows_per_group", c_i32),
                ("rb_mod", c_i32), ("act", c_i32), ("c_fp32", c_i32), ("mode", c_i32),
                ("NB", c_i32), ("IH", c_i32), ("IW", c_i32), ("OH", c_i32), ("OW", c_i32), ("Cin", c_i32),
                ("stride", c_i32), ("pad_t", c_i32), ("pad_l", c_i32), ("upsample", c_i32),
                ("batch", c_i32), ("tile", c_i32), ("split_k", c_i32), ("alpha", c_f32),
                ("gn_ab", c_p), ("gn_images_per_sample", c_i32), ("gn_silu", c_i32),
                ("stats_out", c_p), ("stats_scratch", c_p), ("stats_parts", c_i32), ("ln_eps", c_f32),
                ("w_group_stride", c_i64), ("w_group_rows", c_i32), ("reserved0", c_i32)]


class WinogradInDesc(C.Structure):
    _fields_ = [("x", c_p), ("x2", c_p), ("gn_ab", c_p), ("v", c_p), ("ldx", c_i64), ("ldx2", c_i64), ("v_group_rows", c_i64),
                ("NB", c_i32), ("H", c_i32), ("W", c_i32), ("C", c_i32), ("C1", c_i32), ("gn_images_per_sample", c_i32), ("gn_silu", c_i32), ("upsample", c_i32)]


class WinogradOutDesc(C.Structure):
    _fields_ = [("m", c_p), ("bias", c_p), ("row_bias", c_p), ("residual", c_p), ("y", c_p), ("m_group_rows", c_i64),
                ("ldr", c_i64), ("ldy", c_i64), ("ld_rb", c_i64), ("NB", c_i32), ("H", c_i32), ("W", c_i32), ("Cout", c_i32), ("rows_per_group", c_i32), ("upsample", c_i32)]


class FfnDesc(C.Structure):
    _fields_ = [("x", c_p), ("out", c_p), ("wstream", c_p), ("ldx", c_i64), ("ldo", c_i64),
                ("M", c_i32), ("C", c_i32), ("hidden", c_i32), ("eps", c_f32), ("post_residual", c_p), ("ld_post", c_i64), ("post", c_i32)]


class RowLinDesc(C.Structure):
    _fields_ = [("x", c_p), ("out", c_p), ("residual", c_p), ("wstream", c_p), ("ldx", c_i64), ("ldo", c_i64), ("ldr", c_i64),
                ("M", c_i32), ("N", c_i32), ("K", c_i32), ("layernorm", c_i32), ("frame_bias", c_i32), ("rows_per_frame", c_i32),
                ("frames", c_i32), ("eps", c_f32), ("stats_out", c_p), ("stats_eps", c_f32), ("gn_ab", c_p), ("gn_rows", c_i32)]


class TattnDesc(C.Structure):
    _fields_ = [("x", c_p), ("out", c_p), ("wstream", c_p), ("ldx", c_i64), ("ldo", c_i64),
                ("samples", c_i32), ("HW", c_i32), ("C", c_i32), ("heads", c_i32), ("frames", c_i32), ("eps", c_f32), ("scale", c_f32)]


class XattnDesc(C.Structure):
    _fields_ = [("x", c_p), ("out", c_p), ("wstream", c_p), ("kvstream", c_p), ("ldx", c_i64), ("ldo", c_i64),
                ("M", c_i32), ("rows_per_sample", c_i32), ("C", c_i32), ("heads", c_i32), ("ctx_len", c_i32), ("eps", c_f32), ("scale", c_f32),
                ("pre_residual", c_p), ("ld_pre", c_i64)]


class GroupNormDesc(C.Structure):
    _fields_ = [("x", c_p), ("x2", c_p), ("y", c_p), ("gamma", c_p), ("beta", c_p), ("partials", c_p),
                ("ldx", c_i64), ("ldx2", c_i64), ("ldy", c_i64),
                ("nsamples", c_i32), ("rows_per_sample", c_i32), ("C", c_i32), ("C1", c_i32), ("G", c_i32),
                ("nchunks", c_i32), ("silu", c_i32), ("eps", c_f32), ("ab", c_p), ("stats_only", c_i32)]


class LayerNormDesc(C.Structure):
    _fields_ = [("x", c_p), ("y", c_p), ("gamma", c_p), ("beta", c_p), ("pe", c_p),
                ("ldx", c_i64), ("ldy", c_i64), ("rows", c_i32), ("C", c_i32),
                ("rows_per_frame", c_i32), ("frames", c_i32), ("pe_start", c_i32), ("eps", c_f32)]


class AttentionDesc(C.Structure):
    _fields_ = [("q", c_p), ("k", c_p), ("v", c_p), ("o", c_p),
                ("q_rs", c_i64), ("k_rs", c_i64), ("v_rs", c_i64), ("o_rs", c_i64),
                ("q_outer", c_i64), ("q_step", c_i64), ("kv_outer", c_i64), ("kv_step", c_i64),
                ("o_outer", c_i64), ("o_step", c_i64),
                ("q_inner", c_i32), ("kv_inner", c_i32), ("o_inner", c_i32),
                ("batch", c_i32), ("heads", c_i32), ("head_dim", c_i32), ("seq_q", c_i32), ("seq_k", c_i32),
                ("scale", c_f32), ("causal", c_i32), ("q_bias", c_p), ("k_bias", c_p), ("v_bias", c_p), ("bias_rs", c_i64)]


class StepDesc(C.Structure):
    _fields_ = [("eps_in", c_p), ("latent", c_p), ("latent_ref", c_p), ("delta_q", c_p), ("noise", c_p),
                ("rescale_stats", c_p), ("latent_out", c_p), ("pred_x0", c_p), ("eps_out", c_p),
                ("nbranch", c_i32), ("F", c_i32), ("h", c_i32), ("w", c_i32), ("R", c_i32), ("correct", c_i32),
                ("text_cfg", c_f32), ("img_cfg", c_f32), ("sqrt_a", c_f32), ("sqrt_1ma", c_f32),
                ("c_x0", c_f32), ("c_eps", c_f32), ("c_xt", c_f32), ("c_noise", c_f32), ("guidance_rescale", c_f32),
                ("branch_stride", c_i64)]


class Im2colDesc(C.Structure):
    _fields_ = [("x", c_p), ("x2", c_p), ("out", c_p), ("ldx", c_i64), ("ldx2", c_i64), ("ldo", c_i64),
                ("N", c_i32), ("IH", c_i32), ("IW", c_i32), ("C", c_i32), ("C1", c_i32), ("KH", c_i32), ("KW", c_i32),
                ("stride_h", c_i32), ("stride_w", c_i32), ("pad_h", c_i32), ("pad_w", c_i32), ("OH", c_i32), ("OW", c_i32)]


class CorrLookupDesc(C.Structure):
    _fields_ = [("pyr0", c_p), ("pyr1", c_p), ("pyr2", c_p), ("pyr3", c_p), ("coords", c_p), ("out", c_p), ("ldo", c_i64),
                ("B", c_i32), ("h", c_i32), ("w", c_i32), ("levels", c_i32), ("radius", c_i32)]


# name -> (restype, argtypes); mirrors include/insv2v_hip.h one to one.
SIGNATURES = {
    "insv2v_abi_version": (c_i32, []),
    "insv2v_init": (c_i32, []),
    "insv2v_gemm": (c_i32, [C.POINTER(GemmDesc), c_p]),
    "insv2v_gemm_stats_parts": (c_i32, [C.POINTER(GemmDesc)]),
    "insv2v_set_operand_window": (c_i64, [c_i64]),
    "insv2v_conv3x3_fuses_groupnorm": (c_i32, [C.POINTER(GemmDesc)]),
    "insv2v_winograd_input": (c_i32, [C.POINTER(WinogradInDesc), c_p]),
    "insv2v_winograd_output": (c_i32, [C.POINTER(WinogradOutDesc), c_p]),
    "insv2v_ffn_fused": (c_i32, [C.POINTER(FfnDesc), c_p]),
    "insv2v_ffn_stream_elems": (c_i64, [c_i32, c_i32, c_i32]),
    "insv2v_rowlin": (c_i32, [C.POINTER(RowLinDesc), c_p]),
    "insv2v_rowlin_stream_elems": (c_i64, [c_i32, c_i32]),
    "insv2v_tattn_fused": (c_i32, [C.POINTER(TattnDesc), c_p]),
    "insv2v_tattn_stream_elems": (c_i64, [c_i32, c_i32, c_i32]),
    "insv2v_tattn_attn": (c_i32, [C.POINTER(TattnDesc), c_p]),
    "insv2v_tattn_attn_stream_elems": (c_i64, [c_i32, c_i32, c_i32]),
    "insv2v_xattn_fused": (c_i32, [C.POINTER(XattnDesc), c_p]),
    "insv2v_xattn_stream_elems": (c_i64, [c_i32, c_i32, c_i32]),
    "insv2v_xattn_attn": (c_i32, [C.POINTER(XattnDesc), c_p]),
    "insv2v_xattn_attn_stream_elems": (c_i64, [c_i32, c_i32, c_i32]),
    "insv2v_groupnorm": (c_i32, [C.POINTER(GroupNormDesc), c_p]),
    "insv2v_layernorm": (c_i32, [C.POINTER(LayerNormDesc), c_p]),
    "insv2v_layernorm_stats": (c_i32, [c_p, c_p, c_i64, c_i32, c_i32, c_f32, c_p]),
    "insv2v_attention": (c_i32, [C.POINTER(AttentionDesc), c_p]),
    "insv2v_embed_tokens": (c_i32, [c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_p]),
    "insv2v_softmax_rows": (c_i32, [c_p, c_p, c_i64, c_i64, c_i32, c_i32, c_f32, c_p]),
    "insv2v_timestep_embedding": (c_i32, [c_p, c_p, c_i32, c_i32, c_f32, c_p]),
    "insv2v_build_unet_input": (c_i32, [c_p, c_p, c_p, c_p, c_f32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i64, c_i32, c_p]),
    "insv2v_cfg_step": (c_i32, [C.POINTER(StepDesc), c_p]),
    "insv2v_cfg_stats": (c_i32, [c_p, c_p, c_i32, c_i32, c_i32, c_f32, c_f32, c_i64, c_p]),
    "insv2v_warp_image": (c_i32, [c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_p]),
    "insv2v_tap_gather": (c_i32, [c_p, c_i64, c_p, c_p, c_i64, c_i32, c_i32, c_i32, c_i32, c_p]),
    "insv2v_resize_flow": (c_i32, [c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_p]),
    "insv2v_flow_correction": (c_i32, [c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_f32, c_f32, c_p]),
    "insv2v_nchw_to_nhwc_f16": (c_i32, [c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, c_p]),
    "insv2v_nhwc_to_nchw_f32": (c_i32, [c_p, c_i32, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, c_p]),
    "insv2v_posterior_sample": (c_i32, [c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_f32, c_p]),
    "insv2v_im2col": (c_i32, [C.POINTER(Im2colDesc), c_p]),
    "insv2v_instance_norm": (c_i32, [c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i64, c_i64, c_i32, c_f32, c_i32, c_p]),
    "insv2v_ew": (c_i32, [c_i32, c_p, c_p, c_p, c_p, c_i64, c_i32, c_i64, c_i64, c_i64, c_i64, c_p]),
    "insv2v_avgpool2x2": (c_i32, [c_p, c_p, c_i64, c_i32, c_i32, c_p]),
    "insv2v_corr_lookup": (c_i32, [C.POINTER(CorrLookupDesc), c_p]),
    "insv2v_raft_flow_rows": (c_i32, [c_p, c_p, c_i64, c_p, c_i64, c_i32, c_i32, c_i32, c_i32, c_p]),
    "insv2v_convex_upsample": (c_i32, [c_p, c_p, c_i64, c_p, c_i32, c_i32, c_i32, c_p]),
}

_lib = None


class HipKernelError(RuntimeError):
    pass


def load():
    """Load the shared library (once) and attach signatures.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipKernelError(
            f"{LIB_PATH} not found: build it with `python instruct-video-to-video_amd/build.py` "
            "(or __graft_entry__.build()). There is no CPU fallback for the insv2v hot path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    if lib.insv2v_abi_version() != ABI_VERSION:
        raise HipKernelError("libinsv2v_hip.so ABI version mismatch; rebuild the library")
    _lib = lib
    return lib


def check(status, what):
    if status != 0:
        kind = {-1: "invalid argument", -2: "unsupported configuration"}.get(status, f"hipError {status}")
        raise HipKernelError(f"{what}: {kind}")
