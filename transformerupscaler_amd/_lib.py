"""ctypes binding of libtupscale_hip.so (C ABI in include/tupscale_hip.h).

There is deliberately no fallback: if the library is missing or a symbol is absent the
import fails loudly, and every call checks the returned hipError_t.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int, c_longlong, c_uint, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TUP_LIB_PATH") or os.path.join(_HERE, "libtupscale_hip.so")      # override: A/B of two builds
ABI_VERSION = 14

P = c_void_p
I = c_int
F = c_float
U = c_uint

# name -> argtypes, mirrors include/tupscale_hip.h one to one
SIGNATURES = {
    "tup_abi_version": [],
    "tup_conv3x3_c3_fwd": [P, P, P, P, P, P, I, I, I, I, P],
    "tup_conv3x3_c64_fwd": [P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, P],
    "tup_conv5x5_c64_planar_fwd": [P, P, P, P, P, P, I, I, I, I, I, P],
    "tup_conv3x3_planar_fwd": [P, P, P, P, P, I, I, I, I, I, P],
    "tup_resize_aa_fwd": [P, P, P, P, P, I, P, P, P, I, I, I, I, I, I, I, P],
    "tup_tail_fused_fwd": [P, P, P, P, P, P, P, P, P, P, I, P, P, P, I, I, I, I, I, I, I, I, I, I, P],
    "tup_tail_stream_r2_fwd": [P, P, P, P, P, P, P, I, I, I, I, P],
    "tup_tail_stream_r2_resize_fwd": [P] * 7 + [P, P, P, I, P, P, P, I, P, P] + [I] * 9 + [P],
    "tup_clamp01_fwd": [P, P, c_longlong, P],
    "tup_layernorm_fwd": [P, P, P, P, P, P, I, P],
    "tup_relpos_bias_expand": [P, P, P],
    "tup_window_attn_fwd": [P, P, P, P, I, F, U, P],
    "tup_gemm_tokens_fwd": [P, I, I, P, P, P, P, P, I, I, I, I, I, F, U, P],
    "tup_ln_gemm_fwd": [P, P, P, P, P, P, I, I, P],
    "tup_fused_mlp_fwd": [P, P, P, P, P, P, P, I, P],
    "tup_patch_embed_fwd": [P, P, P, P, I, I, I, P],
    "tup_patch_unembed_fwd": [P, I, P, P, P, P, I, I, I, P],
    # ResidualTransformer
    "tup_rt_patch_embed_fwd": [P, P, P, P, P, I, I, I, P],
    "tup_rt_patch_unembed_fwd": [P, P, P, P, P, I, I, I, P],
    "tup_rt_attention_fwd": [P, P, P, I, I, F, U, P],
    "tup_rt_attention_bwd": [P, P, P, P, P, P, I, I, F, U, P],
    "tup_layernorm128_fwd": [P, P, P, P, P, P, I, P],
    "tup_layernorm128_bwd": [P, P, P, P, P, P, P, P, P, I, P, F, U, P],
    "tup_rt_patch_wgrad": [P, P, P, I, I, I, P],
    "tup_conv3x3_c64_wgrad_s2d": [P, P, P, P, I, I, I, I, I, P],
    "tup_rt_bicubic_bwd": [P] * 10 + [I, I, I, I, I, P],
    "tup_rt_bicubic_bwd_banded": [P] * 7 + [I, P, P, I, P, P, I, I, I, I, I, P, P],
    "tup_relpos_bias_expand_h": [P, P, I, P],
    "tup_relpos_bias_expand_n_h": [P, P, I, P],
    "tup_window_attn_bwd_h": [P, P, P, P, P, P, P, P, I, I, F, U, P],
    "tup_relpos_bias_reduce_h": [P, P, I, P],
    "tup_wt_patch_wgrad": [P, P, P, I, I, I, I, P],
    "tup_window_attn_fwd_h": [P, P, P, P, I, I, F, U, P],
    "tup_wt_patch_embed_fwd": [P, P, P, P, I, I, I, I, P],
    "tup_wt_patch_unembed_fwd": [P, P, P, P, P, I, I, I, I, P],
    "tup_fused_qkv_attn_fwd": [P, P, P, P, P, P, P, I, P],
    "tup_fused_attn_block_fwd": [P, P, P, P, P, P, P, P, I, P],
    "tup_fused_block_fwd": [P] * 10 + [I, P],
    "tup_fused_blocks32_fwd": [P, P, I, I, P],
    "tup_blocks_stream_fwd": [P, P, P, I, I, P],
    "tup_clock_probe": [P, P],
    "tup_pack_gather": [P, P, I, P, P, c_longlong, I, P],
    "tup_adam_step": [P, P, I, P],
    "tup_l1_loss_partial": [P, P, P, c_longlong, I, P],
    "tup_l1_loss_bwd": [P, P, P, P, c_longlong, P],
    "tup_u8hwc_to_f32chw": [P, P, I, I, I, I, P],
    "tup_f32chw_to_u8hwc": [P, P, I, I, I, I, P],
    "tup_bra_compose": [P, P, P, P, P, P, P, P],
    "tup_bra_backward": [P, P, P, P, P, P, P, P, P, I, I, I, P],
    "tup_bra_chain": [P, P, P, P, P, P, P, P, P, P, P],
    "tup_resize_u8_rows": [P, P, P, P, P, I, I, I, I, I, P],
    "tup_resize_u8_cols": [P, P, P, P, P, P, I, I, I, I, I, I, P],
    "tup_rt_bicubic_sum_fwd": [P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, P],
    # backward
    "tup_gemm_wgrad": [P, I, I, P, I, I, P, I, I, I, I, P],
    "tup_gemm_wgrad_bias": [P, I, I, P, I, I, P, I, P, I, I, I, P],
    "tup_patch_wgrad": [P, P, P, I, I, I, I, P],
    "tup_patch_wgrad_bf16": [P, P, P, I, I, I, I, P],
    "tup_colsum": [P, I, I, P, I, I, P, P],
    "tup_layernorm_bwd": [P, P, P, P, P, P, P, P, P, I, P, F, U, P],
    "tup_relpos_bias_expand_n": [P, P, P],
    "tup_window_attn_bwd": [P, P, P, P, P, P, P, P, I, F, U, P],
    "tup_window_attn_bwd_scratch": [I, I],
    "tup_dropout_bwd": [P, P, c_longlong, F, U, P],
    "tup_relpos_bias_reduce": [P, P, P],
    "tup_patch_unembed_bwd": [P, P, P, I, I, I, P],
    "tup_patch_embed_bwd": [P, P, P, I, I, I, P],
    "tup_patch_embed_bwd_merge": [P, P, P, P, P, P, P, I, I, I, P],
    "tup_conv3x3_c64_wgrad": [P, P, P, P, I, I, I, I, I, P],
    "tup_conv3x3_thin_wgrad": [P, P, P, P, I, I, I, P],
    "tup_conv3x3_c3_wgrad": [P, P, P, P, I, I, I, P],
    "tup_conv3x3_planar_wgrad": [P, P, P, P, I, I, I, I, P],
    "tup_conv3x3_planar_dgrad": [P, P, P, I, I, I, I, P],
    "tup_resize_aa_bwd": [P, P, P, P, P, I, P, P, I, P, P, P, P, I, I, I, I, I, P, P],
    "tup_mask_bwd": [P, P, P, P, c_longlong, P, P],
    "tup_feat_grad_combine": [P, P, P, P, P, I, I, I, P],
}


class TupscaleLibraryError(RuntimeError):
    pass


_lib = None


COUNT_RETURNING = {"tup_window_attn_bwd_scratch"}      # return an element count, not a hipError_t


def load():
    """Load the HIP library once; raise TupscaleLibraryError (never fall back) if unusable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TupscaleLibraryError(
            f"{LIB_PATH} not found: build it with `make -C transformerupscaler_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    # The kernels must run in the same HIP runtime instance as the device memory and streams they are
    # handed.  torch bundles its own libamdhip64; importing it first makes the dynamic loader resolve our
    # DT_NEEDED libamdhip64.so.N to that already-loaded copy instead of a second runtime from /opt/rocm.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise TupscaleLibraryError(f"{LIB_PATH} does not export {name}; rebuild it") from e
        fn.argtypes = argtypes
        fn.restype = c_longlong if name in COUNT_RETURNING else c_int
    if lib.tup_abi_version() != ABI_VERSION:
        raise TupscaleLibraryError(f"ABI mismatch: library {lib.tup_abi_version()} != binding {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


_fns = {}


def call(name: str, *args):
    fn = _fns.get(name)
    if fn is None:
        fn = _fns[name] = getattr(load(), name)
    err = fn(*args)
    if err != 0:
        raise RuntimeError(f"{name} failed with hipError_t {err}")
